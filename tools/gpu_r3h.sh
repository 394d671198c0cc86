set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
for w in 0 1; do
echo "== warm $w"
for c in "qkv fwd" "fc1 fwd" "fc1 glu" "fc2 dglu" "fc2 fwd" "out fwd" "fc2 dgrad" "fc1 dgrad" "qkv dgrad" "fc1 wgrad"; do
VITK_WARM=$w timeout -k 10 300 python tools/bench_gemm.py 256 "$c" 2>&1 | grep -v "amdgpu.ids\|^sum" | cut -c1-48,100-190
done
done
run_bench() { timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"; }
VITK_WARM=0 run_bench warm0
VITK_WARM=1 run_bench warm1
VITK_WARM=0 run_bench warm0
VITK_WARM=1 run_bench warm1
