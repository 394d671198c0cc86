set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout -k 10 400 python tools/bench_policy.py 7 > gpurun_out/r3b/policy.txt 2>&1
cat gpurun_out/r3b/policy.txt
for sp in 1 2 3; do
VITK_STORE_POLICY=$sp timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('policy $sp', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
done
timeout -k 10 900 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -5
