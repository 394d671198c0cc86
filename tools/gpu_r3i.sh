set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
timeout -k 10 600 python tools/bench_gemm.py 2>&1 | grep -v "amdgpu.ids" | cut -c1-48,100-190
run_bench() { timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"; }
VITK_LIB=exp/libvitk_base.so run_bench base
run_bench new
VITK_LIB=exp/libvitk_base.so run_bench base
run_bench new
