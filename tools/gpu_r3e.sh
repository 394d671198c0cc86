set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
run_bench() { timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"; }
VITK_LIB=exp/libvitk_base.so run_bench base
run_bench new
VITK_LIB=exp/libvitk_base.so run_bench base
run_bench new
VITK_LIB=exp/libvitk_base.so timeout -k 10 300 python tools/bench_gemm.py 256 "fc2 dglu" 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/bench_gemm.py 256 "fc2 dglu" 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/bench_gemm.py 256 "out fwd" 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/bench_gemm.py 256 "fc2 fwd" 2>&1 | grep -v amdgpu.ids
