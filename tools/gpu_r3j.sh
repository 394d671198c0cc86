set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/full_gpu_tests.log 2>&1 || (tail -30 gpurun_out/full_gpu_tests.log; exit 1)
tail -2 gpurun_out/full_gpu_tests.log
run_bench() { VITK_LIB=$2 timeout -k 10 400 python bench.py --workload $1 --steps $3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"; }
run_bench vit_b14_224_readme_dropout exp/libvitk_base.so 10
run_bench vit_b14_224_readme_dropout vit_amd/libvitk.so 10
run_bench vit_b14_224_swiglu_reg16_fourier exp/libvitk_base.so 10
run_bench vit_b14_224_swiglu_reg16_fourier vit_amd/libvitk.so 10
