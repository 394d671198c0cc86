# Same-box A/B of two builds of libvitk.so on bench.py workloads: the library under test against one kept under exp/
# (exp/ must then travel: take "./exp" out of .gpurunignore for the call).
# usage: bash tools/gpu_ab.sh exp/libvitk_base.so [workload ...]
set -e
cd $GRAFT_REPO_ROOT
BASE=$1; shift
WORKLOADS=${@:-vit_b14_224_swiglu_reg16_fourier}
run_bench() { VITK_LIB=$2 timeout -k 10 500 python bench.py --workload $1 --steps $3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['value'], d['unit'], d['ms_per_step'], 'ms/step; dominant kernel', d['roofline']['launch_ms'], 'ms')"; }
for w in $WORKLOADS; do
  steps=10; [ $w = vit_g14_1024_swiglu_rope ] && steps=4
  for rep in 1 2; do
    run_bench $w $BASE $steps
    run_bench $w vit_amd/libvitk.so $steps
  done
done
