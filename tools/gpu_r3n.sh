set -e
cd $GRAFT_REPO_ROOT
run_bench() { VITK_LIB=$2 timeout -k 10 400 python bench.py --workload $1 --steps $3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"; }
cp vit_amd/libvitk.so /tmp/new.so
for w in vit_b14_224_swiglu_reg16_fourier vit_l16_384_srelu_rope vit_h14_224_droppath_ls_ckpt; do
run_bench $w exp/libvitk_prev.so 10
run_bench $w vit_amd/libvitk.so 10
run_bench $w exp/libvitk_prev.so 10
run_bench $w vit_amd/libvitk.so 10
done
