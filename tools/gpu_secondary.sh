set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3_secondary
run_bench() { VITK_LIB=$2 timeout -k 10 400 python bench.py --workload $1 --steps $3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"; }
for w in vit_l16_384_srelu_rope vit_h14_224_droppath_ls_ckpt; do
run_bench $w exp/libvitk_base.so 10
run_bench $w vit_amd/libvitk.so 10
done
run_bench vit_g14_1024_swiglu_rope exp/libvitk_base.so 4
run_bench vit_g14_1024_swiglu_rope vit_amd/libvitk.so 4
run_bench vit_b14_224_readme_dropout exp/libvitk_base.so 10
run_bench vit_b14_224_readme_dropout vit_amd/libvitk.so 10
