# Secondary measurements of one build on one box: every bench workload (value, and a kernel table from rocprofv3
# --kernel-trace), the GEMM and attention micro-benchmarks, the tile cycle stamps.  usage: bash tools/gpu_final_secondary.sh <out_dir>
set -e
cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/secondary}
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.json; echo
for w in vit_s16_224_fp32_fwd vit_b14_224_readme_dropout vit_l16_384_srelu_rope vit_h14_224_droppath_ls_ckpt vit_g14_1024_swiglu_rope; do
  steps=10; [ $w = vit_g14_1024_swiglu_rope ] && steps=4
  timeout -k 10 500 python bench.py --workload $w --steps $steps --warmup 2 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python3 -c "
import json,sys
d=json.loads(open('$OUT/bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['value'], d['unit'], d['ms_per_step'], 'ms/step')"
  timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$w -o trace -- python3 bench.py --workload $w --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/trace_$w.err
  python3 tools/trace_summary.py $(find $OUT/trace_$w -name "*kernel_trace.csv" | head -1) 5 $OUT/kernel_stats_$w.csv > $OUT/kernel_stats_$w.txt
  rm -rf $OUT/trace_$w
  head -4 $OUT/kernel_stats_$w.txt
done
timeout -k 10 300 python tools/bench_gemm.py 256 2>&1 | grep -v amdgpu.ids > $OUT/gemm_microbench.txt
tail -3 $OUT/gemm_microbench.txt
: > $OUT/attn_bench.txt
for shape in "256 272 12 64" "192 257 16 80" "64 576 16 64" "8 5329 16 88"; do
  timeout -k 10 120 python tools/bench_attn.py $shape 2>&1 | grep -v amdgpu.ids >> $OUT/attn_bench.txt
done
cat $OUT/attn_bench.txt
: > $OUT/stamps.txt
for m in store glu dglu res out fc2 wgrad1 wgrad2 wgradq nn1 nnq; do
  VITK_LIB=exp/libvitk_stamp.so timeout -k 10 120 python tools/stamp_gemm.py $m 2>&1 | grep -v amdgpu.ids >> $OUT/stamps.txt
done
tail -4 $OUT/stamps.txt
