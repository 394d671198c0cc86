# Headline measurements of one build on one box: the bench line, the kernel table of the same command under
# rocprofv3 --kernel-trace, and the hardware-counter passes.   usage: bash tools/gpu_profile_headline.sh <out_dir>
set -e
cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/headline}
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace.err
python3 tools/trace_summary.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) 7 $OUT/kernel_stats.csv > $OUT/kernel_stats.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats.csv
head -24 $OUT/kernel_stats.txt
bash tools/pmc_collect.sh $OUT/pmc > $OUT/pmc.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc $OUT/pmc_summary --steps 5 --raw-dir $OUT/pmc_raw > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
rm -rf $OUT/trace $OUT/pmc
