set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3a/bench_base.json 2> gpurun_out/r3a/bench_base.err
tail -c 600 gpurun_out/r3a/bench_base.json
timeout -k 10 300 python tools/bench_gemm.py > gpurun_out/r3a/gemm_base.txt 2>&1
cat gpurun_out/r3a/gemm_base.txt
timeout -k 10 120 python tools/bench_attn.py > gpurun_out/r3a/attn_base.txt 2>&1
timeout -k 10 120 python tools/bench_attn.py 8 5329 16 88 >> gpurun_out/r3a/attn_base.txt 2>&1
timeout -k 10 120 python tools/bench_attn.py 64 256 16 80 >> gpurun_out/r3a/attn_base.txt 2>&1
cat gpurun_out/r3a/attn_base.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3a/blaslt -o blaslt -- python3 tools/hipblaslt_names.py run > gpurun_out/r3a/blaslt.log 2>&1
python3 tools/hipblaslt_names.py report gpurun_out/r3a/blaslt > gpurun_out/r3a/hipblaslt_kernels.txt
cat gpurun_out/r3a/hipblaslt_kernels.txt
bash tools/pmc_collect.sh gpurun_out/r3a/pmc > gpurun_out/r3a/pmc.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r3a/pmc gpurun_out/r3a/pmc_summary --steps 5 > gpurun_out/r3a/pmc_summary.txt 2>&1
cat gpurun_out/r3a/pmc_summary.txt
