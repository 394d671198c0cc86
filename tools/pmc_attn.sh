#!/bin/bash
# Counter passes over the attention micro-benchmark (run on the GPU box from the repository root).
#   usage: bash tools/pmc_attn.sh out_dir   (environment such as VITK_ATTN_LONG=1 is inherited by the benchmark)
set -u
OUT=${1:-gpurun_out/pmc_attn}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/sq" -o sq -- python3 tools/bench_attn.py > "$OUT/sq.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o fetch -- python3 tools/bench_attn.py > "$OUT/fetch.log" 2>&1 || exit 1
echo done
