#!/bin/bash
# Hardware-counter passes over the benchmark step (run on the GPU box, from the repository root):
#   pass mfma : SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE + kernel trace (durations of the same dispatches)
#   pass fetch: FETCH_SIZE  (L2 memory-side read requests; takes 3 of the 4 TCC slots)
#   pass write: WRITE_SIZE  (takes 2 TCC slots, so it cannot share a pass with FETCH_SIZE)
#   pass l2   : TCC_HIT_sum, TCC_MISS_sum  (hit rate of the per-XCD L2s)
# Each pass is its own rocprofv3 run with counters and the kernel trace only.  tools/pmc_summary.py turns the CSVs
# into profiles/rNN_pmc_summary.{csv,json}.
#   usage: bash tools/pmc_collect.sh [out_dir] [bench.py arguments ...]
set -u
OUT=${1:-gpurun_out/pmc}
shift || true
ARGS=${*:---steps 3 --warmup 2 --no-cpu-baseline}
# One process only: with --gpus N > 1 bench.py starts its ranks as child processes, and under --pmc the profiler's
# preloaded library has already initialised the GPU in the parent by then.
case " $ARGS " in *" --gpus "*|*" --gpus="*) echo "pmc_collect.sh: counter passes run on one GPU; drop --gpus" >&2; exit 2;; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
run_pass() {   # name, counters...
  local name=$1; shift
  echo "[pmc] pass $name: $*"
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o "$name" -- \
    python3 bench.py $ARGS > "$OUT/$name.log" 2>&1
  local rc=$?
  tail -1 "$OUT/$name.log" | cut -c1-200
  return $rc
}
run_pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE || exit 1
run_pass fetch FETCH_SIZE || exit 1
run_pass write WRITE_SIZE || exit 1
run_pass l2 TCC_HIT_sum TCC_MISS_sum || exit 1
find "$OUT" -name "*.csv" | sort
