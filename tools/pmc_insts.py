"""Instruction counts per launch from one rocprofv3 counter pass (millions of wave-level instructions, summed over the chip):
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d out -o a -- python3 tools/bench_dglu.py dglu
  python tools/pmc_insts.py out/.../a_counter_collection.csv
Comparing an epilogue's launch with the plain product of the same shape gives the epilogue's own instruction count."""
import collections
import csv
import sys

csv.field_size_limit(1 << 30)
sums = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
first = None
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    first = first or r["Counter_Name"]
    sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == first:
        calls[k] += 1
for k, v in sums.items():
    if "gemm" in k or "attn" in k:
        c = calls[k] or 1
        print(k, "calls", c, {a: round(b / c / 1e6, 2) for a, b in sorted(v.items())})
