"""Per-kernel summary of a rocprofv3 kernel trace (csv): python tools/trace_summary.py <kernel_trace.csv> <steps> [out.csv]
prints ms per step, calls per step and the average duration per kernel (names shortened as in tools/pmc_summary.py) and
optionally writes the same rows as csv."""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import ROLES, short  # noqa: E402

csv.field_size_limit(1 << 30)


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    d = collections.defaultdict(lambda: [0, 0])
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            d[k][0] += 1
            d[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    rows = [{"kernel": k, "role": ROLES.get(k, ""), "ms_per_step": round(v[1] / steps / 1e6, 4), "calls_per_step": round(v[0] / steps, 2),
             "avg_us": round(v[1] / v[0] / 1e3, 2)} for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])]
    total = sum(r["ms_per_step"] for r in rows)
    print(f"kernel time {total:.3f} ms/step over {steps} steps")
    for r in rows[:20]:
        print(f'{r["ms_per_step"]:8.3f} ms/step {r["calls_per_step"]:7.1f} calls {r["avg_us"]:9.1f} us  {r["kernel"]}  {r["role"]}')
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)


if __name__ == "__main__":
    main()
