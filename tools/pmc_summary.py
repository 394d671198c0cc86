"""Per-kernel summary of the hardware-counter passes taken by tools/pmc_collect.sh.

    python tools/pmc_summary.py gpurun_out/pmc profiles/r02_pmc_summary --steps 5

reads <dir>/{mfma,fetch,write,l2}/*_counter_collection.csv (+ the kernel trace of the mfma pass) and writes
<out>.csv (one row per kernel: calls, summed duration, raw counter sums, derived figures) and <out>.json (the same
rows, the whole-run totals, the formulas, and an identification of the build the passes ran on).

Derived figures (gfx950):
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * CUs * 4)
      SQ_VALU_MFMA_BUSY_CYCLES sums, over every SIMD, the cycles its matrix unit was busy; GRBM_GUI_ACTIVE is
      reported summed over the 8 XCDs, so GRBM_GUI_ACTIVE / 8 is the dispatch's length in shader cycles, and a chip
      whose every matrix unit never idles reads 1.0.  This is busy TIME, not achieved FLOP/s over the dense peak: the
      clock under load is below the 2.4 GHz the peak is quoted at.
  clock_ghz      = GRBM_GUI_ACTIVE / 8 / duration       (reads high on dispatches shorter than ~0.3 ms)
  read_bytes     = 2 * FETCH_SIZE * 1024   (FETCH_SIZE tallies the 128-byte requests of wide reads at 64 bytes)
  write_bytes    = WRITE_SIZE * 1024
  l2_hit_rate    = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)   (all eight per-XCD L2s together; the l2 pass is optional)
"""
import argparse
import collections
import csv
import hashlib
import json
import os
import re
import subprocess
import sys

csv.field_size_limit(1 << 30)
CUS = 256


ROLES = {   # what the instances do in the encoder step (template arguments: trans_a, trans_b, epilogue, output type)
    "gemm_mfma256d_k<true, true, -1, float>": "weight gradients (TN, split-K partials)",
    "gemm_mfma256d_k<false, false, 4, bf16>": "fc1 forward + bias + SwiGLU",
    "gemm_mfma256d_k<false, false, 3, bf16>": "fc1 forward + bias + activation",
    "gemm_mfma256d_k<false, true, 0, bf16>": "input gradients (NN)",
    "gemm_mfma256d_k<false, false, 0, bf16>": "qkv forward",
    "gemm_mfma256d_k<false, false, 1, float>": "fc2 forward + residual (fp32 stream)",
    "gemm_mfma256d_k<false, true, 6, bf16>": "fc2 input gradient + SwiGLU backward",
    "gemm_mfma256d_k<false, true, 5, bf16>": "fc2 input gradient + activation backward",
    "gemm_mfma256d_k<false, false, 7, bf16>": "qkv forward + RoPE on q and k",
    "gemm_mfma_k<false, true, -1, bf16>": "rows cut off a ragged round (NN, split-K partials)",
    "gemm_mfma_k<false, false, 0, bf16>": "rows cut off a ragged round (qkv forward)",
    "gemm_mfma_k<false, true, 6, bf16>": "fc2 input gradient + SwiGLU backward",
    "gemm_mfma_k<false, true, 5, bf16>": "fc2 input gradient + activation backward",
    "gemm_mfma_k<false, false, 1, float>": "out_proj forward + residual (fp32 stream); rows cut off fc2's ragged round",
    "gemm_mfma_k<false, true, 0, bf16>": "small input gradients (NN, 128x128 tiles)",
}


def demangle_templates(name: str) -> str:
    """rocprofv3 leaves names with a __bf16 template argument mangled (_Z15gemm_mfma256d_kILb0ELb0ELi4EDF16bEv...)."""
    m = re.match(r"_Z\d+([A-Za-z_]\w*?)I((?:Lb[01]E|Lin?\d+E|DF16b|f)+)Ev", name)
    if not m:
        return name
    args = []
    for tok in re.findall(r"Lb[01]E|Lin?\d+E|DF16b|f", m.group(2)):
        if tok.startswith("Lb"):
            args.append("true" if tok[2] == "1" else "false")
        elif tok.startswith("Li"):
            args.append(tok[2:-1].replace("n", "-"))
        else:
            args.append("bf16" if tok == "DF16b" else "float")
    return f"void {m.group(1)}<{', '.join(args)}>("


def short(name: str) -> str:
    """Kernel name without argument lists; template arguments are kept for this library's kernels only."""
    name = demangle_templates(name.strip('"'))
    m = re.match(r"(?:void\s+)?([A-Za-z_][\w:]*(?:\(anonymous namespace\)::)?[\w:]*)", name)
    base = m.group(1) if m else name[:60]
    if base.startswith(("gemm_", "attn_", "norm_", "rope_", "colsum", "splitk", "scale_cast", "im2col", "act_", "prefix_",
                        "hold_", "copy_samples")):
        t = re.match(r"(?:void\s+)?[\w:]+<(.*?)>\(", name)
        if t:
            args = re.sub(r"__hip_bfloat16|__bf16|hip_bfloat16", "bf16", t.group(1))
            return f"{base}<{args}>"
        return base
    if "at::native" in name or "at::cuda" in name:
        inner = re.search(r"(\w+_kernel\w*|\w+Functor\w*|CatArrayBatchedCopy\w*|reduce_kernel)", name)
        return "torch:" + (inner.group(1) if inner else base.split("::")[-1])
    return base


def read_counters(path):
    """{dispatch_id: {counter: value}}, {dispatch_id: (kernel, start, end)}"""
    vals, meta = collections.defaultdict(dict), {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            d = int(row["Dispatch_Id"])
            vals[d][row["Counter_Name"]] = vals[d].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            meta[d] = (short(row["Kernel_Name"]), int(row["Start_Timestamp"]), int(row["End_Timestamp"]))
    return vals, meta


def find(dirname, suffix):
    for root, _, files in os.walk(dirname):
        for fn in files:
            if fn.endswith(suffix):
                return os.path.join(root, fn)
    raise FileNotFoundError(f"{suffix} under {dirname}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pmc_dir")
    ap.add_argument("out_prefix")
    ap.add_argument("--steps", type=int, required=True, help="benchmark steps the passes ran (warm-up + timed)")
    ap.add_argument("--command", default="python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline")
    ap.add_argument("--raw-dir", default=None, help="also write the passes' counter rows (kernel names shortened) as "
                    "<raw-dir>/{mfma,fetch,write}_counter_rows.csv")
    a = ap.parse_args()

    PASSES = [n for n in ("mfma", "fetch", "write", "l2") if os.path.isdir(os.path.join(a.pmc_dir, n))]
    if a.raw_dir:   # the counter rows themselves, so that every derived figure can be recomputed from the repository alone
        os.makedirs(a.raw_dir, exist_ok=True)
        for pass_name in PASSES:
            with open(find(os.path.join(a.pmc_dir, pass_name), "counter_collection.csv"), newline="") as f, \
                    open(os.path.join(a.raw_dir, f"{pass_name}_counter_rows.csv"), "w", newline="") as o:
                w = csv.writer(o)
                w.writerow(["Dispatch_Id", "Kernel", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value",
                            "Start_Timestamp", "End_Timestamp"])
                for r in csv.DictReader(f):
                    w.writerow([r["Dispatch_Id"], short(r["Kernel_Name"]), r["Grid_Size"], r["Workgroup_Size"], r["Counter_Name"],
                                r["Counter_Value"], r["Start_Timestamp"], r["End_Timestamp"]])

    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    for pass_name, counters in (("mfma", ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE")),
                                ("fetch", ("FETCH_SIZE",)), ("write", ("WRITE_SIZE",)), ("l2", ("TCC_HIT_sum", "TCC_MISS_sum"))):
        if pass_name not in PASSES:
            continue
        vals, meta = read_counters(find(os.path.join(a.pmc_dir, pass_name), "counter_collection.csv"))
        for d, (kern, t0, t1) in meta.items():
            r = rows[kern]
            r[f"calls_{pass_name}"] += 1
            r[f"ns_{pass_name}"] += t1 - t0
            for c in counters:
                r[c] += vals[d].get(c, 0.0)

    out = []
    for kern, r in rows.items():
        gui, ns = r["GRBM_GUI_ACTIVE"], r["ns_mfma"]
        cyc = gui / 8.0
        out.append({
            "kernel": kern, "role": ROLES.get(kern, ""), "calls_per_step": round(r["calls_mfma"] / a.steps, 2), "ms_per_step": round(ns / a.steps / 1e6, 4),
            "SQ_VALU_MFMA_BUSY_CYCLES": r["SQ_VALU_MFMA_BUSY_CYCLES"], "SQ_BUSY_CYCLES": r["SQ_BUSY_CYCLES"],
            "GRBM_GUI_ACTIVE": gui, "FETCH_SIZE_KB": r["FETCH_SIZE"], "WRITE_SIZE_KB": r["WRITE_SIZE"],
            "mfma_busy_frac": round(r["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * CUS * 4), 4) if cyc else 0.0,
            "clock_ghz": round(cyc / ns, 3) if ns else 0.0,
            "read_bytes_per_call": round(2 * r["FETCH_SIZE"] * 1024 / max(r["calls_fetch"], 1)),
            "write_bytes_per_call": round(r["WRITE_SIZE"] * 1024 / max(r["calls_write"], 1)),
            "TCC_HIT_sum": r["TCC_HIT_sum"], "TCC_MISS_sum": r["TCC_MISS_sum"],
            "l2_hit_rate": round(r["TCC_HIT_sum"] / (r["TCC_HIT_sum"] + r["TCC_MISS_sum"]), 4) if r["TCC_HIT_sum"] + r["TCC_MISS_sum"] else None,
            "hbm_side_gbps": round((2 * r["FETCH_SIZE"] * 1024 / max(r["ns_fetch"], 1) + r["WRITE_SIZE"] * 1024 / max(r["ns_write"], 1)), 1),
        })
    out.sort(key=lambda x: -x["ms_per_step"])
    tot_ns = sum(r["ns_mfma"] for r in rows.values())
    tot_busy = sum(r["SQ_VALU_MFMA_BUSY_CYCLES"] for r in rows.values())
    tot_cyc = sum(r["GRBM_GUI_ACTIVE"] for r in rows.values()) / 8.0
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(here, "vit_amd", "libvitk.so")
    try:
        commit = subprocess.run(["git", "-C", here, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "-C", here, "status", "--porcelain", "--", "vit_amd", "bench.py"],
                                    capture_output=True, text=True).stdout.strip())
    except OSError:
        commit, dirty = "", False
    summary = {
        "command": f"rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- {a.command}  (one pass per counter group: "
                   "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE | FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum; tools/pmc_collect.sh)",
        "steps_in_run": a.steps,
        "build": {"git_commit": commit, "tree_modified_since_commit": dirty,
                  "libvitk_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None},
        "formulas": {"mfma_busy_frac": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)",
                     "clock_ghz": "GRBM_GUI_ACTIVE / 8 / duration_ns",
                     "read_bytes": "2 * FETCH_SIZE * 1024 (gfx950 tallies 128-byte read requests at 64 bytes)",
                     "write_bytes": "WRITE_SIZE * 1024",
                     "l2_hit_rate": "TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)"},
        "whole_run": {"kernel_ms_per_step": round(tot_ns / a.steps / 1e6, 3),
                      "mfma_busy_frac": round(tot_busy / (tot_cyc * CUS * 4), 4) if tot_cyc else 0.0,
                      "mean_clock_ghz": round(tot_cyc / tot_ns, 3) if tot_ns else 0.0},
        "kernels": out,
    }
    with open(a.out_prefix + ".json", "w") as f:
        json.dump(summary, f, indent=1)
    with open(a.out_prefix + ".csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(out[0].keys()))
        w.writeheader()
        w.writerows(out)
    print(json.dumps(summary["whole_run"]))
    for r in out[:14]:
        print(f'{r["ms_per_step"]:8.3f} ms/step {r["calls_per_step"]:6.1f} calls  mfma_busy {r["mfma_busy_frac"]:.3f}  '
              f'{r["clock_ghz"]:.2f} GHz  rd {r["read_bytes_per_call"] / 1e6:8.1f} MB wr {r["write_bytes_per_call"] / 1e6:8.1f} MB  '
              f'L2 hit {r["l2_hit_rate"] if r["l2_hit_rate"] is not None else float("nan"):.3f}  {r["kernel"][:60]}  {r["role"]}')


if __name__ == "__main__":
    sys.exit(main())
