#!/usr/bin/env python3
"""Condense a rocprofv3 --output-format csv directory into a small text summary.

    python tools/prof_summary.py <rocprof_out_dir> <summary.txt> [--delete-raw]

Keeps: the kernel_stats rows (top 12 by total time plus every sea:: kernel), per-dispatch
durations of the sea:: kernels from kernel_trace (count / mean / min / max, VGPR / LDS / grid), and
every PMC counter row of the sea:: kernels (sum and per-dispatch mean).  The raw CSVs are large
(the synthetic-corpus generator alone is thousands of torch dispatches) and are scratch.
"""
import csv
import glob
import os
import shutil
import sys
from collections import defaultdict


def main():
    src, dst = sys.argv[1], sys.argv[2]
    delete = "--delete-raw" in sys.argv[3:]
    lines = [f"# rocprofv3 summary of {os.path.basename(src.rstrip('/'))}"]

    for path in glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(path)))
        lines.append("\n## kernel_stats (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs)")
        keep = rows[:12] + [r for r in rows[12:] if "sea" in r.get("Name", "")]
        for r in keep:
            lines.append(", ".join(str(r.get(k, "")) for k in
                                   ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")))

    for path in glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True):
        per = defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(path)):
            name = r.get("Kernel_Name", "")
            if "sea" not in name:
                continue
            per[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            meta[name] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                                 "Scratch_Size", "Grid_Size", "Workgroup_Size")}
        if per:
            lines.append("\n## kernel_trace, sea:: kernels (per-dispatch duration ns)")
            for name, d in per.items():
                lines.append(f"{name}: n={len(d)} mean={sum(d) / len(d):.0f} min={min(d)} max={max(d)} {meta[name]}")

    for path in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
        acc = defaultdict(list)
        for r in csv.DictReader(open(path)):
            name = r.get("Kernel_Name", "")
            if "sea" not in name:
                continue
            acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
        if acc:
            lines.append("\n## PMC counters, sea:: kernels (per dispatch)")
            for (name, ctr), v in acc.items():
                lines.append(f"{name} {ctr}: dispatches={len(v)} mean={sum(v) / len(v):.1f} min={min(v):.1f} max={max(v):.1f}")

    os.makedirs(os.path.dirname(os.path.abspath(dst)), exist_ok=True)
    with open(dst, "w") as f:
        f.write("\n".join(lines) + "\n")
    if delete:
        shutil.rmtree(src, ignore_errors=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
