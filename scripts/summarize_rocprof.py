#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace and/or PMC counter collection)
into a small per-(kernel, grid) table that can be committed under profiles/.

    python scripts/summarize_rocprof.py gpurun_out/prof_r01/bench_kernel_trace.csv > profiles/r01_kernel_trace_summary.md
    python scripts/summarize_rocprof.py --pmc gpurun_out/pmc_fetch_r01/pmc_counter_collection.csv
"""
import argparse
import collections
import csv
import re
import statistics


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "")[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--pmc", action="store_true")
    ap.add_argument("--only", default="amdr::", help="substring filter on kernel names ('' = all)")
    ap.add_argument("--timeline", type=int, default=0, help="print the last N dispatches in start order: start offset, "
                    "duration, gap to the end of the previous dispatch (us)")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    if a.timeline:
        seq = sorted((r for r in rows if a.only in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))[-a.timeline:]
        t0 = int(seq[0]["Start_Timestamp"])
        prev_end = None
        print("| # | kernel | grid | start us | dur us | gap us |")
        print("|---|---|---|---|---|---|")
        for i, r in enumerate(seq):
            st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            grid = r.get("Grid_Size") or "x".join(r[f"Grid_Size_{c}"] for c in "XYZ")
            gap = "" if prev_end is None else f"{(st - prev_end) / 1e3:.1f}"
            print(f"| {i} | `{short(r['Kernel_Name'])}` | {grid} | {(st - t0) / 1e3:.1f} | {(en - st) / 1e3:.1f} | {gap} |")
            prev_end = en
        return
    if a.pmc:
        agg = collections.defaultdict(list)
        for r in rows:
            if a.only in r["Kernel_Name"]:
                agg[(short(r["Kernel_Name"]), r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        print("| kernel | grid | counter | dispatches | mean value |")
        print("|---|---|---|---|---|")
        for (k, g, c), v in sorted(agg.items()):
            print(f"| `{k}` | {g} | {c} | {len(v)} | {statistics.mean(v):.2f} |")
        return
    agg = collections.defaultdict(list)
    for r in rows:
        if a.only in r["Kernel_Name"]:
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            grid = r.get("Grid_Size") or "x".join(r[f"Grid_Size_{c}"] for c in "XYZ")
            agg[(short(r["Kernel_Name"]), grid, r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?"))].append(dur)
    print("| kernel | grid threads | VGPR | LDS B | dispatches | mean us | median us | min us | max us |")
    print("|---|---|---|---|---|---|---|---|---|")
    for (k, g, v, l), d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"| `{k}` | {g} | {v} | {l} | {len(d)} | {statistics.mean(d)/1e3:.1f} | {statistics.median(d)/1e3:.1f} | "
              f"{min(d)/1e3:.1f} | {max(d)/1e3:.1f} |")


if __name__ == "__main__":
    main()
