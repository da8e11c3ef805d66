#!/usr/bin/env python3
"""Median duration of each kernel of the timed UCC-en hybrid step from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace -d DIR -o kt --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
    python3 scripts/step_kernels.py DIR
Only launches with >= 10 000 workgroup threads x queries are listed (the 37 376-query step)."""
import csv
import glob
import statistics
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = defaultdict(list)
for r in csv.DictReader(open(f)):
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    if grid < 500000:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("amdr::", "")
    d[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for (name, grid), v in sorted(d.items(), key=lambda kv: -statistics.median(kv[1])):
    if len(v) < 50:
        continue
    med = statistics.median(v)
    tot += med
    print(f"{name:44s} grid {grid:9d} n={len(v):4d} median {med:8.1f} us  min {min(v):8.1f}")
print(f"sum of medians {tot:.1f} us")
