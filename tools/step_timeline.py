#!/usr/bin/env python3
"""One step of a bench.py run as a timeline, from a rocprofv3 --kernel-trace csv: start (us from the step's first
kernel), duration, gap to the previous kernel's end, kernel name.  python tools/step_timeline.py <dir> [step index from the end]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
path = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda k: k.replace("void ", "").replace("dst::(anonymous namespace)::", "").split("(")[0][:70]
# a step starts at every ref_sample (or pack when there is none)
marks = [i for i, r in enumerate(rows) if "ref_sample" in r["Kernel_Name"]] or [i for i, r in enumerate(rows) if "pack_kernel" in r["Kernel_Name"]]
a = marks[-back]
b = marks[-back + 1] if back > 1 else len(rows)
t0 = int(rows[a]["Start_Timestamp"])
prev = None
print(f"# {path}: kernels {a}..{b} of {len(rows)}")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {gap:7.1f}  {short(r['Kernel_Name'])}")
    prev = e
print(f"# step span {(prev - t0) / 1e3:.1f} us; next step starts {((int(rows[b]['Start_Timestamp']) - prev) / 1e3) if b < len(rows) else float('nan'):.1f} us later")
