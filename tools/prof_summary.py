#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (tools/profile_gpu.sh) into the summary that is
committed under profiles/: per-kernel stats of the engine's kernels + PMC counters per launch."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src = sys.argv[1]
out = sys.argv[2]
os.makedirs(out, exist_ok=True)
OURS = ("pair_kernel", "finalize_kernel", "pack_kernel", "counts_kernel")


def short(name):
    for k in OURS:
        if k in name:
            if k == "pair_kernel":
                a = name.index("pair_kernel<") + len("pair_kernel<")
                return "pair_kernel<" + name[a:name.index(">", a)].replace("dst::(anonymous namespace)::", "") + ">"
            return k
    return None


lines = []
for path in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    rows = list(csv.DictReader(open(path)))
    lines.append("# rocprofv3 --kernel-trace --stats (bench.py default workload, 1 warmup + 3 steps)")
    lines.append(f"{'kernel':60s} {'calls':>6s} {'total_ms':>12s} {'avg_ms':>12s} {'min_ms':>12s} {'max_ms':>12s} {'pct':>7s}")
    for r in rows:
        s = short(r["Name"])
        if s:
            lines.append(f"{s:60s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:12.3f} {float(r['AverageNs'])/1e6:12.4f} "
                         f"{float(r['MinNs'])/1e6:12.4f} {float(r['MaxNs'])/1e6:12.4f} {float(r['Percentage']):7.2f}")
    other = sum(float(r["TotalDurationNs"]) for r in rows if not short(r["Name"]))
    lines.append(f"{'(torch data-generation kernels, outside the timed region)':60s} {'':>6s} {other/1e6:12.3f}")
    with open(os.path.join(out, "kernel_stats_full.csv"), "w") as fh:
        fh.write(open(path).read())
lines.append("")
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for path in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        acc = defaultdict(lambda: defaultdict(list))
        dur = {}
        meta = {}
        for r in csv.DictReader(open(path)):
            s = short(r["Kernel_Name"])
            if not s:
                continue
            acc[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[(s, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            meta[s] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"],
                       r["Grid_Size"], r["Workgroup_Size"])
        lines.append(f"# rocprofv3 --pmc ({os.path.basename(d)}; bench.py --workload C2: 10000 x 30000 raw), mean per launch")
        for s in acc:
            ms = [v for (k, _), v in dur.items() if k == s]
            m = meta[s]
            lines.append(f"{s}: launches={len(ms)} avg_ms={sum(ms)/len(ms):.4f} vgpr={m[0]} agpr={m[1]} sgpr={m[2]} "
                         f"lds={m[3]} scratch={m[4]} grid={m[5]} wg={m[6]}")
            for c, vals in acc[s].items():
                lines.append(f"    {c:28s} {sum(vals)/len(vals):18.3f}")
# per-launch HBM traffic of the engine's kernels -> traffic.json (read by bench.py)
traffic = []
for tag, wl in (("pmc", "C2"), ("pmcdef", "C3raw")):
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for path in glob.glob(os.path.join(src, f"{tag}_{c}", "*", "*_counter_collection.csv")):
            per = defaultdict(list)
            for r in csv.DictReader(open(path)):
                s = short(r["Kernel_Name"])
                if s and r["Counter_Name"] == c:
                    per[s].append(float(r["Counter_Value"]))
            for s, v in per.items():
                vals.setdefault(s, {})[c] = sum(v) / len(v)
    for s, v in vals.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            traffic.append({"workload": wl, "variant": 0, "kernel": s.split("<")[0], "kernel_full": s,
                            "FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"],
                            "hbm_bytes_per_launch": v["FETCH_SIZE"] * 1024 * 2 + v["WRITE_SIZE"] * 1024,
                            "source": f"{os.path.basename(out.rstrip('/'))}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in "
                                      "separate passes, mean per launch; FETCH_SIZE x2 (gfx950 16-B/lane "
                                      "streaming-read under-count), KiB -> bytes"})
if traffic:
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    lines.append("# HBM traffic per launch (traffic.json): " + "; ".join(
        f"{t['workload']} {t['kernel_full']}: {t['hbm_bytes_per_launch']/1e9:.3f} GB" for t in traffic))
lines.append("")
for path in sorted(glob.glob(os.path.join(src, "*_bench.json"))):
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
        lines.append(f"# {os.path.basename(path)}: value={d['value']:.4e} pairs/s ms_per_step={d['ms_per_step']:.3f} "
                     f"pair_ms={d['kernels_ms']['pair']:.3f} (under the profiler)")
    except Exception as e:  # noqa
        lines.append(f"# {os.path.basename(path)}: unreadable ({e})")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
