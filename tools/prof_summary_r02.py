#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (tools/profile_r02.sh) into profiles/<tag>/: per-kernel rocprofv3 stats of the
engine's kernels, PMC counters per launch, and traffic.json (HBM bytes per launch that bench.py replays as
roofline.traffic).   python tools/prof_summary_r02.py gpurun_out/prof_r02 profiles/r02"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)
OURS = ("consensus_pair_kernel", "pair_kernel", "finalize_kernel", "pack_kernel", "counts_kernel", "index_kernel",
        "ref_sample_kernel", "hot_list_kernel", "site_bucket_kernel", "slot_fill_kernel", "derive_kernel", "sum2_u32_kernel", "add_u32_kernel", "aconst_kernel",
        "scan_block_kernel", "scan_add_kernel", "scan_small_kernel", "scan_mid_kernel", "list_totals_kernel", "planes_from_slots_kernel",
        "range_marks_kernel", "compact_kernel", "report_kernel", "chunk_sums_kernel", "corr_mfma_kernel",
        "run_masks_kernel", "run_known_kernel", "run_panels_kernel", "pack_nibbles_kernel", "number_kernel", "line_kernel")


def short(name):
    for k in OURS:
        if k in name:
            m = re.search(k + r"<(.*?)>\(", name)
            return k + ("<" + m.group(1).replace("dst::(anonymous namespace)::", "") + ">" if m else "")
    return None


lines = []
for path in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    rows = list(csv.DictReader(open(path)))
    lines.append("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline")
    lines.append("# (default workload 50,000 x 30,000: main leg raw/auto, then legs dense raw, tn93 auto, tn93 dense, clades, nruns; the verify passes' dense slabs included)")
    lines.append(f"{'kernel':62s} {'calls':>6s} {'total_ms':>11s} {'avg_ms':>11s} {'min_ms':>11s} {'max_ms':>11s}")
    for r in rows:
        s = short(r["Name"])
        if s:
            lines.append(f"{s:62s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:11.3f} {float(r['AverageNs'])/1e6:11.4f} "
                         f"{float(r['MinNs'])/1e6:11.4f} {float(r['MaxNs'])/1e6:11.4f}")
    with open(os.path.join(out, "kernel_stats_full.csv"), "w") as fh:
        fh.write(open(path).read())
lines.append("")
traffic = defaultdict(dict)
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for path in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        acc = defaultdict(lambda: defaultdict(list))
        dur = defaultdict(list)
        for r in csv.DictReader(open(path)):
            s = short(r["Kernel_Name"])
            if not s:
                continue
            acc[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[s].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        lines.append(f"# rocprofv3 --pmc ({os.path.basename(d)}), mean per launch [min .. max]")
        for s in acc:
            for c, v in acc[s].items():
                lines.append(f"{s:62s} {c:24s} {sum(v)/len(v):18.1f}  [{min(v):.1f} .. {max(v):.1f}]  n={len(v)}  avg_ms={sum(dur[s])/len(dur[s]):.3f}")
                if c in ("FETCH_SIZE", "WRITE_SIZE"):
                    traffic[s][c] = (sum(v) / len(v), min(v), max(v))
        lines.append("")
recs = []
names = {"consensus_pair_kernel<1,": ("C3raw", "consensus_pair_kernel"), "pair_kernel<MRaw": ("C3raw", "pair_kernel"),
         "consensus_pair_kernel<3,": ("C3", "consensus_pair_kernel"), "pair_kernel<MTN93": ("C3", "pair_kernel"),
         "pack_kernel": ("C3raw", "pack_kernel")}
for s, t in traffic.items():
    key = next((k for k in names if s.startswith(k)), None)
    if "FETCH_SIZE" in t and "WRITE_SIZE" in t and key:
        wl, k = names[key]
        f, w = t["FETCH_SIZE"], t["WRITE_SIZE"]
        # FETCH_SIZE x2 for wide coalesced streaming reads (MI355X_MICROARCH.md, HBM); "other access widths are
        # uncalibrated: calibrate on a known byte count in your own access pattern": the pack reads one 128-byte line per
        # LANE (eight 16-byte loads of its own row), and against its known input (n x L bytes) the counter reads x1
        fx = 1 if k == "pack_kernel" else 2
        why = ("FETCH_SIZE x1: the pack's reads are one 128-byte line per lane, not a wave-wide coalesced stream — calibrated against "
               "its known input (the n x L byte matrix), which the counter reports undoubled") if fx == 1 else \
              ("FETCH_SIZE x2 (gfx950 16-B/lane streaming-read under-count; the consensus kernel's reads are mostly 8/16-byte gathers, "
               "for which the x2 is an upper bound)")
        recs.append({"workload": wl, "kernel": k, "kernel_full": s, "FETCH_SIZE_KiB": f[0], "WRITE_SIZE_KiB": w[0],
                     "hbm_bytes_per_launch": f[0] * 1024 * fx + w[0] * 1024,
                     "hbm_bytes_min_max": [f[1] * 1024 * fx + w[1] * 1024, f[2] * 1024 * fx + w[2] * 1024],
                     "source": f"{os.path.basename(out)}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, mean per launch "
                               f"(min-max of the launches {(f[1]*fx+w[1])*1024/1e9:.2f}-{(f[2]*fx+w[2])*1024/1e9:.2f} GB); {why}, KiB -> bytes"})
json.dump(recs, open(os.path.join(out, "traffic.json"), "w"), indent=1)
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
