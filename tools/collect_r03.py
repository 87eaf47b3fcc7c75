#!/usr/bin/env python3
"""gpurun_out/prof_r03* (tools/profile_r03.sh) -> profiles/r03/: summary.txt (default workload, main leg), summary_tn93.txt,
summary_all_legs.txt, kernel_stats_full.csv, traffic.json (what bench.py replays as roofline.traffic), the bench line
printed under rocprofv3.   python tools/collect_r03.py"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles", "r03")
HEADS = {
    "prof_r03": ("summary.txt",
                 "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra   (tools/profile_r03.sh)",
                 "# (default workload 50,000 x 30,000 raw on the default path; the two pair_kernel<MRaw> launches are the verify pass: dense slabs, untimed)"),
    "prof_r03_tn93": ("summary_tn93.txt",
                      "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload C3 --no-extra   (tools/profile_r03.sh)",
                      "# (50,000 x 30,000 tn93 on the default path; the two pair_kernel<MTN93> launches are the verify pass: dense slabs, untimed)"),
    "prof_r03_full": ("summary_all_legs.txt",
                      "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline   (tools/profile_r03.sh)",
                      "# (every leg of the default run: raw auto, dense, tn93 auto, tn93 dense, clades (hybrid), nruns (run records); kernels of one "
                      "name are averaged over the legs that use them)"),
}


def main():
    os.makedirs(OUT, exist_ok=True)
    traffic = []
    for name, (target, h0, h1) in HEADS.items():
        src = os.path.join(ROOT, "gpurun_out", name)
        if not os.path.isdir(src):
            print("missing", src)
            continue
        with tempfile.TemporaryDirectory() as tmp:
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary_r02.py"), src, tmp], check=True,
                           stdout=subprocess.DEVNULL)
            lines = open(os.path.join(tmp, "summary.txt")).read().split("\n")
            lines[0], lines[1] = h0, h1
            open(os.path.join(OUT, target), "w").write("\n".join(lines))
            if name == "prof_r03":
                shutil.copy(os.path.join(tmp, "kernel_stats_full.csv"), os.path.join(OUT, "kernel_stats_full.csv"))
                shutil.copy(os.path.join(src, "trace_bench.json"), os.path.join(OUT, "bench_under_rocprof.json"))
            tj = os.path.join(tmp, "traffic.json")
            if name != "prof_r03_full" and os.path.exists(tj):
                want = "C3raw" if name == "prof_r03" else "C3"
                for r in json.load(open(tj)):
                    if r["workload"] == want and ", -5>" not in r["kernel_full"]:   # (-5: the hybrid's tally launch, not a leg)
                        r["source"] = r["source"].replace(os.path.basename(tmp), "r03")
                        traffic.append(r)
    json.dump(traffic, open(os.path.join(OUT, "traffic.json"), "w"), indent=1)
    for r in traffic:
        print(f"{r['workload']:6s} {r['kernel_full']:44s} {r['hbm_bytes_per_launch'] / 1e9:8.3f} GB per launch")


if __name__ == "__main__":
    main()
