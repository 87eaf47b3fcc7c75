#!/usr/bin/env python3
"""Summarise build/asm/resource_usage.txt (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "build/asm/resource_usage.txt"
txt = open(path).read()
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    m = re.search(r"pair_kernelINS0_(\w+?)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)", name)
    short = f"{m.group(1)} BM={m.group(2)} TN={m.group(3)} W={m.group(4)} INT={m.group(5)}" if m else name[:70]

    def g(key):
        mm = re.search(re.escape(key) + r": (\d+)", b)
        return mm.group(1) if mm else "?"

    print(f"{short:44s} SGPR {g('TotalSGPRs'):>4s} VGPR {g('VGPRs'):>4s} AGPR {g('AGPRs'):>3s} "
          f"scratch {g('ScratchSize [bytes/lane]'):>4s} occ {g('Occupancy [waves/SIMD]')} "
          f"sspill {g('SGPRs Spill')} vspill {g('VGPRs Spill')}")
