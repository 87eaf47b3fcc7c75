#!/usr/bin/env python3
"""f64 instructions of one finalisation, counted in the gfx950 ISA (no GPU needed: hipcc cross-compiles).

    python tools/count_f64_ops.py            # table + JSON on the last line

tools/f64_ops/fin_probe.hip instantiates finalize_pair<measure, close> (distance_amd/csrc/dst_device.hpp) alone in a
kernel; this script compiles it to assembly and counts, in each kernel's own body (the out-of-line fall-backs —
arguments of a logarithm further than 2^-5 from 1, zero denominators — are separate functions and not counted: they do
not run on the alignments the consensus path serves), the v_*_f64 instructions, the f32 reciprocals behind the
divisions, and everything else.  bench.py reads profiles/r03/f64_ops.json (this script's output) for its f64 roofline."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {2: "raw", 3: "jc69", 4: "k80", 5: "tn93"}


def main():
    src = os.path.join(ROOT, "tools", "f64_ops", "fin_probe.hip")
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "probe.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-x", "hip",
                        "--cuda-device-only", "-S", src, "-o", asm], check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    out = {}
    # a kernel's body: from its label to its s_endpgm
    for m in re.finditer(r"^(_ZN3dst5probeILi(\d)ELb([01])EEEvPKjPK15HIP_vector_typeIjLj4EEPd):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        measure, close, body = NAMES[int(m.group(2))], m.group(3) == "1", m.group(4)
        ins = [l.split()[0] for l in body.splitlines() if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
        f64 = [i for i in ins if re.match(r"v_\w+_f64", i) or i in ("v_cvt_f64_u32_e32", "v_cvt_f64_f32_e32", "v_cvt_f32_f64_e32")]
        rec = {"f64": len(f64), "rcp_f32": sum(i.startswith("v_rcp_f32") for i in ins), "rcp_f64": sum(i.startswith("v_rcp_f64") for i in ins),
               "div_scale_fmas_fixup": sum(i.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup")) for i in ins),
               "calls": sum(i.startswith("s_swappc") for i in ins), "all_instructions": len(ins)}
        out.setdefault(measure, {})["close" if close else "epilogue"] = rec
    print(f"{'measure':8s} {'arithmetic':10s} {'f64':>5s} {'rcp_f32':>8s} {'rcp_f64':>8s} {'div seq':>8s} {'calls':>6s} {'all':>6s}")
    for measure in ("raw", "jc69", "k80", "tn93"):
        for kind, rec in sorted(out.get(measure, {}).items()):
            print(f"{measure:8s} {kind:10s} {rec['f64']:5d} {rec['rcp_f32']:8d} {rec['rcp_f64']:8d} {rec['div_scale_fmas_fixup']:8d} "
                  f"{rec['calls']:6d} {rec['all_instructions']:6d}")
    print(json.dumps(out))


if __name__ == "__main__":
    sys.exit(main())
