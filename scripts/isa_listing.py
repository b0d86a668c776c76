#!/usr/bin/env python3
"""Annotated ISA listing of a kernel's loop bodies: every instruction with its class and the issue cost used by the VALU roofline
(scripts/valu_model.py), and per-basic-block sums.   usage: isa_listing.py <workload of valu_model.KERNELS> <out.txt>"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import valu_model as V  # noqa: E402

w, dst = sys.argv[1], sys.argv[2]
tu, flags, name = V.KERNELS[w]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(V.CSRC, "..", "..", "include"), *flags,
       "-S", "--cuda-device-only", "-o", "-", os.path.join(V.CSRC, tu)]
asm = subprocess.run(cmd, capture_output=True, text=True, check=True).stdout
out, infn, depth, label = [], False, 0, "entry"
blocks = {}
for line in asm.splitlines():
    s = line.strip()
    m = re.match(r"^(_Z\w+):", s)
    if m:
        infn = name in m.group(1)
        if infn:
            out.append(f"== {m.group(1)}")
        continue
    if not infn:
        continue
    if s.startswith(".Lfunc_end"):
        break
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        label = m.group(1)
        d = re.search(r"Depth=(\d+)", s)
        depth = int(d.group(1)) if d else 0
        out.append(f"{label}:   ; loop depth {depth}")
        continue
    d = re.search(r"Loop Header: Depth=(\d+)", s)
    if d:
        depth = int(d.group(1))
    if not s or s.startswith((";", ".")):
        continue
    mn = s.split()[0]
    c = V.classify(mn, s)
    kind = c if c else ("lds" if mn.startswith("ds_") else "vmem" if mn.startswith(("global_", "scratch_", "buffer_")) else "salu/other")
    cost = V.COST.get(c, 0.0)
    b = blocks.setdefault(label, {"depth": depth, "full": 0, "vop3": 0, "wide": 0, "lds": 0, "vmem": 0, "cycles": 0.0})
    b["depth"] = depth
    if c:
        b[c] += 1
        b["cycles"] += cost
    elif kind in ("lds", "vmem"):
        b[kind] += 1
    out.append(f"    {s.split(';')[0].rstrip():<78} ; {kind:<10} {cost:.1f}" if c else f"    {s.split(';')[0].rstrip():<78} ; {kind}")
out.append("")
out.append("== per basic block: VALU instructions by class, LDS / VMEM instructions, VALU issue cycles (2.0 / 3.5 / 4.0 per class)")
for lab, b in blocks.items():
    n = b["full"] + b["vop3"] + b["wide"]
    if n >= 20:
        out.append(f"{lab:<12} depth {b['depth']}  valu {n:4d} (full {b['full']:4d} vop3 {b['vop3']:4d} wide {b['wide']:3d})  lds {b['lds']:3d}  vmem {b['vmem']:2d}  issue cycles {b['cycles']:7.1f}")
open(dst, "w").write("\n".join(out) + "\n")
print("\n".join(out[-40:]))
