#!/usr/bin/env python3
"""Static VALU class mix of the sketch kernels' inner loops, from the ISA hipcc generates (CPU only).

    python scripts/valu_model.py [out.json]

For every bench workload's kernel: compile its translation unit to gfx950 assembly, take the instructions of the kernel's basic
blocks that sit in loops of depth >= 2 (the per-block / per-trip code: what a unit executes hundreds of times), and class every
VALU instruction by the issue cost measured in scripts/ubench (cycles one wave-instruction occupies a SIMD with >= 2 waves on it):

    full   2.0   VOP1 / VOP2 encodings (_e32) and v_bitop3_b32
    vop3   3.5   everything else in VOP3 / VOPC-e64 / DPP / SDWA form (v_alignbit, v_cndmask_e64, v_cmp_*_e64, v_and_or, v_bfe, v_mad_i24 ...)
    wide   4.0   64-bit and multiplier ops (v_mad_u64_u32, v_lshl_add_u64, v_mul_lo/hi, v_lshlrev_b64 ...)

cycles_per_inst = the mix's mean.  profile_post.py multiplies it with the measured SQ_INSTS_VALU of the same kernel to get the
VALU-issue roofline fraction (a static mix applied to a dynamic count: an approximation, stated as such in the bench line).
"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "bio_amd", "csrc")
COST = {"full": 2.0, "vop3": 3.5, "wide": 4.0}
WIDE = re.compile(r"^v_(mad_u64_u32|mad_i64_i32|lshl_add_u64|mul_lo_u32|mul_hi_u32|mul_hi_i32|lshlrev_b64|lshrrev_b64|ashrrev_i64|add_f64|mul_f64|fma_f64)")

# workload -> (translation unit, -D flags, substring of the kernel's mangled name)
KERNELS = {
    "minimizer": ("k_minimizer_pk.hip", ["-DBSK_PK_WS(X)=X(11)"], "k_minimizer_pkILi11ELb0"),
    "minimizer400": ("k_minimizer_pkd.hip", ["-DBSK_PKD_WS(X)=X(11)"], "k_minimizer_pkdILi11"),
    "minimizer250": ("k_minimizer_ring.hip", ["-DBSK_RING_WS(X)=X(11)"], "k_minimizer_ringILi11ELi3ELb1"),
    "syncmer": ("k_syncmer_pf.hip", ["-DBSK_SYNPF_WS(X)=X(20)", "-DBSK_SYNPFL_WS(X)="], "k_syncmer_pfILi20"),
    "syncmer250": ("k_syncmer_pf.hip", ["-DBSK_SYNPF_WS(X)=", "-DBSK_SYNPFL_WS(X)=X(20)"], "k_syncmer_pflILi20"),
    "nthash": ("launch.hip", [], "k_nthash_fastILi1"),
    "kmer": ("launch.hip", [], "k_nthash_fastILi2"),
    "simhash": ("launch.hip", [], "k_simhash_fastILi5ELi12"),
    "protmin": ("k_protein.hip", [], "k_prot_minimizer_fastILi5ELi9ELb0"),
    "prothash": ("k_protein.hip", [], "k_prot_hash_fastILi9ELb0"),
}


def classify(mn, line):
    if not mn.startswith("v_"):
        return None
    if WIDE.match(mn):
        return "wide"
    if mn.startswith("v_bitop3"):
        return "full"
    if mn.endswith("_e32") and "dpp" not in line and "sdwa" not in line:
        return "full"
    return "vop3"


def model(tu, flags, name):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(CSRC, "..", "..", "include"), *flags,
           "-S", "--cuda-device-only", "-o", "-", os.path.join(CSRC, tu)]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode:
        raise RuntimeError(p.stderr[-500:])
    cnt = {"full": 0, "vop3": 0, "wide": 0}
    other = {"lds": 0, "vmem": 0, "salu": 0}
    infn, depth = False, 0
    for line in p.stdout.splitlines():
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            infn, depth = name in m.group(1), 0
            continue
        if not infn:
            continue
        if s.startswith(".Lfunc_end"):
            infn = False
            continue
        if re.match(r"^\.LBB\d+_\d+:", s):
            d = re.search(r"Depth=(\d+)", s)
            depth = int(d.group(1)) if d else 0
            continue
        d = re.search(r"Loop Header: Depth=(\d+)", s)
        if d:
            depth = int(d.group(1))
        if depth < 2 or not s or s.startswith((";", ".")):
            continue
        mn = s.split()[0]
        c = classify(mn, s)
        if c:
            cnt[c] += 1
        elif mn.startswith("ds_"):
            other["lds"] += 1
        elif mn.startswith(("global_", "scratch_", "buffer_", "flat_")):
            other["vmem"] += 1
        elif mn.startswith("s_") and not mn.startswith(("s_waitcnt", "s_nop")):
            other["salu"] += 1
    n = sum(cnt.values())
    if not n:
        raise RuntimeError("kernel %s not found in %s" % (name, tu))
    return {"valu_static": cnt, "other_static": other, "vop3_frac": round((cnt["vop3"] + cnt["wide"]) / n, 4),
            "cycles_per_inst": round(sum(cnt[k] * COST[k] for k in cnt) / n, 4), "kernel_symbol_contains": name, "translation_unit": tu}


if __name__ == "__main__":
    out = {"note": __doc__.split("\n\n")[2].strip(), "cost_cycles": COST, "kernels": {}}
    for w, (tu, flags, name) in KERNELS.items():
        try:
            out["kernels"][w] = model(tu, flags, name)
        except Exception as e:  # a kernel that was renamed must not cost the others
            out["kernels"][w] = {"error": str(e)}
        print(w, out["kernels"][w])
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(HERE), "profiles", "valu_model.json")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    json.dump(out, open(dst, "w"), indent=1)
