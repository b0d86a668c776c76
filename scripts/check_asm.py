#!/usr/bin/env python3
"""Two static checks on the hand-written asm of bio_amd/csrc (run by tests/test_abi_and_host.py; CPU only).

1. Every inline-asm block that contains an SALU instruction writing SCC (s_and/s_or/s_xor/s_add/s_sub/s_lshl/s_lshr/s_bcnt/s_cmp ...)
   names "scc" in its clobber list.  (Round 2: a tie chain without the clobber lost whole reads' tuples in 6 % of the fuzz cases --
   the compiler kept a loop condition in SCC across the block.)
2. The loads k_minimizer_pk / k_syncmer_pk issue from inline asm are invisible to the compiler's s_waitcnt pass (on purpose, kernels_pk.hpp).  In the
   generated ISA, between such a load and the hand-written `s_waitcnt vmcnt(0)` that follows it in the text, no instruction may
   mention the load's destination registers: a copy or a use placed there would read registers whose data has not arrived.
   (A linear scan of the text: conservative, it knows nothing of the control flow.)
"""
import re
import subprocess
import sys
import os

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "bio_amd", "csrc")
SCC_WRITERS = re.compile(r"\bs_(and|or|xor|andn2|orn2|nand|nor|xnor|add|sub|addc|subb|lshl|lshr|ashr|bcnt\d|cmp|bitcmp|min|max|abs|not|wqm|quadmask|bfe|mul_hi)\w*")


def asm_blocks(text):
    """Yield (line, body) of every asm(...) / asm volatile(...) statement (balanced parentheses)."""
    for m in re.finditer(r"\basm\s*(volatile)?\s*\(", text):
        i, depth = m.end(), 1
        while depth and i < len(text):
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        yield text.count("\n", 0, m.start()) + 1, text[m.end():i - 1]


def check_scc():
    bad = []
    for fn in sorted(os.listdir(CSRC)):
        if not fn.endswith((".hpp", ".hip", ".cpp")):
            continue
        text = open(os.path.join(CSRC, fn)).read()
        for line, body in asm_blocks(text):
            strings = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', body))
            if SCC_WRITERS.search(strings) and '"scc"' not in body:
                bad.append(f"{fn}:{line}: SALU op that writes SCC without an \"scc\" clobber")
    return bad


def regs(operand_text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", operand_text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", operand_text):
        out.add(int(a))
    return out


def check_hidden_loads(ws=("11",), unit="k_minimizer_pk", macro="BSK_PK_WS", extra=()):
    bad = []
    for w in ws:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(CSRC, "..", "..", "include"),
               f"-D{macro}(X)=X({w})", *extra, "-S", "--cuda-device-only", "-o", "-", os.path.join(CSRC, unit + ".hip")]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            return [f"hipcc failed for w={w}: {p.stderr[-400:]}"]
        pending, in_asm, func = {}, False, "?"
        for n, line in enumerate(p.stdout.splitlines(), 1):
            s = line.strip()
            if re.match(r"^_Z\w+:", s):
                func, pending = s.split(":")[0], {}
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not s or s.startswith((";", ".")) or s.endswith(":"):
                continue
            code = s.split(";")[0]
            if in_asm:
                if code.startswith("global_load") and not code.startswith("global_load_lds"):  # (LDS-DMA has no VGPR destination)
                    dst = code.split()[1].rstrip(",")
                    for r in regs(dst):
                        pending[r] = n
                elif code.startswith("s_waitcnt") and "vmcnt(0)" in code:
                    pending = {}
                continue
            if code.startswith("s_endpgm"):
                pending = {}
            hit = regs(code) & set(pending)
            if hit:
                bad.append(f"w={w} {func} line {n}: `{code.strip()}` touches v{sorted(hit)} while the asm load of line {pending[min(hit)]} is in flight")
                for r in hit:
                    pending.pop(r)
    return bad


def check_reserved(ws=("11",), unit="k_minimizer_ring", macro="BSK_RING_WS", first=144, func_filter="k_minimizer_ring"):
    """3. k_minimizer_ring keeps v144.. out of the register allocator's hands (amdgpu_num_vgpr) and loads into them from inline asm:
    outside asm blocks no instruction of those kernels may name such a register."""
    bad = []
    for w in ws:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(CSRC, "..", "..", "include"),
               f"-D{macro}(X)=X({w})", "-S", "--cuda-device-only", "-o", "-", os.path.join(CSRC, unit + ".hip")]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            return [f"hipcc failed for w={w}: {p.stderr[-400:]}"]
        in_asm, func = False, "?"
        for n, line in enumerate(p.stdout.splitlines(), 1):
            s = line.strip()
            if re.match(r"^_Z\w+:", s):
                func = s.split(":")[0]
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if in_asm or func_filter not in func or not s or s.startswith((";", ".")) or s.endswith(":"):
                continue
            hit = [r for r in regs(s.split(";")[0]) if r >= first]
            if hit:
                bad.append(f"w={w} {func} line {n}: `{s.split(';')[0].strip()}` names reserved v{sorted(hit)}")
    return bad


PK_WS = tuple(str(w) for w in range(2, 14))       # BSK_PK_WS, BSK_RING_WS
SYNPK_WS = tuple(str(w) for w in range(4, 25))    # BSK_SYNPK_WS (4..20: k_syncmer_pk and k_syncmer_pkl), BSK_SYNPKL_WS (21..24: k_syncmer_pkl only)


def run_all(pk_ws, syn_ws, ring_ws, jobs=None):
    """Every (kernel family, width) is one hipcc -S run: in parallel."""
    from concurrent.futures import ThreadPoolExecutor
    tasks = [lambda w=w: check_hidden_loads((w,)) for w in pk_ws]
    tasks += [lambda w=w: check_hidden_loads((w,), "k_syncmer_pk", "BSK_SYNPKL_WS", ("-DBSK_SYNPK_WS(X)=" + ("X(%s)" % w if int(w) <= 20 else ""),)) for w in syn_ws]
    # the fused-emit syncmer kernels (round 6): k_syncmer_pf (k - s = 8..20, LDS-DMA: nothing to check but the build) and k_syncmer_pfl (8..24: the
    # register form of the prefetch, hand-placed waits as k_syncmer_pkl)
    tasks += [lambda w=w: check_hidden_loads((w,), "k_syncmer_pf", "BSK_SYNPFL_WS", ("-DBSK_SYNPF_WS(X)=" + ("X(%s)" % w if int(w) <= 20 else ""),)) for w in syn_ws if int(w) >= 8]
    tasks += [lambda w=w: check_hidden_loads((w,), "k_minimizer_ring", "BSK_RING_WS") + check_reserved((w,)) for w in ring_ws]
    with ThreadPoolExecutor(max_workers=jobs or max(2, (os.cpu_count() or 4))) as ex:
        res = list(ex.map(lambda f: f(), tasks))
    return [e for r in res for e in r]


if __name__ == "__main__":
    # usage: check_asm.py            -> w = 11 (pk, ring), k - s = 20 (syncmer)
    #        check_asm.py 2 11 13    -> those pk widths (+ ring 11, syncmer 20)
    #        check_asm.py all        -> every instantiation the library ships (pk / ring 2..13, syncmer 4..24)
    args = sys.argv[1:]
    if args == ["all"]:
        errs = check_scc() + run_all(PK_WS, SYNPK_WS, PK_WS)
    else:
        errs = check_scc() + run_all(tuple(args) or ("11",), ("20",), ("11",))
    print("\n".join(errs) if errs else "asm checks: ok (%s)" % (" ".join(args) or "default widths"))
    sys.exit(1 if errs else 0)
