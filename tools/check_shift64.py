#!/usr/bin/env python3
"""Static guard against the gfx950 64-bit-shift fault (probe: tools/hazard24/run_shift64_probe.sh).

On MI355X `v_lshlrev_b64`, `v_lshrrev_b64` and `v_ashrrev_i64` return wrong results when the per-lane shift amount sits in the
LAST vector register of the wave's allocation (allocation = next_free_vgpr rounded up to 8) and other waves are resident on the
SIMD.  LLVM knows the fault as `hasShift64HighRegBug` and works around it in GCNHazardRecognizer::fixShift64HighRegBug, but
enables that only for gfx90a (`GFX90AInsts && !GFX940Insts`); ROCm 7.2 therefore emits such shifts for gfx950 unguarded.

This tool compiles a translation unit of csrc/ to assembly (device only, no GPU needed) and lists every 64-bit shift whose
amount register is the last register of its kernel's allocation.  tests/test_abi_cpu.py requires the list to be empty for every
kernel of libacx.so (the kernels keep it empty by declaring a few registers more than they use: ACX_VGPR_PAD, acx_common.h).

    python3 tools/check_shift64.py [acx_search.hip ...] [-- extra hipcc flags]
    python3 tools/check_shift64.py --asm file.s ...      (csrc/Makefile runs this on the assembly of every object it links: the build fails on a hit)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ac-solver_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-kernarg-preload-count=14", "--cuda-device-only", "-S"]
SHIFT = re.compile(r"^\s*(v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64)\s+v\[\d+:\d+\],\s*(\S+?),")


def scan_asm(text):
    """-> ({kernel: (next_free_vgpr, accum_offset or None)}, [(function, opcode, amount operand)]) for one assembly file"""
    nfree, sites, cur, in_desc = {}, [], None, None
    for line in text.split("\n"):
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m and not line.startswith(".L"):
            cur = m.group(1)
            continue
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            in_desc = m.group(1)
            nfree[in_desc] = [0, None]
            continue
        if in_desc:
            m = re.match(r"\s*\.amdhsa_next_free_vgpr\s+(\d+)", line)
            if m:
                nfree[in_desc][0] = int(m.group(1))
            m = re.match(r"\s*\.amdhsa_accum_offset\s+(\d+)", line)
            if m:
                nfree[in_desc][1] = int(m.group(1))
            if ".end_amdhsa_kernel" in line:
                in_desc = None
            continue
        m = SHIFT.match(line)
        if m and cur:
            sites.append((cur, m.group(1), m.group(2)))
    return {k: tuple(v) for k, v in nfree.items()}, sites


def function_bodies(text):
    """{symbol: body text} of every function of an assembly file (kernels and the functions they call)"""
    bodies, cur, buf = {}, None, []
    for line in text.split("\n"):
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m and not line.startswith(".L"):
            if cur:
                bodies[cur] = "\n".join(buf)
            cur, buf = m.group(1), []
            continue
        if cur:
            buf.append(line)
    if cur:
        bodies[cur] = "\n".join(buf)
    return bodies


def risky_sites(text):
    """The 64-bit shifts whose VGPR amount is the last register of an allocation block of the kernel they are in: the last register
    of the whole allocation (next_free_vgpr rounded up to the granule of 8) and, in a kernel that also holds accumulation registers
    (the unified file of gfx90a+: the architectural VGPRs end at accum_offset, the AGPRs follow), the last architectural one.  A
    shift with a VGPR amount in a function that is NOT a kernel (the micro-batch of acx_greedy.h) runs inside the allocation of
    the kernel that calls it: it is checked against the allocation of EVERY kernel whose code mentions the function's symbol,
    directly or through another such function; a function nobody seems to call is held to the rule that covers any caller (the
    amount must not be the last register of a granule of 8)."""
    nfree, sites = scan_asm(text)

    def tops_of(kernel):
        nf, acc = nfree[kernel]
        tops = {(nf + 7) // 8 * 8 - 1}
        if acc is not None and acc < nf:  # AGPRs in use: the architectural registers are v0 .. v(accum_offset - 1)
            tops.add(acc - 1)
        return tops

    callee_sites = [s for s in sites if s[0] not in nfree and re.fullmatch(r"v(\d+)", s[2])]
    callers = {}
    if callee_sites:
        bodies = function_bodies(text)
        callees = {s[0] for s in callee_sites}
        mentions = {f: {g for g, body in bodies.items() if g != f and f in body} for f in callees}
        for f in callees:
            seen, todo, ks = set(), [f], set()
            while todo:
                x = todo.pop()
                for g in mentions.get(x, {h for h, body in bodies.items() if h != x and x in body}):
                    if g in nfree:
                        ks.add(g)
                    elif g not in seen:
                        seen.add(g)
                        todo.append(g)
            callers[f] = ks
    bad = []
    for fn, op, amt in sites:
        m = re.fullmatch(r"v(\d+)", amt)
        if not m:
            continue  # an SGPR / literal amount
        idx = int(m.group(1))
        if fn in nfree:
            if idx in tops_of(fn):
                bad.append((fn, op, amt, nfree[fn][0]))
        elif callers.get(fn):
            for k in callers[fn]:
                if idx in tops_of(k):
                    bad.append((fn, op, amt, nfree[k][0]))
                    break
        elif idx % 8 == 7:
            bad.append((fn, op, amt, -1))
    return bad, {k: v[0] for k, v in nfree.items()}, sites


def compile_to_asm(tu, extra=()):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "tu.s")
        p = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + list(extra) + ["-o", out, os.path.join(CSRC, tu)], capture_output=True, text=True, cwd=CSRC)
        if p.returncode != 0:
            raise RuntimeError(p.stderr[-2000:])
        return open(out).read()


def report(name, text):
    bad, nfree, sites = risky_sites(text)
    demangle = subprocess.run(["c++filt"], input="\n".join(b[0] for b in bad), capture_output=True, text=True).stdout.split("\n")
    print(f"== {name}: {len(nfree)} kernels, {len(sites)} 64-bit shifts with a register amount checked, {len(bad)} at the top of an allocation")
    for (fn, op, amt, nf), dn in zip(bad, demangle):
        print(f"   {op} amount {amt} with next_free_vgpr {nf}: {dn[:140]}" if nf >= 0 else f"   {op} amount {amt} in a non-kernel function, in the last register of a granule of 8 (could be the last of its caller's allocation): {dn[:140]}")
    return len(bad)


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--asm":  # csrc/Makefile: the device assembly of the object that is about to be linked (-save-temps)
        sys.exit(1 if sum(report(os.path.basename(f), open(f).read()) for f in args[1:]) else 0)
    extra = []
    if "--" in args:
        k = args.index("--")
        args, extra = args[:k], args[k + 1:]
    tus = args or ["acx_step.hip", "acx_search.hip", "acx_shard.hip", "acx_ball.hip", "acx_simplex.hip", "acx_policy.hip"]
    total = 0
    for tu in tus:
        total += report(tu, compile_to_asm(tu, extra + (["-fno-slp-vectorize"] if tu == "acx_policy.hip" else [])))
    sys.exit(1 if total else 0)
