#!/usr/bin/env python3
"""Static guard against the gfx950 64-bit-shift fault (probe: tools/hazard24/run_shift64_probe.sh).

On MI355X `v_lshlrev_b64`, `v_lshrrev_b64` and `v_ashrrev_i64` return wrong results when the per-lane shift amount sits in the
LAST vector register of the wave's allocation (allocation = next_free_vgpr rounded up to 8) and other waves are resident on the
SIMD.  LLVM knows the fault as `hasShift64HighRegBug` and works around it in GCNHazardRecognizer::fixShift64HighRegBug, but
enables that only for gfx90a (`GFX90AInsts && !GFX940Insts`); ROCm 7.2 therefore emits such shifts for gfx950 unguarded.

This tool compiles a translation unit of csrc/ to assembly (device only, no GPU needed) and lists every 64-bit shift whose
amount register is the last register of its kernel's allocation.  tests/test_abi_cpu.py requires the list to be empty for every
kernel of libacx.so (the kernels keep it empty by declaring a few registers more than they use: ACX_VGPR_PAD, acx_common.h).

    python3 tools/check_shift64.py [acx_search.hip ...] [-- extra hipcc flags]"""
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
    """-> ({kernel: next_free_vgpr}, [(function, opcode, amount operand)]) for one assembly file"""
    nfree, sites, cur, in_desc = {}, [], None, None
    for line in text.split("\n"):
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m and not line.startswith(".L"):
            cur = m.group(1)
            continue
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            in_desc = m.group(1)
            continue
        if in_desc:
            m = re.match(r"\s*\.amdhsa_next_free_vgpr\s+(\d+)", line)
            if m:
                nfree[in_desc] = int(m.group(1))
            if ".end_amdhsa_kernel" in line:
                in_desc = None
            continue
        m = SHIFT.match(line)
        if m and cur:
            sites.append((cur, m.group(1), m.group(2)))
    return nfree, sites


def risky_sites(text):
    """the 64-bit shifts whose VGPR amount is the last register of the allocation of the kernel they are in"""
    nfree, sites = scan_asm(text)
    bad = []
    for fn, op, amt in sites:
        m = re.fullmatch(r"v(\d+)", amt)
        if not m or fn not in nfree:
            continue  # an SGPR / literal amount, or a non-kernel function (everything is inlined in this library)
        alloc = (nfree[fn] + 7) // 8 * 8
        if int(m.group(1)) == alloc - 1:
            bad.append((fn, op, amt, nfree[fn]))
    return bad, nfree, sites


def compile_to_asm(tu, extra=()):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "tu.s")
        p = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + list(extra) + ["-o", out, os.path.join(CSRC, tu)], capture_output=True, text=True, cwd=CSRC)
        if p.returncode != 0:
            raise RuntimeError(p.stderr[-2000:])
        return open(out).read()


if __name__ == "__main__":
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        k = args.index("--")
        args, extra = args[:k], args[k + 1:]
    tus = args or ["acx_step.hip", "acx_search.hip", "acx_shard.hip", "acx_ball.hip", "acx_simplex.hip", "acx_policy.hip"]
    total = 0
    for tu in tus:
        bad, nfree, sites = risky_sites(compile_to_asm(tu, extra + (["-fno-slp-vectorize"] if tu == "acx_policy.hip" else [])))
        demangle = subprocess.run(["c++filt"], input="\n".join(b[0] for b in bad), capture_output=True, text=True).stdout.split("\n")
        print(f"== {tu}: {len(nfree)} kernels, {len(sites)} 64-bit shifts with a register amount checked, {len(bad)} at the top of an allocation")
        for (fn, op, amt, nf), name in zip(bad, demangle):
            print(f"   {op} amount {amt} with next_free_vgpr {nf}: {name[:140]}")
        total += len(bad)
    sys.exit(1 if total else 0)
