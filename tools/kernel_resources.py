#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of a .hip translation unit, from hipcc's own resource remarks
(-Rpass-analysis=kernel-resource-usage; device-only compile, no GPU needed).

    python3 tools/kernel_resources.py [acx_search.hip ...]        # default: every TU of libacx.so

Also the build-time guard of DESIGN.md's 24-VGPR hazard: tests/test_abi_cpu.py calls `resources()` and fails when a kernel
that inlines the packed-word move code is built with fewer than 32 VGPRs."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ac-solver_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-kernarg-preload-count=14", "--cuda-device-only", "-c",
         "-Rpass-analysis=kernel-resource-usage", "-o", os.devnull]
KEYS = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
        "LDS Size [bytes/block]": "lds"}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", re.sub(r"^void ", "", o)).replace("acx::", "") for o in out[:len(names)]]


def resources(tu, extra=()):
    """-> {kernel name: {"vgpr": .., "sgpr": .., "scratch": .., "occupancy": .., "lds": ..}} for one .hip file of csrc/;
    `extra`: more compiler flags (e.g. -DACX_NO_VGPR_PAD: the kernels' own register need, without the pads)"""
    p = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + list(extra) + [os.path.join(CSRC, tu)], capture_output=True, text=True, cwd=CSRC)
    if p.returncode != 0:
        raise RuntimeError(p.stderr[-2000:])
    res, cur = {}, None
    for line in p.stderr.split("\n"):
        m = re.search(r"remark: (?:\s*)([A-Za-z \[\]/]+): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = res.setdefault(v, {})
        elif cur is not None and k in KEYS:
            cur[KEYS[k]] = int(v)
    names = list(res)
    return dict(zip(demangle(names), (res[n] for n in names)))


if __name__ == "__main__":
    tus = sys.argv[1:] or ["acx_step.hip", "acx_search.hip", "acx_shard.hip", "acx_ball.hip", "acx_simplex.hip"]
    for tu in tus:
        print(f"== {tu}")
        for name, r in sorted(resources(tu).items()):
            print(f"  {name:70s} vgpr {r.get('vgpr', 0):3d}  sgpr {r.get('sgpr', 0):3d}  scratch {r.get('scratch', 0):4d}  lds {r.get('lds', 0):6d}  occupancy {r.get('occupancy', 0)}")
