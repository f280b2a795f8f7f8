"""Builds tests/hostshim/libacx_hostshim.so with the ROCm clang (host only).  Test infrastructure."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libacx_hostshim.so")
SRC = os.path.join(HERE, "acx_hostshim.cpp")
CSRC = os.path.join(HERE, "..", "..", "ac-solver_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build():
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("acx_word.h", "acx_bytes.h", "acx_keys.h")]
    if os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(d) for d in deps):
        return SO
    subprocess.check_call([CLANG, "-O2", "-std=c++17", "-fPIC", "-shared", "-o", SO, SRC])
    return SO
