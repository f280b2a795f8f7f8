#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...] -> ac-solver_amd/lib/var_NAME.so (A/B experiments; select with ACX_LIB=...)
# Goes through csrc/Makefile, so a variant gets the same flags and the same 64-bit-shift check as libacx.so
# (SHIFT64_CHECK=0 in the environment for the deliberate reproducer builds).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
make -s -j6 -C "$ROOT/ac-solver_amd/csrc" OUT=../lib/var_$name.so BUILD=/tmp/acx_var/$name EXTRA="$*" SHIFT64_CHECK=${SHIFT64_CHECK:-1}
echo built ../lib/var_$name.so
