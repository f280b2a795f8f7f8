#!/bin/bash
# run_shift64_probe.sh -- on the GPU box: bash tools/hazard24/run_shift64_probe.sh
# Builds the probe with the shift amount in the LAST register of the wave's allocation (v7 of 8, v15 of 16, v23 of 24, v31 of 32)
# and, same instructions, with one more register declared (the allocation grows by a granule of 8), for the three 64-bit shifts.
set -e
cd "$(dirname "$0")"
B=/tmp/shift64; mkdir -p $B
LLVM=/opt/rocm/lib/llvm/bin
/opt/rocm/bin/hipcc -O2 -o $B/probe shift64_probe.cpp
objs=""
for op in v_lshrrev_b64 v_lshlrev_b64 v_ashrrev_i64; do
  for amt in 7 15 23 31; do
    for extra in 0 1; do
      nv=$((amt + 1 + extra)); acc=$(( (nv + 3) / 4 * 4 ))
      name=$B/${op}_amt${amt}_nvgpr${nv}
      sed -e "s/@OP@/$op/g" -e "s/@AMT@/$amt/g" -e "s/@NVGPR@/$nv/g" -e "s/@ACC@/$acc/g" shift64_probe.s.in > $name.s
      $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $name.s -o $name.o
      $LLVM/ld.lld -shared $name.o -o $name.hsaco
      objs="$objs $name.hsaco"
    done
  done
done
# control: the amount in a register that is NOT the last of a granule, allocation exactly filled (v6 under test as well)
$B/probe $objs
