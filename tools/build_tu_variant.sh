#!/bin/bash
# build_tu_variant.sh TU NAME [extra hipcc flags...] -> ac-solver_amd/lib/var_NAME.so with only acx_TU.hip recompiled with the extra
# flags (the other translation units come from the current build); select with ACX_LIB=...  The recompiled object goes through
# csrc/Makefile's rule, i.e. through the same 64-bit-shift check as the objects of libacx.so.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
tu=$1; name=$2; shift 2
cd "$ROOT/ac-solver_amd/csrc"
make -s -j6 >/dev/null
B=/tmp/acx_var/${tu}_$name
mkdir -p $B
make -s BUILD=$B EXTRA="$*" SHIFT64_CHECK=${SHIFT64_CHECK:-1} $B/acx_$tu.o
objs=""
for f in step search search_greedy search_many shard shard_run ball simplex policy; do
  if [ $f = $tu ]; then objs="$objs $B/acx_$f.o"; else objs="$objs _build/acx_$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib/var_$name.so $objs -ldl
echo built ../lib/var_$name.so
