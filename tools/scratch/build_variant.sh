#!/bin/bash
# tools/scratch/build_variant.sh NAME [-DFLAG ...]: ac-solver_amd/lib/var_NAME.so = libacx.so with acx_shard.hip rebuilt under the flags
# (run with ACX_LIB=ac-solver_amd/lib/var_NAME.so).  The other objects come from the last `make`.
set -e
cd "$(dirname "$0")/../../ac-solver_amd/csrc"
name=$1; shift
src=${ACX_VARIANT_SRC:-acx_shard}
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=14"
/opt/rocm/bin/hipcc $F "$@" -c $src.hip -o /tmp/var_$name.o
objs=""
for o in acx_step acx_search acx_shard acx_ball acx_simplex acx_policy; do
  if [ $o = $src ]; then objs="$objs /tmp/var_$name.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib/var_$name.so $objs
