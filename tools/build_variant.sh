#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...] -> ac-solver_amd/lib/var_NAME.so (A/B experiments; select with ACX_LIB=...)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
cd "$ROOT/ac-solver_amd/csrc"
F=${ACX_BASEFLAGS:-"--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=14"}
mkdir -p /tmp/acx_var
for f in acx_step acx_search acx_shard acx_ball acx_simplex acx_policy; do /opt/rocm/bin/hipcc $F "$@" -c $f.hip -o /tmp/acx_var/${f}_$name.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib/var_$name.so /tmp/acx_var/acx_step_$name.o /tmp/acx_var/acx_search_$name.o /tmp/acx_var/acx_shard_$name.o /tmp/acx_var/acx_ball_$name.o /tmp/acx_var/acx_simplex_$name.o /tmp/acx_var/acx_policy_$name.o
echo built ../lib/var_$name.so
