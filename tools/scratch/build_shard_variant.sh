#!/bin/bash
# build_shard_variant.sh NAME [extra hipcc flags...] -> ac-solver_amd/lib/var_NAME.so with only acx_shard.hip recompiled
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
cd "$ROOT/ac-solver_amd/csrc"
F=${ACX_BASEFLAGS:-"--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=14"}
mkdir -p /tmp/acx_var
/opt/rocm/bin/hipcc $F "$@" -c acx_shard.hip -o /tmp/acx_var/acx_shard_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib/var_$name.so acx_step.o acx_search.o /tmp/acx_var/acx_shard_$name.o acx_ball.o acx_simplex.o acx_policy.o
echo built ../lib/var_$name.so
