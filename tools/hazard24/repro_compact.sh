#!/bin/bash
# repro_compact.sh -- the round-2 sighting of the register-allocation hazard, reproducible in seconds.
# Builds libacx with k_bfs_compact as it was in commit 57c6f83, where the fault was found (-DACX_HAZARD_REPRO: the write loop
# of that commit, 2048-candidate tiles, and a clobber of v31 in place of the register pad: with the normal-form cyclical move
# code the kernel then uses and declares exactly 32 vector registers) and runs the same BFS five times:
#   on the GPU box:  bash tools/hazard24/repro_compact.sh
# Expected: the repro library returns a different (nodes, expanded) pair on every run and never the oracle's; the shipped
# library (same source, 48 registers declared for that kernel) returns the oracle's numbers every time.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT"
[ -f ac-solver_amd/lib/var_hazard.so ] || bash tools/build_search_variant.sh hazard -DACX_HAZARD_REPRO -DACX_COMPACT_ITEMS=8
echo "== shipped library"
python3 tools/hazard24/debug_repeat.py 145 1e6 1 5
echo "== k_bfs_compact as in commit 57c6f83: exactly its 32 registers declared"
ACX_LIB=ac-solver_amd/lib/var_hazard.so python3 tools/hazard24/debug_repeat.py 145 1e6 1 5
