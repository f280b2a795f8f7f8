#!/bin/bash
# repro_compact.sh -- the round-2 sighting of the register-allocation hazard, reproducible in seconds.
# Builds libacx with k_bfs_compact<u64, normal-form cyclical move> declaring EXACTLY the 32 vector registers its code uses
# (-DACX_HAZARD_REPRO replaces the kernel's ACX_VGPR_PAD by a clobber of v31) and runs the same BFS five times:
#   on the GPU box:  bash tools/hazard24/repro_compact.sh
# Expected: the repro library returns a different (nodes, expanded) pair on every run and never the oracle's; the shipped
# library (same source, 48 registers declared for that kernel) returns the oracle's numbers every time.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT"
[ -f ac-solver_amd/lib/var_hazard.so ] || bash tools/build_search_variant.sh hazard -DACX_HAZARD_REPRO
echo "== shipped library"
python3 tools/debug_repeat.py 145 1e6 1 5
echo "== k_bfs_compact with exactly its 32 registers declared"
ACX_LIB=ac-solver_amd/lib/var_hazard.so python3 tools/debug_repeat.py 145 1e6 1 5
