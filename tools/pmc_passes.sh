#!/bin/bash
# pmc_passes.sh OUTDIR CMD...  -- the rocprofv3 counter passes behind profiles/*_pmc_summary.txt, one counter group per run
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; no tracing options next to --pmc).  Run on the GPU box:
#   cd /tmp && export TMPDIR=/tmp && bash $GRAFT_REPO_ROOT/tools/pmc_passes.sh $GRAFT_REPO_ROOT/gpurun_out/pmc python3 $GRAFT_REPO_ROOT/tools/bfs_only.py 1e8
# then  python3 tools/pmc_sum.py OUTDIR  prints per-kernel SUMS over all dispatches.
out=$1; shift
mkdir -p "$out"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    rocprofv3 --pmc $grp --output-format csv -d "$out/pass$i" -- "$@" > "$out/pass$i.log" 2>&1 || echo "pass $i failed (see $out/pass$i.log)"
done
