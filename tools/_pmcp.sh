cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pol_pmc2; rm -rf $O; mkdir -p $O
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES" "SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/p$i -- python3 $GRAFT_REPO_ROOT/tools/policy_only.py > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/p$i.log; }
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $O k_policy 2>&1 | head -60
