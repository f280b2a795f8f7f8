#!/bin/bash
# refresh_profiles_r3.sh -- on the GPU box: the round-3 search evidence behind profiles/r3_{shard,bfs,greedy}_*.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# sharded BFS at world 1, 1e8 nodes (tools/shard_bench.py: one warm-up + three timed searches; the fused search runs twice in front)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/shard_kt -- python3 $R/tools/shard_bench.py 1e8 21 > $O/shard_kt.log 2>&1
bash $R/tools/pmc_passes.sh $O/shard_pmc python3 $R/tools/shard_bench.py 1e8 21
python3 $R/tools/pmc_sum.py $O/shard_pmc > $O/shard_pmc_summary.txt 2>&1
# greedy_search AK(3) 1e7 (two searches)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/greedy_kt -- python3 $R/tools/greedy_only.py 1e7 2 > $O/greedy_kt.log 2>&1
bash $R/tools/pmc_passes.sh $O/greedy_pmc python3 $R/tools/greedy_only.py 1e7 2
python3 $R/tools/pmc_sum.py $O/greedy_pmc > $O/greedy_pmc_summary.txt 2>&1
# fused BFS 1e8 (two searches)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bfs_kt -- python3 $R/tools/bfs_only.py 1e8 > $O/bfs_kt.log 2>&1
bash $R/tools/pmc_passes.sh $O/bfs_pmc python3 $R/tools/bfs_only.py 1e8
python3 $R/tools/pmc_sum.py $O/bfs_pmc > $O/bfs_pmc_summary.txt 2>&1
find $O -name "*.csv" -size +8M -delete
find $O -name "*kernel_trace.csv" -delete
ls -la $O
# the Miller-Schupp sweeps (BASELINE config 4 shape on one GPU): k_bfs_multi / k_greedy_multi
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sweep_bfs_kt -- python3 $R/tools/ms_sweep.py bfs 1e6 16 1 together > $O/sweep_bfs.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sweep_greedy_kt -- python3 $R/tools/ms_sweep.py greedy 1e6 16 0 together > $O/sweep_greedy.log 2>&1
find $O -name "*kernel_trace.csv" -delete
