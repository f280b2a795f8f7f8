#!/bin/bash
# profile_r6.sh WHAT... -- on the GPU box: round-6 evidence behind profiles/r6_*.  WHAT: shard | greedy | bfs | sweep | gsweep | policy
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for what in "$@"; do
  cmd=
  case $what in
    shard)   cmd="python3 $R/tools/shard_bench.py 1e8 21" ;;
    greedy)  cmd="python3 $R/tools/greedy_only.py 1e7 2" ;;
    bfs)     cmd="python3 $R/tools/bfs_only.py 1e8" ;;
    sweep)   cmd="python3 $R/tools/ms_sweep.py bfs 1e6 16 1 together" ;;
    gsweep)  cmd="python3 $R/tools/ms_sweep.py greedy 1e6 16 0 together" ;;
    policy)  cmd="python3 $R/tools/policy_only.py" ;;
    *)       echo "unknown target $what" >&2; continue ;;
  esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${what}_kt -- $cmd > $O/${what}_kt.log 2>&1
  bash $R/tools/pmc_passes.sh $O/${what}_pmc $cmd
  python3 $R/tools/pmc_sum.py $O/${what}_pmc > $O/${what}_pmc_summary.txt 2>&1
done
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -delete
ls -la $O
# tracked copies: gpurun_out/r6prof/profiles/r6_* (copy them to profiles/)
mkdir -p $O/profiles
for what in "$@"; do
  case $what in shard) n=shard_1e8 ;; greedy) n=greedy_1e7 ;; bfs) n=bfs_1e8 ;; sweep) n=ms_sweep_bfs ;; gsweep) n=ms_sweep_greedy ;; policy) n=policy ;; esac
  cp $(find $O/${what}_kt -name "*kernel_stats.csv" | head -1) $O/profiles/r6_${n}_kernel_stats.csv
  cp $O/${what}_pmc_summary.txt $O/profiles/r6_${n}_pmc_summary.txt
done
