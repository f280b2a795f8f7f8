mkdir -p gpurun_out/many
R=$GRAFT_REPO_ROOT
for mode in fused multi fused multi; do echo "== $mode" >> $R/gpurun_out/many/ab2.log; ACX_BFS_MANY=$mode python3 tools/ms_sweep_warm.py bfs 1e6 2>&1 | tail -2 >> $R/gpurun_out/many/ab2.log; done
cat $R/gpurun_out/many/ab2.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/many/prof -o sweep -- python3 $R/tools/ms_sweep_warm.py bfs 1e6 > $R/gpurun_out/many/prof.log 2>&1
cd $R
find gpurun_out/many/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} head -20 {} | cut -c1-200
grep "bfs run" gpurun_out/many/prof.log
find gpurun_out/many/prof -name "*kernel_trace.csv" -delete
