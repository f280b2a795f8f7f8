mkdir -p gpurun_out/many
L=gpurun_out/many/ab3.log
for rep in 1 2; do
for w in 1 2 3 7; do echo "== workers $w" >> $L; ACX_SWEEP_WORKERS=$w python3 tools/ms_sweep_warm.py bfs 1e6 2>&1 | tail -2 >> $L; done
for ld in 1.5 1.2; do echo "== workers 1 load $ld" >> $L; ACX_BFS_MANY_LOAD=$ld ACX_SWEEP_WORKERS=1 python3 tools/ms_sweep_warm.py bfs 1e6 2>&1 | tail -2 >> $L; done
for b in 8192 16384 65536; do echo "== workers 1 bmax $b" >> $L; ACX_BFS_MANY_BMAX=$b ACX_SWEEP_WORKERS=1 python3 tools/ms_sweep_warm.py bfs 1e6 2>&1 | tail -2 >> $L; done
echo "== workers 1 bmax 16384 load 1.5" >> $L; ACX_BFS_MANY_LOAD=1.5 ACX_BFS_MANY_BMAX=16384 ACX_SWEEP_WORKERS=1 python3 tools/ms_sweep_warm.py bfs 1e6 2>&1 | tail -2 >> $L
done
cat $L
