# many_ab.sh: the level-synchronous multi-search BFS against round 3's one-workgroup-per-search kernel: tests, then sweep times
mkdir -p gpurun_out/many
python -m pytest tests/test_gpu_search.py tests/test_gpu_sweeps.py -m gpu -x -q -k "many or config4 or bfs_sweep" > gpurun_out/many/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/many/tests.log
tail -15 gpurun_out/many/tests.log
for rep in 1 2; do
  for mode in fused multi; do
    echo "== $mode (rep $rep)" >> gpurun_out/many/sweep.log
    ACX_BFS_MANY=$mode ACX_DEBUG=1 timeout 300 python tools/ms_sweep_warm.py bfs 1e6 2>&1 | grep -v amdgpu | tail -12 >> gpurun_out/many/sweep.log
  done
done
for b in 16384 65536; do
  echo "== fused bmax $b" >> gpurun_out/many/sweep.log
  ACX_BFS_MANY_BMAX=$b timeout 300 python tools/ms_sweep_warm.py bfs 1e6 2>&1 | grep -v amdgpu | tail -2 >> gpurun_out/many/sweep.log
done
cat gpurun_out/many/sweep.log
