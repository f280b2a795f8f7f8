mkdir -p gpurun_out/r4_17
timeout 300 python tools/ms_sweep_warm.py bfs 1e6 > gpurun_out/r4_17/sweep.log 2>&1
ACX_LIB=$GRAFT_REPO_ROOT/ac-solver_amd/lib/var_nopf.so timeout 300 python tools/ms_sweep_warm.py bfs 1e6 > gpurun_out/r4_17/sweep_nopf.log 2>&1
timeout 300 python tools/ms_sweep_warm.py bfs 1e6 >> gpurun_out/r4_17/sweep.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_search.py tests/test_gpu_sweeps.py tests/test_gpu_search_fuzz.py tests/test_gpu_determinism.py -m gpu -x -q -k "config4 or many or sweep or fuzz or miller or determin or fixtures" > gpurun_out/r4_17/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4_17/tests.log
grep -v amdgpu gpurun_out/r4_17/sweep.log; echo nopf; grep -v amdgpu gpurun_out/r4_17/sweep_nopf.log; tail -3 gpurun_out/r4_17/tests.log
