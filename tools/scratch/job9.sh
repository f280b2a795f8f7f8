mkdir -p gpurun_out/r4_10
ACX_DEBUG=1 timeout 300 python tools/greedy_only.py 1e7 3 > gpurun_out/r4_10/greedy.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_search.py tests/test_gpu_sweeps.py tests/test_gpu_search_fuzz.py tests/test_gpu_determinism.py -m gpu -x -q -k "not sharded" > gpurun_out/r4_10/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4_10/tests.log
ACX_LIB=$GRAFT_REPO_ROOT/ac-solver_amd/lib/var_gprof.so ACX_DEBUG=1 timeout 300 python tools/greedy_only.py 1e7 1 > gpurun_out/r4_10/greedy_prof.log 2>&1
timeout 300 python tools/ms_sweep_warm.py greedy 1e6 > gpurun_out/r4_10/gsweep.log 2>&1

tail -4 gpurun_out/r4_10/tests.log; grep -E "nodes/s|status=" gpurun_out/r4_10/greedy.log; grep acx_greedy gpurun_out/r4_10/greedy_prof.log | tail -7; cat gpurun_out/r4_10/gsweep.log
