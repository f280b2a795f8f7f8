mkdir -p gpurun_out/r4_7
ACX_LIB=$GRAFT_REPO_ROOT/ac-solver_amd/lib/var_gprof.so ACX_DEBUG=1 timeout 300 python tools/greedy_only.py 1e7 2 > gpurun_out/r4_7/greedy_prof.log 2>&1
python bench.py > gpurun_out/r4_7/bench.json 2> gpurun_out/r4_7/bench.err
grep acx_greedy gpurun_out/r4_7/greedy_prof.log | tail -8
