mkdir -p gpurun_out/gchain
export ACX_LIB=$GRAFT_REPO_ROOT/ac-solver_amd/lib/var_rank2k.so
ACX_DEBUG=1 timeout 300 python3 tools/greedy_only.py 1e7 1 2>&1 | grep -E "nodes/s|hand-offs|status=" | tail -4
timeout 900 python -m pytest tests/test_gpu_search.py -m gpu -x -q -k "greedy or fixtures or config3 or L25 or determin" 2>&1 | tail -3
