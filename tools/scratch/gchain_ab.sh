mkdir -p gpurun_out/gchain
timeout 1500 python -m pytest tests/test_gpu_search.py tests/test_gpu_sweeps.py -m gpu -x -q -k "greedy or fixtures or config3 or L25 or determin" > gpurun_out/gchain/tests4.log 2>&1; tail -3 gpurun_out/gchain/tests4.log
timeout 300 python3 tools/greedy_only.py 1e7 3 2>&1 | tail -2
