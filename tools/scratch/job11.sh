mkdir -p gpurun_out/r4_16
timeout 600 python tools/shard_bench.py 1e8 21 > gpurun_out/r4_16/shard.log 2>&1
timeout 300 python tools/bfs_only.py 1e8 > gpurun_out/r4_16/bfs.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_search.py tests/test_gpu_determinism.py tests/test_gpu_search_fuzz.py -m gpu -x -q -k "not sharded and not greedy" > gpurun_out/r4_16/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4_16/tests.log
python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r4_16/bench.json 2> gpurun_out/r4_16/bench.err
grep -v amdgpu gpurun_out/r4_16/shard.log gpurun_out/r4_16/bfs.log; tail -3 gpurun_out/r4_16/tests.log
