mkdir -p gpurun_out/r4_5
for v in libacx var_bmB var_bmC; do echo "== $v" >> gpurun_out/r4_5/sweep.log; ACX_LIB=$GRAFT_REPO_ROOT/ac-solver_amd/lib/$v.so timeout 300 python tools/ms_sweep_warm.py bfs 1e6 >> gpurun_out/r4_5/sweep.log 2>&1; done
timeout 900 python -m pytest tests/test_gpu_search.py -m gpu -x -q -k "sharded or host_side or rccl or routing" > gpurun_out/r4_5/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4_5/tests.log
timeout 600 python tools/shard_bench.py 1e8 21 > gpurun_out/r4_5/shard.log 2>&1
python bench.py --no-search --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r4_5/bench_env.json 2> gpurun_out/r4_5/bench_env.err
ACX_PPO_PHASES=1 timeout 900 python tools/train_probe.py 32 4 > gpurun_out/r4_5/train.log 2>&1
cat gpurun_out/r4_5/sweep.log; tail -3 gpurun_out/r4_5/tests.log; cat gpurun_out/r4_5/shard.log; grep -E "ppo phases|per update" gpurun_out/r4_5/train.log
