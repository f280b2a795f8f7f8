mkdir -p gpurun_out/r4_6
timeout 1200 python -m pytest tests/test_gpu_sweeps.py tests/test_gpu_search.py tests/test_gpu_ppo.py -m gpu -x -q > gpurun_out/r4_6/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4_6/tests.log
timeout 300 python tools/ms_sweep_warm.py bfs 1e6 > gpurun_out/r4_6/sweep.log 2>&1
ACX_SWEEP_STAGGER_MS=0 timeout 300 python tools/ms_sweep_warm.py bfs 1e6 > gpurun_out/r4_6/sweep_nostagger.log 2>&1
timeout 300 python tools/ms_sweep_warm.py greedy 1e6 > gpurun_out/r4_6/gsweep.log 2>&1
ACX_PPO_PHASES=1 timeout 900 python tools/train_probe.py 32 4 > gpurun_out/r4_6/train.log 2>&1
bash tools/profile_r4.sh sweep gsweep > gpurun_out/r4_6/prof.log 2>&1
tail -3 gpurun_out/r4_6/tests.log; cat gpurun_out/r4_6/sweep.log gpurun_out/r4_6/sweep_nostagger.log gpurun_out/r4_6/gsweep.log; grep -E "ppo phases|per update" gpurun_out/r4_6/train.log
