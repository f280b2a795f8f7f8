mkdir -p gpurun_out/soak
for seed in 12 13 14 15 16 17 18 19 20 21 22 23; do ACX_FUZZ_SEED=$seed timeout 900 python -m pytest tests/test_gpu_search_fuzz.py -m gpu -x -q 2>&1 | tail -1; done
for sl in 1 3 7 33; do ACX_GREEDY_SLOTS=$sl timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_sweeps.py -m gpu -x -q -k "many or greedy_sweep or paths_file or groups" 2>&1 | tail -1; done
python tools/scratch/greedy_soak.py 5 40 2>&1 | tail -1
timeout 900 python -m ac_solver.agents.ppo --num-envs 8192 --tile-initial-states --fused-policy --num-steps 32 --total-timesteps 10485760 --horizon-length 200 2>&1 | tail -2
