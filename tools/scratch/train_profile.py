"""cProfile of two PPO updates at the config-5 per-GPU shape (where does the host time of an update go?)"""
import cProfile, os, pstats, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import torch
from ac_solver.agents.ppo import train_ppo
os.chdir(tempfile.mkdtemp())
N, T = 131072, 32
args = ["--num-envs", str(N), "--num-steps", str(T), "--tile-initial-states", "--fused-policy", "--horizon-length", "200", "--num-minibatches", "4"]
train_ppo(args + ["--total-timesteps", str(2 * T * N)])
pr = cProfile.Profile()
pr.enable()
train_ppo(args + ["--total-timesteps", str(4 * T * N)])
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
