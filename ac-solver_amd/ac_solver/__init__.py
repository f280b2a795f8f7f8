"""ac_solver -- MI355X-native drop-in for the Andrews-Curtis hot path of shehper/AC-Solver.

Same import paths and names as the reference package (ac_solver/__init__.py:1-6) for the
environment and the searches; the PPO trainer (`train_ppo`) is the *caller* of this path and is
out of scope here -- the reference's own ac_solver/agents package runs
unchanged on top of `ac_solver.envs`.
"""
from ac_solver.envs.ac_env import ACEnv, ACEnvConfig
from ac_solver.search.breadth_first import bfs
from ac_solver.search.greedy import greedy_search

__all__ = ["ACEnv", "ACEnvConfig", "bfs", "greedy_search"]
