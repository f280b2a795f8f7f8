"""ac_solver -- MI355X-native drop-in for the Andrews-Curtis hot path of shehper/AC-Solver.

Same import paths and names as the reference package (ac_solver/__init__.py:1-6): the environment, the two
searches and `train_ppo`.  The environment step and the search frontiers run as HIP kernels behind the C ABI of
libacx (include/acx.h); `train_ppo` (ac_solver/agents) is this build's PPO loop on the device-resident
environments and is imported lazily, so that `import ac_solver` does not pull in torch.
"""
from ac_solver.envs.ac_env import ACEnv, ACEnvConfig
from ac_solver.search.breadth_first import bfs
from ac_solver.search.greedy import greedy_search

__all__ = ["ACEnv", "ACEnvConfig", "bfs", "greedy_search", "train_ppo"]


def __getattr__(name):
    if name == "train_ppo":
        from ac_solver.agents.ppo import train_ppo

        return train_ppo
    raise AttributeError(f"module 'ac_solver' has no attribute {name!r}")
