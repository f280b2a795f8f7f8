"""Environment construction for the PPO trainer (reference: ac_solver/agents/environment.py:19-127).

`get_env` returns an `ACVecEnv` -- every environment stepped by one HIP kernel launch, rewards clipped inside the
kernel -- in place of `gym.vector.SyncVectorEnv([make_env(...)])`; the other five return values are the reference's.
`make_env` keeps the reference's thunk for a single `ACEnv` (wrapped with gymnasium's NormalizeReward /
TransformReward when gymnasium is installed; a minimal clip wrapper otherwise).
"""
import numpy as np

from ac_solver._gym import HAVE_GYMNASIUM
from ac_solver.agents.utils import load_initial_states_from_text_file
from ac_solver.envs.ac_env import ACEnv, ACEnvConfig
from ac_solver.envs.utils import change_max_relator_length_of_presentation, convert_relators_to_presentation


class TransformReward:
    """The one thing gymnasium.wrappers.TransformReward does here: reward -> f(reward)."""

    def __init__(self, env, f):
        self.env, self.f = env, f

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        return obs, self.f(reward), terminated, truncated, info

    def __getattr__(self, name):
        return getattr(self.env, name)


def make_env(presentation, args):
    def thunk():
        config = ACEnvConfig.from_dict({"initial_state": presentation, "horizon_length": args.horizon_length,
                                        "use_supermoves": args.use_supermoves})
        env = ACEnv(config)
        if args.norm_rewards:
            if not HAVE_GYMNASIUM:
                raise NotImplementedError("--norm-rewards needs gymnasium's NormalizeReward wrapper")
            import gymnasium as gym

            env = gym.wrappers.NormalizeReward(env, gamma=args.gamma)
        if args.clip_rewards:
            assert args.min_rew < args.max_rew, "min_rew must be less than max_rew"
            clip = lambda reward: np.clip(reward, args.min_rew, args.max_rew)  # noqa: E731
            if HAVE_GYMNASIUM:
                import gymnasium as gym

                env = gym.wrappers.TransformReward(env, clip)
            else:
                env = TransformReward(env, clip)
        return env

    return thunk


def get_env(args, device=None, rank=0, world=1):
    """-> (envs, initial_states, curr_states, success_record, ACMoves_hist, states_processed)
    One process per GPU (`rank` of `world`): the Miller-Schupp states are dealt rank::world, environment i of rank r
    starts from state r + i * world, so the ranks together cover num_envs * world distinct states."""
    from ac_solver.envs.vec_env import ACVecEnv

    if args.use_supermoves:
        raise NotImplementedError("ACEnv with supermoves is not yet implemented.")  # ac_env.py:62-65
    if args.fixed_init_state:
        presentation = convert_relators_to_presentation(args.relator1, args.relator2, args.max_relator_length)
        initial_states = [presentation]
        rows = np.repeat(np.asarray(presentation, np.int8)[None], args.num_envs, axis=0)
        curr_states = [0] * args.num_envs
    else:
        initial_states = load_initial_states_from_text_file(states_type=args.states_type)
        n_states = len(initial_states)
        if not getattr(args, "tile_initial_states", False):  # the reference's limit (environment.py:80-83)
            assert args.num_envs * world <= n_states, \
                "Expect number of environments to be less than number of distinct initial states for now"
        args.max_relator_length = 36  # max(4n + 2) over 1 <= n <= 7 (environment.py:87)
        initial_states = [change_max_relator_length_of_presentation(s, args.max_relator_length) for s in initial_states]
        # --tile-initial-states: more environments than states (BASELINE config 5: 131 072 per GPU) start from state i mod n
        curr_states = [(rank + i * world) % n_states for i in range(args.num_envs)]
        rows = np.asarray(initial_states, np.int8)[curr_states]
    clip = None
    if args.clip_rewards:
        assert args.min_rew < args.max_rew, "min_rew must be less than max_rew"
        if not args.norm_rewards:  # with --norm-rewards the clip follows the normalisation (training loop), as make_env stacks the wrappers
            clip = (args.min_rew, args.max_rew)
    supermoves = [[int(a) for a in q.split(",")] for q in getattr(args, "supermoves", "").split(";") if q.strip()] or None
    # with the fused policy kernel the rollout keeps int8 observation rows (a quarter of the bytes written and read per step);
    # the update widens a minibatch to f32 when it needs it (training.py)
    envs = ACVecEnv(rows, horizon_length=args.horizon_length, obs_dtype="int8" if getattr(args, "fused_policy", False) else "float32", clip_rewards=clip, record_actions=True,
                    final_info=False, device=device, supermoves=supermoves)
    states_processed = set(curr_states)
    success_record = {"solved": set(), "unsolved": set(range(len(initial_states)))}
    ACMoves_hist = {}
    return envs, initial_states, curr_states, success_record, ACMoves_hist, states_processed
