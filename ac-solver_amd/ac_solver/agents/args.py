"""Command-line arguments of the PPO trainer: the flags, defaults and derived fields of the reference's
ac_solver/agents/args.py:10-296 (`distutils.strtobool` replaced: it left the standard library in Python 3.12)."""
import argparse

# (flag, type, default, help); "bool" flags follow the reference's `--flag [true|false]` convention (nargs="?", const=True)
_FLAGS = [
    ("--exp-name", str, "args", "the name of this experiment"),
    ("--seed", int, 1, "seed of the experiment"),
    ("--torch-deterministic", "bool", True, "if toggled, `torch.backends.cudnn.deterministic=False`"),
    ("--cuda", "bool", True, "if toggled, cuda will be enabled by default"),
    ("--wandb-log", "bool", False, "if toggled, this experiment will be tracked with Weights and Biases"),
    ("--wandb-project-name", str, "AC-Solver-PPO", "the wandb's project name"),
    ("--wandb-entity", str, None, "the entity (team) of wandb's project"),
    ("--fixed-init-state", "bool", False, "start every rollout from the presentation given by --relator1 / --relator2 instead of the Miller-Schupp set"),
    ("--states-type", str, "all", "which Miller-Schupp presentations to load: solved or all"),
    ("--repeat-solved-prob", float, 0.25, "probability of choosing an already solved state once every state has been attempted"),
    ("--max-relator-length", int, 7, "the maximum length a relator is allowed to take when acted on by AC moves"),
    ("--relator1", "ints", [1, 1, -2, -2, -2], "first relator of the initial presentation (default: AK(2))"),
    ("--relator2", "ints", [1, 2, 1, -2, -1, -2], "second relator of the initial presentation (default: AK(2))"),
    ("--horizon-length", int, 2000, "number of environment steps after which a rollout is truncated"),
    ("--use_supermoves", "bool", False, "whether to use supermoves or not"),
    ("--nodes-counts", "ints", [256, 256], "widths of the hidden layers of the actor and the critic"),
    ("--is-loss-clip", "bool", True, "clipped surrogate objective (True) or KL-penalty objective (False)"),
    ("--beta", float, 0.9, "initial KL-penalty coefficient (KL-penalty objective only)"),
    ("--total-timesteps", int, 200000, "total timesteps of the experiment"),
    ("--learning-rate", float, 2.5e-4, "the (maximum) learning rate of the optimizer"),
    ("--warmup-period", float, 0.0, "fraction of the updates used for a linear learning-rate warm-up"),
    ("--lr-decay", str, "linear", "learning-rate schedule after the warm-up: linear or cosine"),
    ("--min-lr-frac", float, 0.0, "fraction of the maximum learning rate to anneal to"),
    ("--num-envs", int, 4, "the number of parallel environments"),
    ("--num-steps", int, 2000, "the number of steps per environment per policy rollout"),
    ("--anneal-lr", "bool", True, "toggle learning-rate annealing"),
    ("--gamma", float, 0.99, "the discount factor gamma"),
    ("--gae-lambda", float, 0.95, "the lambda of generalized advantage estimation"),
    ("--num-minibatches", int, 4, "the number of mini-batches"),
    ("--update-epochs", int, 1, "the K epochs to update the policy"),
    ("--norm-adv", "bool", True, "toggle advantage normalization"),
    ("--norm-rewards", "bool", False, "normalize rewards by a running estimate of the return variance (gymnasium NormalizeReward)"),
    ("--clip-rewards", "bool", True, "clip rewards to [min-rew, max-rew]"),
    ("--min-rew", int, -10, "lower reward clip"),
    ("--max-rew", int, 1000, "upper reward clip"),
    ("--clip-coef", float, 0.2, "the surrogate clipping coefficient"),
    ("--clip-vloss", "bool", True, "use a clipped loss for the value function"),
    ("--ent-coef", float, 0.01, "coefficient of the entropy"),
    ("--vf-coef", float, 0.5, "coefficient of the value function"),
    ("--max-grad-norm", float, 0.5, "the maximum norm for gradient clipping"),
    ("--target-kl", float, 0.01, "the target KL divergence threshold"),
    ("--epsilon", float, 0.00001, "epsilon of the Adam optimizer"),
    # not in the reference: lifts its `num_envs <= number of initial states` assert (agents/environment.py:80-83)
    ("--tile-initial-states", "bool", False, "allow more environments than initial states: environment i starts from state i mod n"),
    ("--supermoves", str, "", "opt-in supermoves, e.g. '11,4,2;0,1': action 12 + s runs the s-th ';'-separated list of base moves as one step "
                               "(the reference's --use_supermoves has no implementation to follow: it raises NotImplementedError, here too)"),
    ("--fused-policy", "bool", False, "rollouts sample through the fused MFMA policy kernel (bf16 operands, f32 accumulation) instead of the f32 torch modules"),
]


def _to_bool(text):
    v = str(text).strip().lower()
    if v in ("y", "yes", "t", "true", "on", "1"):
        return True
    if v in ("n", "no", "f", "false", "off", "0"):
        return False
    raise argparse.ArgumentTypeError(f"invalid truth value {text!r}")


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="PPO on the Andrews-Curtis environment")
    for flag, kind, default, text in _FLAGS:
        if kind == "bool":
            parser.add_argument(flag, type=_to_bool, default=default, nargs="?", const=True, help=text)
        elif kind == "ints":
            parser.add_argument(flag, type=int, nargs="+", default=default, help=text)
        else:
            parser.add_argument(flag, type=kind, default=default, help=text)
    args = parser.parse_args(argv)
    args.batch_size = int(args.num_envs * args.num_steps)
    args.minibatch_size = int(args.batch_size // args.num_minibatches)
    assert 0.0 <= args.warmup_period <= 1.0, "warmup period should be less than 1.0 as it is the fraction of total timesteps"
    assert args.lr_decay in ["linear", "cosine"], f"lr-decay must be linear or cosine, not {args.lr_decay}. Other LR schedules not supported yet"
    assert 0.0 <= args.min_lr_frac <= 1.0, "min-lr-frac is the fraction of maximum lr to which we anneal."
    return args
