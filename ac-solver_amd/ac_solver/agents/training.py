"""PPO training loop on the device-resident environments (reference: ac_solver/agents/training.py:18-410).

Same algorithm and bookkeeping as the reference (learning-rate schedule, GAE, clipped / KL-penalty objective,
clipped value loss, curriculum over the initial states, ACMoves_hist, checkpoints, optional wandb scalars); what
changes is where the rollout lives: `ACVecEnv.step` writes the float32 observation, the (clipped) reward and the
terminated flag of every environment straight into the [num_steps, num_envs, ...] rollout tensors on the GPU, the
policy samples on the GPU, and the host only sees the indices of environments that finished an episode.
With torch.distributed initialised (one process per GPU) gradients are averaged over the ranks (data parallel).
"""
import math
import os
import random
import uuid
from collections import deque
from os import makedirs
from os.path import join

import numpy as np
import torch
from torch import nn


def get_curr_lr(n_update, lr_decay, warmup, max_lr, min_lr, total_updates):
    """Learning rate of update `n_update` (1-based): linear warm-up over the first `warmup` fraction of the updates,
    then a linear or cosine decay from max_lr to min_lr (training.py:18-65)."""
    k, last = n_update - 1, total_updates - 1
    warm_end = last * warmup
    if warm_end > 0 and k <= warm_end:
        return max_lr * k / warm_end
    if lr_decay == "linear":
        frac = (k - warm_end) / (last - warm_end)
        return max_lr + (min_lr - max_lr) * frac
    if lr_decay == "cosine":
        frac = (k - warm_end) / (last - warm_end)
        return min_lr + (max_lr - min_lr) * (1 + math.cos(frac * math.pi)) / 2
    raise NotImplementedError("Only 'linear' and 'cosine' lr-schedules are available.")


def compute_gae(rewards, values, dones, next_value, next_done, gamma, gae_lambda):
    """Generalised advantage estimation over a [T, N] rollout (training.py:241-258): dones[t] is 1 when the episode
    that produced obs[t] had just ended.  -> (advantages, returns)"""
    T = rewards.shape[0]
    advantages = torch.zeros_like(rewards)
    last = torch.zeros_like(next_value)
    for t in reversed(range(T)):
        nonterminal = 1.0 - (next_done if t == T - 1 else dones[t + 1])
        nextvalues = next_value if t == T - 1 else values[t + 1]
        delta = rewards[t] + gamma * nextvalues * nonterminal - values[t]
        advantages[t] = last = delta + gamma * gae_lambda * nonterminal * last
    return advantages, advantages + values


def choose_next_state(states_processed, n_states, success_record, round1_complete, repeat_solved_prob, stride=1):
    """Curriculum step of the reference (training.py:199-221): first walk through the initial states in order; once
    every state has been started at least once, pick an unsolved state with probability 1 - repeat_solved_prob
    (always, while nothing is solved), otherwise a solved one.  -> (next state index, round1_complete)
    `stride` > 1 (data parallel, one process per GPU): the states are dealt rank::world, so a rank's first round walks its
    own residue class and the ranks together start every state once."""
    round1_complete = round1_complete or max(states_processed) + stride > n_states - 1
    if not round1_complete:
        return max(states_processed) + stride, round1_complete
    if len(success_record["solved"]) == 0 or (success_record["unsolved"] and random.uniform(0, 1) > repeat_solved_prob):
        return random.choice(list(success_record["unsolved"])), round1_complete
    return random.choice(list(success_record["solved"])), round1_complete


class Curriculum:
    """`choose_next_state` for a whole run, with the same decisions and the same draws from `random` as calling that function
    once per finished episode (training.py:199-221 of the reference) -- but without its per-call O(n_states) work: the
    reference takes max(states_processed) over the whole set and builds list(unsolved) / list(solved) for every episode, which
    at 131 072 environments (hundreds of finished episodes per rollout step) was most of an update's wall time.  Here the
    maximum is kept as a number and the two lists are rebuilt only when their set changed (list(set) of an unchanged set is
    the same list, so random.choice picks the same element).  The caller's sets stay the containers that are updated."""

    def __init__(self, states_processed, n_states, success_record, repeat_solved_prob, stride=1):
        self.processed, self.n_states, self.rec, self.p, self.stride = states_processed, n_states, success_record, repeat_solved_prob, stride
        self.max_processed = max(states_processed)
        self.round1_complete = False
        self._lists = {"solved": None, "unsolved": None}
        self._sizes = {"solved": -1, "unsolved": -1}

    def _list(self, which):
        st = self.rec[which]
        if self._lists[which] is None or self._sizes[which] != len(st):  # (a state only ever moves unsolved -> solved: sizes tell)
            self._lists[which], self._sizes[which] = list(st), len(st)
        return self._lists[which]

    def mark_solved(self, s):
        if s in self.rec["unsolved"]:
            self.rec["unsolved"].remove(s)
            self.rec["solved"].add(s)
            self._lists["solved"] = self._lists["unsolved"] = None

    def next_state(self):
        self.round1_complete = self.round1_complete or self.max_processed + self.stride > self.n_states - 1
        if not self.round1_complete:
            nxt = self.max_processed + self.stride
        elif len(self.rec["solved"]) == 0 or (self.rec["unsolved"] and random.uniform(0, 1) > self.p):
            nxt = random.choice(self._list("unsolved"))
        else:
            nxt = random.choice(self._list("solved"))
        self.processed.add(nxt)
        if nxt > self.max_processed:
            self.max_processed = nxt
        return nxt

    # ---- a rollout step's finished episodes at once ------------------------------------------------------------------------------
    _C_MIN = 48  # below this many draws Python's own generator is cheaper than moving its state to libacx_trainer and back

    # The state of Python's global generator while libacx_trainer draws from it: taken out once (begin_borrow: 625 words, ~60 us with the
    # conversions), advanced in place by every acxt_py_curriculum_draws call, put back by end_borrow -- the training loop borrows it
    # for a whole rollout (nothing else draws from `random` in there); on its own, finish_episodes borrows per call.
    _borrowed = None

    def begin_borrow(self):
        import ctypes as C

        ver, internal, gauss = random.getstate()
        self._borrowed = (ver, gauss, np.array(internal[:624], dtype=np.uint32), C.c_int32(internal[624]))

    def end_borrow(self):
        if self._borrowed is not None:
            ver, gauss, mt, pos = self._borrowed
            self._borrowed = None
            random.setstate((ver, tuple(mt.tolist()) + (pos.value,), gauss))

    def _draws(self, n):
        """next_state() n times after the first round, with both lists fixed: the same draws from `random`, taken in libacx_trainer
        (acxt_py_curriculum_draws restates CPython's Random.random / choice) on the state of the global generator.  -> states [n]"""
        if n < self._C_MIN and self._borrowed is None:
            return [self.next_state() for _ in range(n)]
        import ctypes as C

        from ac_solver import _acx
        from ac_solver.agents import _host

        mine = self._borrowed is None
        if mine:
            self.begin_borrow()
        _, _, mt, pos = self._borrowed
        which, index = np.empty(n, np.uint8), np.empty(n, np.int64)
        solved, unsolved = self._list("solved"), self._list("unsolved")
        _host.check(_host.lib.acxt_py_curriculum_draws(mt.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos), n, len(solved), len(unsolved), float(self.p),
                                                       _acx.ptr(which, C.c_uint8), _acx.ptr(index, C.c_int64)), "acxt_py_curriculum_draws")
        if mine:
            self.end_borrow()
        key = ("arrays", len(solved), len(unsolved))
        if getattr(self, "_arr_key", None) != key:  # (a state only ever moves unsolved -> solved: the sizes tell)
            self._arr = (np.asarray(unsolved, np.int64), np.asarray(solved, np.int64))
            self._arr_key = key
        un, so = self._arr
        out = np.where(which == 0, un[np.minimum(index, max(len(un) - 1, 0))] if len(un) else 0, so[np.minimum(index, max(len(so) - 1, 0))] if len(so) else 0)
        nxt = out.tolist()
        self.processed.update(nxt)
        self.max_processed = max(self.max_processed, max(nxt))
        return nxt

    def finish_episodes(self, current, done, on_done):
        """The reference's bookkeeping (training.py:172-224) for the episodes a rollout step ended, in environment order: `current`
        [k] the curriculum states those environments ran, `done` [k] whether the episode reached the trivial presentation; for every
        such episode `on_done(position, state)` is called (the shortest-path record) after the state has been marked solved.
        -> the next state of each environment [k].  Same decisions, same draws from `random`, as mark_solved / next_state called
        episode by episode; the episodes between two changes of the solved set take their draws in one call of libacx."""
        k = len(current)
        out = []
        done_pos = np.flatnonzero(done).tolist()
        a = 0
        for pos in done_pos + [k]:
            # episodes a .. pos - 1 draw with the lists as they are; episode pos (if any) first changes them
            if pos < k:
                s = current[pos]
                changes = s in self.rec["unsolved"]
                if not changes:
                    on_done(pos, s)  # (solved before: only the path record may change, the lists do not)
                    continue
            n = pos - a
            if n > 0:
                if not self.round1_complete:
                    # first round: the states in order, `stride` apart, while they last (then the rule flips for good)
                    self.round1_complete = self.max_processed + self.stride > self.n_states - 1
                while n > 0 and not self.round1_complete:
                    out.append(self.next_state())
                    n -= 1
                    self.round1_complete = self.round1_complete or self.max_processed + self.stride > self.n_states - 1
                if n > 0:
                    out.extend(self._draws(n))
            a = pos
            if pos < k:
                self.mark_solved(current[pos])
                on_done(pos, current[pos])
        # (every `continue` above left its episodes in the pending range a .. : they were drawn with the segment that followed)
        return out


class RunningReturnNormalizer:
    """Per-environment reward normalisation as gymnasium 0.28.1's `NormalizeReward` wrapper does it around every single
    env (environment.py:44-46 of the reference): a discounted return is accumulated, its running variance is tracked
    (Welford / Chan update with one sample per step, initial mean 0, variance 1, count 1e-4) and the reward is divided by
    sqrt(variance + 1e-8).  gymnasium is a third-party dependency that is not part of the reference tree: this follows
    its published algorithm and is not pinned by a fixture."""

    def __init__(self, n, gamma, device, epsilon=1e-8):
        self.gamma, self.epsilon = gamma, epsilon
        self.returns = torch.zeros(n, device=device, dtype=torch.float64)
        self.mean = torch.zeros(n, device=device, dtype=torch.float64)
        self.var = torch.ones(n, device=device, dtype=torch.float64)
        self.count = torch.full((n,), 1e-4, device=device, dtype=torch.float64)

    def __call__(self, rewards, terminated):
        self.returns = self.returns * self.gamma * (1.0 - terminated.to(torch.float64)) + rewards.to(torch.float64)
        delta = self.returns - self.mean
        tot = self.count + 1.0
        self.mean = self.mean + delta / tot
        self.var = (self.var * self.count + delta * delta * self.count / tot) / tot
        self.count = tot
        return (rewards.to(torch.float64) / torch.sqrt(self.var + self.epsilon)).to(rewards.dtype)


def _average_gradients(params, world):
    """One all-reduce per minibatch over ONE flat bucket (the policy + critic are ~0.6 MB: a single RCCL call over xGMI)."""
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    torch.distributed.all_reduce(flat)
    flat /= world
    o = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n


def _global_mean(x, world):
    """Mean over the ranks of a scalar tensor: every rank must take the SAME early-stop / KL-penalty decision, or the
    per-minibatch all-reduces of the ranks stop pairing up (the reference is single-process and has no such step)."""
    y = x.detach().clone().reshape(1)
    torch.distributed.all_reduce(y)
    return (y / world)[0]


def _share_success_record(success_record, n_states, device):
    """Union of the ranks' solved sets (one all-reduce of an n_states mask per update): the curriculum of every rank
    then samples from what ANY rank has solved."""
    mask = torch.zeros(n_states, dtype=torch.int32, device=device)
    if success_record["solved"]:
        mask[torch.as_tensor(sorted(success_record["solved"]), device=device)] = 1
    torch.distributed.all_reduce(mask, op=torch.distributed.ReduceOp.MAX)
    solved = set(torch.nonzero(mask).flatten().tolist())
    success_record["solved"] |= solved
    success_record["unsolved"] -= solved


class _MinibatchOrder:
    """The minibatch permutations of the updates, computed ahead of time.  The reference shuffles np.arange(batch_size) once per epoch
    with the global NumPy generator it seeded at the top of the update (training.py:121, 273-275: np.random.seed(seed + update) ...
    np.random.shuffle(b_inds)); nothing draws from that generator in between, so the permutations are a function of the seed alone.
    At BASELINE config 5's shape a shuffle of 4 Mi indices takes ~100 ms of host time, and each minibatch's slice went to the device
    through a synchronous copy -- with the GPU idle meanwhile (60 of an update's 123 ms).  Here a thread computes them through libacx
    (acxt_np_shuffle_epochs: NumPy's legacy algorithm restated, pinned against numpy in tests/test_agents_cpu.py, and -- unlike
    np.random.shuffle, which holds the GIL for all of its run -- off the interpreter lock), ONE UPDATE AHEAD: the permutations of
    update u + 1 are started at the top of update u (two pinned buffers take turns), go to the device in one copy, and a minibatch
    is a slice of that."""

    def __init__(self, batch_size, epochs, device):
        self.batch_size, self.epochs, self.device = batch_size, epochs, device
        self.on_gpu = str(device).startswith("cuda") and torch.cuda.is_available()
        self.host = [torch.empty((epochs, batch_size), dtype=torch.int64, pin_memory=self.on_gpu) for _ in range(2)]
        self.dev = torch.empty((epochs, batch_size), dtype=torch.int64, device=device) if self.on_gpu else None
        self.jobs = {}       # seed -> (thread, buffer index)
        self.copied = [None, None]  # per buffer: event behind the last copy out of it

    def prefetch(self, seed):
        import threading

        if seed in self.jobs:
            return
        busy = {job[1] for job in self.jobs.values()}
        k = 0 if 0 not in busy else 1
        assert k not in busy, "more than two updates' permutations in flight"
        if self.copied[k] is not None:
            self.copied[k].synchronize()  # the permutations this buffer held have left it

        failed = []  # an exception on the worker thread would die with it and leave the never-written buffer as "permutations"

        def work(out=self.host[k].numpy()):
            import ctypes as C

            from ac_solver import _acx
            from ac_solver.agents import _host

            try:
                _host.check(_host.lib.acxt_np_shuffle_epochs(int(seed) & 0xFFFFFFFF, self.batch_size, self.epochs, _acx.ptr(out, C.c_int64)), "acxt_np_shuffle_epochs")
            except BaseException as e:  # noqa: BLE001 -- re-raised by get() on the training thread
                failed.append(e)

        th = threading.Thread(target=work)
        th.start()
        self.jobs[seed] = (th, k, failed)

    def get(self, seed):
        """-> [epochs, batch_size] int64 on the training device: the permutations of the update seeded with `seed`"""
        self.prefetch(seed)
        th, k, failed = self.jobs.pop(seed)
        th.join()
        if failed:
            raise failed[0]
        if not self.on_gpu:
            return self.host[k]
        self.dev.copy_(self.host[k], non_blocking=True)
        self.copied[k] = torch.cuda.Event()
        self.copied[k].record()
        return self.dev


class _Phases:
    """ACX_PPO_PHASES=1: wall time per phase of an update (a device synchronisation at every phase edge, so the run itself is
    slower), printed by rank 0 at the end of the training loop; off: no-ops."""

    def __init__(self, device):
        import os

        self.on = bool(os.environ.get("ACX_PPO_PHASES")) and torch.cuda.is_available() and str(device).startswith("cuda")
        self.acc, self.t = {}, None

    def start(self):
        if self.on:
            import time

            torch.cuda.synchronize()
            self.t = time.perf_counter()

    def lap(self, name):
        if self.on:
            import time

            torch.cuda.synchronize()
            now = time.perf_counter()
            self.acc[name] = self.acc.get(name, 0.0) + now - self.t
            self.t = now

    def report(self, updates):
        if self.on and updates:
            tot = sum(self.acc.values())
            print("[ppo phases] per update: " + ", ".join(f"{k} {v / updates * 1e3:.1f} ms" for k, v in self.acc.items()) + f"; sum {tot / updates * 1e3:.1f} ms")


class _StatsStream:
    """refresh_behaviour_stats WHILE the rollout runs: the f32 forward over rollout rows t0 .. t1 - 1 is enqueued on a second stream as
    soon as those rows are final (behind an event of the rollout's stream), so the ~23 ms of GEMMs per update at BASELINE config 5's
    shape fill the GPU time the rollout leaves idle -- its own kernels are 5 ms, the rest of its ~20 ms is the host's episode
    bookkeeping.  Groups of `group` rows: fewer, fuller GEMMs and an eighth of the launches of a row at a time."""

    def __init__(self, agent, device, group=4):
        self.agent, self.group = agent, max(1, int(group))
        self.side = torch.cuda.Stream(device) if device.type == "cuda" else None
        self.t0 = 0

    def rows_final(self, t1, obs, actions, logprobs, values, last=False):
        """rows self.t0 .. t1 - 1 of obs / actions are final on the current stream (and the sampling kernel has written its logprobs /
        values for them): recompute those of a full group (or, with `last`, of what is left) in f32, in place."""
        if t1 - self.t0 < self.group and not (last and t1 > self.t0):
            return
        t0, self.t0 = self.t0, t1
        if self.side is None:
            refresh_behaviour_stats(self.agent, obs[t0:t1], actions[t0:t1], logprobs[t0:t1], values[t0:t1])
            return
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            refresh_behaviour_stats(self.agent, obs[t0:t1], actions[t0:t1], logprobs[t0:t1], values[t0:t1])

    def join(self):
        """the current stream waits for everything enqueued so far; the next rollout starts at row 0"""
        self.t0 = 0
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)


def refresh_behaviour_stats(agent, obs, actions, logprobs, values):
    """logprobs[t], values[t] <- the f32 agent's log pi(actions[t] | obs[t]) and V(obs[t]) for every rollout row t (in place)."""
    T, N = actions.shape
    rows = max(1, min(T, (1 << 20) // max(N, 1)))  # ~1 Mi samples per forward pass: 32 passes of 131 072 rows cost 25 ms at BASELINE config 5's shape, 4 of 1 Mi cost less (fewer, fuller GEMMs)
    with torch.no_grad():
        for t0 in range(0, T, rows):
            t1 = min(T, t0 + rows)
            _, lp, _, v = agent.get_action_and_value(obs[t0:t1].reshape((-1,) + obs.shape[2:]).float(), actions[t0:t1].reshape(-1))
            logprobs[t0:t1].copy_(lp.view(t1 - t0, N))
            values[t0:t1].copy_(v.view(t1 - t0, N))


def ppo_training_loop(envs, args, device, optimizer, agent, curr_states, success_record, ACMoves_hist, states_processed,
                      initial_states, progress=True, rollout_log=None):
    """`rollout_log`, when a list, receives per update a dict of CPU copies of the rollout tensors (tests)."""
    T, N = args.num_steps, args.num_envs
    obs_shape = envs.single_observation_space.shape
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    world = torch.distributed.get_world_size() if dist_on else 1
    rank = torch.distributed.get_rank() if dist_on else 0

    # rollout storage: the environment kernel writes rows t+1 of obs / term and row t of rewards
    fused_on = bool(getattr(args, "fused_policy", False))
    obs_dtype = torch.int8 if fused_on else torch.float32  # (get_env creates the environments with the matching obs_dtype)
    obs = torch.zeros((T + 1, N) + obs_shape, dtype=obs_dtype, device=device)
    term = torch.zeros((T + 1, N), dtype=torch.bool, device=device)
    trunc = torch.zeros(N, dtype=torch.bool, device=device)
    actions = torch.zeros((T, N), dtype=torch.int64, device=device)
    logprobs = torch.zeros((T, N), device=device)
    rewards = torch.zeros((T, N), device=device)
    values = torch.zeros((T, N), device=device)
    init_rows = np.asarray(initial_states, np.int8)
    init_table = torch.as_tensor(init_rows, device=device).to(obs_dtype)  # [n_states, 2L]
    init_rows_dev = torch.as_tensor(init_rows, device=device)  # int8 rows for ACVecEnv.reset_envs_device
    ns_pin = torch.empty(N, dtype=torch.int64).pin_memory() if device.type == "cuda" else None
    ep_return = torch.zeros(N, device=device)
    ep_length = torch.zeros(N, device=device)

    normalizer = RunningReturnNormalizer(N, args.gamma, device) if args.norm_rewards else None
    global_step = 0
    obs[0].copy_(envs.reset()[0])
    num_updates = args.total_timesteps // args.batch_size
    episode = 0
    returns_queue, lengths_queue = deque([0], maxlen=100), deque([0], maxlen=100)
    round1_complete = False
    beta = None if args.is_loss_clip else args.beta
    params = list(agent.parameters())
    fused = None
    if fused_on:  # opt-in: rollout inference on the matrix cores (agents/fused_policy.py); the update stays f32 torch
        from ac_solver.agents.fused_policy import FusedPolicy

        fused = FusedPolicy(agent, int(np.prod(obs_shape)), seed=args.seed)
    # data parallel: the Miller-Schupp states are dealt rank::world (get_env), so the first curriculum round of a rank
    # walks its own residue class; a single fixed initial state has nothing to deal
    stride = world if dist_on and world > 1 and len(initial_states) > 1 else 1
    curriculum = Curriculum(states_processed, len(initial_states), success_record, args.repeat_solved_prob, stride)

    run_name = f"{args.exp_name}_ppo-ffn-nodes_{args.nodes_counts}_{uuid.uuid4()}"
    out_dir = f"out/{run_name}"
    wandb = None
    if args.wandb_log and rank == 0:
        import wandb  # noqa: F811  (optional dependency)

        wandb.init(project=args.wandb_project_name, entity=args.wandb_entity, name=run_name, config=vars(args), save_code=True)
    if rank == 0:
        print(f"total number of timesteps: {args.total_timesteps}, updates: {num_updates}")
    updates = range(1, num_updates + 1)
    if progress and rank == 0:
        try:
            from tqdm import tqdm

            updates = tqdm(updates, desc="Training Progress", total=num_updates)
        except ImportError:  # pragma: no cover
            pass
    stats = {}
    stats_stream = _StatsStream(agent, device, int(os.environ.get("ACX_PPO_STATS_GROUP", "4"))) if fused_on and os.environ.get("ACX_PPO_STATS_STREAM", "1") != "0" else None
    order = _MinibatchOrder(args.batch_size, args.update_epochs, device)
    ph = _Phases(device)
    n_updates_done = 0

    for update in updates:
        ph.start()
        random.seed(args.seed + update)
        np.random.seed(args.seed + update)
        order.prefetch(args.seed + update)  # this update's minibatch permutations (under way since the previous update, except for the first) ...
        if update < num_updates:
            order.prefetch(args.seed + update + 1)  # ... and the next one's: a whole update to hide their ~100 ms of host time in
        torch.manual_seed(args.seed + update)
        if args.anneal_lr:
            optimizer.param_groups[0]["lr"] = get_curr_lr(update, args.lr_decay, args.warmup_period, args.learning_rate,
                                                          args.learning_rate * args.min_lr_frac, num_updates)
        events = []  # (step, env, next curriculum state) of this rollout
        reset_err_any = None
        curriculum.begin_borrow()  # the rollout's curriculum draws advance the generator's state inside libacx; back in `random` after the rollout
        try:  # (whatever the rollout raises, the generator state goes back into `random`: a caller that catches and continues must not find it stale)
            if fused is not None:
                fused.refresh()  # the weights of the last update

            # ---------------------------------------------------------------- rollout (device resident) ----
            for step in range(T):
                global_step += N * world
                if fused is not None:
                    fused.sample(obs[step], actions[step], logprobs[step], values[step])
                    action = actions[step]
                    if stats_stream is not None:  # rows up to `step` are final: their f32 behaviour statistics start on the second stream
                        stats_stream.rows_final(step + 1, obs, actions, logprobs, values, last=step == T - 1)
                else:
                    with torch.no_grad():
                        action, logprob, _, value = agent.get_action_and_value(obs[step])
                    actions[step] = action
                    logprobs[step] = logprob
                    values[step] = value.flatten()
                envs.step(action, out=(obs[step + 1], rewards[step], term[step + 1], trunc), check_errors=False)
                ph.lap("policy+env")
                if normalizer is not None:  # NormalizeReward, then TransformReward(clip) as make_env stacks them
                    rewards[step] = normalizer(rewards[step], term[step + 1])
                    if args.clip_rewards:
                        rewards[step].clamp_(args.min_rew, args.max_rew)
                ep_return += rewards[step]
                ep_length += 1
                fin = term[step + 1] | trunc
                # ---- episodes ended: bookkeeping of the reference (training.py:167-224) on the finished envs only ----
                # two read-backs per step: how many (the size of `idx`), then index, done flag, return and length of each in one block
                idx = torch.nonzero(fin).flatten()
                if idx.numel() == 0:
                    ph.lap("episode bookkeeping")
                    continue
                packed = torch.stack((idx.double(), term[step + 1][idx].double(), ep_return[idx].double(), ep_length[idx].double())).cpu().numpy()
                idx_h = packed[0].astype(np.int64).tolist()
                done_np = packed[1] != 0
                ret_h, len_h = packed[2].tolist(), packed[3].tolist()  # (float32 values, exact in float64: what .tolist() of the tensors gives)
                current = [curr_states[i] for i in idx_h]

                def on_done(k, s):  # the shortest action sequence seen for state s (strictly shorter replaces; the first of a length stays)
                    # an episode's action list is as long as the episode: only a record-setting one is read back from the device
                    if s not in ACMoves_hist or int(len_h[k]) < len(ACMoves_hist[s]):
                        moves = envs.get_actions(idx_h[k], finished=True)
                        if len(moves) != int(len_h[k]):
                            raise RuntimeError(f"env {idx_h[k]}: an episode of {int(len_h[k])} steps with {len(moves)} recorded actions")
                        if s not in ACMoves_hist or len(moves) < len(ACMoves_hist[s]):
                            ACMoves_hist[s] = moves

                new_states = curriculum.finish_episodes(current, done_np, on_done)
                for i, nxt in zip(idx_h, new_states):
                    curr_states[i] = nxt
                if rollout_log is not None:
                    events.extend((step, i, nxt) for i, nxt in zip(idx_h, new_states))
                returns_queue.extend(ret_h)
                lengths_queue.extend(len_h)
                episode += len(idx_h)
                round1_complete = curriculum.round1_complete
                ep_return[idx] = 0
                ep_length[idx] = 0
                # the finished envs restart from their next curriculum state (envs.envs[i].reset(options={"starting_state": ...})): one
                # upload (the states' numbers, through a pinned buffer), rows gathered and environments reset on the device
                if device.type == "cuda":
                    k = len(new_states)
                    ns_pin[:k].copy_(torch.from_numpy(np.asarray(new_states, np.int64)))
                    ns_dev = ns_pin[:k].to(device, non_blocking=True)
                    reset_err = envs.reset_envs_device(idx, init_rows_dev.index_select(0, ns_dev))
                    reset_err_any = reset_err.any() if reset_err_any is None else reset_err_any | reset_err.any()
                else:
                    ns_dev = torch.as_tensor(new_states, device=device)
                    envs.reset_envs(idx_h, init_rows[new_states])
                obs[step + 1].index_copy_(0, idx, init_table.index_select(0, ns_dev))
                ph.lap("episode bookkeeping")
        finally:
            curriculum.end_borrow()
        if reset_err_any is not None and bool(reset_err_any):  # (cannot happen: the rows are the validated initial states)
            raise ValueError("a curriculum state is not a valid presentation (ACEnv.reset)")
        envs._raise_on_errors()
        if rollout_log is not None:
            rollout_log.append({"obs": obs.cpu().numpy().copy(), "actions": actions.cpu().numpy().copy(), "rewards": rewards.cpu().numpy().copy(),
                                "term": term.cpu().numpy().copy(), "events": list(events)})

        if not args.norm_rewards:
            rewards /= envs.max_reward
            normalized_returns = np.array(returns_queue) / envs.max_reward
            normalized_lengths = np.array(lengths_queue) / args.horizon_length
        else:
            normalized_returns, normalized_lengths = np.array(returns_queue), np.array(lengths_queue)

        dones = term.to(torch.float32)
        if fused is not None:
            # The rollout's log-probabilities and values came out of the bf16 matrix-core kernel (|dlogp| up to ~0.1 against
            # the f32 modules).  The update differentiates the f32 modules, and the reference forms its importance ratio
            # against log-probabilities of THAT policy (training.py:283-291): so the behaviour statistics the update uses are
            # recomputed here with one f32 forward over the batch -- the ratio of the first minibatch of the first epoch is then
            # exactly 1, as in the reference.  One row of the rollout at a time keeps the temporaries at [N, 256].
            # Since round 4 that forward runs on a second stream beside the rollout (_StatsStream); here only its tail is waited for.
            ph.lap("rollout tail")
            if stats_stream is not None:
                stats_stream.join()
            else:
                refresh_behaviour_stats(agent, obs, actions, logprobs, values)
            ph.lap("behaviour stats (f32)")
        with torch.no_grad():
            next_value = agent.get_value(obs[T].float()).reshape(-1)
            advantages, returns = compute_gae(rewards, values, dones[:T], next_value, dones[T], args.gamma, args.gae_lambda)

        ph.lap("gae")
        b_obs = obs[:T].reshape((-1,) + obs_shape)
        b_logprobs, b_actions = logprobs.reshape(-1), actions.reshape(-1)
        b_advantages, b_returns, b_values = advantages.reshape(-1), returns.reshape(-1), values.reshape(-1)

        # ---------------------------------------------------------------- policy / value update ----
        perms = order.get(args.seed + update)
        clipfracs = []
        for epoch in range(args.update_epochs):
            for start in range(0, args.batch_size, args.minibatch_size):
                mb = perms[epoch, start:start + args.minibatch_size]
                _, newlogprob, entropy, newvalue = agent.get_action_and_value(b_obs[mb].float(), b_actions[mb])
                logratio = newlogprob - b_logprobs[mb]
                ratio = logratio.exp()
                kl_var = (ratio - 1) - logratio  # E[kl_var] approximates KL(pi_old || pi)
                with torch.no_grad():
                    if epoch == 0 and start == 0:  # before any step of this update the ratio is 1 (reference: training.py:283-291)
                        ratio0_dev = (ratio - 1.0).abs().max()
                    approx_kl = kl_var.mean()
                    clipfracs.append(((ratio - 1.0).abs() > args.clip_coef).float().mean())
                mb_adv = b_advantages[mb]
                if args.norm_adv:
                    mb_adv = (mb_adv - mb_adv.mean()) / (mb_adv.std() + 1e-8)
                if args.is_loss_clip:
                    pg_loss = torch.max(-mb_adv * ratio, -mb_adv * torch.clamp(ratio, 1 - args.clip_coef, 1 + args.clip_coef)).mean()
                else:
                    pg_loss = (-mb_adv * ratio + beta * kl_var).mean()
                newvalue = newvalue.view(-1)
                if args.clip_vloss:
                    v_clipped = b_values[mb] + torch.clamp(newvalue - b_values[mb], -args.clip_coef, args.clip_coef)
                    v_loss = 0.5 * torch.max((newvalue - b_returns[mb]) ** 2, (v_clipped - b_returns[mb]) ** 2).mean()
                else:
                    v_loss = 0.5 * ((newvalue - b_returns[mb]) ** 2).mean()
                entropy_loss = entropy.mean()
                loss = pg_loss - args.ent_coef * entropy_loss + v_loss * args.vf_coef
                optimizer.zero_grad()
                loss.backward()
                if dist_on and world > 1:
                    _average_gradients(params, world)
                nn.utils.clip_grad_norm_(agent.parameters(), args.max_grad_norm)
                optimizer.step()
            if dist_on and world > 1:  # one decision for all ranks (see _global_mean)
                approx_kl = _global_mean(approx_kl, world)
            if args.is_loss_clip:
                if args.target_kl is not None and approx_kl > args.target_kl:
                    break
            else:
                beta = beta / 2 if approx_kl < args.target_kl / 1.5 else (beta * 2 if approx_kl > args.target_kl * 1.5 else beta)
        ph.lap("update (f32 torch)")
        if dist_on and world > 1:
            _share_success_record(success_record, len(initial_states), device)

        # the next rollout continues where this one stopped
        obs[0].copy_(obs[T])
        term[0].copy_(term[T])

        # explained variance on the device (np.var's population variance, as the reference computes it on host copies: two 16 MB
        # read-backs per update at BASELINE config 5's shape, 20 ms)
        var_y = float(b_returns.var(unbiased=False))
        explained_var = np.nan if var_y == 0 else 1 - float((b_returns - b_values).var(unbiased=False)) / var_y
        stats = {
            "charts/global_step": global_step, "charts/episode": episode,
            "charts/normalized_returns_mean": float(normalized_returns.mean()), "charts/normalized_lengths_mean": float(normalized_lengths.mean()),
            "charts/learning_rate": optimizer.param_groups[0]["lr"], "charts/solved": len(success_record["solved"]),
            "charts/unsolved": len(success_record["unsolved"]),
            "charts/highest_solved": max(success_record["solved"]) if success_record["solved"] else -1,
            "losses/value_loss": v_loss.item(), "losses/policy_loss": pg_loss.item(), "losses/entropy_loss": entropy_loss.item(),
            "losses/approx_kl": approx_kl.item(), "losses/explained_variance": explained_var,
            "losses/clipfrac": float(torch.stack(clipfracs).mean()), "debug/advantages_mean": float(b_advantages.mean()),
            "debug/advantages_std": float(b_advantages.std()), "debug/ratio0_maxdev": float(ratio0_dev),
        }
        if wandb is not None:
            wandb.log(stats)
        if update % 100 == 0 and dist_on and world > 1:  # the checkpoint holds the shortest solution any rank has found
            gathered = [None] * world
            torch.distributed.all_gather_object(gathered, ACMoves_hist)
            for other in gathered:
                for s_idx, moves in other.items():
                    if s_idx not in ACMoves_hist or len(moves) < len(ACMoves_hist[s_idx]):
                        ACMoves_hist[s_idx] = moves
        if update % 100 == 0 and rank == 0:  # a checkpoint every 100 updates (training.py:384-408)
            makedirs(out_dir, exist_ok=True)
            checkpoint = {
                "critic": agent.critic.state_dict(), "actor": agent.actor.state_dict(), "optimizer": optimizer.state_dict(),
                "update": update, "episode": episode, "config": vars(args), "mean_return": normalized_returns.mean(),
                "success_record": success_record, "value_loss": v_loss.item(), "policy_loss": pg_loss.item(),
                "entropy_loss": entropy_loss.item(), "approx_kl": approx_kl.item(), "explained_var": explained_var,
                "clipfrac": stats["losses/clipfrac"], "global_step": global_step, "round1_complete": round1_complete,
                "curr_states": curr_states, "states_processed": states_processed, "ACMoves_hist": ACMoves_hist,
                "supermoves": ({12 + k: q for k, q in enumerate(envs.supermoves)} if getattr(envs, "supermoves", None) else None),
            }
            print(f"saving checkpoint to {out_dir}")
            torch.save(checkpoint, join(out_dir, "ckpt.pt"))
        ph.lap("stats + tail")
        n_updates_done += 1
    if rank == 0:
        ph.report(n_updates_done)
    return stats
