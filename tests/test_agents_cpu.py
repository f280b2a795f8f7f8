"""CPU tests of the PPO counterpart (ac_solver/agents): what the reference's tests/agents/test_ppo.py pins -- argparse
defaults, layer shapes, build_network, Agent outputs, get_curr_lr -- plus GAE and the curriculum step."""
import random
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from torch import nn

from ac_solver.agents.args import parse_args
from ac_solver.agents.ppo_agent import Agent, build_network, initialize_layer
from ac_solver.agents.training import choose_next_state, compute_gae, get_curr_lr


def _envs(width=14, n_actions=12):
    return SimpleNamespace(single_observation_space=SimpleNamespace(shape=(width,)), single_action_space=SimpleNamespace(n=n_actions, shape=()))


def test_parse_args_defaults_and_derived_fields():
    a = parse_args(["--exp-name", "test_exp", "--seed", "42"])
    assert (a.exp_name, a.seed, a.torch_deterministic, a.cuda, a.wandb_log) == ("test_exp", 42, True, True, False)
    d = parse_args([])
    assert (d.exp_name, d.seed, d.states_type, d.repeat_solved_prob, d.max_relator_length) == ("args", 1, "all", 0.25, 7)
    assert d.relator1 == [1, 1, -2, -2, -2] and d.relator2 == [1, 2, 1, -2, -1, -2] and d.nodes_counts == [256, 256]
    assert (d.horizon_length, d.num_envs, d.num_steps, d.total_timesteps) == (2000, 4, 2000, 200000)
    assert (d.learning_rate, d.warmup_period, d.lr_decay, d.min_lr_frac, d.anneal_lr) == (2.5e-4, 0.0, "linear", 0.0, True)
    assert (d.gamma, d.gae_lambda, d.num_minibatches, d.update_epochs) == (0.99, 0.95, 4, 1)
    assert (d.norm_adv, d.norm_rewards, d.clip_rewards, d.min_rew, d.max_rew) == (True, False, True, -10, 1000)
    assert (d.clip_coef, d.clip_vloss, d.ent_coef, d.vf_coef, d.max_grad_norm, d.target_kl, d.epsilon) == (0.2, True, 0.01, 0.5, 0.5, 0.01, 1e-5)
    assert (d.is_loss_clip, d.beta, d.fixed_init_state, d.use_supermoves) == (True, 0.9, False, False)
    assert d.batch_size == 8000 and d.minibatch_size == 2000
    b = parse_args(["--cuda", "false", "--wandb-log", "--nodes-counts", "64", "32", "--num-envs", "8", "--num-steps", "10"])
    assert b.cuda is False and b.wandb_log is True and b.nodes_counts == [64, 32] and b.batch_size == 80 and b.minibatch_size == 20
    with pytest.raises(AssertionError):
        parse_args(["--lr-decay", "step"])


def test_layers_and_network():
    layer = initialize_layer(nn.Linear(4, 2))
    assert layer.weight.shape == torch.Size([2, 4]) and layer.bias.shape == torch.Size([2]) and float(layer.bias.abs().sum()) == 0.0
    layers = build_network([4, 8, 2], 1.0)
    assert len(layers) == 3 and isinstance(layers[0], nn.Linear) and isinstance(layers[1], nn.Tanh) and isinstance(layers[2], nn.Linear)
    deep = build_network([50, 256, 256, 12])
    assert [type(m).__name__ for m in deep] == ["Linear", "Tanh", "Linear", "Tanh", "Linear"]
    assert float(deep[-1].weight.norm()) < float(deep[0].weight.norm())  # policy head: gain 0.01


def test_agent_shapes():
    agent = Agent(_envs(), [256, 256])
    assert agent.critic_nodes == [14, 256, 256, 1] and agent.actor_nodes == [14, 256, 256, 12]
    obs = torch.randn(4, 14)
    assert agent.get_value(obs).shape == torch.Size([4, 1])
    action, log_prob, entropy, value = agent.get_action_and_value(obs)
    assert action.shape == log_prob.shape == entropy.shape == torch.Size([4]) and value.shape == torch.Size([4, 1])
    _, lp2, _, _ = agent.get_action_and_value(obs, action)
    assert torch.allclose(lp2, log_prob)


@pytest.mark.parametrize("lr_decay, warmup, n_update, expected_lr",
                         [("linear", 0.1, 1, 0.0), ("linear", 0.0, 1, 2.5e-04), ("cosine", 0.0, 159, 0.0), ("cosine", 0.1, 100, 9.36e-05)])
def test_get_curr_lr_reference_cases(lr_decay, warmup, n_update, expected_lr):
    assert np.isclose(get_curr_lr(n_update, lr_decay, warmup, 2.5e-4, 0.0, 1000 // 40), expected_lr, atol=1e-4)


def test_get_curr_lr_schedule_shape():
    lrs = [get_curr_lr(k, "linear", 0.2, 1.0, 0.1, 101) for k in range(1, 102)]
    assert lrs[0] == 0.0 and abs(lrs[20] - 1.0) < 1e-12 and abs(lrs[-1] - 0.1) < 1e-12 and all(a >= b for a, b in zip(lrs[20:], lrs[21:]))
    cos = [get_curr_lr(k, "cosine", 0.0, 1.0, 0.0, 11) for k in range(1, 12)]
    assert abs(cos[0] - 1.0) < 1e-12 and abs(cos[5] - 0.5) < 1e-12 and abs(cos[-1]) < 1e-12
    with pytest.raises(NotImplementedError):
        get_curr_lr(5, "step", 0.0, 1.0, 0.0, 10)


def test_gae_against_plain_recursion():
    rng = np.random.default_rng(0)
    T, N, gamma, lam = 7, 5, 0.99, 0.95
    r, v = rng.normal(size=(T, N)), rng.normal(size=(T, N))
    d = (rng.random((T, N)) < 0.3).astype(np.float64)
    nv, nd = rng.normal(size=N), (rng.random(N) < 0.3).astype(np.float64)
    adv = np.zeros((T, N))
    for n in range(N):
        last = 0.0
        for t in reversed(range(T)):
            nonterm = 1.0 - (nd[n] if t == T - 1 else d[t + 1, n])
            nextv = nv[n] if t == T - 1 else v[t + 1, n]
            delta = r[t, n] + gamma * nextv * nonterm - v[t, n]
            last = adv[t, n] = delta + gamma * lam * nonterm * last
    a, ret = compute_gae(*(torch.tensor(x) for x in (r, v, d, nv, nd)), gamma, lam)
    assert np.allclose(a.numpy(), adv) and np.allclose(ret.numpy(), adv + v)


def test_curriculum_walks_then_samples():
    rec = {"solved": set(), "unsolved": set(range(5))}
    processed = {0, 1}
    nxt, r1 = choose_next_state(processed, 5, rec, False, 0.25)
    assert (nxt, r1) == (2, False)
    processed |= {2, 3, 4}
    random.seed(0)
    nxt, r1 = choose_next_state(processed, 5, rec, False, 0.25)
    assert r1 is True and nxt in rec["unsolved"]  # nothing solved yet: always an unsolved state
    rec = {"solved": {1}, "unsolved": {0, 2, 3, 4}}
    random.seed(1)
    picks = [choose_next_state(processed, 5, rec, True, 0.25)[0] for _ in range(400)]
    frac_solved = sum(p == 1 for p in picks) / 400
    assert 0.15 < frac_solved < 0.35
    rec = {"solved": {0, 1, 2, 3, 4}, "unsolved": set()}
    assert choose_next_state(processed, 5, rec, True, 0.25)[0] in rec["solved"]


def test_legacy_path_encoding_and_literal_files(tmp_path):
    """data_files: the published greedy_search_paths.txt stores (action + 1, length) with a (0, length) root"""
    from ac_solver.search.miller_schupp.data_files import FILES, from_legacy_path, read_literals, to_legacy_path, write_literals

    path = [(-1, 7), (5, 7), (8, 7), (3, 5), (9, 3), (2, 2)]
    legacy = to_legacy_path(path)
    assert legacy == [(0, 7), (6, 7), (9, 7), (4, 5), (10, 3), (3, 2)] and from_legacy_path(legacy) == path
    rows = [[-1, 2, 1, -2, -2, 0, -1, 2, 0, 0], legacy]
    write_literals(rows, str(tmp_path / "x.txt"))
    assert read_literals(str(tmp_path / "x.txt")) == [rows[0], legacy]
    assert set(FILES) == {"all_presentations.txt", "greedy_solved_presentations.txt", "greedy_search_paths.txt", "bfs_solved_presentations.txt"}


def test_running_return_normalizer_matches_the_scalar_recursion():
    from ac_solver.agents.training import RunningReturnNormalizer

    rng = np.random.default_rng(1)
    T, N, gamma = 30, 3, 0.99
    rew = rng.normal(size=(T, N)) * 5
    term = rng.random((T, N)) < 0.2
    norm = RunningReturnNormalizer(N, gamma, torch.device("cpu"))
    got = np.stack([norm(torch.tensor(rew[t]), torch.tensor(term[t])).numpy() for t in range(T)])
    for n in range(N):  # gymnasium's RunningMeanStd.update_from_moments with a batch of one sample, per env
        mean, var, count, ret = 0.0, 1.0, 1e-4, 0.0
        for t in range(T):
            ret = ret * gamma * (1.0 - float(term[t, n])) + rew[t, n]
            delta, tot = ret - mean, count + 1.0
            new_mean = mean + delta / tot
            m2 = var * count + 0.0 + delta ** 2 * count * 1.0 / tot
            mean, var, count = new_mean, m2 / tot, tot
            assert np.isclose(got[t, n], rew[t, n] / np.sqrt(var + 1e-8), rtol=1e-12), (t, n)


def test_fused_policy_fragments_decode_back_to_the_layer():
    """agents/fused_policy._fragments writes a layer as the A operands of v_mfma_f32_32x32x16_bf16 in the order csrc/acx_policy.hip
    consumes them.  Decoded with the instruction's operand layout (lane = 32 h + row holds k = 8 h + j of the k-step) and the
    kernel's K order for layers fed by a hidden layer, the fragments give back scale * W (rounded to bf16) and the bias as hi + lo."""
    import torch

    from ac_solver.agents.fused_policy import TANH_SCALE, _fragments

    torch.manual_seed(5)
    for out, inp, rows, cols, hidden, scale in ((256, 50, 256, 64, False, TANH_SCALE), (256, 256, 256, 256, True, TANH_SCALE), (12, 256, 32, 256, True, 1.0)):
        w, b = torch.randn(out, inp), torch.randn(out)
        nks = cols // 16
        f = _fragments(w, b, rows, cols, hidden, scale).float().view(rows // 32, 1 + nks, 2, 32, 8)  # [ob][step][h][row][j]
        dense = torch.zeros(rows, cols)
        for ks in range(nks):
            for h in range(2):
                for j in range(8):
                    k = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3) if hidden else 16 * ks + 8 * h + j
                    dense[:, k] = f[:, 1 + ks, h, :, j].reshape(-1)
        want = torch.zeros(rows, cols)
        want[:out, :inp] = (w * scale).to(torch.bfloat16).float()
        assert torch.equal(dense, want)
        bias = f[:, 0, 0, :, 0].reshape(-1) + f[:, 0, 0, :, 1].reshape(-1)
        assert float((bias[:out] - b * scale).abs().max()) < 1e-4 and float(bias[out:].abs().sum()) == 0.0
        assert float(f[:, 0, 1].abs().max()) == 0.0 and float(f[:, 0, 0, :, 2:].abs().max()) == 0.0
        if hidden:  # every k-step of a hidden-fed layer contracts 16 distinct hidden units, all 256 exactly once
            seen = sorted(32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3) for ks in range(nks) for h in range(2) for j in range(8))
            assert seen == list(range(256))


def test_curriculum_object_equals_per_episode_choose_next_state():
    """training.Curriculum (no per-episode O(n_states) work) takes the same decisions and the same draws from `random` as
    calling choose_next_state once per finished episode, as the reference's loop does (training.py:199-221)."""
    import random

    from ac_solver.agents.training import Curriculum

    for stride in (1, 3):
        proc_a = set(range(0, 8 * stride, stride))
        rec_a = {"solved": set(), "unsolved": set(range(60))}
        proc_b, rec_b = set(proc_a), {"solved": set(), "unsolved": set(range(60))}
        cur = Curriculum(proc_b, 60, rec_b, 0.3, stride)
        ev = random.Random(1)
        events = [(ev.random() < 0.2, ev.randrange(60)) for _ in range(3000)]
        random.seed(5)
        start = random.getstate()
        out_a, r1 = [], False
        for done, s in events:
            if done and s in rec_a["unsolved"]:
                rec_a["unsolved"].remove(s)
                rec_a["solved"].add(s)
            nxt, r1 = choose_next_state(proc_a, 60, rec_a, r1, 0.3, stride=stride)
            proc_a.add(nxt)
            out_a.append(nxt)
        random.setstate(start)
        out_b = []
        for done, s in events:
            if done:
                cur.mark_solved(s)
            out_b.append(cur.next_state())
        assert out_a == out_b and proc_a == proc_b and rec_a == rec_b and r1 == cur.round1_complete


def test_py_curriculum_draws_are_pythons_own():
    """acxt_py_curriculum_draws (csrc/trainer/acx_trainer.cpp -> libacx_trainer.so, host only): CPython's random.uniform / random.choice restated on the state of the
    global generator -- the same decisions as the Python expression of choose_next_state, and the same generator state afterwards."""
    import ctypes as C
    import random

    from ac_solver import _acx
    from ac_solver.agents import _host

    for seed, n_solved, n_unsolved, p, n in ((1, 0, 17, 0.25, 300), (2, 5, 0, 0.25, 300), (3, 1, 1, 0.5, 500), (4, 640, 550, 0.25, 4000),
                                             (5, 3, 1190, 0.0, 1000), (6, 1189, 1, 1.0, 1000), (7, 2 ** 20 + 1, 2 ** 31, 0.3, 2000)):
        random.seed(seed)
        for _ in range(seed * 37):  # (a generator in mid-block)
            random.random()
        ver, internal, gauss = random.getstate()
        want = []
        for _ in range(n):
            if n_solved == 0 or (n_unsolved and random.uniform(0, 1) > p):
                want.append((0, random.choice(range(n_unsolved))))
            else:
                want.append((1, random.choice(range(n_solved))))
        after = random.getstate()
        mt = np.array(internal[:624], dtype=np.uint32)
        pos = C.c_int32(internal[624])
        which, index = np.empty(n, np.uint8), np.empty(n, np.int64)
        assert _host.lib.acxt_py_curriculum_draws(mt.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos), n, n_solved, n_unsolved, p,
                                                _acx.ptr(which, C.c_uint8), _acx.ptr(index, C.c_int64)) == 0
        assert list(zip(which.tolist(), index.tolist())) == want
        assert (ver, tuple(mt.tolist()) + (pos.value,), gauss) == after
    mt = np.zeros(624, np.uint32)
    pos = C.c_int32(624)
    one = np.empty(1, np.uint8), np.empty(1, np.int64)
    assert _host.lib.acxt_py_curriculum_draws(mt.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos), 1, 0, 0, 0.5, _acx.ptr(one[0], C.c_uint8),
                                            _acx.ptr(one[1], C.c_int64)) == _host.E_INVAL


def test_finished_episodes_of_a_step_at_once_equal_one_by_one():
    """Curriculum.finish_episodes (a rollout step's finished episodes in one call; the draws between two changes of the solved set in
    libacx) = mark_solved / next_state episode by episode: next states, the sets, the shortest-path callbacks in order, and the state
    of `random` afterwards.  Steps of 1 .. 3000 finished episodes, through the first round and past it, with and without new solves."""
    import random

    from ac_solver.agents.training import Curriculum

    for stride, n_states, p in ((1, 1190, 0.25), (2, 400, 0.6), (1, 60, 0.0)):
        ev = random.Random(11 + stride)
        steps = []
        for t in range(40):
            k = ev.choice((1, 3, 40, 47, 48, 49, 300, 3000))
            steps.append([(ev.random() < (0.3 if t % 3 else 0.002), ev.randrange(n_states)) for _ in range(k)])

        def run(batched, borrow=False):
            proc = set(range(0, 8 * stride, stride))
            rec = {"solved": set(), "unsolved": set(range(n_states))}
            cur = Curriculum(proc, n_states, rec, p, stride)
            random.seed(3)
            outs, calls = [], []
            for t, step in enumerate(steps):
                if borrow and t % 8 == 0:  # (the training loop: the generator's state stays in libacx for a whole rollout)
                    cur.end_borrow()
                    cur.begin_borrow()
                current, done = [s for _, s in step], np.array([d for d, _ in step])
                if batched:
                    outs.append(cur.finish_episodes(current, done, lambda k, s: calls.append((len(outs), k, s))))
                else:
                    nxt = []
                    for k, (d, s) in enumerate(step):
                        if d:
                            cur.mark_solved(s)
                            calls.append((len(outs), k, s))
                        nxt.append(cur.next_state())
                    outs.append(nxt)
            cur.end_borrow()
            return outs, calls, proc, rec, cur.round1_complete, cur.max_processed, random.getstate()

        a, b, c = run(False), run(True), run(True, borrow=True)
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and a[3] == b[3] and a[4:] == b[4:]
        assert a[0] == c[0] and a[1] == c[1] and a[2] == c[2] and a[3] == c[3] and a[4:] == c[4:]
        assert a[4] and len(a[3]["solved"]) > 5


def test_split_k_linear_has_the_gradients_of_nn_linear():
    """agents/ppo_agent.py: the large-batch Linear forms its weight gradient as a sum of per-chunk products (split-K); output and all
    three gradients must be those of torch's linear (float64: equal up to summation order).  The layers are nn.Linear subclasses with
    the same state_dict keys, and below the row threshold (or on the CPU) they ARE nn.Linear."""
    import torch

    from ac_solver.agents.ppo_agent import Linear, _LinearSplitK

    torch.manual_seed(0)
    for rows, fin, fout in ((8192 * 4, 50, 256), (4096 * 8, 256, 12), (2048 * 3, 256, 1)):
        x = torch.randn(rows, fin, dtype=torch.float64, requires_grad=True)
        w = torch.randn(fout, fin, dtype=torch.float64, requires_grad=True)
        b = torch.randn(fout, dtype=torch.float64, requires_grad=True)
        y = _LinearSplitK.apply(x, w, b)
        g = torch.randn_like(y)
        y.backward(g)
        got = (y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone())
        x.grad = w.grad = b.grad = None
        y2 = torch.nn.functional.linear(x, w, b)
        y2.backward(g)
        for a, c in zip(got, (y2.detach(), x.grad, w.grad, b.grad)):
            assert torch.allclose(a, c, rtol=1e-12, atol=1e-10)
    lin = Linear(4, 3)
    assert isinstance(lin, torch.nn.Linear) and set(lin.state_dict()) == {"weight", "bias"}
    xs = torch.randn(5, 4)
    assert torch.equal(lin(xs), torch.nn.functional.linear(xs, lin.weight, lin.bias))


def test_minibatch_order_is_the_references_shuffle_sequence():
    """training._MinibatchOrder: the permutations a thread computes beside the rollout are exactly what the reference's loop produces
    with the global generator it seeded at the top of the update (np.random.seed(s); np.random.shuffle(b_inds) once per epoch)."""
    import numpy as np
    import torch

    from ac_solver.agents.training import _MinibatchOrder

    for seed, batch, epochs in ((3, 1000, 1), (11, 4096, 3)):
        order = _MinibatchOrder(batch, epochs, torch.device("cpu"))
        order.prefetch(seed)
        order.prefetch(seed + 1)  # (the next update's, one ahead, in the other buffer)
        got = order.get(seed).numpy().copy()
        nxt = order.get(seed + 1).numpy()
        np.random.seed(seed + 1)
        b_next = np.arange(batch)
        np.random.shuffle(b_next)
        assert np.array_equal(nxt[0], b_next)
        np.random.seed(seed)
        b_inds = np.arange(batch)
        for e in range(epochs):
            np.random.shuffle(b_inds)
            assert np.array_equal(got[e], b_inds)


def test_libacx_shuffle_is_numpys_legacy_shuffle():
    """acxt_np_shuffle_epochs (csrc/trainer/acx_trainer.cpp -> libacx_trainer.so, a host utility without device work): np.random.seed(s) followed by one
    np.random.shuffle of the same array per epoch, bit for bit -- MT19937 seeded by an integer, Fisher-Yates from the top with
    masked rejection sampling.  Pinned against numpy itself, including seeds above 2^31 and sizes around powers of two."""
    import ctypes as C

    import numpy as np

    from ac_solver import _acx
    from ac_solver.agents import _host

    for seed, n, epochs in ((1, 10, 1), (7, 1000, 3), (123456, 100003, 2), (2**31 + 5, 4096, 1), (3, 1, 2), (42, 2, 4), (0, 65537, 1), (2**32 - 1, 65535, 2)):
        out = np.empty((epochs, n), np.int64)
        assert _host.lib.acxt_np_shuffle_epochs(seed, n, epochs, _acx.ptr(out, C.c_int64)) == 0
        np.random.seed(seed)
        a = np.arange(n)
        for e in range(epochs):
            np.random.shuffle(a)
            assert np.array_equal(out[e], a), (seed, n, e)
    assert _host.lib.acxt_np_shuffle_epochs(1, 0, 1, _acx.ptr(np.empty(1, np.int64), C.c_int64)) == _host.E_INVAL
