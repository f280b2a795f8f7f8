"""Pure Python / NumPy restatement of the reference's hot path -- TEST INFRASTRUCTURE and the third CPU-baseline leg of
bench.py ("what a user of the reference gets": one interpreter, NumPy slicing per move).

Written from the lane-level specification in SURVEY.md appendix B, function for function with the reference's call graph
(each function cites the file:line it stands for); pinned against the same golden fixtures as the C oracle by
tests/test_oracle_golden.py::test_numpy_restatement_*.  Only tests/ and bench.py's cpu_baseline leg import this module.
"""
import heapq
from collections import deque

import numpy as np


def _word(presentation, L, k):
    half = presentation[k * L:(k + 1) * L]
    return half[half != 0]


def simplify_relator(relator, max_relator_length, cyclical=False, padded=True):
    """envs/utils.py:175-245: free reduction (delete an adjacent inverse pair, step back), optional cyclic reduction."""
    relator = np.array(relator)
    assert (relator[len(relator[relator != 0]):] == 0).all(), "expect all zeros to be at the right end"
    rel = relator[relator != 0]
    pos = 0
    while pos < len(rel) - 1:
        if rel[pos] == -rel[pos + 1]:
            rel = np.delete(rel, [pos, pos + 1])
            pos = max(pos - 1, 0)
        else:
            pos += 1
    if cyclical and len(rel) > 0:
        strip = 0
        while strip < len(rel) // 2 and rel[strip] == -rel[len(rel) - 1 - strip]:  # stops before the middle of the word
            strip += 1
        if strip:
            rel = rel[strip:len(rel) - strip]
    n = len(rel)
    if padded:
        assert max_relator_length >= n, "Increase max length! Word length is bigger than maximum allowed length."
        rel = np.pad(rel, (0, max_relator_length - n))
    return rel, n


def is_array_valid_presentation(array):
    """envs/utils.py:9-60"""
    array = np.asarray(array)
    if array.ndim != 1 or len(array) == 0 or len(array) % 2:
        return False
    L = len(array) // 2
    for k in (0, 1):
        half = array[k * L:(k + 1) * L]
        n = int(np.count_nonzero(half))
        if n == 0 or (half[n:] != 0).any() or (half[:n] == 0).any():
            return False
    return True


def simplify_presentation(presentation, max_relator_length, lengths_of_words, cyclical=True):
    """envs/utils.py:248-280: both relators through simplify_relator"""
    presentation = np.array(presentation)
    assert is_array_valid_presentation(presentation), f"{presentation} is not a valid presentation"
    lengths_of_words = list(lengths_of_words)
    for k in (0, 1):
        rel, n = simplify_relator(presentation[k * max_relator_length:(k + 1) * max_relator_length], max_relator_length, cyclical=cyclical,
                                  padded=True)
        presentation[k * max_relator_length:(k + 1) * max_relator_length] = rel
        lengths_of_words[k] = n
    return presentation, lengths_of_words


def concatenate_relators(presentation, max_relator_length, i, j, sign, lengths):
    """envs/ac_moves.py:4-76: r_i <- r_i r_j^sign with the junction cancelled, unless the result is too long"""
    assert i in (0, 1) and j in (0, 1) and i == 1 - j
    assert sign in (1, -1)
    L = max_relator_length
    presentation = np.array(presentation)
    lengths = list(lengths)
    w1 = _word(presentation, L, i)
    w2 = _word(presentation, L, j)
    if sign == -1:
        w2 = -w2[::-1]
    acc = 0
    while acc < min(len(w1), len(w2)) and w1[-1 - acc] == -w2[acc]:
        acc += 1
    new = len(w1) + len(w2) - 2 * acc
    if new <= L:
        lengths[i] = new
        presentation[i * L:i * L + len(w1) - acc] = w1[:len(w1) - acc]
        presentation[i * L + len(w1) - acc:i * L + new] = w2[acc:]
        presentation[i * L + new:(i + 1) * L] = 0
    return presentation, lengths


def conjugate(presentation, max_relator_length, i, j, sign, lengths):
    """envs/ac_moves.py:79-156: r_i <- g r_i g^-1, g = sign * (j + 1), with the cancellations at both ends"""
    assert i in (0, 1) and j in (1, 2)
    assert sign in (1, -1)
    L = max_relator_length
    presentation = np.array(presentation)
    lengths = list(lengths)
    rel = presentation[i * L:(i + 1) * L]
    n = int(np.count_nonzero(rel))
    g = sign * j
    start_cancel = 1 if rel[0] == -g else 0
    end_cancel = 1 if rel[n - 1] == g else 0
    new = n + 2 - 2 * (start_cancel + end_cancel)
    if new <= L:
        lengths[i] = new
        body = rel[start_cancel:n - end_cancel].copy()
        out = np.concatenate(([g] if not start_cancel else [], body, [-g] if not end_cancel else [])).astype(presentation.dtype)
        presentation[i * L:i * L + new] = out
        presentation[i * L + new:(i + 1) * L] = 0
    return presentation, lengths


_MOVES = [("cat", 1, 0, 1), ("cat", 0, 1, -1), ("cat", 1, 0, -1), ("cat", 0, 1, 1),
          ("conj", 1, 1, -1), ("conj", 0, 2, -1), ("conj", 1, 2, -1), ("conj", 0, 1, 1),
          ("conj", 1, 1, 1), ("conj", 0, 2, 1), ("conj", 1, 2, 1), ("conj", 0, 1, -1)]


def ACMove(move_id, presentation, max_relator_length, lengths, cyclical=True):
    """envs/ac_moves.py:159-231: one of the twelve moves, then simplify_presentation"""
    assert 0 <= move_id <= 11
    kind, i, j, sign = _MOVES[move_id]
    fn = concatenate_relators if kind == "cat" else conjugate
    presentation, lengths = fn(presentation, max_relator_length, i, j, sign, lengths)
    return simplify_presentation(presentation, max_relator_length, lengths, cyclical=cyclical)


class Env:
    """envs/ac_env.py:57-134 (ACEnv.step / reset) on top of ACMove"""

    def __init__(self, initial_state, horizon_length=1000):
        self.initial_state = np.array(initial_state, dtype=np.int8)
        self.L = len(self.initial_state) // 2
        self.horizon_length = horizon_length
        self.max_reward = horizon_length * self.L * 2
        self.reset()

    def reset(self):
        self.state = self.initial_state.copy()
        self.lengths = [int(np.count_nonzero(self.state[:self.L])), int(np.count_nonzero(self.state[self.L:]))]
        self.count_steps = 0
        return self.state

    def step(self, action):
        self.state, self.lengths = ACMove(int(action), self.state, self.L, self.lengths, cyclical=True)
        done = sum(self.lengths) == 2
        reward = self.max_reward * done - sum(self.lengths) * (1 - done)
        self.count_steps += 1
        return self.state, reward, done, self.count_steps >= self.horizon_length


def bfs(presentation, max_nodes_to_explore=10000, cyclically_reduce_after_moves=False):
    """search/breadth_first.py:15-97"""
    presentation = np.array(presentation, dtype=np.int8)
    L = len(presentation) // 2
    lengths = [int(np.count_nonzero(presentation[:L])), int(np.count_nonzero(presentation[L:]))]
    start = tuple(presentation.tolist())
    seen = {start}
    queue = deque([(start, tuple(lengths), [(-1, sum(lengths))])])
    while queue:
        state, lens, path = queue.popleft()
        for action in range(12):
            new_state, new_lens = ACMove(action, np.array(state, dtype=np.int8), L, list(lens), cyclical=cyclically_reduce_after_moves)
            total = sum(new_lens)
            if total == 2:
                return True, path + [(action, total)]
            key = tuple(new_state.tolist())
            if key not in seen:
                seen.add(key)
                queue.append((key, tuple(new_lens), path + [(action, total)]))
        if len(seen) >= max_nodes_to_explore:
            break
    return False, None


def greedy_search(presentation, max_nodes_to_explore=10000, cyclically_reduce_after_moves=False):
    """search/greedy.py:15-121: priority (total length, path length, state tuple)"""
    presentation = np.array(presentation, dtype=np.int8)
    L = len(presentation) // 2
    lengths = [int(np.count_nonzero(presentation[:L])), int(np.count_nonzero(presentation[L:]))]
    start = tuple(presentation.tolist())
    seen = {start}
    heap = [(sum(lengths), 0, start, tuple(lengths), [(-1, sum(lengths))])]
    path, total = None, 0
    while heap:
        _, depth, state, lens, path = heapq.heappop(heap)
        for action in range(12):
            new_state, new_lens = ACMove(action, np.array(state, dtype=np.int8), L, list(lens), cyclical=cyclically_reduce_after_moves)
            total = sum(new_lens)
            if total == 2:
                return True, path + [(action, total)]
            key = tuple(new_state.tolist())
            if key not in seen:
                seen.add(key)
                heapq.heappush(heap, (total, depth + 1, key, tuple(new_lens), path + [(action, total)]))
        if len(seen) >= max_nodes_to_explore:
            break
    return False, (path + [(11, total)]) if path is not None else None
