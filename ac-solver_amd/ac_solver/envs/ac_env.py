"""Andrews-Curtis environment for balanced presentations with two generators -- drop-in for
ac_solver/envs/ac_env.py (`ACEnvConfig`, `ACEnv`).

`ACEnv` keeps the reference's gymnasium surface (attributes, return conventions, exceptions) but
its state lives on the GPU: every `step` is one launch of libacx's environment kernel (the same
kernel that advances 65 536 environments at a time in `ACVecEnv`), every `reset` one upload.
"""
from dataclasses import dataclass, field
from typing import Union
import ctypes as C

import sys

import numpy as np

from ac_solver import _acx
from ac_solver._gym import Box, Discrete, Env
from ac_solver.envs.utils import is_array_valid_presentation


@dataclass
class ACEnvConfig:
    """Reference: ac_env.py:14-53.  Default state is the trivial presentation <x, y>."""

    initial_state: Union[np.ndarray, list] = field(default_factory=lambda: np.array([1, 0, 2, 0]))
    horizon_length: any = 1000
    use_supermoves: any = False

    def __post_init__(self):
        if isinstance(self.initial_state, list):
            self.initial_state = np.array(self.initial_state)
        if not isinstance(self.initial_state, np.ndarray):
            raise TypeError("initial_state must be a numpy array")
        if self.initial_state.ndim != 1:
            raise ValueError("initial_state must be a 1-dimensional array")
        if len(self.initial_state) % 2 != 0:
            raise ValueError("initial state must have even length")
        if not is_array_valid_presentation(self.initial_state):
            raise ValueError("initial state must be a valid presentation")

    @property
    def max_relator_length(self):
        return len(self.initial_state) // 2

    @classmethod
    def from_dict(cls, config_dict):
        defaults = cls()
        return cls(
            initial_state=np.array(config_dict.get("initial_state", defaults.initial_state)),
            horizon_length=config_dict.get("horizon_length", defaults.horizon_length),
            use_supermoves=config_dict.get("use_supermoves", defaults.use_supermoves),
        )


def normalise_supermoves(spec):
    """`use_supermoves` in its opt-in form -- a dict {action id: [base moves]} with ids 12, 13, ... or a list of move lists --
    -> list of move lists (supermove s is action 12 + s).  The reference only knows False (and raises for anything else,
    ac_env.py:62-65); True keeps raising here as well: it names no supermoves."""
    if isinstance(spec, dict):
        ids = sorted(spec)
        if ids != list(range(12, 12 + len(ids))):
            raise ValueError("supermove ids must be 12, 13, ... without gaps")
        seqs = [list(spec[k]) for k in ids]
    else:
        seqs = [list(q) for q in spec]
    for q in seqs:
        if not q or any(int(a) not in range(12) for a in q):
            raise ValueError("a supermove is a non-empty list of base moves 0..11")
    return [[int(a) for a in q] for q in seqs]


def upload_supermoves(handle_ptr, seqs, stream=None):
    moves = np.array([a for q in seqs for a in q], np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(q) for q in seqs])]).astype(np.int32)
    _acx.check(_acx.lib.acx_env_set_supermoves(handle_ptr, _acx.ptr(moves, C.c_uint8), _acx.ptr(offs, C.c_int32), len(seqs), stream), "acx_env_set_supermoves")


class _Handle:
    """Owns one acx_env; frees it with the Python object."""

    def __init__(self, n, L, horizon, flags):
        if L > 64:  # (the reference takes any length; here a relator is one 128-bit device word)
            raise ValueError(f"max_relator_length = {L} is not supported: the device environment keeps a relator in one 128-bit word (max_relator_length <= 64)")
        _acx.require_device()
        self.ptr = _acx.lib.acx_env_create(n, L, int(horizon), flags)
        if not self.ptr:
            raise _acx.AcxError(f"acx_env_create failed: {_acx.last_error()}")

    def __del__(self):
        # not during interpreter shutdown: the HIP runtime may already be tearing down (a hipFree then can block forever)
        if getattr(self, "ptr", None) and not sys.is_finalizing():
            _acx.lib.acx_env_destroy(self.ptr)
            self.ptr = None


class ACEnv(Env):
    """Reference: ac_env.py:56-134.  reward = max_reward * done - total_length * (1 - done);
    done <=> total length 2; truncated <=> count_steps >= horizon_length; no self-reset."""

    def __init__(self, config: ACEnvConfig = None):
        config = ACEnvConfig() if config is None else config
        self.n_gen = 2
        self.max_relator_length = config.max_relator_length
        self.initial_state = config.initial_state
        self.horizon_length = config.horizon_length
        # the reference's behaviour (ac_env.py:62-65) for every value it knows; a dict / list of move lists is this
        # build's opt-in extension (include/acx.h: acx_env_set_supermoves)
        self.supermoves = None
        if config.use_supermoves:
            if config.use_supermoves is True or not isinstance(config.use_supermoves, (dict, list, tuple)):
                raise NotImplementedError("ACEnv with supermoves is not yet implemented in this library.")
            self.supermoves = normalise_supermoves(config.use_supermoves)
        L = self.max_relator_length
        if L > 64:
            raise ValueError(f"max_relator_length = {L} is not supported: the device environment keeps a relator in one 128-bit word (max_relator_length <= 64)")
        if np.any(np.abs(self.initial_state) > self.n_gen):
            raise ValueError("ACEnv is a two-generator environment: letters must be in {+-1, +-2}")

        self.observation_space = Box(np.full(2 * L, -self.n_gen, dtype=np.int8), np.full(2 * L, self.n_gen, dtype=np.int8), dtype=np.int8)
        self.action_space = Discrete(12 + (len(self.supermoves) if self.supermoves else 0))
        self.max_reward = self.horizon_length * L * self.n_gen

        self._dtype = self.initial_state.dtype
        self._h = _Handle(1, L, self.horizon_length, 0)
        # host side of step(): the arrays acx_env_step_host reads and fills and their ctypes pointers, made once
        act, obs, rew = np.zeros(1, np.int64), np.zeros((1, 2 * L), np.int8), np.zeros(1, np.float32)
        done, trunc, err = np.zeros(1, np.uint8), np.zeros(1, np.uint8), np.zeros(1, np.uint8)
        self._io = (act, obs, done, trunc, err)
        self._io_keep = rew
        self._io_ptrs = (_acx.ptr(act, C.c_int64), _acx.ptr(obs, C.c_int8), _acx.ptr(rew, C.c_float), _acx.ptr(done, C.c_uint8),
                         _acx.ptr(trunc, C.c_uint8), None, 0, _acx.ptr(err, C.c_uint8), None)
        if self.supermoves:
            upload_supermoves(self._h.ptr, self.supermoves)
        row = _acx.as_i8_rows(self.initial_state.reshape(1, -1))
        _acx.check(_acx.lib.acx_env_set_initial(self._h.ptr, _acx.ptr(row, C.c_int8), None, 1, None), "acx_env_set_initial")
        self.state = np.copy(self.initial_state)
        self.count_steps = 0
        self.lengths = [int(np.count_nonzero(self.state[:L])), int(np.count_nonzero(self.state[L:]))]
        self.actions = []

    def step(self, action):
        self.actions += [action]
        L = self.max_relator_length
        assert action in range(0, self.action_space.n), f"Expect n to be in range 0-{self.action_space.n - 1} (both inclusive); got {action}"
        act, obs, done, trunc, err = self._io
        act[0] = int(action)
        _acx.check(_acx.lib.acx_env_step_host(self._h.ptr, *self._io_ptrs), "acx_env_step_host")  # one launch, one synchronisation
        if err[0]:
            # the reference's ACMove raised before state/lengths/count_steps were touched (ac_env.py:97);
            # the kernel left the device state and counter untouched as well
            raise (IndexError if err[0] == _acx.ERR_INDEX else AssertionError)(
                f"move {action} empties a relator of {self.state}: not a valid presentation")
        self.state = obs[0].astype(self._dtype)
        self.lengths = [int(np.count_nonzero(obs[0, :L])), int(np.count_nonzero(obs[0, L:]))]
        self.count_steps += 1
        is_done = bool(done[0])
        # the reference's own expression on exact Python ints (ac_env.py:101-102); the kernel's float32 reward is what the rollout
        # tensors carry and is only exact below 2^24 (max_reward = horizon_length * max_relator_length * 2 can be larger)
        reward = self.max_reward * is_done - sum(self.lengths) * (1 - is_done)
        return self.state, reward, is_done, bool(trunc[0]), ({"actions": self.actions.copy()} if is_done else {})

    def reset(self, *, seed=None, options=None):
        L = self.max_relator_length
        start = options["starting_state"] if options and "starting_state" in options else self.initial_state
        self.state = np.copy(start)
        row = _acx.as_i8_rows(np.asarray(self.state).reshape(1, -1))
        _acx.check(_acx.lib.acx_env_reset(self._h.ptr, _acx.ptr(row, C.c_int8), None, 1, None), "acx_env_reset")
        self.lengths = [int(np.count_nonzero(self.state[:L])), int(np.count_nonzero(self.state[L:]))]
        self.count_steps = 0
        self.actions = []
        return self.state, {}

    def render(self):
        pass
