"""`ACVecEnv` -- n Andrews-Curtis environments stepped by one HIP kernel launch, with observations,
rewards and done flags written straight into PyTorch-ROCm tensors.

This is the build's counterpart of what the reference's PPO loop consumes from
`gym.vector.SyncVectorEnv([make_env(...)])` (ac_solver/agents/environment.py:19-127,
agents/training.py:80-228): `reset()`, `step(actions)`, `single_observation_space`,
`single_action_space`, `envs[i].reset(options={"starting_state": ...})`, `envs[0].max_reward`,
`infos["final_info"][i]["actions"]`, `close()`.  gymnasium 0.28.1 vector semantics are followed:
a finished env is reset inside `step`, the returned observation is the reset one, and the terminal
observation / info travel in `infos["final_observation"]` / `infos["final_info"]` with boolean masks
`_final_observation` / `_final_info`.

Differences by design: observations / rewards / flags are device tensors (no host round trip per
step); reward clipping (`TransformReward(np.clip)`, environment.py:48-52) is fused into the kernel;
`infos` is only materialised when `final_info=True` (it needs one tiny device->host copy per step).
"""
import ctypes as C

import numpy as np

from ac_solver import _acx
from ac_solver._gym import Box, Discrete
from ac_solver.envs.ac_env import _Handle
from ac_solver.envs.utils import are_rows_valid_presentations, is_array_valid_presentation

_ACTION_DTYPES = None


def _torch():
    import torch

    return torch


class _EnvView:
    """What `envs.envs[i]` exposes to the reference's training loop (training.py:224,233)."""

    def __init__(self, vec, i):
        self._vec, self._i = vec, i

    @property
    def max_reward(self):
        return self._vec.max_reward

    @property
    def state(self):
        return self._vec.get_states([self._i])[0]

    @property
    def lengths(self):
        L = self._vec.max_relator_length
        s = self.state
        return [int(np.count_nonzero(s[:L])), int(np.count_nonzero(s[L:]))]

    @property
    def count_steps(self):
        return int(self._vec.get_counts([self._i])[0])

    @property
    def actions(self):
        return self._vec.get_actions(self._i)

    def reset(self, *, seed=None, options=None):
        start = options["starting_state"] if options and "starting_state" in options else None
        self._vec.reset_envs([self._i], None if start is None else np.asarray(start)[None])
        return self.state, {}


class _EnvList:
    def __init__(self, vec):
        self._vec = vec

    def __len__(self):
        return self._vec.num_envs

    def __getitem__(self, i):
        if not -self._vec.num_envs <= i < self._vec.num_envs:
            raise IndexError(i)
        return _EnvView(self._vec, i % self._vec.num_envs)


class ACVecEnv:
    def __init__(self, initial_states, horizon_length=1000, obs_dtype="int8", clip_rewards=None, record_actions=True,
                 final_info=True, device=None, reward_dtype="float32", supermoves=None):
        """`reward_dtype="float64"`: `step` (without `out`) returns the rewards as float64, the dtype gymnasium's
        SyncVectorEnv hands to the reference's training loop; the kernel always writes float32 (exact: rewards are integers
        below 2^24 in magnitude, or clipped)."""
        torch = _torch()
        self._reward_f64 = {"float32": False, "float64": True}[reward_dtype]
        _acx.require_device()
        states = np.asarray(initial_states)
        if states.ndim != 2 or states.shape[1] % 2:
            raise ValueError("initial_states must be [num_envs, 2 * max_relator_length]")
        if not are_rows_valid_presentations(states).all():  # is_array_valid_presentation on every row, vectorised
            raise ValueError("initial state must be a valid presentation")  # ACEnvConfig.__post_init__
        self.num_envs, width = states.shape
        self.max_relator_length = L = width // 2
        self.horizon_length = int(horizon_length)
        self.max_reward = self.horizon_length * L * 2
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.obs_torch_dtype = {"int8": torch.int8, "float32": torch.float32}[obs_dtype]
        self._obs_code = _acx.I8 if obs_dtype == "int8" else _acx.F32
        self.clip = (0.0, 0.0) if clip_rewards is None else (float(clip_rewards[0]), float(clip_rewards[1]))
        if clip_rewards is not None:
            assert self.clip[0] < self.clip[1], "min_rew must be less than max_rew"
        self.final_info = final_info
        self.record_actions = record_actions
        self.single_observation_space = Box(np.full(2 * L, -2, np.int8), np.full(2 * L, 2, np.int8), dtype=np.int8)
        self.supermoves = None
        if supermoves:  # opt-in: action 12 + s = the s-th list of base moves as one step
            from ac_solver.envs.ac_env import normalise_supermoves

            self.supermoves = normalise_supermoves(supermoves)
        n_actions = 12 + (len(self.supermoves) if self.supermoves else 0)
        self.single_action_space = Discrete(n_actions)
        self.observation_space = Box(np.full((self.num_envs, 2 * L), -2, np.int8), np.full((self.num_envs, 2 * L), 2, np.int8), dtype=np.int8)
        self.action_space = Discrete(n_actions)
        with torch.cuda.device(self.device):
            self._h = _Handle(self.num_envs, L, self.horizon_length, _acx.ENV_RECORD_ACTIONS if record_actions else 0)
            rows = _acx.as_i8_rows(states)
            _acx.check(_acx.lib.acx_env_set_initial(self._h.ptr, _acx.ptr(rows, C.c_int8), None, self.num_envs, None), "acx_env_set_initial")
            if self.supermoves:
                from ac_solver.envs.ac_env import upload_supermoves

                upload_supermoves(self._h.ptr, self.supermoves)
        self.initial_states = rows
        self.envs = _EnvList(self)
        n = self.num_envs
        self._obs = torch.empty((n, 2 * L), dtype=self.obs_torch_dtype, device=self.device)
        self._final = torch.empty((n, 2 * L), dtype=self.obs_torch_dtype, device=self.device)
        self._rew = torch.empty(n, dtype=torch.float32, device=self.device)
        self._done = torch.empty(n, dtype=torch.bool, device=self.device)
        self._trunc = torch.empty(n, dtype=torch.bool, device=self.device)

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return _torch().cuda.current_stream(self.device).cuda_stream

    def _actions_tensor(self, actions):
        torch = _torch()
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(np.asarray(actions), device=self.device)
        if actions.device != self.device:
            actions = actions.to(self.device)
        code = {torch.uint8: _acx.U8, torch.int32: _acx.I32, torch.int64: _acx.I64, torch.int8: _acx.I8}.get(actions.dtype)
        if code is None:
            actions, code = actions.to(torch.int64), _acx.I64
        if actions.numel() != self.num_envs:
            raise ValueError(f"expected {self.num_envs} actions, got {actions.numel()}")
        return actions.contiguous(), code

    def _raise_on_errors(self):
        err = np.empty(self.num_envs, np.uint8)
        _acx.check(_acx.lib.acx_env_get_errors(self._h.ptr, _acx.ptr(err, C.c_uint8), 1, self._stream()))
        if err.any():
            i = int(np.flatnonzero(err)[0])
            raise (IndexError if err[i] == _acx.ERR_INDEX else AssertionError)(
                f"env {i}: the move empties a relator (the reference's ACMove raises here)")

    # ------------------------------------------------------------------ gymnasium-style surface
    def reset(self, *, seed=None, options=None):
        """All envs back to their initial states.  -> (obs tensor [n, 2L], {})"""
        _acx.check(_acx.lib.acx_env_reset(self._h.ptr, None, None, self.num_envs, self._stream()), "acx_env_reset")
        return self.observe(), {}

    def observe(self, out=None):
        out = self._obs if out is None else out
        _acx.check(_acx.lib.acx_env_observe(self._h.ptr, out.data_ptr(), self._obs_code, self._stream()), "acx_env_observe")
        return out

    def step(self, actions, out=None, check_errors=True):
        """One ACEnv.step on every env.  `out` may supply preallocated (obs, reward, terminated, truncated)
        tensors, e.g. slices of the PPO rollout buffers.  -> (obs, reward, terminated, truncated, infos)"""
        act, code = self._actions_tensor(actions)
        obs, rew, done, trunc = (self._obs, self._rew, self._done, self._trunc) if out is None else out
        _acx.check(_acx.lib.acx_env_step(self._h.ptr, act.data_ptr(), code, obs.data_ptr(), self._obs_code, rew.data_ptr(), self.clip[0],
                                         self.clip[1], done.data_ptr(), trunc.data_ptr(), self._final.data_ptr() if self.final_info else None,
                                         1, self._stream()), "acx_env_step")
        infos = {}
        if self.final_info:
            fin = (done | trunc).cpu().numpy()
            if check_errors:
                self._raise_on_errors()
            if fin.any():
                n = self.num_envs
                final_obs = np.full(n, None, dtype=object)
                final_info = np.full(n, None, dtype=object)
                done_h = done.cpu().numpy()
                rows = self._final[_torch().as_tensor(np.flatnonzero(fin), device=self.device)].cpu().numpy()
                for k, i in enumerate(np.flatnonzero(fin)):
                    final_obs[i] = rows[k]
                    final_info[i] = {"actions": self.get_actions(int(i), finished=True)} if (done_h[i] and self.record_actions) else {}
                infos = {"final_observation": final_obs, "_final_observation": fin.copy(), "final_info": final_info, "_final_info": fin.copy()}
        if self._reward_f64 and out is None:
            rew = rew.to(_torch().float64)
        return obs, rew, done, trunc, infos

    def rollout(self, tape, reward=None, terminated=None, truncated=None, autoreset=True):
        """T fused steps from an action tape [T, n] (uint8 device tensor); optional [T, n] output tensors."""
        torch = _torch()
        assert tape.dtype == torch.uint8 and tape.is_contiguous() and tape.shape[1] == self.num_envs
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        _acx.check(_acx.lib.acx_env_rollout(self._h.ptr, tape.data_ptr(), tape.shape[0], p(reward), self.clip[0], self.clip[1], p(terminated),
                                            p(truncated), int(autoreset), self._stream()), "acx_env_rollout")

    def close(self):
        self._h = None

    # ------------------------------------------------------------------ host-side access
    def reset_envs(self, indices, states=None):
        """ACEnv.reset for some envs: states None -> initial state, else options={'starting_state': row}."""
        idx = np.ascontiguousarray(indices, np.int64)
        rows = None if states is None else _acx.as_i8_rows(states)
        _acx.check(_acx.lib.acx_env_reset(self._h.ptr, None if rows is None else _acx.ptr(rows, C.c_int8), _acx.ptr(idx, C.c_int64), len(idx),
                                          self._stream()), "acx_env_reset")

    def reset_envs_device(self, indices, states, rowerr=None):
        """reset_envs from device tensors, queued on the current stream without a copy or a synchronisation: `indices` int64 [k]
        (in range -- not checked), `states` int8 [k, 2L] rows (or None -> initial state).  -> the uint8 [k] tensor that will hold a
        non-zero byte for every row that is not a presentation over {+-1, +-2} (such an env keeps its state)."""
        torch = _torch()
        assert indices.dtype == torch.int64 and indices.is_contiguous()
        k = indices.numel()
        if states is not None:
            assert states.dtype == torch.int8 and states.is_contiguous() and tuple(states.shape) == (k, 2 * self.max_relator_length)
        if rowerr is None:
            rowerr = torch.empty(max(k, 1), dtype=torch.uint8, device=indices.device)
        _acx.check(_acx.lib.acx_env_reset_device(self._h.ptr, None if states is None else states.data_ptr(), indices.data_ptr(), k, rowerr.data_ptr(),
                                                 self._stream()), "acx_env_reset_device")
        return rowerr[:k]

    def _get(self, indices):
        idx = np.ascontiguousarray(np.arange(self.num_envs) if indices is None else indices, np.int64)
        st = np.empty((len(idx), 2 * self.max_relator_length), np.int8)
        ln = np.empty((len(idx), 2), np.int32)
        ct = np.empty(len(idx), np.int32)
        _acx.check(_acx.lib.acx_env_get(self._h.ptr, _acx.ptr(idx, C.c_int64), len(idx), _acx.ptr(st, C.c_int8), _acx.ptr(ln, C.c_int32),
                                        _acx.ptr(ct, C.c_int32), self._stream()), "acx_env_get")
        return st, ln, ct

    def get_states(self, indices=None):
        return self._get(indices)[0]

    def get_lengths(self, indices=None):
        return self._get(indices)[1]

    def get_counts(self, indices=None):
        return self._get(indices)[2]

    def get_actions(self, i, finished=False):
        """info['actions'] of env i: since its last reset, or (finished=True) of the episode the last step ended."""
        cap = max(self.horizon_length, 16)
        buf = np.empty(cap, np.int32)
        n = C.c_int64()
        _acx.check(_acx.lib.acx_env_get_actions(self._h.ptr, i, int(finished), _acx.ptr(buf, C.c_int32), cap, C.byref(n), self._stream()),
                   "acx_env_get_actions")
        return buf[: n.value].tolist()
