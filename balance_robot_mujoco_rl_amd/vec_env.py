"""BalanceVecEnv -- the batched environments behind Stable-Baselines3's VecEnv API.

The reference trains with `PPO("MlpPolicy", env=gym.make(id))` (src/sb_rl.py:63-71, :500, :552 of the reference);
SB3 wraps that single env in a DummyVecEnv.  An object that already IS a VecEnv is used as-is, which is the drop-in
seam: `PPO("MlpPolicy", env=BalanceVecEnv("Env03-v2", 65536))`.

Contract mirrored (SB3 2.x `VecEnv`):
  reset() -> obs[N,6] f32 ; step_async(actions[N,2]) ; step_wait() -> (obs, rewards[N] f32, dones[N] bool, infos)
  dones = terminated | truncated; a done env is already reset and obs is the first observation of its new episode;
  infos[i]["terminal_observation"], infos[i]["TimeLimit.truncated"], infos[i]["episode"] = {"r","l","t"} (what
  the reference's Monitor wrapper provides, sb_rl.py:501).
Envs shard over GPUs by contiguous index ranges, one BatchedSim (one C-ABI handle, one stream) per device; there
is no collective on the step path, only the host-side concatenation of the per-device outputs.
"""
import time

import numpy as np

from .registry import spec


class LazyInfos(list):
    """the `infos` list of one step.  Every env that did not finish shares ONE empty dict; the dicts of the finished
    envs ("terminal_observation", "TimeLimit.truncated", "episode") are built on the first element access / iteration,
    from arrays that are always available eagerly as `done_indices`, `terminal_observations`, `time_limit_truncated`,
    `episode_returns`, `episode_lengths` (what an array-based consumer should read instead: at 65,536 envs under a random
    policy ~2,300 episodes end per step, and 2,300 Python dicts cost more than the GPU step)."""

    __slots__ = ("done_indices", "terminal_observations", "time_limit_truncated", "episode_returns", "episode_lengths",
                 "_t", "_pending")

    def __init__(self, n, empty, idx, tob, tl, ret, length, t):
        super().__init__([empty] * n)
        self.done_indices, self.terminal_observations, self.time_limit_truncated = idx, tob, tl
        self.episode_returns, self.episode_lengths, self._t = ret, length, t
        self._pending = idx.size > 0

    def _materialise(self):
        if self._pending:
            self._pending = False
            rows, tl = list(self.terminal_observations), self.time_limit_truncated.tolist()
            rets, lens, t = self.episode_returns.tolist(), self.episode_lengths.tolist(), self._t
            for k, i in enumerate(self.done_indices.tolist()):
                list.__setitem__(self, i, {"terminal_observation": rows[k], "TimeLimit.truncated": tl[k],
                                           "episode": {"r": rets[k], "l": lens[k], "t": t}})

    def __getitem__(self, i):
        self._materialise()
        return list.__getitem__(self, i)

    def __iter__(self):
        self._materialise()
        return list.__iter__(self)

    def __eq__(self, other):
        self._materialise()
        return list.__eq__(self, other)

    def __reduce__(self):
        self._materialise()
        return (list, (list(list.__iter__(self)),))

    # every other way into the list goes through the C slots of `list`, which would see (or overwrite) the placeholder of a
    # finished env: materialise first.  Unfinished envs still SHARE one empty dict -- a wrapper that wants to write into
    # infos[i] of an unfinished env must assign a fresh dict (infos[i] = {...}), as SB3's own wrappers do.
    def _m(name):
        def f(self, *a, **k):
            self._materialise()
            return getattr(list, name)(self, *a, **k)
        f.__name__ = name
        return f

    for _n in ("__setitem__", "__delitem__", "__contains__", "__reversed__", "__add__", "__mul__", "__rmul__", "__len__", "__ne__",
               "copy", "index", "count", "sort", "reverse", "pop", "insert", "append", "extend", "remove", "__iadd__", "__imul__"):
        locals()[_n] = _m(_n)
    del _m, _n

try:  # SB3 is optional: the class is a real VecEnv subclass when it is importable
    from stable_baselines3.common.vec_env import VecEnv as _VecEnvBase  # type: ignore
    _HAVE_SB3 = True
except Exception:  # pragma: no cover - SB3 is not installed in the build container
    _HAVE_SB3 = False

    class _VecEnvBase:  # structurally identical minimal base
        def __init__(self, num_envs, observation_space, action_space):
            self.num_envs = num_envs
            self.observation_space = observation_space
            self.action_space = action_space
            self.render_mode = None
            self.reset_infos = [{} for _ in range(num_envs)]
            self._seeds = [None for _ in range(num_envs)]
            self._options = [{} for _ in range(num_envs)]

        def step(self, actions):
            self.step_async(actions)
            return self.step_wait()

try:
    from gymnasium.spaces import Box as _Box  # type: ignore
except Exception:  # pragma: no cover

    class _Box:
        """tiny stand-in for gymnasium.spaces.Box (metadata only; the reference never enforces its spaces)"""

        def __init__(self, low, high, dtype=np.float32):
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
            self.shape = self.low.shape
            self.dtype = np.dtype(dtype)

        def sample(self):
            return np.random.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


def make_spaces():
    """spaces of envs/RobotBaseEnv.py:50-54 (observation) and :74-85 (action) of the reference"""
    hi = np.array([2 * np.pi, 2 * np.pi, 1.0, 1.0, 1.0, 1.0], dtype=np.float32)
    return _Box(-hi, hi, dtype=np.float32), _Box(-np.ones(2, np.float32), np.ones(2, np.float32), dtype=np.float32)


def shard_ranges(num_envs, num_shards):
    """contiguous, near-equal env-index ranges [(start, count), ...]"""
    base, rem = divmod(num_envs, num_shards)
    out, start = [], 0
    for s in range(num_shards):
        cnt = base + (1 if s < rem else 0)
        out.append((start, cnt))
        start += cnt
    return [r for r in out if r[1] > 0]


class BalanceVecEnv(_VecEnvBase):
    metadata = {"render_modes": [], "render_fps": 200}

    def __init__(self, env_id, num_envs, devices=None, seed=0, obs_noise=None, max_episode_steps=0, env_index_base=0,
                 sparse_infos=True, _sims=None):
        self.spec_ = spec(env_id)
        self.env_id = env_id
        self.render_mode = None
        self._sparse = sparse_infos
        if _sims is None:  # product path: HIP only
            import torch
            from .sim import BatchedSim, BrsError
            if not torch.cuda.is_available():
                raise BrsError("BalanceVecEnv needs a HIP device (no CPU fallback)")
            if devices is None:
                devices = [torch.cuda.current_device()]
            self._torch = torch
            self._sims = [BatchedSim(env_id, cnt, device=dev, seed=seed, env_index_base=env_index_base + start,
                                     auto_reset=True, obs_noise=obs_noise, max_episode_steps=max_episode_steps)
                          for dev, (start, cnt) in zip(devices, shard_ranges(num_envs, len(devices)))]
            # a device listed more than once (devices=[0, 0]) gets one handle per entry, each on its own stream, so that
            # the shards' kernels overlap on that GPU; a device listed once uses the current stream
            repeated = {d for d in devices if list(devices).count(d) > 1}
            self._streams = [torch.cuda.Stream(torch.device("cuda", d)) if d in repeated else None for d in devices][:len(self._sims)]
            for d in repeated:  # the handles' buffers were zero-filled on the default stream: finish that first
                torch.cuda.synchronize(torch.device("cuda", d))
        else:  # tests inject stand-ins with the same surface
            self._torch = None
            self._sims = list(_sims)
            self._streams = [None] * len(self._sims)
        self._ranges = []
        start = 0
        for s in self._sims:
            self._ranges.append((start, s.n))
            start += s.n
        assert start == num_envs, "shards must cover num_envs"
        obs_space, act_space = make_spaces()
        self._ep_ret = np.zeros(num_envs, np.float64)
        self._ep_len = np.zeros(num_envs, np.int64)
        self._t0 = time.time()
        self._actions = None
        self._empty = {}
        super().__init__(num_envs, obs_space, act_space)

    # ------------------------------------------------------------------ helpers
    def _to_numpy(self, t):
        return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)

    def _sync_streams(self):
        for st in self._streams:
            if st is not None:
                st.synchronize()

    def _gather(self, parts):
        return parts[0] if len(parts) == 1 else np.concatenate(parts, axis=0)

    # ------------------------------------------------------------------ VecEnv API
    def _on(self, st):
        import contextlib
        return contextlib.nullcontext() if st is None else self._torch.cuda.stream(st)

    def reset(self):
        outs = []
        if self._torch is None:  # injected stand-ins (tests)
            outs = [self._to_numpy(s.reset()).copy() for s in self._sims]
        else:
            for s, st in zip(self._sims, self._streams):
                with self._on(st):
                    outs.append(s.reset_host().copy())
        obs = self._gather(outs)
        self._ep_ret[:] = 0
        self._ep_len[:] = 0
        self.reset_infos = [{} for _ in range(self.num_envs)]
        return obs.astype(np.float32, copy=False)

    def step_async(self, actions):
        a = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, 2)
        self._pending = []
        for s, st, (start, cnt) in zip(self._sims, self._streams, self._ranges):  # enqueue on every device before waiting on any
            if self._torch is None:
                self._pending.append(s.step(a[start:start + cnt]))
            else:  # pinned H2D of the actions, the step kernel and ONE packed pinned D2H, all asynchronous on the shard's stream
                with self._on(st):
                    s.step_host_async(a[start:start + cnt])

    def _collect(self):
        """-> obs, rew, term, trunc, tob : fresh host arrays for obs / rew / flags, tob possibly a view of pinned staging"""
        if self._torch is None:
            obs, rew, term, trunc, tob = ([] for _ in range(5))
            for o, r, te, tr, to in self._pending:
                obs.append(self._to_numpy(o).copy()); rew.append(self._to_numpy(r).copy())
                term.append(self._to_numpy(te).astype(bool)); trunc.append(self._to_numpy(tr).astype(bool))
                tob.append(self._to_numpy(to))
            return self._gather(obs), self._gather(rew), self._gather(term), self._gather(trunc), self._gather(tob)
        views = [s.step_host_wait() for s in self._sims]  # (obs, tob, rew, term, trunc) views of each shard's pinned mirror
        if len(views) == 1:
            o, to, r, te, tr = views[0]
            return o.copy(), r.copy(), te.astype(bool), tr.astype(bool), to
        cat = lambda k: np.concatenate([v[k] for v in views], axis=0)
        return cat(0), cat(2), cat(3).astype(bool), cat(4).astype(bool), cat(1)

    def step_wait(self):
        obs, rew, term, trunc, tob = self._collect()
        dones = term | trunc
        self._ep_ret += rew
        self._ep_len += 1
        idx = np.flatnonzero(dones)
        now = round(time.time() - self._t0, 6)
        lazy = LazyInfos(self.num_envs, self._empty, idx, tob[idx], (trunc[idx] & ~term[idx]), self._ep_ret[idx], self._ep_len[idx], now)
        if idx.size:
            self._ep_ret[idx] = 0
            self._ep_len[idx] = 0
        if self._sparse:
            infos = lazy
        else:  # plain list of distinct dicts (slow at large N; kept for consumers that mutate infos)
            infos = [dict(d) for d in lazy]
        return obs, rew, dones, infos

    def close(self):
        for s in self._sims:
            s.close()

    def seed(self, seed=None):
        # streams are keyed at construction (Philox(seed, global env index)); nothing to reseed per call
        return [seed for _ in range(self.num_envs)]

    def _indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        if isinstance(indices, int):
            return [indices]
        return indices

    def get_attr(self, attr_name, indices=None):
        val = {"render_mode": None, "spec": self.spec_, "env_id": self.env_id,
               "max_episode_steps": self.spec_.max_episode_steps}.get(attr_name, getattr(self, attr_name, None))
        return [val for _ in self._indices(indices)]

    def set_attr(self, attr_name, value, indices=None):
        raise AttributeError("per-env attributes are fixed at construction in the batched simulator")

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        raise NotImplementedError(f"env_method({method_name!r}): envs are lanes of a GPU kernel, not Python objects")

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False for _ in self._indices(indices)]

    def get_images(self):
        return [None for _ in range(self.num_envs)]

    def render(self, mode=None):
        return None  # no GL on the GPU box; the reference's viewer overlays are out of scope


class BalanceVectorEnv:
    """Gymnasium-style vector API over the same simulators: reset(seed) -> (obs, infos),
    step(a) -> (obs, rewards, terminated, truncated, infos) with same-step auto-reset and infos["final_observation"] /
    infos["_final_observation"] (gymnasium 0.29 convention).  Thin adapter over BalanceVecEnv."""

    def __init__(self, env_id, num_envs, **kwargs):
        self._v = BalanceVecEnv(env_id, num_envs, **kwargs)
        self.num_envs = num_envs
        self.single_observation_space, self.single_action_space = self._v.observation_space, self._v.action_space
        self.observation_space, self.action_space = self.single_observation_space, self.single_action_space

    def reset(self, seed=None, options=None):
        return self._v.reset(), {}

    def step(self, actions):
        a = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, 2)
        v = self._v
        v.step_async(a)
        obs, rew, term, trunc, tob = v._collect()
        tob = np.array(tob, copy=True)
        done = term | trunc
        infos = {}
        if done.any():
            final = np.empty(self.num_envs, dtype=object)
            for i in np.flatnonzero(done):
                final[i] = tob[i]
            infos = {"final_observation": final, "_final_observation": done.copy()}
        return obs, rew, term, trunc, infos

    def close(self):
        self._v.close()
