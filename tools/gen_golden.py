#!/usr/bin/env python3
"""Golden-vector generator for the Python-side half of the hot path.

Runs ONLY in the build container (it needs /root/reference).  It imports the
reference's own env classes with *stub* `mujoco` / `gymnasium` modules (physics
stubbed out: `mj_step` replays a scripted post-step state), drives
step()/reset_model()/set_block_pos_vel() on seeded inputs and records

    inputs  (state before, action, every uniform the code drew, scripted post-step state)
    outputs (obs, reward, terminated, ctrl, qpos written, block state, ...)

as plain JSON under tests/golden/.  Only DATA is written; no reference source is
copied.  The committed fixtures pin oracle/ (tests/test_oracle_envlogic.py), and
through the oracle the HIP path.

What is pinned here:  SURVEY.md §8 rows a1, a4-a10, a12, a13 (control law,
termination, pitch/yaw, noise draw order, finite-difference pitch rate, obs,
reward, reset pose + quaternion slot mix-up, block state machine, block throw).
What is NOT pinned: mj_step itself (MuJoCo is not installed anywhere reachable).

usage: python tools/gen_golden.py          (writes tests/golden/envlogic.json)
"""
import json
import math
import os
import sys
import types

import numpy as np

REF_SRC = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "envlogic.json")

sys.dont_write_bytecode = True


# --------------------------------------------------------------------------------------
# stub modules
# --------------------------------------------------------------------------------------
class _Hook:
    """what the stubbed mj_step does: replay the next scripted post-step state"""
    on_step = None
    calls = []


def _mj_step(model, data, nstep=1):
    _Hook.calls.append(("mj_step", nstep))
    if _Hook.on_step is not None:
        _Hook.on_step(data, nstep)


def _install_stubs():
    mj = types.ModuleType("mujoco")
    mj.mj_step = _mj_step
    mj.mj_rnePostConstraint = lambda m, d: _Hook.calls.append(("mj_rnePostConstraint",))
    mj.mj_forward = lambda m, d: None
    mj.mjtGridPos = types.SimpleNamespace(mjGRID_TOPRIGHT=1)
    sys.modules["mujoco"] = mj

    gym = types.ModuleType("gymnasium")
    utils = types.ModuleType("gymnasium.utils")

    class EzPickle:
        def __init__(self, *a, **k):
            pass

    utils.EzPickle = EzPickle
    gym.utils = utils

    spaces = types.ModuleType("gymnasium.spaces")

    class Box:
        def __init__(self, low, high, dtype=np.float32, shape=None):
            self.low, self.high, self.dtype = np.asarray(low), np.asarray(high), dtype
            self.shape = self.low.shape

    spaces.Box = Box
    gym.spaces = spaces

    envs = types.ModuleType("gymnasium.envs")
    envs_mj = types.ModuleType("gymnasium.envs.mujoco")

    class MujocoEnv:
        def __init__(self, model_path, frame_skip, observation_space=None, render_mode=None, **kw):
            self.model_path = model_path
            self.frame_skip = frame_skip
            self.observation_space = observation_space
            self.render_mode = render_mode
            self._set_action_space()

    envs_mj.MujocoEnv = MujocoEnv
    envs.mujoco = envs_mj
    gym.envs = envs

    reg = types.ModuleType("gymnasium.envs.registration")
    _registry = {}

    def register(id, entry_point=None, max_episode_steps=None, reward_threshold=None, **kw):
        _registry[id] = dict(entry_point=entry_point, max_episode_steps=max_episode_steps,
                             reward_threshold=reward_threshold)

    reg.register = register
    reg.make = reg.pprint_registry = reg.spec = lambda *a, **k: None
    reg.registry = _registry
    envs.registration = reg

    for name, mod in [("gymnasium", gym), ("gymnasium.utils", utils), ("gymnasium.spaces", spaces),
                      ("gymnasium.envs", envs), ("gymnasium.envs.mujoco", envs_mj),
                      ("gymnasium.envs.registration", reg)]:
        sys.modules[name] = mod
    return _registry


# --------------------------------------------------------------------------------------
# fake mjData with the named accessors the reference uses
# --------------------------------------------------------------------------------------
class _View:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _Act:
    def __init__(self):
        self._ctrl = np.zeros(1)

    @property
    def ctrl(self):
        return self._ctrl

    @ctrl.setter
    def ctrl(self, v):
        self._ctrl = np.asarray(v, dtype=np.float64).copy()


class FakeData:
    def __init__(self, nq, nv):
        self.qpos = np.zeros(nq)
        self.qvel = np.zeros(nv)
        self.time = 0.0
        self.xquat_robot = np.array([1.0, 0, 0, 0])
        self.xpos_robot = np.zeros(3)
        self._act = {"motor_l_wheel": _Act(), "motor_r_wheel": _Act()}

    def body(self, name):
        assert name == "robot_body"
        return _View(xquat=self.xquat_robot, xpos=self.xpos_robot)

    def joint(self, name):
        if name == "robot_body_joint":
            return _View(qpos=self.qpos[0:7], qvel=self.qvel[0:6])
        if name == "torso_l_wheel":
            return _View(qpos=self.qpos[7:8], qvel=self.qvel[6:7])
        if name == "torso_r_wheel":
            return _View(qpos=self.qpos[8:9], qvel=self.qvel[7:8])
        if name == "block_joint":
            return _View(qpos=self.qpos[9:16], qvel=self.qvel[8:14])
        raise KeyError(name)

    def actuator(self, name):
        return self._act[name]


def _attach(env, nq, nv, gym_seed):
    """give a stub-constructed env the attributes MujocoEnv would have made"""
    env.data = FakeData(nq, nv)
    geoms = {n: _View(friction=np.array([1.0, 0.005, 0.0001])) for n in ('floor', 'l_wheel_geom', 'r_wheel_geom')}
    env.model = _View(nq=nq, nv=nv, geom=lambda name: geoms[name], _geoms=geoms)
    q0 = np.zeros(nq)
    q0[3] = 1.0
    if nq == 16:
        q0[12] = 1.0
    env.init_qpos = q0
    env.init_qvel = np.zeros(nv)
    env.np_random = np.random.default_rng(gym_seed)

    def set_state(qpos, qvel):
        # what gymnasium's MujocoEnv.set_state + mj_forward leave behind for the accessors
        env.data.qpos[:] = qpos
        env.data.qvel[:] = qvel
        q = np.array(qpos[3:7], dtype=np.float64)
        env.data.xquat_robot[:] = q / np.linalg.norm(q)
        env.data.xpos_robot[:] = qpos[0:3]

    env.set_state = set_state
    env._update_camera_follow = lambda: None
    return env


class _UniformLog:
    """wrap np.random.random so every draw of the global RNG is recorded"""

    def __init__(self):
        self.draws = []
        self._orig = np.random.random

    def __enter__(self):
        def rnd(*a, **k):
            v = self._orig(*a, **k)
            self.draws.append(float(v))
            return v

        np.random.random = rnd
        return self

    def __exit__(self, *a):
        np.random.random = self._orig


def _rand_quat(rng, tilt=1.0):
    """unit quaternion (w,x,y,z): random yaw, bounded roll/pitch"""
    q = rng.normal(size=4)
    q[1:3] *= tilt
    return q / np.linalg.norm(q)


def L(x):
    return [float(v) for v in np.asarray(x, dtype=np.float64).ravel()]


def _extras(e):
    """per-env scalars of the later variants: wheel/floor friction (Env02), target speed schedule + pitch offset (Env01_v3)"""
    return dict(friction=float(e.model._geoms['l_wheel_geom'].friction[0]), floor_friction=float(e.model._geoms['floor'].friction[0]),
                delay_target_speed=float(getattr(e, 'delay_target_speed', 0.0)), pitch_offset=float(getattr(e, 'pitch_offset', 0.0)),
                target_wheel_speed=float(e.target_wheel_speed))


# --------------------------------------------------------------------------------------
def main():
    registry = _install_stubs()
    sys.path.insert(0, REF_SRC)
    import balance_robot  # noqa: F401  (fills the stub registry)
    from balance_robot.envs.env01_v1 import Env01
    from balance_robot.envs.env01_v2 import Env01_v2
    from balance_robot.envs.env03_v1 import Env03
    from balance_robot.envs.env03_v2 import Env03_v2
    from balance_robot.envs.env01_v3 import Env01_v3
    from balance_robot.envs.env02_v1 import Env02
    from balance_robot.envs import RobotBaseEnv as rb

    out = {"_about": "generated by tools/gen_golden.py from the reference's env classes with physics stubbed; data only"}
    out["registry"] = {k: v for k, v in registry.items()}
    out["constants"] = dict(PITCH_MAX=rb.PITCH_MAX, PITCH_DOT_MAX=rb.PITCH_DOT_MAX,
                            WHEEL_SPEED_MAX=rb.WHEEL_SPEED_MAX,
                            WHEEL_SPEED_DELTA_MAX=rb.WHEEL_SPEED_DELTA_MAX, YAW_MAX=rb.YAW_MAX)

    rng = np.random.default_rng(20250228)

    # ---------------------------------------------------------------- (1) pitch / yaw
    env = _attach(Env01(), 9, 8, 0)
    cases = []
    quats = [_rand_quat(rng, t) for t in (0.05, 0.2, 0.5, 1.0) for _ in range(16)]
    quats += [np.array([1.0, 0, 0, 0]), np.array([0.0, 1, 0, 0]), np.array([0.0, 0, 0, 1]),
              np.array([-0.3, 0.1, 0.2, 0.9]) / np.linalg.norm([-0.3, 0.1, 0.2, 0.9]),
              np.array([2.0, 0.2, -0.1, 0.4])]  # un-normalised on purpose
    for q in quats:
        env.data.xquat_robot[:] = q
        cases.append(dict(xquat=L(q), pitch=float(env.get_pitch()), yaw=float(env.get_yaw())))
    out["pitch_yaw"] = cases
    # gimbal lock of as_euler('xyz') (the wheel axis vertical: second angle +-pi/2): within 1e-7 rad scipy zeroes the yaw and puts
    # the whole rotation about the vertical into the pitch (and warns); just outside it does not
    import warnings
    cases = []
    from scipy.spatial.transform import Rotation as _Rot
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for sgn in (1.0, -1.0):
            for off in (0.0, 1e-9, 3e-8, -3e-8, 3e-7, -3e-7, 1e-5, 1e-3):
                for a, c in ((0.3, 0.7), (-1.2, 2.0), (2.9, 1.5), (0.0, 0.0)):
                    x, y, z, w = _Rot.from_euler("xyz", [a, sgn * (np.pi / 2 - off), c]).as_quat()
                    q = np.array([w, x, y, z]) * (1.0 if (a, c) != (2.9, 1.5) else 1.7)  # one family un-normalised
                    env.data.xquat_robot[:] = q
                    cases.append(dict(xquat=L(q), pitch=float(env.get_pitch()), yaw=float(env.get_yaw())))
    out["pitch_yaw_gimbal"] = cases

    # ---------------------------------------------------------------- (2) reward (no-noise class)
    cases = []
    for _ in range(48):
        env.data.xquat_robot[:] = _rand_quat(rng, 0.3)
        env.data.qvel[:] = rng.normal(size=8) * np.array([1, 1, 1, 3, 3, 3, 40, 40])
        cases.append(dict(xquat=L(env.data.xquat_robot), qvel=L(env.data.qvel), reward=float(env._get_reward())))
    out["reward"] = cases

    # ---------------------------------------------------------------- (3) scripted step sequences
    def run_steps(cls, nq, nv, np_seed, nsteps, block=False, act_scale=1.0, slow_block_at=(), time0=0.0):
        np.random.seed(np_seed)
        with _UniformLog() as ctor_log:
            e = _attach(cls(), nq, nv, np_seed + 1000)
        rec = dict(cls=cls.__name__, nq=nq, nv=nv, ctor_uniforms=list(ctor_log.draws))
        if block:
            rec["attack_side_front"] = bool(getattr(e, "attack_side_front", False))
            rec["block_delay"] = float(e.block_delay)
        # reset through the reference's reset_model (gym RNG draws are recorded separately)
        pre = []
        if cls.__name__ == "Env01_v3":  # reset_model draws two scalars from the seeded generator first
            pre = [float(e.np_random.uniform(low=-10.0, high=10)), float(e.np_random.uniform(low=-0.0349066, high=0.0349066))]
        gym_draw = e.np_random.uniform(size=nq, low=-0.01, high=0.01)
        e.np_random = np.random.default_rng(np_seed + 1000)  # rewind so reset_model draws the same
        rec["reset_gym_scalars"] = pre
        with _UniformLog() as log:
            obs0 = e.reset_model()
        rec["reset"] = dict(gym_uniform=L(gym_draw), uniforms=list(log.draws), qpos=L(e.data.qpos),
                            qvel=L(e.data.qvel), xquat=L(e.data.xquat_robot), xpos=L(e.data.xpos_robot),
                            obs=L(obs0), time=float(e.data.time), extras=_extras(e),
                            block_timer=None if not block else e.block_delay_time_start)
        e.data.time = time0  # (Env01_v3's schedule keys on data.time at the start of step)
        rec["time0"] = time0
        steps = []
        srng = np.random.default_rng(np_seed + 7)
        for k in range(nsteps):
            a = (srng.uniform(-1, 1, size=2) * act_scale).astype(np.float32)
            pre = dict(qvel=L(e.data.qvel), xquat=L(e.data.xquat_robot), xpos=L(e.data.xpos_robot),
                       time=float(e.data.time), qpos=L(e.data.qpos))
            # scripted "physics": what the stubbed mj_step leaves in data
            post_q = _rand_quat(srng, 0.25 if k % 7 else 1.2)
            post_qvel = srng.normal(size=nv) * (np.array([1, 1, 1, 3, 3, 3, 40, 40] + [2, 2, 2, 5, 5, 5] * (nv == 14)))
            if block and k in slow_block_at:
                post_qvel[8:11] = srng.normal(size=3) * 0.03
            post_xpos = srng.normal(size=3) * 0.3
            post_qpos = np.array(e.data.qpos)
            post_qpos[0:3] = post_xpos + srng.normal(size=3) * 1e-5
            if block:
                post_qpos[9:12] = np.array(e.data.qpos[9:12]) + srng.normal(size=3) * 0.01

            def on_step(data, nstep, post_q=post_q, post_qvel=post_qvel, post_xpos=post_xpos, post_qpos=post_qpos):
                data.qvel[:] = post_qvel
                data.qpos[:] = post_qpos
                data.xquat_robot[:] = post_q
                data.xpos_robot[:] = post_xpos
                for _ in range(nstep):
                    data.time += 0.00002

            _Hook.on_step = on_step
            _Hook.calls.clear()
            with _UniformLog() as log:
                ob, rew, term, trunc, info = e.step(a)
            steps.append(dict(action=L(a), pre=pre,
                              post=dict(qvel=L(post_qvel), xquat=L(post_q), xpos=L(post_xpos), qpos=L(post_qpos)),
                              uniforms=list(log.draws), calls=[list(c) for c in _Hook.calls],
                              ctrl=[float(e.data.actuator("motor_l_wheel").ctrl[0]),
                                    float(e.data.actuator("motor_r_wheel").ctrl[0])],
                              obs=L(ob), reward=float(rew), terminated=bool(term), truncated=bool(trunc),
                              time=float(e.data.time), extras=_extras(e),
                              qpos_after=L(e.data.qpos), qvel_after=L(e.data.qvel),
                              block_timer=None if not block else e.block_delay_time_start))
        rec["steps"] = steps
        # a second reset in the middle of an episode: last_time/last_pitch are NOT cleared
        e.np_random = np.random.default_rng(np_seed + 2000)
        g2 = np.random.default_rng(np_seed + 2000)
        pre2 = []
        if cls.__name__ == "Env01_v3":
            pre2 = [float(g2.uniform(low=-10.0, high=10)), float(g2.uniform(low=-0.0349066, high=0.0349066))]
        gym_draw = g2.uniform(size=nq, low=-0.01, high=0.01)
        rec["reset2_gym_scalars"] = pre2
        e.data.time = 0.0  # what mj_resetData does
        with _UniformLog() as log:
            obs1 = e.reset_model()
        rec["reset2"] = dict(gym_uniform=L(gym_draw), uniforms=list(log.draws), qpos=L(e.data.qpos),
                             qvel=L(e.data.qvel), xquat=L(e.data.xquat_robot), obs=L(obs1), extras=_extras(e))
        return rec

    seqs = []
    seqs.append(run_steps(Env01, 9, 8, 11, 12))
    seqs.append(run_steps(Env01_v2, 9, 8, 12, 12))
    seqs.append(run_steps(Env01_v2, 9, 8, 13, 12, act_scale=3.0))  # actions outside [-1,1]: env does not clip
    seqs.append(run_steps(Env03, 16, 14, 14, 14, block=True, slow_block_at=(3, 9)))
    for s in (15, 16, 17, 18):
        seqs.append(run_steps(Env03_v2, 16, 14, s, 14, block=True, slow_block_at=(2, 5)))
    out["sequences"] = seqs
    import contextlib, io
    seqs2 = []
    with contextlib.redirect_stdout(io.StringIO()):  # the two classes print() on reset
        seqs2.append(run_steps(Env02, 9, 8, 21, 10))
        seqs2.append(run_steps(Env02, 9, 8, 22, 10, act_scale=2.0))
        for sd, t0 in ((23, 0.0), (24, 0.99), (25, 2.98), (26, 4.48), (27, 5.48)):
            seqs2.append(run_steps(Env01_v3, 9, 8, sd, 10, time0=t0))
    out["sequences_f3"] = seqs2

    # ---------------------------------------------------------------- (4) Env03-v2 block timer over many steps
    # time accumulates 250 x 2e-5 per step in fp64 exactly as MuJoCo does; the `> block_delay`
    # comparison at the nominal boundary is decided by that accumulation.
    np.random.seed(99)
    e = _attach(Env03_v2(), 16, 14, 5)
    e.reset_model()
    timeline = []

    def on_step(data, nstep):
        for _ in range(nstep):
            data.time += 0.00002

    _Hook.on_step = on_step
    e.data.qvel[8:11] = 0.0  # block slow from the start -> removed on step 1, respawn after > 0.5 s
    trng = np.random.default_rng(4242)
    for k in range(230):
        # block slow on every step: removed as soon as it is live, respawned after > 0.5 s
        e.data.qvel[8:11] = 0.0
        e.data.xquat_robot[:] = _rand_quat(trng, 0.2)
        e.data.xpos_robot[:] = trng.normal(size=3) * 0.2
        pre = dict(xquat=L(e.data.xquat_robot), xpos=L(e.data.xpos_robot))
        with _UniformLog() as log:
            e.step(np.zeros(2, dtype=np.float32))
        timeline.append(dict(time=float(e.data.time),
                             timer=None if e.block_delay_time_start is None else float(e.block_delay_time_start),
                             pre=pre, uniforms=list(log.draws),
                             block_qpos=L(e.data.qpos[9:16]), block_qvel=L(e.data.qvel[8:14])))
    out["block_timer_timeline"] = dict(block_delay=0.5, attack_side_front=bool(e.attack_side_front), rows=timeline)

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=None, separators=(",", ":"))
    print("wrote", os.path.normpath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
