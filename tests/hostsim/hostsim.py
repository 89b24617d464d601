"""ctypes wrapper of tests/hostsim/libbrs_hostsim.so -- the kernel source compiled for the HOST (tests only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_DIR, "libbrs_hostsim.so")
VARIANTS = {"Env01-v1": 0, "Env01-v2": 1, "Env03-v1": 2, "Env03-v2": 3, "Env01-v3": 4, "Env02-v1": 5}
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", _DIR, "-s"])
        L = C.CDLL(_LIB)
        vp, dp, fp, u8p = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint8)
        L.hs_create.restype = vp
        L.hs_set_threads.argtypes = [C.c_int]
        L.hs_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
        for name, args in dict(hs_destroy=[vp], hs_nq=[vp], hs_nv=[vp], hs_set_state=[vp, dp, dp, dp, dp],
                               hs_get_state=[vp, dp, dp, dp, dp], hs_get_aux=[vp, dp], hs_set_aux=[vp, dp],
                               hs_get_xpose=[vp, dp, dp], hs_set_xpose=[vp, dp, dp], hs_physics=[vp, dp, C.c_int],
                               hs_reset=[vp, u8p, fp], hs_step=[vp, fp, fp, fp, u8p, u8p, fp],
                               hs_script=[vp, C.c_int, dp, C.c_int], hs_step_stub=[vp, C.c_int, fp, dp, dp, dp, dp, fp, fp, u8p, u8p, dp], hs_script_remaining=[vp, C.c_int]).items():
            getattr(L, name).argtypes = args
        L.hs_pitch_yaw.argtypes = [dp, C.c_int, dp, dp]
        _lib = L
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


class HostSim:
    """same surface as oracle.Oracle, running the kernel's own source on the CPU in float or double"""

    def __init__(self, variant, num_envs, seed=0, env_index_base=0, auto_reset=False, noise=None, max_episode_steps=0,
                 substeps=0, timestep=0.0, double=False, threads=1):
        if isinstance(variant, str):
            variant = VARIANTS[variant]
        self.L = lib()
        self.L.hs_set_threads(int(threads))  # process-wide OpenMP setting (the per-env loops are data-parallel)
        nz = -1 if noise is None else int(bool(noise))
        self.h = self.L.hs_create(variant, num_envs, int(double), seed, env_index_base, int(auto_reset), nz,
                                  max_episode_steps, substeps, timestep)
        self.n = num_envs
        self.nq, self.nv = self.L.hs_nq(self.h), self.L.hs_nv(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.hs_destroy(self.h)
            self.h = None

    def set_state(self, qpos=None, qvel=None, warm=None, time=None):
        c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        qpos, qvel, warm, time = c(qpos), c(qvel), c(warm), c(time)
        self.L.hs_set_state(self.h, _p(qpos, C.c_double), _p(qvel, C.c_double), _p(warm, C.c_double), _p(time, C.c_double))

    def get_state(self):
        qpos = np.zeros((self.n, self.nq)); qvel = np.zeros((self.n, self.nv)); warm = np.zeros((self.n, self.nv)); t = np.zeros(self.n)
        self.L.hs_get_state(self.h, _p(qpos, C.c_double), _p(qvel, C.c_double), _p(warm, C.c_double), _p(t, C.c_double))
        return qpos, qvel, warm, t

    def get_aux(self):
        a = np.zeros((self.n, 14))
        self.L.hs_get_aux(self.h, _p(a, C.c_double))
        return a

    def set_aux(self, aux):
        a = np.ascontiguousarray(aux, dtype=np.float64)
        self.L.hs_set_aux(self.h, _p(a, C.c_double))

    def get_xpose(self):
        xq = np.zeros((self.n, 4)); xp = np.zeros((self.n, 3))
        self.L.hs_get_xpose(self.h, _p(xq, C.c_double), _p(xp, C.c_double))
        return xq, xp

    def set_xpose(self, xquat=None, xpos=None):
        c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        xquat, xpos = c(xquat), c(xpos)
        self.L.hs_set_xpose(self.h, _p(xquat, C.c_double), _p(xpos, C.c_double))

    def physics(self, ctrl, nsub):
        c = np.ascontiguousarray(ctrl, dtype=np.float64).reshape(self.n, 2)
        self.L.hs_physics(self.h, _p(c, C.c_double), int(nsub))

    def reset(self, mask=None):
        obs = np.zeros((self.n, 6), np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.L.hs_reset(self.h, _p(m, C.c_uint8), _p(obs, C.c_float))
        return obs

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n, 2)
        obs = np.zeros((self.n, 6), np.float32); tob = np.zeros((self.n, 6), np.float32)
        rew = np.zeros(self.n, np.float32); te = np.zeros(self.n, np.uint8); tr = np.zeros(self.n, np.uint8)
        self.L.hs_step(self.h, _p(a, C.c_float), _p(obs, C.c_float), _p(rew, C.c_float), _p(te, C.c_uint8), _p(tr, C.c_uint8), _p(tob, C.c_float))
        return obs, rew, te.astype(bool), tr.astype(bool), tob

    def step_stub(self, env, action, qpos, qvel, xquat, xpos):
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        a = np.ascontiguousarray(action, dtype=np.float32)
        qpos, qvel, xquat, xpos = c(qpos), c(qvel), c(xquat), c(xpos)
        obs = np.zeros(6, np.float32); rew = np.zeros(1, np.float32); te = np.zeros(1, np.uint8); tr = np.zeros(1, np.uint8)
        ctrl = np.zeros(2)
        self.L.hs_step_stub(self.h, env, _p(a, C.c_float), _p(qpos, C.c_double), _p(qvel, C.c_double), _p(xquat, C.c_double),
                            _p(xpos, C.c_double), _p(obs, C.c_float), _p(rew, C.c_float), _p(te, C.c_uint8), _p(tr, C.c_uint8),
                            _p(ctrl, C.c_double))
        return obs, float(rew[0]), bool(te[0]), bool(tr[0]), ctrl

    def script_uniforms(self, env, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        self.L.hs_script(self.h, env, _p(u, C.c_double), len(u))

    def script_remaining(self, env):
        return self.L.hs_script_remaining(self.h, env)


def pitch_yaw(xquat, double=False):
    """Sim<R>::pitch_yaw of the kernel source on one accessor quaternion (w, x, y, z)"""
    q = np.ascontiguousarray(xquat, dtype=np.float64)
    p, y = C.c_double(), C.c_double()
    lib().hs_pitch_yaw(_p(q, C.c_double), int(double), C.byref(p), C.byref(y))
    return p.value, y.value
