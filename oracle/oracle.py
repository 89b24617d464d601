"""ctypes wrapper of the CPU fp64 oracle (oracle/brs_oracle.c).  TEST INFRASTRUCTURE ONLY.

May be imported from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from
the product package.  `build()` compiles the library with gcc if it is missing or stale.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_DIR, "libbrs_oracle.so")

ENV01_V1, ENV01_V2, ENV03_V1, ENV03_V2 = 0, 1, 2, 3
VARIANTS = {"Env01-v1": 0, "Env01-v2": 1, "Env03-v1": 2, "Env03-v2": 3, "Env01-v3": 4, "Env02-v1": 5}
AUX_COLS = 14
FLAG_AUTO_RESET, FLAG_NOISE_ON, FLAG_NOISE_OFF = 1, 2, 4
MAXNV, MAXCON = 14, 40


class Contact(C.Structure):
    _fields_ = [("dist", C.c_double), ("pos", C.c_double * 3), ("frame", C.c_double * 9),
                ("body1", C.c_int), ("body2", C.c_int), ("mu", C.c_double), ("solref", C.c_double * 2),
                ("solimp", C.c_double * 5), ("margin", C.c_double)]


class ForwardOut(C.Structure):
    _fields_ = [("nv", C.c_int), ("ncon", C.c_int), ("nefc", C.c_int), ("solver_iter", C.c_int),
                ("M", C.c_double * (MAXNV * MAXNV)), ("bias", C.c_double * MAXNV), ("passive", C.c_double * MAXNV),
                ("actuator", C.c_double * MAXNV), ("qacc_smooth", C.c_double * MAXNV), ("qacc", C.c_double * MAXNV),
                ("qfrc_constraint", C.c_double * MAXNV), ("qacc_integ", C.c_double * MAXNV),
                ("xquat", C.c_double * 4), ("xpos", C.c_double * 3), ("con", Contact * MAXCON),
                ("efc_D", C.c_double * (4 * MAXCON)), ("efc_aref", C.c_double * (4 * MAXCON)),
                ("efc_force", C.c_double * (4 * MAXCON)), ("energy_kin", C.c_double), ("energy_pot", C.c_double)]


class ModelInfo(C.Structure):
    _fields_ = [("body_mass", C.c_double * 5), ("body_inertia", (C.c_double * 3) * 5), ("body_ipos", (C.c_double * 3) * 5),
                ("invweight0", (C.c_double * 2) * 5), ("meaninertia", C.c_double), ("nq", C.c_int), ("nv", C.c_int)]


def build(force=False):
    src = [os.path.join(_DIR, f) for f in ("brs_oracle.c", "brs_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _DIR, "-s", "-B", "libbrs_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, dp, fp, u8p = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint8)
        L.bo_create.restype = vp
        L.bo_create.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int64, C.c_uint32, C.c_int, C.c_int, C.c_double]
        L.bo_destroy.argtypes = [vp]
        L.bo_nq.argtypes = [vp]
        L.bo_nv.argtypes = [vp]
        L.bo_model_info_get.argtypes = [vp, C.POINTER(ModelInfo)]
        L.bo_set_threads.argtypes = [vp, C.c_int]
        L.bo_reset.argtypes = [vp, u8p, fp]
        L.bo_step.argtypes = [vp, fp, fp, fp, u8p, u8p, fp]
        L.bo_get_state.argtypes = [vp, dp, dp, dp, dp]
        L.bo_set_state.argtypes = [vp, dp, dp, dp, dp]
        L.bo_get_aux.argtypes = [vp, dp]
        L.bo_set_aux.argtypes = [vp, dp]
        L.bo_get_xpose.argtypes = [vp, dp, dp]
        L.bo_set_xpose.argtypes = [vp, dp, dp]
        L.bo_physics.argtypes = [vp, dp, C.c_int]
        L.bo_forward.argtypes = [vp, C.c_int, dp, C.POINTER(ForwardOut)]
        L.bo_script_uniforms.argtypes = [vp, C.c_int, dp, C.c_int]
        L.bo_script_remaining.argtypes = [vp, C.c_int]
        L.bo_stub_physics.argtypes = [vp, C.c_int, dp, dp, dp, dp]
        L.bo_pitch_yaw.argtypes = [dp, dp, dp]
        ip = C.POINTER(C.c_int)
        L.bo_box_box_points.argtypes = [dp, C.c_double, dp, dp, C.c_double, dp, dp, dp, ip, dp, ip]
        L.bo_set_boxbox_keep_all.argtypes = [C.c_int]
        L.bo_set_boxbox_max.argtypes = [C.c_int]
        L.bo_box_cyl_point.argtypes = [dp, dp, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp, dp]
        L.bo_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.bo_uniform.restype = C.c_double
        L.bo_uniform.argtypes = [C.c_uint64, C.c_int64, C.c_uint32, C.c_int]
        L.bo_euler_slot_quat.argtypes = [C.c_double, C.c_double, C.c_double, dp]
        _lib = L
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _u8(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_uint8))


def pitch_yaw(xquat):
    q = np.ascontiguousarray(xquat, dtype=np.float64)
    p, y = C.c_double(), C.c_double()
    lib().bo_pitch_yaw(_dp(q), C.byref(p), C.byref(y))
    return p.value, y.value


def box_box_points(sT, s, cg, RTB, margin):
    """block<->torso generator on its own (torso-geom frame): (points[<=4,3], dists, normal, code, raw[<=16,4])"""
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    sT, cg, RTB = c(sT), c(cg), c(RTB).reshape(9)
    pos = np.zeros(24); dist = np.zeros(8); nrm = np.zeros(3); raw = np.zeros(64)
    code, nraw = C.c_int(), C.c_int()
    n = lib().bo_box_box_points(_dp(sT), float(s), _dp(cg), _dp(RTB), float(margin), _dp(pos), _dp(dist), _dp(nrm),
                                C.byref(code), _dp(raw), C.byref(nraw))
    return pos.reshape(8, 3)[:n].copy(), dist[:n].copy(), nrm, code.value, raw.reshape(16, 4)[:nraw.value].copy()


def set_boxbox_keep_all(on):
    """study switch (process-wide): box-box patches keep all <= 8 clipped points, like MuJoCo, instead of the 4 deepest"""
    lib().bo_set_boxbox_keep_all(int(bool(on)))


def set_boxbox_max(n):
    """study switch (process-wide): box-box patches keep the n deepest clipped points (4 = the specification, 8 = all)"""
    lib().bo_set_boxbox_max(int(n))


def box_cyl_point(d, RTB, s, r, hl, margin):
    """block<->wheel generator on its own (torso frame, wheel centre at the origin, axis x): (pos, normal, dist) or None"""
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    d, RTB = c(d), c(RTB).reshape(9)
    pos = np.zeros(3); nrm = np.zeros(3); dist = C.c_double()
    if not lib().bo_box_cyl_point(_dp(d), _dp(RTB), float(s), float(r), float(hl), float(margin), _dp(pos), _dp(nrm), C.byref(dist)):
        return None
    return pos, nrm, dist.value


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().bo_philox4x32_10(c, k, o)
    return list(o)


def euler_slot_quat(a, b, c):
    q = np.zeros(4)
    lib().bo_euler_slot_quat(a, b, c, _dp(q))
    return q


class Oracle:
    """N independent fp64 envs stepped on the CPU."""

    def __init__(self, variant, num_envs, seed=0, env_index_base=0, auto_reset=False, noise=None,
                 max_episode_steps=0, substeps=0, timestep=0.0, threads=1):
        if isinstance(variant, str):
            variant = VARIANTS[variant]
        flags = (FLAG_AUTO_RESET if auto_reset else 0)
        if noise is True:
            flags |= FLAG_NOISE_ON
        elif noise is False:
            flags |= FLAG_NOISE_OFF
        self.L = lib()
        self.h = self.L.bo_create(variant, num_envs, seed, env_index_base, flags, max_episode_steps, substeps, timestep)
        if not self.h:
            raise ValueError("bo_create failed")
        self.n = num_envs
        self.nq = self.L.bo_nq(self.h)
        self.nv = self.L.bo_nv(self.h)
        self.L.bo_set_threads(self.h, threads)

    def close(self):
        if self.h:
            self.L.bo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def model_info(self):
        mi = ModelInfo()
        self.L.bo_model_info_get(self.h, C.byref(mi))
        return dict(body_mass=np.array(mi.body_mass), body_inertia=np.array(mi.body_inertia),
                    body_ipos=np.array(mi.body_ipos), invweight0=np.array(mi.invweight0),
                    meaninertia=mi.meaninertia, nq=mi.nq, nv=mi.nv)

    def reset(self, mask=None):
        obs = np.zeros((self.n, 6), np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.L.bo_reset(self.h, _u8(m), _fp(obs))
        return obs

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n, 2)
        obs = np.zeros((self.n, 6), np.float32)
        tob = np.zeros((self.n, 6), np.float32)
        rew = np.zeros(self.n, np.float32)
        term = np.zeros(self.n, np.uint8)
        trunc = np.zeros(self.n, np.uint8)
        self.L.bo_step(self.h, _fp(a), _fp(obs), _fp(rew), _u8(term), _u8(trunc), _fp(tob))
        return obs, rew, term.astype(bool), trunc.astype(bool), tob

    def get_state(self):
        qpos = np.zeros((self.n, self.nq)); qvel = np.zeros((self.n, self.nv))
        warm = np.zeros((self.n, self.nv)); time = np.zeros(self.n)
        self.L.bo_get_state(self.h, _dp(qpos), _dp(qvel), _dp(warm), _dp(time))
        return qpos, qvel, warm, time

    def set_state(self, qpos=None, qvel=None, warm=None, time=None):
        c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        qpos, qvel, warm, time = c(qpos), c(qvel), c(warm), c(time)
        self.L.bo_set_state(self.h, _dp(qpos), _dp(qvel), _dp(warm), _dp(time))

    def get_aux(self):
        aux = np.zeros((self.n, AUX_COLS))
        self.L.bo_get_aux(self.h, _dp(aux))
        return aux

    def set_aux(self, aux):
        a = np.ascontiguousarray(aux, dtype=np.float64)
        self.L.bo_set_aux(self.h, _dp(a))

    def get_xpose(self):
        xq = np.zeros((self.n, 4)); xp = np.zeros((self.n, 3))
        self.L.bo_get_xpose(self.h, _dp(xq), _dp(xp))
        return xq, xp

    def set_xpose(self, xquat=None, xpos=None):
        c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        xquat, xpos = c(xquat), c(xpos)
        self.L.bo_set_xpose(self.h, _dp(xquat), _dp(xpos))

    def physics(self, ctrl, nsub):
        c = np.ascontiguousarray(ctrl, dtype=np.float64).reshape(self.n, 2)
        self.L.bo_physics(self.h, _dp(c), int(nsub))

    def forward(self, env=0, ctrl=(0.0, 0.0)):
        out = ForwardOut()
        c = np.array(ctrl, dtype=np.float64)
        self.L.bo_forward(self.h, env, _dp(c), C.byref(out))
        nv = out.nv
        d = dict(nv=nv, ncon=out.ncon, nefc=out.nefc, solver_iter=out.solver_iter,
                 M=np.array(out.M).reshape(MAXNV, MAXNV)[:nv, :nv].copy())
        for k in ("bias", "passive", "actuator", "qacc_smooth", "qacc", "qfrc_constraint", "qacc_integ"):
            d[k] = np.array(getattr(out, k))[:nv].copy()
        d["xquat"] = np.array(out.xquat); d["xpos"] = np.array(out.xpos)
        d["contacts"] = [dict(dist=c_.dist, pos=np.array(c_.pos), frame=np.array(c_.frame).reshape(3, 3),
                              body1=c_.body1, body2=c_.body2, mu=c_.mu, margin=c_.margin) for c_ in out.con[:out.ncon]]
        for k in ("efc_D", "efc_aref", "efc_force"):
            d[k] = np.array(getattr(out, k))[:out.nefc].copy()
        d["energy_kin"] = out.energy_kin; d["energy_pot"] = out.energy_pot
        return d

    def script_uniforms(self, env, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        self.L.bo_script_uniforms(self.h, env, _dp(u), len(u))

    def script_remaining(self, env):
        return self.L.bo_script_remaining(self.h, env)

    def stub_physics(self, env, qpos, qvel, xquat, xpos):
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        qpos, qvel, xquat, xpos = c(qpos), c(qvel), c(xquat), c(xpos)
        self.L.bo_stub_physics(self.h, env, _dp(qpos), _dp(qvel), _dp(xquat), _dp(xpos))
