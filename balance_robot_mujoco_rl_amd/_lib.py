"""ctypes binding of libbrs_hip.so (C ABI: include/brs.h).  Loading never needs a GPU; brs_create does."""
import ctypes as C
import hashlib
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libbrs_hip.so")
SRC = os.path.join(_PKG, "csrc", "brs_kernels.hip")
SRC_POLICY = os.path.join(_PKG, "csrc", "brs_policy.hip")
HEADERS = [os.path.join(_PKG, "csrc", h) for h in ("brs_core.hpp", "brs_model.hpp", "brs_state.hpp")] + \
          [os.path.join(os.path.dirname(_PKG), "include", h) for h in ("brs.h", "brs_policy.h")]

# every symbol include/brs.h declares
SYMBOLS = ["brs_create", "brs_destroy", "brs_last_error", "brs_sizes", "brs_reset", "brs_step", "brs_physics",
           "brs_get_state", "brs_set_state", "brs_get_aux", "brs_set_aux", "brs_get_xpose", "brs_set_xpose",
           "brs_step_bytes_per_env", "brs_step_kernel_name", "brs_build_id",
           # include/brs_policy.h
           "brs_policy_create", "brs_policy_destroy", "brs_policy_last_error", "brs_policy_set_weights",
           "brs_policy_use_device_weights", "brs_policy_act", "brs_policy_value", "brs_rollout_bootstrap", "brs_gae"]
POLICY_NPARAM = (64 * 6 + 64 + 64 * 64 + 64 + 2 * 64 + 2) + (64 * 6 + 64 + 64 * 64 + 64 + 64 + 1) + 2


class BrsConfig(C.Structure):
    _fields_ = [("variant", C.c_int32), ("num_envs", C.c_int32), ("device", C.c_int32), ("flags", C.c_uint32),
                ("seed", C.c_uint64), ("env_index_base", C.c_int64), ("max_episode_steps", C.c_int32),
                ("substeps", C.c_int32), ("timestep", C.c_double), ("block_threads", C.c_int32), ("reserved", C.c_int32)]


FLAG_AUTO_RESET, FLAG_NOISE_ON, FLAG_NOISE_OFF, FLAG_NO_LANE_GROUPING = 1, 2, 4, 8


def hipcc_path():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.sep not in p or os.path.exists(p)):
            return p
    return "hipcc"


def build(force=False, verbose=False, out=None, extra_flags=()):
    """compile the HIP kernels + C ABI for gfx950 in-tree (hipcc cross-compiles without a GPU).  `out` / `extra_flags`: an A/B
    build next to the product library (tools/ab_build.py; loaded only when BRS_HIP_LIB points at it)"""
    srcs = [SRC, SRC_POLICY] + HEADERS
    if out is not None:
        return _compile(out, verbose, list(extra_flags), tag="_" + os.path.splitext(os.path.basename(out))[0])
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if not (force or stale):
        return LIB_PATH
    return _compile(LIB_PATH, verbose, [], tag="")


def _compile(lib_path, verbose, extra_flags, tag):
    hipcc = hipcc_path()
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    # step kernels: -ffast-math on the DEVICE side only (the host-side state conversion keeps IEEE semantics): no IEEE
    # division/sqrt expansions and free reassociation inside the fp32 force path (parity budget is 1e-4, rounding noise 1e-7);
    # NaN detection in the kernel is done on the bit pattern.
    sim_flags = ["-Xarch_device", "-ffast-math", "-Xarch_device", "-fgpu-flush-denormals-to-zero",
                 # the dense algebra is packed by hand (V2 -> v_pk_fma_f32); the SLP vectoriser's extra packing of the scalar
                 # code only adds pack/unpack moves (measured: +13 % env-steps/s without it)
                 "-Xarch_device", "-fno-slp-vectorize",
                 # one wave per SIMD at 512 registers: there is no occupancy to protect, yet the default strategy schedules
                 # for register pressure and leaves serial chains (a dependent VALU instruction issues ~1.7x slower than
                 # an independent one for a lone wave).  The ILP strategy: +4.3 % Env03, +2.7 % Env01 (same-box A/B)
                 "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
    sim_flags += os.environ.get("BRS_EXTRA_HIPCC_FLAGS", "").split()  # ablation builds (e.g. -DBRS_NO_COUPLED); not for production
    sim_flags += extra_flags
    if verbose:
        sim_flags.append("-Rpass-analysis=kernel-resource-usage")
    obj_sim, obj_pol = os.path.join(_PKG, "csrc", f"brs_kernels{tag}.o"), os.path.join(_PKG, "csrc", f"brs_policy{tag}.o")
    # build id = hash of every source the library is made from + the flags: ties a committed rocprof summary to the code that ran
    hsh = hashlib.sha256()
    for f in sorted([SRC, SRC_POLICY] + HEADERS):
        hsh.update(open(f, "rb").read())
    hsh.update(" ".join(base[1:] + [a for a in sim_flags if not a.startswith("-Rpass")]).encode())
    sim_flags = sim_flags + [f'-DBRS_BUILD_ID="{hsh.hexdigest()[:16]}"']
    subprocess.check_call(base + sim_flags + ["-c", "-o", obj_sim, SRC])
    # policy / GAE kernels: IEEE math (tanh, exp, log at libm accuracy): the parity test is rtol 1e-5 against fp32 torch
    subprocess.check_call(base + ["-c", "-o", obj_pol, SRC_POLICY])
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, obj_sim, obj_pol])
    return lib_path


_lib = None


def lib():
    """load the shared library (raises if it has not been built: the product has no fallback path)"""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("BRS_HIP_LIB") or LIB_PATH  # BRS_HIP_LIB: another BUILD of the same HIP library (same-box A/B runs)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(path)
    vp, dp, fp, u8p = C.c_void_p, C.POINTER(C.c_double), C.c_void_p, C.c_void_p
    L.brs_create.argtypes = [C.POINTER(BrsConfig), C.POINTER(vp)]
    L.brs_destroy.argtypes = [vp]
    L.brs_last_error.argtypes = [vp]
    L.brs_last_error.restype = C.c_char_p
    L.brs_sizes.argtypes = [C.c_int32] + [C.POINTER(C.c_int32)] * 4
    L.brs_reset.argtypes = [vp, u8p, fp, vp]
    L.brs_step.argtypes = [vp, fp, fp, fp, u8p, u8p, fp, vp]
    L.brs_physics.argtypes = [vp, fp, C.c_int32, vp]
    L.brs_get_state.argtypes = [vp, dp, dp, dp, dp]
    L.brs_set_state.argtypes = [vp, dp, dp, dp, dp]
    L.brs_get_aux.argtypes = [vp, dp]
    L.brs_set_aux.argtypes = [vp, dp]
    L.brs_get_xpose.argtypes = [vp, dp, dp]
    L.brs_set_xpose.argtypes = [vp, dp, dp]
    L.brs_step_bytes_per_env.argtypes = [vp]
    L.brs_step_bytes_per_env.restype = C.c_int64
    L.brs_step_kernel_name.argtypes = [vp]
    L.brs_step_kernel_name.restype = C.c_char_p
    L.brs_build_id.argtypes = []
    L.brs_build_id.restype = C.c_char_p
    i32, u64, i64, u32, f32 = C.c_int32, C.c_uint64, C.c_int64, C.c_uint32, C.c_float
    L.brs_policy_create.argtypes = [i32, C.POINTER(vp)]
    L.brs_policy_destroy.argtypes = [vp]
    L.brs_policy_last_error.argtypes = [vp]
    L.brs_policy_last_error.restype = C.c_char_p
    L.brs_policy_set_weights.argtypes = [vp, C.POINTER(C.c_float)]
    L.brs_policy_use_device_weights.argtypes = [vp, vp]
    L.brs_policy_act.argtypes = [vp, i32, vp, u64, i64, u32, i32, vp, vp, vp, vp, vp, vp]
    L.brs_policy_value.argtypes = [vp, i32, vp, vp, vp]
    L.brs_rollout_bootstrap.argtypes = [vp, i32, vp, vp, vp, f32, vp, vp]
    L.brs_gae.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp]
    _lib = L
    return L


def build_id():
    """id of the loaded library's build (include/brs.h: brs_build_id)"""
    return lib().brs_build_id().decode()
