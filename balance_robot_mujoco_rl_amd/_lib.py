"""ctypes binding of libbrs_hip.so (C ABI: include/brs.h).  Loading never needs a GPU; brs_create does."""
import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libbrs_hip.so")
SRC = os.path.join(_PKG, "csrc", "brs_kernels.hip")
HEADERS = [os.path.join(_PKG, "csrc", h) for h in ("brs_core.hpp", "brs_model.hpp", "brs_state.hpp")] + \
          [os.path.join(os.path.dirname(_PKG), "include", "brs.h")]

# every symbol include/brs.h declares
SYMBOLS = ["brs_create", "brs_destroy", "brs_last_error", "brs_sizes", "brs_reset", "brs_step", "brs_physics",
           "brs_get_state", "brs_set_state", "brs_get_aux", "brs_set_aux", "brs_get_xpose", "brs_set_xpose",
           "brs_step_bytes_per_env", "brs_step_kernel_name"]


class BrsConfig(C.Structure):
    _fields_ = [("variant", C.c_int32), ("num_envs", C.c_int32), ("device", C.c_int32), ("flags", C.c_uint32),
                ("seed", C.c_uint64), ("env_index_base", C.c_int64), ("max_episode_steps", C.c_int32),
                ("substeps", C.c_int32), ("timestep", C.c_double), ("block_threads", C.c_int32), ("reserved", C.c_int32)]


FLAG_AUTO_RESET, FLAG_NOISE_ON, FLAG_NOISE_OFF = 1, 2, 4


def hipcc_path():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.sep not in p or os.path.exists(p)):
            return p
    return "hipcc"


def build(force=False, verbose=False):
    """compile the HIP kernels + C ABI for gfx950 in-tree (hipcc cross-compiles without a GPU)"""
    srcs = [SRC] + HEADERS
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if not (force or stale):
        return LIB_PATH
    # -ffast-math on the DEVICE side only (the host-side state conversion keeps IEEE semantics): no IEEE division/sqrt expansions and free reassociation inside the fp32 force path
    # (parity budget is 1e-4, rounding noise 1e-7); NaN detection in the kernel is done on the bit pattern.
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-Xarch_device", "-ffast-math",
           "-Xarch_device", "-fgpu-flush-denormals-to-zero",
           # the dense algebra is packed by hand (V2 -> v_pk_fma_f32); the SLP vectoriser's extra packing of the scalar
           # code only adds pack/unpack moves (measured: +13 % env-steps/s without it)
           "-Xarch_device", "-fno-slp-vectorize", "-fPIC", "-shared", "-o", LIB_PATH, SRC]
    cmd += os.environ.get("BRS_EXTRA_HIPCC_FLAGS", "").split()  # ablation builds (e.g. -DBRS_NO_COUPLED); not for production
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib():
    """load the shared library (raises if it has not been built: the product has no fallback path)"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
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
    _lib = L
    return L
