"""ctypes binding of libgem_hip.so (include/gem_hip.h).

There is no CPU fallback: if the HIP library has not been built (`python -c "import
__graft_entry__ as g; g.build()"`) importing the compute path raises, and every entry point that
returns non-zero raises `GemError` with `gem_last_error()`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GEM_HIP_LIB: developer override to A/B two builds of the library on the same box
LIB_PATH = os.environ.get("GEM_HIP_LIB") or os.path.join(_HERE, "_lib", "libgem_hip.so")

GEM_MAX_HIDDEN, GEM_MAX_POLY, GEM_MAX_JOINTS = 8, 16, 16
STAGE_LOCAL, STAGE_GLOBAL = 0, 1
PRECISION = {"f32": 0, "bf16x3": 1, "bf16": 2}


class GemError(RuntimeError):
    pass


class GemConfig(C.Structure):
    _fields_ = [("seq_len", C.c_int32), ("n_joints", C.c_int32), ("latent_dim", C.c_int32), ("n_hidden", C.c_int32),
                ("hidden", C.c_int32 * GEM_MAX_HIDDEN), ("heat_h", C.c_int32), ("heat_w", C.c_int32),
                ("n_poly", C.c_int32), ("poly", C.c_double * GEM_MAX_POLY), ("cx", C.c_double), ("cy", C.c_double),
                ("parents", C.c_int32 * GEM_MAX_JOINTS), ("max_windows", C.c_int32), ("device", C.c_int32)]


class GemEnergyWeights(C.Structure):
    _fields_ = [("w3d", C.c_double), ("smooth", C.c_double), ("bone", C.c_double), ("vae", C.c_double),
                ("reproj", C.c_double)]


class GemLbfgsOpts(C.Structure):
    _fields_ = [("lr", C.c_double), ("max_iter", C.c_int32), ("max_eval", C.c_int32), ("history", C.c_int32),
                ("reserved", C.c_int32), ("tol_grad", C.c_double), ("tol_change", C.c_double), ("c1", C.c_double),
                ("c2", C.c_double), ("ls_tol_change", C.c_double)]


class GemTrainOpts(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("weight_decay", C.c_double),
                ("kld_weight", C.c_double), ("bn_momentum", C.c_double), ("recon_sum", C.c_int32), ("reserved", C.c_int32)]


class GemPickleArray(C.Structure):
    _fields_ = [("offset", C.c_int64), ("nbytes", C.c_int64), ("dtype", C.c_int32), ("ndim", C.c_int32), ("fortran", C.c_int32),
                ("key", C.c_int32), ("shape", C.c_int64 * 4)]


DT_F32, DT_F64 = 0, 1
PICKLE_UNSUPPORTED = 2


class GemWindowStats(C.Structure):
    _fields_ = [("n_iter", C.c_int32), ("func_evals", C.c_int32), ("final_loss", C.c_float), ("status", C.c_int32)]


# name -> (restype, argtypes); every symbol include/gem_hip.h declares
_P = C.c_void_p
SIGNATURES = {
    "gem_last_error": (C.c_char_p, []),
    "gem_version": (C.c_int, []),
    "gem_create": (C.c_int, [C.POINTER(GemConfig), C.POINTER(_P)]),
    "gem_destroy": (None, [_P]),
    "gem_set_precision": (C.c_int, [_P, C.c_int]),
    "gem_load_vae": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "gem_mean_bone_length": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "gem_encode": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P]),
    "gem_decode": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P]),
    "gem_energy_grad": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.POINTER(GemEnergyWeights), _P, _P, _P, _P, _P]),
    "gem_optimize_stage": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.POINTER(GemEnergyWeights),
                                     C.POINTER(GemLbfgsOpts), _P, _P, _P]),
    "gem_optimize_windows": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.POINTER(GemEnergyWeights),
                                       C.POINTER(GemEnergyWeights), C.POINTER(GemLbfgsOpts), _P, _P, _P, _P]),
    "gem_set_texel_cache": (C.c_int, [_P, C.c_int]),
    "gem_graph_enable": (C.c_int, [_P, C.c_int]),
    "gem_graph_stats": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gem_read_trace": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "gem_merge_windows": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "gem_calculate_errors": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.POINTER(C.c_double), _P, _P]),
    "gem_calculate_errors_chunks": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.POINTER(C.c_double), _P, _P]),
    "gem_lift_skeleton": (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "gem_set_lanes": (C.c_int, [_P, C.c_int]),
    "gem_pickle_scan": (C.c_int, [_P, C.c_int64, C.POINTER(C.c_char_p), C.c_int, C.POINTER(GemPickleArray), C.c_int64, C.POINTER(C.c_int64)]),
    "gem_pickle_gather_f64": (C.c_int, [_P, C.c_int64, C.POINTER(GemPickleArray), C.c_int64, _P]),
    "gem_heat_gather": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "gem_chunk_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.POINTER(_P)]),
    "gem_chunk_close": (None, [_P]),
    "gem_chunk_bytes": (C.c_int64, [_P]),
    "gem_chunk_count": (C.c_int64, [_P, C.c_int]),
    "gem_chunk_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64)]),
    "gem_chunk_offsets": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
    "gem_chunk_gather_f64": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
    "gem_file_stage": (C.c_int, [C.c_char_p, C.c_int, _P, _P, C.c_int64, C.c_int64, C.POINTER(C.c_int64), _P]),
    "gem_profile_enable": (C.c_int, [_P, C.c_int]),
    "gem_profile_read": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "gem_profile_kernels": (C.c_int, [_P, C.c_int, C.c_char_p, C.c_int]),
    "gem_trainer_create": (C.c_int, [C.POINTER(GemConfig), C.POINTER(_P)]),
    "gem_trainer_destroy": (None, [_P]),
    "gem_trainer_sizes": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gem_trainer_upload": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
    "gem_trainer_download": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
    "gem_trainer_set_step": (C.c_int, [_P, C.c_int64]),
    "gem_trainer_step": (C.c_int, [_P, C.c_int, _P, _P, C.POINTER(GemTrainOpts), C.c_int, _P, _P]),
    "gem_trainer_arena": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "gem_trainer_apply": (C.c_int, [_P, C.POINTER(GemTrainOpts), C.c_double, _P]),
}

_lib = None


def load_library(path=None):
    """dlopen the HIP library and bind every declared symbol; raises GemError if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise GemError("HIP library %s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback for the optimiser)" % p)
    # PyTorch-ROCm ships its own libamdhip64.so.7 / libhsa-runtime64; it must be the HIP runtime of the
    # process (device pointers and streams come from torch), so make sure it is loaded first: the
    # DT_NEEDED entry of libgem_hip.so then binds to that copy by soname.
    import torch  # noqa: F401
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError here = header and library out of sync
        fn.restype, fn.argtypes = res, args
    if path is None:
        _lib = lib
    return lib


def check(rc, lib=None):
    if rc != 0:
        msg = (lib or load_library()).gem_last_error()
        raise GemError(msg.decode() if msg else "libgem_hip call failed")


def default_lbfgs_opts(lr=2.0, max_iter=25, tol_change=1e-6):
    """torch.optim.LBFGS defaults as instantiated at optimizer.py:261-262."""
    return GemLbfgsOpts(lr=float(lr), max_iter=int(max_iter), max_eval=int(max_iter) * 5 // 4, history=100, reserved=0,
                        tol_grad=1e-7, tol_change=float(tol_change), c1=1e-4, c2=0.9, ls_tol_change=1e-9)
