"""ctypes loader for the C-ABI library (include/pft.h).  No fallback: if the HIP extension is not
built, or there is no GPU, the product path raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PFT_LIB_PATH: another build of the same HIP library (A/B timing of compile-time variants, tools/build_variant.py)
LIB_PATH = os.environ.get("PFT_LIB_PATH") or os.path.join(_HERE, "_build", "libpft_hip.so")

PFT_ABI_VERSION = 4
K_RESAMPLE, K_AABB, K_CROP, K_OCTREE, K_LIKELIHOOD, K_POPULATION, K_PACK, K_COUNT = range(8)

STATUS = {0: "ok", 1: "invalid argument", 2: "no input cloud", 3: "no reference cloud", 4: "no usable HIP device",
          5: "HIP error", 6: "capacity exceeded", 7: "invalid state"}


class PftError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        super().__init__("pft status %d (%s) %s" % (status, STATUS.get(status, "?"), detail))


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("device_id", C.c_int32), ("stream", C.c_void_p),
        ("stream_is_external", C.c_int32),
        ("particle_num", C.c_int32), ("iteration_num", C.c_int32),
        ("step_noise_cov", C.c_double * 6), ("initial_noise_cov", C.c_double * 6),
        ("initial_noise_mean", C.c_double * 6),
        ("alpha", C.c_double), ("resample_likelihood_thr", C.c_double), ("max_distance", C.c_double),
        ("octree_resolution", C.c_double), ("distance_weight", C.c_double), ("hsv_weight", C.c_double),
        ("h_weight", C.c_double), ("s_weight", C.c_double), ("v_weight", C.c_double),
        ("hsv_pcl180_argorder", C.c_int32), ("use_normal", C.c_int32), ("seed", C.c_uint64),
        ("rank", C.c_int32), ("world_size", C.c_int32),
        ("max_reference_points", C.c_uint32), ("max_input_points", C.c_uint32),
        ("kld_adaptive", C.c_int32), ("maximum_particle_num", C.c_int32), ("kld_delta", C.c_double),
        ("kld_epsilon", C.c_double), ("kld_bin_size", C.c_double * 6), ("motion_ratio", C.c_double),
        ("exact_nearest", C.c_int32),
    ]


class FilterConfig(C.Structure):
    """pft_filter_config (include/pft_filters.h)"""
    _fields_ = [
        ("abi_version", C.c_uint32), ("device_id", C.c_int32), ("stream", C.c_void_p),
        ("stream_is_external", C.c_int32),
        ("pass_enable", C.c_int32), ("pass_field", C.c_int32), ("pass_min", C.c_float), ("pass_max", C.c_float),
        ("pass_negative", C.c_int32),
        ("voxel_mode", C.c_int32), ("leaf_size", C.c_float * 3), ("approx_hist_size", C.c_uint32),
        ("max_points", C.c_uint32),
    ]


VOXEL_NONE, VOXEL_APPROX, VOXEL_EXACT = 0, 1, 2

# every symbol include/*.h declare: (name, restype, argtypes)
_vp, _sz, _i32, _u32, _u64, _f64 = C.c_void_p, C.c_size_t, C.c_int32, C.c_uint32, C.c_uint64, C.c_double
_P = C.POINTER
SYMBOLS = [
    ("pft_config_default", None, [_P(Config)]),
    ("pft_status_string", C.c_char_p, [C.c_int]),
    ("pft_create", C.c_int, [_P(Config), _P(_vp)]),
    ("pft_destroy", None, [_vp]),
    ("pft_last_error_string", C.c_char_p, [_vp]),
    ("pft_set_reference", C.c_int, [_vp, _vp, _sz]),
    ("pft_set_trans", C.c_int, [_vp, _vp]),
    ("pft_set_input", C.c_int, [_vp, _vp, _sz]),
    ("pft_set_input_device", C.c_int, [_vp, _vp, _sz]),
    ("pft_compute", C.c_int, [_vp]),
    ("pft_get_result", C.c_int, [_vp, _vp]),
    ("pft_get_particles", C.c_int, [_vp, _vp, _sz, _P(_sz)]),
    ("pft_to_matrix", None, [_vp, _vp]),
    ("pft_to_state", None, [_vp, _vp]),
    ("pft_get_fit_ratio", C.c_int, [_vp, _P(_f64)]),
    ("pft_synchronize", C.c_int, [_vp]),
    ("pft_dist_bind", C.c_int, [_vp, _vp, _vp, _vp]),
    ("pft_dist_begin_frame", C.c_int, [_vp]),
    ("pft_dist_phase_a", C.c_int, [_vp, C.c_int]),
    ("pft_dist_phase_b", C.c_int, [_vp]),
    ("pft_dist_phase_c", C.c_int, [_vp]),
    ("pft_set_particles", C.c_int, [_vp, _vp, _sz]),
    ("pft_eval_weights", C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp]),
    ("pft_debug_get_bbox", C.c_int, [_vp, _vp]),
    ("pft_debug_get_crop", C.c_int, [_vp, _vp, _sz, _P(_sz)]),
    ("pft_debug_get_octree", C.c_int, [_vp, _P(_i32), _vp, _vp, _P(_u32), _P(_u32)]),
    ("pft_debug_get_point_keys", C.c_int, [_vp, _vp, _sz]),
    ("pft_debug_get_scan_stats", C.c_int, [_vp, _P(_u64), _P(_u64)]),
    ("pft_debug_set_limits", C.c_int, [_vp, _u32, C.c_int]),
    ("pft_debug_inject_error", C.c_int, [_vp, _u32]),
    ("pft_debug_state_save", C.c_int, [_vp]),
    ("pft_debug_state_restore", C.c_int, [_vp]),
    ("pft_debug_get_host_stat", C.c_int, [_vp, _vp]),
    ("pft_debug_get_ticks", C.c_int, [_vp, _vp]),
    ("pft_debug_get_descent_stats", C.c_int, [_vp, _vp]),
    ("pft_debug_aabb_support_subset", C.c_int, [_vp, C.c_size_t, _vp, _vp]),
    ("pft_debug_likelihood_occupancy", C.c_int, []),
    ("pft_debug_normalize", C.c_int, [_vp, _vp, _sz, _P(_f64)]),
    ("pft_debug_alias", C.c_int, [_vp, _vp, _sz, _vp, _vp]),
    ("pft_debug_weighted_mean", C.c_int, [_vp, _vp, _sz, _vp]),
    ("pft_debug_init_particles", C.c_int, [_vp, _vp, _u32, _sz, _vp]),
    ("pft_debug_resample", C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp, _u32, _u32, _sz, _vp]),
    ("pft_debug_pose_to_matrix", C.c_int, [_vp, _vp, _sz, _vp]),
    ("pft_debug_kld_resample", C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp, _u32, _vp, _vp, _P(_u32), _P(_u32)]),
    ("pft_kld_normal_quantile", _f64, [_f64]),
    ("pft_kld_bound", _f64, [C.c_int, _f64, _f64]),
    ("pft_profile_enable", C.c_int, [_vp, C.c_int]),
    ("pft_profile_get", C.c_int, [_vp, C.c_int, _P(_f64), _P(_u64)]),
    ("pft_profile_reset", C.c_int, [_vp]),
    ("pft_kernel_name", C.c_char_p, [C.c_int]),
    # include/pft_filters.h
    ("pft_filter_default_config", None, [_P(FilterConfig)]),
    ("pft_filter_create", C.c_int, [_P(FilterConfig), _P(_vp)]),
    ("pft_filter_destroy", None, [_vp]),
    ("pft_filter_last_error_string", C.c_char_p, [_vp]),
    ("pft_filter_apply", C.c_int, [_vp, _vp, _sz]),
    ("pft_filter_apply_device", C.c_int, [_vp, _vp, _sz]),
    ("pft_filter_counts", C.c_int, [_vp, _P(_sz), _P(_sz)]),
    ("pft_filter_output_device", C.c_int, [_vp, _P(_vp), _P(_sz)]),
    ("pft_filter_get_output", C.c_int, [_vp, _vp, _sz, _P(_sz)]),
    ("pft_filter_get_pass_indices", C.c_int, [_vp, _vp, _sz, _P(_sz)]),
    ("pft_filter_last_ms", C.c_int, [_vp, _P(_f64)]),
]

# exported by the diagnostic variant library only (tools/build_variant.py diag -DPFT_DIAG): bound when present
DIAG_SYMBOLS = [
    ("pft_debug_set_ablate", None, [C.c_int]),
]

_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64;
    libpft_hip.so asks for libamdhip64.so.7 by SONAME.  If torch is imported first the loader hands us
    torch's copy and all is well; if we load /opt/rocm's copy first, torch later finds no GPU.  So when
    PyTorch is installed (it is not imported here) its copy is loaded first, whatever the import order."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(p):
        try:
            C.CDLL(p, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """dlopen the HIP extension. Raises if it was not built: the product has no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -m pcl_tracking_amd.build` (hipcc, gfx950). "
            "pcl_tracking_amd has no CPU fallback." % LIB_PATH)
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        f = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        f.restype = res
        f.argtypes = args
    for name, res, args in DIAG_SYMBOLS:
        f = getattr(L, name, None)
        if f is not None:
            f.restype = res
            f.argtypes = args
    _lib = L
    return L
