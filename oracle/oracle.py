"""ctypes wrapper over oracle/_build/libpft_oracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
PARITY UNPINNED: the oracle restates PCL 1.8.0 (absent from the container); see pft_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpft_oracle.so")

POINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4"), ("rgba", "<u4"), ("pad", "<u4", (3,))]
)
PARTICLE_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4"),
     ("roll", "<f4"), ("pitch", "<f4"), ("yaw", "<f4"), ("weight", "<f4")]
)
assert POINT_DTYPE.itemsize == 32 and PARTICLE_DTYPE.itemsize == 32


class Config(C.Structure):
    _fields_ = [
        ("particle_num", C.c_int32), ("iteration_num", C.c_int32),
        ("step_cov", C.c_double * 6), ("init_cov", C.c_double * 6), ("init_mean", C.c_double * 6),
        ("alpha", C.c_double), ("max_distance", C.c_double), ("octree_resolution", C.c_double),
        ("distance_weight", C.c_double), ("hsv_weight", C.c_double),
        ("h_weight", C.c_double), ("s_weight", C.c_double), ("v_weight", C.c_double),
        ("hsv_pcl180_argorder", C.c_int32), ("threads", C.c_int32), ("emulate_pcl_alloc", C.c_int32),
        ("seed", C.c_uint64),
        ("kld_adaptive", C.c_int32), ("kld_max_particles", C.c_int32), ("kld_delta", C.c_double),
        ("kld_epsilon", C.c_double), ("kld_bin_size", C.c_double * 6), ("motion_ratio", C.c_double),
        ("exact_nearest", C.c_int32),
    ]


def build(force=False):
    """Compile the C restatement with the committed Makefile (building the checker is not using it)."""
    srcs = [os.path.join(_HERE, f) for f in ("pft_oracle.c", "pft_oracle_filters.c", "pft_oracle_app.c", "pft_oracle.h", "Makefile")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    vp, f32, f64, i32, u32, u64, sz = C.c_void_p, C.c_float, C.c_double, C.c_int32, C.c_uint32, C.c_uint64, C.c_size_t
    P = C.POINTER
    L.orc_config_default.argtypes = [P(Config)]
    L.orc_get_transformation.argtypes = [f32] * 6 + [vp]
    L.orc_to_state.argtypes = [vp, vp]
    L.orc_transform_cloud.argtypes = [vp, sz, vp, vp]
    L.orc_div_table.argtypes = [C.c_int]
    L.orc_div_table.restype = C.c_int
    L.orc_rgb2hsv.argtypes = [C.c_int] * 3 + [P(f32)] * 3
    L.orc_rgb2hsv_int.argtypes = [C.c_int] * 3 + [P(C.c_int)] * 3
    L.orc_hsv_coherence.argtypes = [P(Config), u32, u32]
    L.orc_hsv_coherence.restype = f64
    L.orc_distance_coherence.argtypes = [P(Config), vp, vp]
    L.orc_distance_coherence.restype = f64
    L.orc_octree_build.argtypes = [vp, sz, f64, C.c_int]
    L.orc_octree_build.restype = vp
    L.orc_octree_free.argtypes = [vp]
    L.orc_octree_info.argtypes = [vp, P(C.c_int), vp, P(sz), P(sz)]
    L.orc_octree_point_key.argtypes = [vp, sz, vp]
    L.orc_octree_approx_nearest.argtypes = [vp, vp, P(C.c_int), P(f32)]
    L.orc_octree_approx_nearest.restype = C.c_int
    L.orc_normalize_weights.argtypes = [vp, sz, f64, P(f64)]
    L.orc_gen_alias_table.argtypes = [vp, sz, vp, vp]
    L.orc_weighted_mean.argtypes = [vp, sz, vp]
    L.orc_normalize_weights_tree.argtypes = [vp, sz, f64, P(f64)]
    L.orc_weighted_mean_tree.argtypes = [vp, sz, vp]
    L.orc_philox4x32.argtypes = [vp, vp, vp]
    L.orc_u53.argtypes = [u32, u32]
    L.orc_u53.restype = f64
    L.orc_rng_normal_pair.argtypes = [u64, u32, u32, u32, u32, P(f64), P(f64)]
    L.orc_rng_uniform.argtypes = [u64, u32, u32, u32, u32]
    L.orc_rng_uniform.restype = f64
    L.orc_init_particles.argtypes = [P(Config), vp, u32, sz, vp]
    L.orc_resample.argtypes = [P(Config), vp, sz, vp, vp, vp, u32, u32, sz, vp]
    L.orc_tracker_create.argtypes = [P(Config)]
    L.orc_tracker_create.restype = vp
    L.orc_tracker_destroy.argtypes = [vp]
    L.orc_tracker_set_reference.argtypes = [vp, vp, sz]
    L.orc_tracker_set_trans.argtypes = [vp, vp]
    L.orc_tracker_set_input.argtypes = [vp, vp, sz]
    L.orc_tracker_compute.argtypes = [vp]
    L.orc_tracker_compute.restype = C.c_int
    L.orc_tracker_get_result.argtypes = [vp, vp]
    L.orc_tracker_get_particles.argtypes = [vp, vp, sz]
    L.orc_tracker_get_particles.restype = sz
    L.orc_tracker_set_particles.argtypes = [vp, vp, sz]
    L.orc_tracker_fit_ratio.argtypes = [vp]
    L.orc_tracker_fit_ratio.restype = f64
    L.orc_tracker_eval_weights.argtypes = [vp, vp, sz, vp, vp, vp, vp, sz, vp, P(C.c_int), vp, P(u64), P(u64)]
    L.orc_tracker_eval_weights.restype = sz
    L.orc_tracker_stage_times.argtypes = [vp, vp]
    L.orc_tracker_set_matrix_override.argtypes = [vp, vp]
    L.orc_tracker_set_bbox_override.argtypes = [vp, vp]
    L.orc_tracker_set_bbox_only.argtypes = [vp, C.c_int]
    L.orc_tracker_set_trig_mode.argtypes = [vp, C.c_int]
    L.orc_tracker_set_sum_mode.argtypes = [vp, C.c_int]
    L.orc_kld_normal_quantile.argtypes = [f64]
    L.orc_kld_normal_quantile.restype = f64
    L.orc_kld_bound.argtypes = [C.c_int, f64, f64]
    L.orc_kld_bound.restype = f64
    L.orc_kld_resample.argtypes = [P(Config), vp, sz, vp, vp, vp, u32, vp, vp, P(i32)]
    L.orc_kld_resample.restype = sz
    L.orc_pass_through.argtypes = [vp, sz, C.c_int, f32, f32, C.c_int, vp]
    L.orc_pass_through.restype = sz
    L.orc_approx_voxel_grid.argtypes = [vp, sz, vp, u32, vp]
    L.orc_approx_voxel_grid.restype = sz
    L.orc_voxel_grid.argtypes = [vp, sz, vp, vp]
    L.orc_voxel_grid.restype = sz
    L.orc_remove_zero_points.argtypes = [vp, sz, vp]
    L.orc_remove_zero_points.restype = sz
    L.orc_compute_3d_centroid.argtypes = [vp, sz, C.c_int, vp]
    L.orc_compute_3d_centroid.restype = sz
    L.orc_recentre_model.argtypes = [vp, sz, vp, vp, vp]
    L.orc_object_position.argtypes = [vp, sz, vp, vp, vp]
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_config(**kw):
    c = Config()
    lib().orc_config_default(C.byref(c))
    for k, v in kw.items():
        if k in ("step_cov", "init_cov", "init_mean", "kld_bin_size"):
            for i in range(6):
                getattr(c, k)[i] = float(v[i])
        else:
            setattr(c, k, v)
    return c


def get_transformation(x, y, z, roll, pitch, yaw):
    m = np.zeros(16, np.float32)
    lib().orc_get_transformation(x, y, z, roll, pitch, yaw, _ptr(m))
    return m.reshape(4, 4)


def to_state(m):
    m = np.ascontiguousarray(m, np.float32).reshape(16)
    out = np.zeros(1, PARTICLE_DTYPE)
    lib().orc_to_state(_ptr(m), _ptr(out))
    return out[0]


def transform_cloud(pts, m):
    pts = np.ascontiguousarray(pts, POINT_DTYPE)
    m = np.ascontiguousarray(m, np.float32).reshape(16)
    out = np.zeros_like(pts)
    lib().orc_transform_cloud(_ptr(pts), len(pts), _ptr(m), _ptr(out))
    return out


def rgb2hsv(r, g, b):
    h, s, v = C.c_float(), C.c_float(), C.c_float()
    lib().orc_rgb2hsv(r, g, b, C.byref(h), C.byref(s), C.byref(v))
    return h.value, s.value, v.value


def rgb2hsv_int(r, g, b):
    h, s, v = C.c_int(), C.c_int(), C.c_int()
    lib().orc_rgb2hsv_int(r, g, b, C.byref(h), C.byref(s), C.byref(v))
    return h.value, s.value, v.value


def hsv_coherence(cfg, src_rgba, tgt_rgba):
    return lib().orc_hsv_coherence(C.byref(cfg), int(src_rgba), int(tgt_rgba))


def distance_coherence(cfg, s, t):
    s = np.ascontiguousarray(s, POINT_DTYPE).reshape(1)
    t = np.ascontiguousarray(t, POINT_DTYPE).reshape(1)
    return lib().orc_distance_coherence(C.byref(cfg), _ptr(s), _ptr(t))


class Octree:
    def __init__(self, pts, resolution=0.01, emulate_pcl_alloc=0):
        self.pts = np.ascontiguousarray(pts, POINT_DTYPE)
        self.h = lib().orc_octree_build(_ptr(self.pts), len(self.pts), resolution, emulate_pcl_alloc)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_octree_free(self.h)
            self.h = None

    def info(self):
        d = C.c_int()
        b = np.zeros(6, np.float64)
        lc, bc = C.c_size_t(), C.c_size_t()
        lib().orc_octree_info(self.h, C.byref(d), _ptr(b), C.byref(lc), C.byref(bc))
        return dict(depth=d.value, min=b[:3].copy(), max=b[3:].copy(), leaves=lc.value, branches=bc.value)

    def point_keys(self):
        k = np.zeros((len(self.pts), 3), np.uint32)
        tmp = np.zeros(3, np.uint32)
        for i in range(len(self.pts)):
            lib().orc_octree_point_key(self.h, i, _ptr(tmp))
            k[i] = tmp
        return k

    def approx_nearest(self, q):
        q = np.ascontiguousarray(q, POINT_DTYPE).reshape(-1)
        idx = np.full(len(q), -1, np.int32)
        d2 = np.full(len(q), np.inf, np.float32)
        i, d = C.c_int(), C.c_float()
        for n in range(len(q)):
            if lib().orc_octree_approx_nearest(self.h, C.c_void_p(q[n:n + 1].ctypes.data), C.byref(i), C.byref(d)):
                idx[n], d2[n] = i.value, d.value
        return idx, d2


def normalize_weights(w, alpha=15.0):
    w = np.array(w, np.float32, copy=True)
    fr = C.c_double()
    lib().orc_normalize_weights(_ptr(w), len(w), alpha, C.byref(fr))
    return w, fr.value


def normalize_weights_tree(w, alpha=15.0):
    """normalizeWeight with the weight sum taken as the adjacent-pair tree (sum mode 1)"""
    w = np.array(w, np.float32, copy=True)
    fr = C.c_double()
    lib().orc_normalize_weights_tree(_ptr(w), len(w), alpha, C.byref(fr))
    return w, fr.value


def gen_alias_table(w):
    w = np.ascontiguousarray(w, np.float32)
    a = np.zeros(len(w), np.int32)
    q = np.zeros(len(w), np.float64)
    lib().orc_gen_alias_table(_ptr(w), len(w), _ptr(a), _ptr(q))
    return a, q


def weighted_mean(p):
    p = np.ascontiguousarray(p, PARTICLE_DTYPE)
    out = np.zeros(1, PARTICLE_DTYPE)
    lib().orc_weighted_mean(_ptr(p), len(p), _ptr(out))
    return out[0]


def weighted_mean_tree(p):
    """update() with the six weighted-pose sums taken as adjacent-pair trees in double (sum mode 1)"""
    p = np.ascontiguousarray(p, PARTICLE_DTYPE)
    out = np.zeros(1, PARTICLE_DTYPE)
    lib().orc_weighted_mean_tree(_ptr(p), len(p), _ptr(out))
    return out[0]


def philox4x32(ctr, key):
    ctr = np.ascontiguousarray(ctr, np.uint32)
    key = np.ascontiguousarray(key, np.uint32)
    out = np.zeros(4, np.uint32)
    lib().orc_philox4x32(_ptr(ctr), _ptr(key), _ptr(out))
    return out


def rng_normal_pair(seed, pid, slot, epoch, purpose):
    a, b = C.c_double(), C.c_double()
    lib().orc_rng_normal_pair(seed, pid, slot, epoch, purpose, C.byref(a), C.byref(b))
    return a.value, b.value


def rng_uniform(seed, pid, slot, epoch, purpose):
    return lib().orc_rng_uniform(seed, pid, slot, epoch, purpose)


def init_particles(cfg, rep, id_offset, n_local):
    rep = np.ascontiguousarray(rep, PARTICLE_DTYPE).reshape(1)
    out = np.zeros(n_local, PARTICLE_DTYPE)
    lib().orc_init_particles(C.byref(cfg), _ptr(rep), id_offset, n_local, _ptr(out))
    return out


def resample(cfg, old, a, q, rep, epoch, id_offset=0, n_local=None):
    old = np.ascontiguousarray(old, PARTICLE_DTYPE)
    a = np.ascontiguousarray(a, np.int32)
    q = np.ascontiguousarray(q, np.float64)
    rep = np.ascontiguousarray(rep, PARTICLE_DTYPE).reshape(1)
    n_local = len(old) if n_local is None else n_local
    out = np.zeros(n_local, PARTICLE_DTYPE)
    lib().orc_resample(C.byref(cfg), _ptr(old), len(old), _ptr(a), _ptr(q), _ptr(rep), epoch, id_offset, n_local,
                       _ptr(out))
    return out


def kld_normal_quantile(u):
    return lib().orc_kld_normal_quantile(u)


def kld_bound(k, delta=0.99, epsilon=0.2):
    return lib().orc_kld_bound(k, delta, epsilon)


def kld_resample(cfg, old, a, q, motion, epoch):
    """KLDAdaptiveParticleFilterTracker::resample -> (new particles, bins (n,6), distinct-bin count k)"""
    old = np.ascontiguousarray(old)
    out = np.zeros(cfg.kld_max_particles, PARTICLE_DTYPE)
    bins = np.zeros((cfg.kld_max_particles, 6), np.int32)
    k = C.c_int32()
    a = np.ascontiguousarray(a, np.int32)
    q = np.ascontiguousarray(q, np.float64)
    motion = np.ascontiguousarray(motion)
    n = lib().orc_kld_resample(C.byref(cfg), _ptr(old), len(old), _ptr(a), _ptr(q), _ptr(motion), epoch,
                               _ptr(out), _ptr(bins), C.byref(k))
    return out[:n].copy(), bins[:n].copy(), k.value


def pass_through(pts, field="z", lo=0.0, hi=10.0, negative=False):
    """PassThrough(field, [lo, hi]) -> indices kept (auto_tracking.cpp:536-547 uses z in [0, 10])"""
    pts = np.ascontiguousarray(pts)
    idx = np.zeros(len(pts), np.int32)
    n = lib().orc_pass_through(_ptr(pts), len(pts), "xyz".index(field), lo, hi, int(negative), _ptr(idx))
    return idx[:n].copy()


def approx_voxel_grid(pts, leaf=0.01, hist_size=512):
    """ApproximateVoxelGrid(leaf) (auto_tracking.cpp:563-575) -> output cloud"""
    pts = np.ascontiguousarray(pts)
    out = np.zeros(max(1, len(pts)), pts.dtype)
    lf = np.asarray([leaf] * 3 if np.isscalar(leaf) else leaf, np.float32)
    n = lib().orc_approx_voxel_grid(_ptr(pts), len(pts), _ptr(lf), hist_size, _ptr(out))
    return out[:n].copy()


def voxel_grid(pts, leaf=0.01):
    """VoxelGrid(leaf) (auto_tracking.cpp:549-561) -> output cloud, or None when PCL would refuse the leaf size"""
    pts = np.ascontiguousarray(pts)
    out = np.zeros(max(1, len(pts)), pts.dtype)
    lf = np.asarray([leaf] * 3 if np.isscalar(leaf) else leaf, np.float32)
    n = lib().orc_voxel_grid(_ptr(pts), len(pts), _ptr(lf), _ptr(out))
    if n == C.c_size_t(-1).value:
        return None
    return out[:n].copy()


def remove_zero_points(pts):
    """removeZeroPoints (/root/reference/src/auto_tracking.cpp:577-595)"""
    pts = np.ascontiguousarray(pts, POINT_DTYPE)
    out = np.zeros(max(len(pts), 1), POINT_DTYPE)
    n = lib().orc_remove_zero_points(_ptr(pts), len(pts), _ptr(out))
    return out[:n].copy()


def compute_3d_centroid(pts, is_dense=True):
    """pcl::compute3DCentroid<PointT, float> -> (centroid[4] float32, number of points used)"""
    pts = np.ascontiguousarray(pts, POINT_DTYPE)
    c = np.zeros(4, np.float32)
    n = lib().orc_compute_3d_centroid(_ptr(pts), len(pts), 1 if is_dense else 0, _ptr(c))
    return c, n


def recentre_model(pts, centroid):
    """:663-668 -> (re-centred cloud, trans 4x4 float32 as handed to setTrans)"""
    pts = np.ascontiguousarray(pts, POINT_DTYPE)
    c = np.ascontiguousarray(centroid, np.float32)
    out = np.zeros(max(len(pts), 1), POINT_DTYPE)
    t = np.zeros(16, np.float32)
    lib().orc_recentre_model(_ptr(pts), len(pts), _ptr(c), _ptr(out), _ptr(t))
    return out[:len(pts)].copy(), t.reshape(4, 4)


def object_position(reference_full, result):
    """drawResult + viz_cb: (moved cloud, centroid[4]) of the full-resolution model under the result pose"""
    ref = np.ascontiguousarray(reference_full, POINT_DTYPE)
    r = np.ascontiguousarray(result, PARTICLE_DTYPE).reshape(1)
    moved = np.zeros(max(len(ref), 1), POINT_DTYPE)
    c = np.zeros(4, np.float32)
    lib().orc_object_position(_ptr(ref), len(ref), _ptr(r), _ptr(moved), _ptr(c))
    return moved[:len(ref)].copy(), c


class Tracker:
    """Mirror of the calls /root/reference/src/auto_tracking.cpp makes on the PCL tracker."""

    def __init__(self, cfg=None):
        self.cfg = cfg if cfg is not None else default_config()
        self.h = lib().orc_tracker_create(C.byref(self.cfg))
        self._keep = {}

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_tracker_destroy(self.h)
            self.h = None

    def set_reference(self, pts):
        pts = np.ascontiguousarray(pts, POINT_DTYPE)
        lib().orc_tracker_set_reference(self.h, _ptr(pts), len(pts))
        self.M = len(pts)

    def set_trans(self, m):
        m = np.ascontiguousarray(m, np.float32).reshape(16)
        lib().orc_tracker_set_trans(self.h, _ptr(m))

    def set_input(self, pts):
        pts = np.ascontiguousarray(pts, POINT_DTYPE)
        self._keep["input"] = pts
        lib().orc_tracker_set_input(self.h, _ptr(pts), len(pts))

    def compute(self):
        return lib().orc_tracker_compute(self.h)

    def get_result(self):
        out = np.zeros(1, PARTICLE_DTYPE)
        lib().orc_tracker_get_result(self.h, _ptr(out))
        return out[0]

    def get_particles(self):
        n = lib().orc_tracker_get_particles(self.h, None, 0)
        out = np.zeros(n, PARTICLE_DTYPE)
        lib().orc_tracker_get_particles(self.h, _ptr(out), n)
        return out

    def set_particles(self, p):
        p = np.ascontiguousarray(p, PARTICLE_DTYPE)
        lib().orc_tracker_set_particles(self.h, _ptr(p), len(p))

    def fit_ratio(self):
        return lib().orc_tracker_fit_ratio(self.h)

    def set_sum_mode(self, mode):
        """tests only: 0 = PCL's sequential weight sum and weighted mean (default); 1 = the adjacent-pair tree order the
        product specifies for its parallel reductions"""
        lib().orc_tracker_set_sum_mode(self.h, int(mode))

    def set_trig_mode(self, mode):
        """tests only: 0 = A1 with cosf / sinf as PCL (default); 1 = double sin / cos rounded to float, as the
        product's device code forms the matrix (identical matrices on both sides: bit-level long-run comparison)"""
        lib().orc_tracker_set_trig_mode(self.h, int(mode))

    def stage_times(self):
        s = np.zeros(7, np.float64)
        lib().orc_tracker_stage_times(self.h, _ptr(s))
        return s

    def bbox_of(self, particles):
        """calcBoundingBox over the transformed reference clouds of these particles:
        x_min,x_max,y_min,y_max,z_min,z_max"""
        p = np.ascontiguousarray(particles, PARTICLE_DTYPE)
        bbox = np.zeros(6, np.float64)
        lib().orc_tracker_set_bbox_only(self.h, 1)
        lib().orc_tracker_eval_weights(self.h, _ptr(p), len(p), None, None, None, None, 0, _ptr(bbox), None, None,
                                       None, None)
        lib().orc_tracker_set_bbox_only(self.h, 0)
        return bbox

    def eval_weights(self, particles, want_nn=False, mats=None, bbox=None):
        """mats: optional (P,3,4) or (P,4,4) float32 matrices overriding toEigenMatrix(particle);
        bbox: optional crop box overriding calcBoundingBox (sharded-path tests)"""
        p = np.ascontiguousarray(particles, PARTICLE_DTYPE)
        bb = None if bbox is None else np.ascontiguousarray(bbox, np.float64)
        lib().orc_tracker_set_bbox_override(self.h, _ptr(bb))
        m16 = None
        if mats is not None:
            mats = np.asarray(mats, np.float32)
            m16 = np.zeros((len(p), 4, 4), np.float32)
            m16[:, 3, 3] = 1.0
            m16[:, :mats.shape[1], :] = mats
            m16 = np.ascontiguousarray(m16)
        lib().orc_tracker_set_matrix_override(self.h, _ptr(m16))
        P, M = len(p), self.M
        N = len(self._keep["input"])
        raw = np.zeros(P, np.float32)
        nn_idx = np.zeros(P * M, np.int32) if want_nn else None
        nn_d2 = np.zeros(P * M, np.float32) if want_nn else None
        crop = np.zeros(max(N, 1), np.int32)
        bbox = np.zeros(6, np.float64)
        depth = C.c_int()
        ob = np.zeros(6, np.float64)
        sq, sp = C.c_uint64(), C.c_uint64()
        nc = lib().orc_tracker_eval_weights(self.h, _ptr(p), P, _ptr(raw), _ptr(nn_idx), _ptr(nn_d2), _ptr(crop), N,
                                            _ptr(bbox), C.byref(depth), _ptr(ob), C.byref(sq), C.byref(sp))
        lib().orc_tracker_set_matrix_override(self.h, None)
        lib().orc_tracker_set_bbox_override(self.h, None)
        return dict(raw=raw, nn_idx=None if nn_idx is None else nn_idx.reshape(P, M),
                    nn_d2=None if nn_d2 is None else nn_d2.reshape(P, M), crop_idx=crop[:nc].copy(), bbox=bbox,
                    octree_depth=depth.value, octree_min=ob[:3].copy(), octree_max=ob[3:].copy(),
                    scan_queries=sq.value, scan_points=sp.value)
