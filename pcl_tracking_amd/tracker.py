"""Python mirror of the PCL tracker interface the reference drives (same member names and argument
meaning as the calls at /root/reference/src/auto_tracking.cpp:201-254, 270, 309-310, 673-676, 691-693),
implemented over the C ABI of include/pft.h.  All compute runs in the HIP library; nothing here
computes on the CPU."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Config, PftError
from .scene import PARTICLE_DTYPE, POINT_DTYPE


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class DistanceCoherence:
    """pcl::tracking::DistanceCoherence (auto_tracking.cpp:240-242)"""

    def __init__(self):
        self.weight = 1.0

    def setWeight(self, w):
        self.weight = float(w)


class HSVColorCoherence:
    """pcl::tracking::HSVColorCoherence (auto_tracking.cpp:244-247)"""

    def __init__(self):
        self.weight, self.h_weight, self.s_weight, self.v_weight = 1.0, 1.0, 1.0, 0.0

    def setWeight(self, w):
        self.weight = float(w)

    def setHWeight(self, w):
        self.h_weight = float(w)

    def setSWeight(self, w):
        self.s_weight = float(w)

    def setVWeight(self, w):
        self.v_weight = float(w)


class OctreeSearch:
    """pcl::search::Octree(resolution) (auto_tracking.cpp:250-252)"""

    def __init__(self, resolution):
        self.resolution = float(resolution)


class ApproxNearestPairPointCloudCoherence:
    """pcl::tracking::ApproxNearestPairPointCloudCoherence (auto_tracking.cpp:235-253).  As upstream,
    the class keeps its own search::Octree(0.01); setSearchMethod is accepted and only its resolution
    (which the reference sets to the same 0.01) is used."""

    def __init__(self):
        self.point_coherences = []
        self.maximum_distance = float("inf")
        self.resolution = 0.01

    def addPointCoherence(self, c):
        self.point_coherences.append(c)

    def setSearchMethod(self, search):
        self.resolution = search.resolution

    def setMaximumDistance(self, d):
        self.maximum_distance = float(d)


class NearestPairPointCloudCoherence(ApproxNearestPairPointCloudCoherence):
    """pcl::tracking::NearestPairPointCloudCoherence: the true nearest neighbour instead of the greedy octree
    descent -- the alternative auto_tracking.cpp keeps commented out (:237-238, :249)"""
    exact = True


class ParticleFilterTracker:
    """pcl::tracking::ParticleFilterOMPTracker<PointXYZRGBA, ParticleXYZRPY>, fixed particle number."""

    def __init__(self, threads=16, device_id=0, stream=None, seed=1, rank=0, world_size=1):
        self._L = _lib.load()
        self._cfg = Config()
        self._L.pft_config_default(C.byref(self._cfg))
        self._cfg.device_id = device_id
        self._cfg.stream = stream
        self._cfg.stream_is_external = 0 if stream is None else 1  # 0 is a valid handle: the default stream
        self._cfg.seed = seed
        self._cfg.rank = rank
        self._cfg.world_size = world_size
        self._h = None
        self._trans = np.eye(4, dtype=np.float32)
        self._ref = None
        self._keep = None
        self.threads = threads  # OpenMP thread count of the reference; meaningless on the GPU

    # ---- configuration (before the first compute) ----
    def _cfg_guard(self):
        if self._h is not None:
            raise PftError(7, "configuration is fixed once the handle exists")

    def setTrans(self, m):
        self._trans = np.ascontiguousarray(m, np.float32).reshape(4, 4)
        if self._h is not None:
            self._check(self._L.pft_set_trans(self._h, _ptr(self._trans)))

    def setStepNoiseCovariance(self, cov):
        self._cfg_guard()
        for i in range(6):
            self._cfg.step_noise_cov[i] = float(cov[i])

    def setInitialNoiseCovariance(self, cov):
        self._cfg_guard()
        for i in range(6):
            self._cfg.initial_noise_cov[i] = float(cov[i])

    def setInitialNoiseMean(self, mean):
        self._cfg_guard()
        for i in range(6):
            self._cfg.initial_noise_mean[i] = float(mean[i])

    def setIterationNum(self, n):
        self._cfg_guard()
        self._cfg.iteration_num = int(n)

    def setParticleNum(self, n):
        self._cfg_guard()
        self._cfg.particle_num = int(n)

    def setResampleLikelihoodThr(self, v):
        self._cfg_guard()
        self._cfg.resample_likelihood_thr = float(v)

    def setUseNormal(self, b):
        self._cfg_guard()
        self._cfg.use_normal = 1 if b else 0

    def setAlpha(self, a):
        self._cfg_guard()
        self._cfg.alpha = float(a)

    def setMinIndices(self, n):
        pass  # only read when use_normal_ is true (auto_tracking.cpp:676)

    def setCloudCoherence(self, coh):
        self._cfg_guard()
        self._cfg.max_distance = coh.maximum_distance
        self._cfg.octree_resolution = coh.resolution
        self._cfg.exact_nearest = 1 if getattr(coh, "exact", False) else 0
        kinds = [type(c) for c in coh.point_coherences]
        if kinds != [DistanceCoherence, HSVColorCoherence]:
            raise PftError(1, "supported point coherences: DistanceCoherence then HSVColorCoherence "
                              "(auto_tracking.cpp:240-247)")
        d, h = coh.point_coherences
        self._cfg.distance_weight = d.weight
        self._cfg.hsv_weight = h.weight
        self._cfg.h_weight, self._cfg.s_weight, self._cfg.v_weight = h.h_weight, h.s_weight, h.v_weight

    # ---- handle ----
    def _check(self, status):
        if status != 0:
            detail = self._L.pft_last_error_string(self._h).decode() if self._h else ""
            raise PftError(status, detail)

    def _ensure(self):
        if self._h is None:
            h = C.c_void_p()
            st = self._L.pft_create(C.byref(self._cfg), C.byref(h))
            if st != 0:
                raise PftError(st)
            self._h = h
            self._check(self._L.pft_set_trans(self._h, _ptr(self._trans)))
            if self._ref is not None:
                self._check(self._L.pft_set_reference(self._h, _ptr(self._ref), len(self._ref)))

    def close(self):
        if self._h is not None:
            self._L.pft_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data ----
    def setReferenceCloud(self, cloud):
        self._ref = np.ascontiguousarray(cloud, POINT_DTYPE)
        if self._h is not None:
            self._check(self._L.pft_set_reference(self._h, _ptr(self._ref), len(self._ref)))

    def setInputCloud(self, cloud):
        self._ensure()
        cloud = np.ascontiguousarray(cloud, POINT_DTYPE)
        self._check(self._L.pft_set_input(self._h, _ptr(cloud), len(cloud)))

    def setInputCloudDevice(self, device_ptr, n, keepalive=None):
        """input cloud already resident in HBM (PCL 32-byte layout)"""
        self._ensure()
        self._keep = keepalive
        self._check(self._L.pft_set_input_device(self._h, C.c_void_p(device_ptr), n))

    def compute(self):
        self._ensure()
        self._check(self._L.pft_compute(self._h))

    def synchronize(self):
        self._check(self._L.pft_synchronize(self._h))

    def getResult(self):
        out = np.zeros(1, PARTICLE_DTYPE)
        self._check(self._L.pft_get_result(self._h, _ptr(out)))
        return out[0]

    def getParticles(self):
        n = C.c_size_t()
        self._check(self._L.pft_get_particles(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, PARTICLE_DTYPE)
        if n.value:
            self._check(self._L.pft_get_particles(self._h, _ptr(out), n.value, C.byref(n)))
        return out

    def toEigenMatrix(self, particle):
        p = np.ascontiguousarray(particle, PARTICLE_DTYPE).reshape(1)
        m = np.zeros(16, np.float32)
        self._L.pft_to_matrix(_ptr(p), _ptr(m))
        return m.reshape(4, 4)

    def getFitRatio(self):
        v = C.c_double()
        self._check(self._L.pft_get_fit_ratio(self._h, C.byref(v)))
        return v.value

    # ---- test hooks (stage-level parity against the oracle) ----
    def setParticles(self, p):
        self._ensure()
        p = np.ascontiguousarray(p, PARTICLE_DTYPE)
        self._check(self._L.pft_set_particles(self._h, _ptr(p), len(p)))

    def evalWeights(self, particles, want_nn=False):
        self._ensure()
        p = np.ascontiguousarray(particles, PARTICLE_DTYPE)
        P, M = len(p), len(self._ref)
        raw = np.zeros(P, np.float32)
        nn_idx = np.zeros((P, M), np.int32) if want_nn else None
        nn_d2 = np.zeros((P, M), np.float32) if want_nn else None
        self._check(self._L.pft_eval_weights(self._h, _ptr(p), P, _ptr(raw), _ptr(nn_idx), _ptr(nn_d2)))
        bbox = np.zeros(6, np.float32)
        self._check(self._L.pft_debug_get_bbox(self._h, _ptr(bbox)))
        n = C.c_size_t()
        self._check(self._L.pft_debug_get_crop(self._h, None, 0, C.byref(n)))
        crop = np.zeros(n.value, np.int32)
        if n.value:
            self._check(self._L.pft_debug_get_crop(self._h, _ptr(crop), n.value, C.byref(n)))
        depth, nl, nn = C.c_int32(), C.c_uint32(), C.c_uint32()
        mn, mx = np.zeros(3), np.zeros(3)
        self._check(self._L.pft_debug_get_octree(self._h, C.byref(depth), _ptr(mn), _ptr(mx), C.byref(nl), C.byref(nn)))
        keys = np.zeros((n.value, 3), np.uint32)
        if n.value:
            self._check(self._L.pft_debug_get_point_keys(self._h, _ptr(keys), n.value))
        q, s = C.c_uint64(), C.c_uint64()
        self._check(self._L.pft_debug_get_scan_stats(self._h, C.byref(q), C.byref(s)))
        return dict(raw=raw, nn_idx=nn_idx, nn_d2=nn_d2, bbox=bbox, crop_idx=crop, octree_depth=depth.value,
                    octree_min=mn, octree_max=mx, n_leaves=nl.value, n_words=nn.value, point_keys=keys,
                    scan_queries=q.value, scan_points=s.value)

    def debugSetLimits(self, max_words=0, sorted_npass=0):
        """error-path tests: lower the octree node capacity / fix the sorted builder's radix passes"""
        self._ensure()
        self._check(self._L.pft_debug_set_limits(self._h, int(max_words), int(sorted_npass)))

    def debugStateSave(self):
        """checkpoint of the filter state between two frames (HBM-resident)"""
        self._check(self._L.pft_debug_state_save(self._h))

    def debugStateRestore(self):
        """back to the checkpoint: one kernel on the handle's stream"""
        self._check(self._L.pft_debug_state_restore(self._h))

    def debugInjectError(self, bits):
        """error-path tests: OR `bits` into the device-side error flags right after the next crop launch"""
        self._ensure()
        self._check(self._L.pft_debug_inject_error(self._h, int(bits)))

    def debugHostStat(self):
        """the pinned status block: last crop size, last depth, flags of the last failed iteration, unreported flags"""
        self._ensure()
        out = np.zeros(4, np.uint32)
        self._check(self._L.pft_debug_get_host_stat(self._h, _ptr(out)))
        return out

    def debugNormalize(self, w):
        self._ensure()
        w = np.array(w, np.float32, copy=True)
        fr = C.c_double()
        self._check(self._L.pft_debug_normalize(self._h, _ptr(w), len(w), C.byref(fr)))
        return w, fr.value

    def debugAlias(self, w):
        self._ensure()
        w = np.ascontiguousarray(w, np.float32)
        a = np.zeros(len(w), np.int32)
        q = np.zeros(len(w), np.float64)
        self._check(self._L.pft_debug_alias(self._h, _ptr(w), len(w), _ptr(a), _ptr(q)))
        return a, q

    def debugWeightedMean(self, p):
        self._ensure()
        p = np.ascontiguousarray(p, PARTICLE_DTYPE)
        out = np.zeros(1, PARTICLE_DTYPE)
        self._check(self._L.pft_debug_weighted_mean(self._h, _ptr(p), len(p), _ptr(out)))
        return out[0]

    def debugInitParticles(self, rep, id_offset, n_local):
        self._ensure()
        rep = np.ascontiguousarray(rep, PARTICLE_DTYPE).reshape(1)
        out = np.zeros(n_local, PARTICLE_DTYPE)
        self._check(self._L.pft_debug_init_particles(self._h, _ptr(rep), id_offset, n_local, _ptr(out)))
        return out

    def debugResample(self, old, a, q, rep, epoch, id_offset=0, n_local=None):
        self._ensure()
        old = np.ascontiguousarray(old, PARTICLE_DTYPE)
        a = np.ascontiguousarray(a, np.int32)
        q = np.ascontiguousarray(q, np.float64)
        rep = np.ascontiguousarray(rep, PARTICLE_DTYPE).reshape(1)
        n_local = len(old) if n_local is None else n_local
        out = np.zeros(n_local, PARTICLE_DTYPE)
        self._check(self._L.pft_debug_resample(self._h, _ptr(old), len(old), _ptr(a), _ptr(q), _ptr(rep), epoch,
                                               id_offset, n_local, _ptr(out)))
        return out

    def debugPoseToMatrix(self, p):
        self._ensure()
        p = np.ascontiguousarray(p, PARTICLE_DTYPE)
        m = np.zeros((len(p), 12), np.float32)
        self._check(self._L.pft_debug_pose_to_matrix(self._h, _ptr(p), len(p), _ptr(m)))
        return m.reshape(len(p), 3, 4)

    # ---- per-kernel HIP-event timing ----
    def profileEnable(self, on=True):
        self._ensure()
        self._check(self._L.pft_profile_enable(self._h, 1 if on else 0))

    def profileReset(self):
        self._check(self._L.pft_profile_reset(self._h))

    def profileGet(self):
        out = {}
        for k in range(_lib.K_COUNT):
            ms, n = C.c_double(), C.c_uint64()
            self._check(self._L.pft_profile_get(self._h, k, C.byref(ms), C.byref(n)))
            out[self._L.pft_kernel_name(k).decode()] = (ms.value, n.value)
        return out


class KLDAdaptiveParticleFilterOMPTracker(ParticleFilterTracker):
    """pcl::tracking::KLDAdaptiveParticleFilterOMPTracker<PointXYZRGBA, ParticleXYZRPY>: what auto_tracking.cpp runs
    unless use_fixed is set (:207-222, :821).  setParticleNum is the initial count; every resample draws until the
    KL bound is met (at most setMaximumParticleNum)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._cfg.kld_adaptive = 1

    def setMaximumParticleNum(self, n):
        self._cfg_guard()
        self._cfg.maximum_particle_num = int(n)

    def setDelta(self, d):
        self._cfg_guard()
        self._cfg.kld_delta = float(d)

    def setEpsilon(self, e):
        self._cfg_guard()
        self._cfg.kld_epsilon = float(e)

    def setBinSize(self, bin_size):
        """bin_size: the six pose steps (x, y, z, roll, pitch, yaw), a ParticleXYZRPY upstream"""
        self._cfg_guard()
        if hasattr(bin_size, "dtype") and bin_size.dtype == PARTICLE_DTYPE:
            bin_size = [float(np.asarray(bin_size).reshape(-1)[0][k]) for k in ("x", "y", "z", "roll", "pitch", "yaw")]
        for i in range(6):
            self._cfg.kld_bin_size[i] = float(bin_size[i])

    def debugKldResample(self, old, a, q, motion, epoch):
        """test hook: the KLD resample alone, with an explicit alias table -> (particles, bins (n,6), k)"""
        self._ensure()
        old = np.ascontiguousarray(old, PARTICLE_DTYPE)
        a = np.ascontiguousarray(a, np.int32)
        q = np.ascontiguousarray(q, np.float64)
        motion = np.ascontiguousarray(motion, PARTICLE_DTYPE).reshape(1)
        cap = self._cfg.maximum_particle_num
        out = np.zeros(cap, PARTICLE_DTYPE)
        bins = np.zeros((cap, 6), np.int32)
        n, k = C.c_uint32(), C.c_uint32()
        self._check(self._L.pft_debug_kld_resample(self._h, _ptr(old), len(old), _ptr(a), _ptr(q), _ptr(motion), epoch,
                                                   _ptr(out), _ptr(bins), C.byref(n), C.byref(k)))
        return out[:n.value].copy(), bins[:n.value].copy(), k.value


def make_reference_tracker(particle_num=400, seed=1, kld=False, **kw):
    """A tracker configured exactly as /root/reference/src/auto_tracking.cpp:187-254 does; kld=True takes the
    use_fixed == false branch (:207-222), the reference's runtime default."""
    if kld:
        t = KLDAdaptiveParticleFilterOMPTracker(threads=16, seed=seed, **kw)
        t.setMaximumParticleNum(500)
        t.setDelta(0.99)
        t.setEpsilon(0.2)
        t.setBinSize([0.1] * 6)
    else:
        t = ParticleFilterTracker(threads=16, seed=seed, **kw)
    step = [0.015 * 0.015] * 6
    step[3] *= 40.0
    step[4] *= 40.0
    step[5] *= 40.0
    t.setTrans(np.eye(4, dtype=np.float32))
    t.setStepNoiseCovariance(step)
    t.setInitialNoiseCovariance([0.00001] * 6)
    t.setInitialNoiseMean([0.0] * 6)
    t.setIterationNum(2)
    t.setParticleNum(particle_num)
    t.setResampleLikelihoodThr(0.00)
    t.setUseNormal(False)
    coherence = ApproxNearestPairPointCloudCoherence()
    coherence.addPointCoherence(DistanceCoherence())
    color = HSVColorCoherence()
    color.setWeight(0.1)
    coherence.addPointCoherence(color)
    coherence.setSearchMethod(OctreeSearch(0.01))
    coherence.setMaximumDistance(0.1)
    t.setCloudCoherence(coherence)
    return t
