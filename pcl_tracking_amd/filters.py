"""Python mirror of the PCL filter classes the reference runs in front of the tracker (same member names
as the calls at /root/reference/src/auto_tracking.cpp:536-575), over the C ABI of include/pft_filters.h.
All compute runs in the HIP library; nothing here computes on the CPU.

    pass_ = PassThrough(); pass_.setFilterFieldName("z"); pass_.setFilterLimits(0, 10)
    pass_.setInputCloud(cloud); kept = pass_.filter()
    grid = ApproximateVoxelGrid(); grid.setLeafSize(0.01, 0.01, 0.01); grid.setInputCloud(kept); out = grid.filter()

InputFilter fuses PassThrough + voxel grid into one device pipeline whose output stays in HBM
(`filterDevice` -> (device pointer, n), ready for ParticleFilterTracker.setInputCloudDevice)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import VOXEL_APPROX, VOXEL_EXACT, VOXEL_NONE, FilterConfig, PftError
from .scene import POINT_DTYPE


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class InputFilter:
    """PassThrough and / or a voxel grid as one device pipeline (one handle of include/pft_filters.h)."""

    def __init__(self, device_id=0, stream=None):
        self._L = _lib.load()
        self._cfg = FilterConfig()
        self._L.pft_filter_default_config(C.byref(self._cfg))
        self._cfg.device_id = device_id
        if stream is not None:
            self._cfg.stream = stream
            self._cfg.stream_is_external = 1
        self._h = None
        self._cloud = None
        self._dev = None

    # -- configuration (handle is created lazily, re-created when the configuration changes) --
    def _set(self, **kw):
        for k, v in kw.items():
            setattr(self._cfg, k, v)
        self.close()

    def setPassThrough(self, field="z", lo=0.0, hi=10.0, negative=False, enable=True):
        self._set(pass_enable=int(enable), pass_field="xyz".index(field), pass_min=lo, pass_max=hi,
                  pass_negative=int(negative))

    def setVoxelMode(self, mode):
        self._set(voxel_mode=mode)

    def setLeafSize(self, lx, ly=None, lz=None):
        ly = lx if ly is None else ly
        lz = lx if lz is None else lz
        self._set(leaf_size=(C.c_float * 3)(lx, ly, lz))

    def setHistorySize(self, n):
        self._set(approx_hist_size=int(n))

    def _check(self, status):
        if status != 0:
            detail = self._L.pft_filter_last_error_string(self._h).decode() if self._h else ""
            raise PftError(status, detail)

    def _ensure(self):
        if self._h is None:
            h = C.c_void_p()
            st = self._L.pft_filter_create(C.byref(self._cfg), C.byref(h))
            if st != 0:
                raise PftError(st, "pft_filter_create")
            self._h = h

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._L.pft_filter_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- PCL-style use --
    def setInputCloud(self, cloud):
        cloud = np.ascontiguousarray(cloud)
        assert cloud.dtype == POINT_DTYPE
        self._cloud, self._dev = cloud, None

    def setInputCloudDevice(self, device_ptr, n, keepalive=None):
        self._cloud, self._dev = None, (int(device_ptr), int(n), keepalive)

    def _apply(self):
        self._ensure()
        if self._dev is not None:
            self._check(self._L.pft_filter_apply_device(self._h, C.c_void_p(self._dev[0]), self._dev[1]))
        elif self._cloud is not None:
            self._check(self._L.pft_filter_apply(self._h, _ptr(self._cloud), len(self._cloud)))
        else:
            raise PftError(2, "filter() without an input cloud")

    def counts(self):
        a, b = C.c_size_t(), C.c_size_t()
        self._check(self._L.pft_filter_counts(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def filter(self):
        """run the pipeline, return the output cloud on the host"""
        self._apply()
        _, n_out = self.counts()
        out = np.zeros(n_out, POINT_DTYPE)
        n = C.c_size_t()
        self._check(self._L.pft_filter_get_output(self._h, _ptr(out), n_out, C.byref(n)))
        return out

    def filterDevice(self):
        """run the pipeline, return (device pointer, n) of the output cloud in HBM (valid until the next call)"""
        self._apply()
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.pft_filter_output_device(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def passIndices(self):
        """indices of the points PassThrough kept (after filter())"""
        n_pass, _ = self.counts()
        idx = np.zeros(n_pass, np.int32)
        n = C.c_size_t()
        self._check(self._L.pft_filter_get_pass_indices(self._h, _ptr(idx), n_pass, C.byref(n)))
        return idx

    def lastMilliseconds(self):
        ms = C.c_double()
        self._check(self._L.pft_filter_last_ms(self._h, C.byref(ms)))
        return ms.value


class PassThrough(InputFilter):
    """pcl::PassThrough<PointXYZRGBA> (auto_tracking.cpp:536-547)"""

    def __init__(self, **kw):
        super().__init__(**kw)
        self._set(voxel_mode=VOXEL_NONE, pass_enable=1)
        self._lim = (0.0, 10.0)
        self._field = "z"
        self._neg = False

    def setFilterFieldName(self, name):
        self._field = name
        self.setPassThrough(self._field, self._lim[0], self._lim[1], self._neg)

    def setFilterLimits(self, lo, hi):
        self._lim = (float(lo), float(hi))
        self.setPassThrough(self._field, self._lim[0], self._lim[1], self._neg)

    def setFilterLimitsNegative(self, neg):
        self._neg = bool(neg)
        self.setPassThrough(self._field, self._lim[0], self._lim[1], self._neg)

    def setKeepOrganized(self, keep):
        if keep:
            raise NotImplementedError("keep_organized = true is not on the reference's path (auto_tracking.cpp:543)")


class ApproximateVoxelGrid(InputFilter):
    """pcl::ApproximateVoxelGrid<PointXYZRGBA> (auto_tracking.cpp:563-575)"""

    def __init__(self, **kw):
        super().__init__(**kw)
        self._set(voxel_mode=VOXEL_APPROX, pass_enable=0)


class VoxelGrid(InputFilter):
    """pcl::VoxelGrid<PointXYZRGBA> (auto_tracking.cpp:549-561)"""

    def __init__(self, **kw):
        super().__init__(**kw)
        self._set(voxel_mode=VOXEL_EXACT, pass_enable=0)


def make_reference_input_filter(**kw):
    """PassThrough z in [0, 10] + ApproximateVoxelGrid(0.01): the reference's per-frame front end
    (auto_tracking.cpp:637, 683), fused"""
    return InputFilter(**kw)
