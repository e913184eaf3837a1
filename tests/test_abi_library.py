"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/*.h declare (no compute call is made here: there is no GPU in the build container)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from pcl_tracking_amd import build

    build.build()  # hipcc cross-compiles gfx950 without a GPU
    from pcl_tracking_amd import _lib

    return _lib


def declared_functions():
    names = set()
    inc = os.path.join(ROOT, "include")
    for h in sorted(os.listdir(inc)):
        if not h.endswith(".h"):
            continue
        src = open(os.path.join(inc, h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"#ifdef PFT_DIAG.*?#endif", "", src, flags=re.S)  # diagnostic variant library only
        names |= set(re.findall(r"\b(pft_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_every_declared_symbol_is_exported_and_bound(lib):
    L = lib.load()
    names = declared_functions()
    assert len(names) >= 35
    bound = {n for n, _, _ in lib.SYMBOLS}
    for n in names:
        assert hasattr(L, n), "library does not export %s" % n
        assert n in bound, "ctypes binding missing for %s" % n
    assert bound <= set(names)


def test_pod_layouts_match_pcl():
    from pcl_tracking_amd import scene

    assert scene.POINT_DTYPE.itemsize == 32 and scene.POINT_DTYPE.fields["rgba"][1] == 16
    assert scene.PARTICLE_DTYPE.itemsize == 32 and scene.PARTICLE_DTYPE.fields["weight"][1] == 28


def test_config_defaults_are_the_reference_values(lib):
    L = lib.load()
    c = lib.Config()
    L.pft_config_default(C.byref(c))
    # /root/reference/src/auto_tracking.cpp:187-253
    assert c.particle_num == 400 and c.iteration_num == 2
    assert list(c.step_noise_cov) == [0.015 * 0.015] * 3 + [0.015 * 0.015 * 40.0] * 3
    assert list(c.initial_noise_cov) == [0.00001] * 6 and list(c.initial_noise_mean) == [0.0] * 6
    assert c.max_distance == 0.1 and c.octree_resolution == 0.01 and c.hsv_weight == 0.1
    assert (c.alpha, c.distance_weight, c.h_weight, c.s_weight, c.v_weight) == (15.0, 1.0, 1.0, 1.0, 0.0)
    assert c.abi_version == lib.PFT_ABI_VERSION
    # the KLD-adaptive branch, :207-219 (off by default: the north-star path is the fixed tracker)
    assert c.kld_adaptive == 0 and c.maximum_particle_num == 500 and (c.kld_delta, c.kld_epsilon) == (0.99, 0.2)
    assert list(c.kld_bin_size) == [0.1] * 6 and c.motion_ratio == 0.25


def test_host_helpers_match_oracle(lib, orc):
    """pft_to_matrix / pft_to_state are host-side float helpers (toEigenMatrix / toState)"""
    import numpy as np

    from pcl_tracking_amd import scene

    L = lib.load()
    rng = np.random.default_rng(0)
    for _ in range(20):
        p = np.zeros(1, scene.PARTICLE_DTYPE)
        vals = rng.uniform(-1.5, 1.5, 6).astype(np.float32)
        for k, v in zip(("x", "y", "z", "roll", "pitch", "yaw"), vals):
            p[k] = v
        m = np.zeros(16, np.float32)
        L.pft_to_matrix(p.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p))
        np.testing.assert_array_equal(m.reshape(4, 4), orc.get_transformation(*vals))
        back = np.zeros(1, scene.PARTICLE_DTYPE)
        L.pft_to_state(m.ctypes.data_as(C.c_void_p), back.ctypes.data_as(C.c_void_p))
        assert back[0].tobytes() == orc.to_state(m).tobytes()


def test_filter_config_defaults_are_the_reference_values(lib):
    L = lib.load()
    c = lib.FilterConfig()
    L.pft_filter_default_config(C.byref(c))
    # /root/reference/src/auto_tracking.cpp:536-547 (z in [0, 10]) and :563-575 (ApproximateVoxelGrid 0.01)
    assert (c.pass_enable, c.pass_field, c.pass_min, c.pass_max, c.pass_negative) == (1, 2, 0.0, 10.0, 0)
    assert c.voxel_mode == lib.VOXEL_APPROX and [round(v, 6) for v in c.leaf_size] == [0.01] * 3
    assert c.approx_hist_size == 512 and c.max_points == 960 * 540


def test_no_gpu_means_loud_failure(lib):
    """without a usable device pft_create refuses; there is no CPU fallback to fall into"""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = lib.load()
    c = lib.Config()
    L.pft_config_default(C.byref(c))
    h = C.c_void_p()
    assert L.pft_create(C.byref(c), C.byref(h)) == 4  # PFT_ERR_NO_DEVICE
    assert not h.value
    fc = lib.FilterConfig()
    L.pft_filter_default_config(C.byref(fc))
    assert L.pft_filter_create(C.byref(fc), C.byref(h)) == 4 and not h.value
    from pcl_tracking_amd import scene, tracker
    from pcl_tracking_amd._lib import PftError

    t = tracker.make_reference_tracker(particle_num=16)
    t.setReferenceCloud(scene.make_model(64))
    with pytest.raises(PftError):
        t.setInputCloud(scene.make_model(64))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pcl_tracking_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pft_oracle" not in txt and "import oracle" not in txt, os.path.join(dirpath, f)


def test_product_library_has_no_wrong_result_switches(lib):
    """the stage-ablation mask and the octree-reuse switch are compiled only into the diagnostic
    variant (-DPFT_DIAG); libpft_hip.so neither reads their environment variables nor exports the setter."""
    blob = open(lib.LIB_PATH, "rb").read()
    for needle in (b"PFT_ABLATE", b"SKIP_OCTREE", b"pft_debug_set_ablate"):
        assert blob.count(needle) == 0, needle
    assert not hasattr(lib.load(), "pft_debug_set_ablate")
