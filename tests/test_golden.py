"""Committed fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the oracle):
CPU: the oracle still reproduces them bit for bit; GPU: the HIP path reproduces them through the C ABI.
The fixtures pin the oracle restatement, not PCL (parity unpinned: tests/golden/README.md)."""
import os

import numpy as np
import pytest

from pcl_tracking_amd import scene

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = ("x", "y", "z", "roll", "pitch", "yaw")


def load(name):
    z = np.load(os.path.join(G, name + ".npz"))
    d = {k: z[k] for k in z.files}
    d["model"] = d["model"].view(scene.POINT_DTYPE)
    d["cloud"] = d["cloud"].view(scene.POINT_DTYPE)
    if "particles" in d:
        d["particles"] = np.ascontiguousarray(d["particles"]).view(scene.PARTICLE_DTYPE).reshape(-1)
    return d


@pytest.mark.parametrize("name", ["eval_small", "eval_ragged"])
def test_oracle_reproduces_eval_fixture(orc, name):
    d = load(name)
    t = orc.Tracker(orc.default_config(particle_num=len(d["particles"]), threads=1, emulate_pcl_alloc=0))
    t.set_reference(d["model"])
    t.set_trans(scene.initial_trans())
    t.set_input(d["cloud"])
    ev = t.eval_weights(d["particles"], want_nn=True, mats=d["mats"])
    for k in ("raw", "nn_idx", "nn_d2", "crop_idx", "bbox", "octree_min", "octree_max"):
        np.testing.assert_array_equal(ev[k], d[k], err_msg=k)
    assert ev["octree_depth"] == int(d["octree_depth"])
    w, fit = orc.normalize_weights(d["raw"])
    np.testing.assert_array_equal(w, d["weights"])
    a, q = orc.gen_alias_table(d["weights"])
    np.testing.assert_array_equal(a, d["alias_a"])
    np.testing.assert_array_equal(q, d["alias_q"])


def test_oracle_reproduces_track_fixture(orc):
    d = load("track_small")
    t = orc.Tracker(orc.default_config(particle_num=int(d["P"]), seed=int(d["seed"]), threads=1, emulate_pcl_alloc=0))
    t.set_reference(d["model"])
    t.set_trans(scene.initial_trans())
    t.set_input(d["cloud"])
    for f in range(len(d["results"])):
        assert t.compute() == 0
        np.testing.assert_array_equal(np.frombuffer(t.get_result().tobytes(), np.float32), d["results"][f])
    np.testing.assert_array_equal(t.get_particles().view(np.float32).reshape(-1, 8),
                                  d["particles"].view(np.float32).reshape(-1, 8))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["eval_small", "eval_ragged"])
def test_gpu_matches_eval_fixture(name):
    from pcl_tracking_amd import tracker

    d = load(name)
    P = len(d["particles"])
    g = tracker.make_reference_tracker(particle_num=P)
    g.setReferenceCloud(d["model"])
    g.setTrans(scene.initial_trans())
    g.setInputCloud(d["cloud"])
    ev = g.evalWeights(d["particles"], want_nn=True)
    # the GPU forms its matrices with double sin/cos rounded to float; the fixture's came from cosf/sinf
    mats = g.debugPoseToMatrix(d["particles"])
    same = np.all(mats.reshape(P, -1) == d["mats"].reshape(P, -1), axis=1)
    assert same.mean() > 0.5
    np.testing.assert_array_equal(ev["crop_idx"], d["crop_idx"])
    assert ev["octree_depth"] == int(d["octree_depth"])
    np.testing.assert_array_equal(ev["octree_min"], d["octree_min"])
    np.testing.assert_array_equal(ev["nn_idx"][same], d["nn_idx"][same])  # bit-exact where the matrix is
    np.testing.assert_array_equal(ev["nn_d2"][same], d["nn_d2"][same])
    assert (ev["nn_idx"] == d["nn_idx"]).mean() > 0.999
    np.testing.assert_allclose(ev["raw"], d["raw"], atol=1e-3, rtol=0)
    w, fit = g.debugNormalize(d["raw"])
    assert np.abs(w.view(np.int32) - d["weights"].view(np.int32)).max() <= 1
    a, q = g.debugAlias(d["weights"])
    np.testing.assert_array_equal(a, d["alias_a"])
    np.testing.assert_allclose(q, d["alias_q"], atol=1e-9)


@pytest.mark.gpu
def test_gpu_matches_track_fixture():
    from pcl_tracking_amd import tracker

    d = load("track_small")
    g = tracker.make_reference_tracker(particle_num=int(d["P"]), seed=int(d["seed"]))
    g.setReferenceCloud(d["model"])
    g.setTrans(scene.initial_trans())
    g.setInputCloud(d["cloud"])
    for f in range(len(d["results"])):
        g.compute()
        r = g.getResult()
        want = d["results"][f]
        for i, k in enumerate(KEYS):
            j = i if i < 3 else i + 1  # x,y,z,w,roll,pitch,yaw,weight
            assert abs(float(r[k]) - float(want[j])) < 1e-4, (f, k)


# ---- input front end: tests/golden/filters_small.npz ----
def load_filters():
    z = np.load(os.path.join(G, "filters_small.npz"))
    d = {k: z[k] for k in z.files}
    for k in ("frame", "approx512", "approx64", "exact"):
        c = np.zeros(len(d[k]), scene.POINT_DTYPE)
        c["x"], c["y"], c["z"] = (d[k][:, j].copy().view(np.float32) for j in range(3))
        c["rgba"] = d[k][:, 3]
        c["w"] = 1.0
        d[k] = c
    d["leaf"] = float(d["leaf"])
    return d


def test_oracle_reproduces_filter_fixture(orc):
    d = load_filters()
    idx = orc.pass_through(d["frame"], "z", 0.0, 10.0)
    np.testing.assert_array_equal(idx, d["pass_idx"])
    kept = d["frame"][idx]
    assert orc.approx_voxel_grid(kept, d["leaf"], 512).tobytes() == d["approx512"].tobytes()
    assert orc.approx_voxel_grid(kept, d["leaf"], 64).tobytes() == d["approx64"].tobytes()
    assert orc.voxel_grid(kept, d["leaf"]).tobytes() == d["exact"].tobytes()


@pytest.mark.gpu
def test_hip_reproduces_filter_fixture():
    from pcl_tracking_amd import filters

    d = load_filters()
    f = filters.make_reference_input_filter()
    f.setLeafSize(d["leaf"])
    f.setInputCloud(d["frame"])
    assert f.filter().tobytes() == d["approx512"].tobytes()
    np.testing.assert_array_equal(f.passIndices(), d["pass_idx"])
    f.setHistorySize(64)
    f.setInputCloud(d["frame"])
    assert f.filter().tobytes() == d["approx64"].tobytes()
    g = filters.VoxelGrid()
    g.setLeafSize(d["leaf"])
    g.setInputCloud(d["frame"][d["pass_idx"]])
    assert g.filter().tobytes() == d["exact"].tobytes()


# ---- KLD-adaptive resample: tests/golden/kld_small.npz ----
def load_kld():
    z = np.load(os.path.join(G, "kld_small.npz"))
    d = {k: z[k] for k in z.files}
    for k in ("old", "motion", "particles_0", "particles_5"):
        d[k] = np.ascontiguousarray(d[k]).view(scene.PARTICLE_DTYPE).reshape(-1)
    return d


def test_oracle_reproduces_kld_fixture(orc):
    d = load_kld()
    cfg = orc.default_config(kld_adaptive=1, seed=int(d["seed"]))
    for epoch in (0, 5):
        p, bins, k = orc.kld_resample(cfg, d["old"], d["alias_a"], d["alias_q"], d["motion"], epoch)
        assert p.tobytes() == d["particles_%d" % epoch].tobytes()
        np.testing.assert_array_equal(bins, d["bins_%d" % epoch])
        assert k == int(d["k_%d" % epoch])


@pytest.mark.gpu
def test_hip_reproduces_kld_fixture():
    from pcl_tracking_amd import tracker

    d = load_kld()
    g = tracker.make_reference_tracker(particle_num=len(d["old"]), seed=int(d["seed"]), kld=True)
    for epoch in (0, 5):
        p, bins, k = g.debugKldResample(d["old"], d["alias_a"], d["alias_q"], d["motion"], epoch)
        want = d["particles_%d" % epoch]
        assert len(p) == len(want) and k == int(d["k_%d" % epoch])
        np.testing.assert_array_equal(bins, d["bins_%d" % epoch])
        for c in KEYS:  # Box-Muller's log / sin / cos in double: ocml against glibc
            assert np.abs(p[c] - want[c]).max() <= 1e-6
