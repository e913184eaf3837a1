"""The C++ host side (pcl_tracking_amd/include/pft/particle_filter_tracker.hpp + examples/auto_tracking_amd.cpp):
the reference is C++, so the mirror of the PCL classes it drives is C++ too.  CPU: it compiles and links
against the C-ABI library.  GPU: the ROS-free driver, fed a model cluster and frames as PCL-layout binary
files, reports the same poses as the Python binding (both are thin layers over the same C ABI)."""
import os
import subprocess

import numpy as np
import pytest

from pcl_tracking_amd import scene


def test_cpp_mirror_compiles_and_links():
    from pcl_tracking_amd import build

    exe = build.build_example()
    assert os.path.exists(exe) and os.access(exe, os.X_OK)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def test_cpp_mirror_uses_the_reference_call_names():
    """every tracker/coherence member the reference calls exists in the mirror"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "pcl_tracking_amd", "include", "pft", "particle_filter_tracker.hpp")).read()
    for name in ("setTrans", "setStepNoiseCovariance", "setInitialNoiseCovariance", "setInitialNoiseMean",
                 "setIterationNum", "setParticleNum", "setResampleLikelihoodThr", "setUseNormal", "setCloudCoherence",
                 "getParticles", "getResult", "toEigenMatrix", "setReferenceCloud", "setMinIndices", "setInputCloud",
                 "compute", "addPointCoherence", "setWeight", "setSearchMethod", "setMaximumDistance",
                 "ParticleFilterOMPTracker", "ApproxNearestPairPointCloudCoherence", "DistanceCoherence",
                 "HSVColorCoherence", "KLDAdaptiveParticleFilterOMPTracker", "setMaximumParticleNum", "setDelta",
                 "setEpsilon", "setBinSize", "NearestPairPointCloudCoherence"):
        assert name in hdr, name
    # the filter classes of cloud_cb's front end (auto_tracking.cpp:536-575)
    fh = open(os.path.join(root, "pcl_tracking_amd", "include", "pft", "filters.hpp")).read()
    for name in ("PassThrough", "ApproximateVoxelGrid", "VoxelGrid", "setFilterFieldName", "setFilterLimits",
                 "setKeepOrganized", "setLeafSize", "setInputCloud", "filter"):
        assert name in fh, name


@pytest.mark.gpu
def test_cpp_driver_matches_python_binding(tmp_path):
    from pcl_tracking_amd import build, tracker

    exe = build.build_example()
    model = scene.make_model(512)
    # the segmented cluster as create_model.cpp would hand it over: in the camera frame
    cluster = model.copy()
    off = np.array(scene.model_gt_pose()[:3], np.float32)
    for k, name in enumerate(("x", "y", "z")):
        cluster[name] = cluster[name] + off[k]
    frames = [scene.make_scene(50000, obj_pose=scene.advance_pose(scene.GT_POSE, 3 * f))[:15000] for f in range(3)]
    cluster.tofile(tmp_path / "model.bin")
    paths = []
    for i, fr in enumerate(frames):
        fr.tofile(tmp_path / ("frame%d.bin" % i))
        paths.append(str(tmp_path / ("frame%d.bin" % i)))
    r = subprocess.run([exe, str(tmp_path / "model.bin")] + paths + ["--particles", "1000", "--seed", "6", "--model-leaf", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = [list(map(float, line.split("pose")[1].split("t =")[0].split())) for line in r.stdout.splitlines()
           if line.startswith("frame")]
    assert len(got) == 3
    # the same steps in Python: float centroid (sequential float sums as the driver does), re-centre, track
    s = np.zeros(3, np.float32)
    for p in cluster:
        s[0] += p["x"]
        s[1] += p["y"]
        s[2] += p["z"]
    c = s / np.float32(len(cluster))
    ref = cluster.copy()
    for k, name in enumerate(("x", "y", "z")):
        ref[name] = ref[name] - c[k]
    trans = np.eye(4, dtype=np.float32)
    trans[:3, 3] = c
    t = tracker.make_reference_tracker(particle_num=1000, seed=6)
    t.setReferenceCloud(ref)
    t.setTrans(trans)
    for f in range(3):
        t.setInputCloud(frames[f])
        t.compute()
        res = t.getResult()
        want = [float(res[k]) for k in ("x", "y", "z", "roll", "pitch", "yaw")]
        np.testing.assert_allclose(got[f], want, atol=2e-6)


@pytest.mark.gpu
def test_cpp_driver_raw_frames_go_through_the_device_front_end(tmp_path):
    """--raw: sensor frame -> PassThrough + ApproximateVoxelGrid on the device -> tracker, all in HBM; the poses
    equal those of the driver fed the frame the Python binding filtered"""
    from pcl_tracking_amd import build, filters

    exe = build.build_example()
    model = scene.make_model(512)
    cluster = model.copy()
    off = np.array(scene.model_gt_pose()[:3], np.float32)
    for k, name in enumerate(("x", "y", "z")):
        cluster[name] = cluster[name] + off[k]
    raw = scene.make_depth_frame(480, 270)
    f = filters.make_reference_input_filter()
    f.setInputCloud(raw)
    down = f.filter()
    cluster.tofile(tmp_path / "model.bin")
    raw.tofile(tmp_path / "raw.bin")
    down.tofile(tmp_path / "down.bin")
    outs, errs = [], []
    for args in ([str(tmp_path / "raw.bin"), "--raw"], [str(tmp_path / "down.bin")]):
        r = subprocess.run([exe, str(tmp_path / "model.bin")] + args + ["--particles", "600", "--seed", "2", "--model-leaf", "0"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append([line for line in r.stdout.splitlines() if line.startswith("frame")])
        errs.append(r.stderr)
    assert len(outs[0]) == 1 and outs[0] == outs[1]
    assert "after downsampled: %d data points" % len(down) in errs[0]


@pytest.mark.gpu
def test_cpp_driver_kld_branch_matches_python_binding(tmp_path):
    """--kld: the use_fixed == false branch of initialize_trackers() (auto_tracking.cpp:207-222)"""
    from pcl_tracking_amd import build, tracker

    exe = build.build_example()
    model = scene.make_model(512)
    off = np.array(scene.model_gt_pose()[:3], np.float32)
    for k, name in enumerate(("x", "y", "z")):  # the cluster in the camera frame, as create_model.cpp hands it over
        model[name] = model[name] + off[k]
    frame = scene.make_scene(50000)[:20000]
    model.tofile(tmp_path / "model.bin")
    frame.tofile(tmp_path / "frame.bin")
    r = subprocess.run([exe, str(tmp_path / "model.bin"), str(tmp_path / "frame.bin"), str(tmp_path / "frame.bin"),
                        "--kld", "--seed", "8", "--model-leaf", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = [list(map(float, line.split("pose")[1].split("t =")[0].split())) for line in r.stdout.splitlines()
           if line.startswith("frame")]
    assert len(got) == 2
    s = np.zeros(3, np.float32)
    for p in model:
        s[0] += p["x"]
        s[1] += p["y"]
        s[2] += p["z"]
    c = s / np.float32(len(model))
    ref = model.copy()
    for k, name in enumerate(("x", "y", "z")):
        ref[name] = ref[name] - c[k]
    trans = np.eye(4, dtype=np.float32)
    trans[:3, 3] = c
    t = tracker.make_reference_tracker(particle_num=400, seed=8, kld=True)
    t.setReferenceCloud(ref)
    t.setTrans(trans)
    for f in range(2):
        t.setInputCloud(frame)
        t.compute()
        res = t.getResult()
        np.testing.assert_allclose(got[f], [float(res[k]) for k in ("x", "y", "z", "roll", "pitch", "yaw")], atol=2e-6)


def write_pcd(path, cloud, binary):
    """PCD v0.7 with FIELDS x y z rgba, as pcl::PCDWriter::write<PointXYZRGBA> lays it out"""
    n = len(cloud)
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgba\nSIZE 4 4 4 4\nTYPE F F F U\n"
           "COUNT 1 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n" % (n, n, "binary" if binary else "ascii"))
    with open(path, "wb") as f:
        f.write(hdr.encode())
        if binary:
            rec = np.zeros(n, np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgba", "<u4")]))
            for k in ("x", "y", "z", "rgba"):
                rec[k] = cloud[k]
            f.write(rec.tobytes())
        else:
            for p in cloud:
                f.write(("%.9g %.9g %.9g %d\n" % (p["x"], p["y"], p["z"], p["rgba"])).encode())


@pytest.mark.gpu
def test_cpp_driver_pcd_models_gridsample_and_centroid(tmp_path):
    """PCD model in (ascii and binary), removeZeroPoints -> centroid -> re-centre -> gridSample
    (VoxelGrid on the device, :672) -> tracking -> drawResult's centroid of the moved full-resolution model"""
    from pcl_tracking_amd import build, filters, tracker

    exe = build.build_example()
    model = scene.make_model(4000)
    off = np.array(scene.model_gt_pose()[:3], np.float32)
    for k, name in enumerate(("x", "y", "z")):
        model[name] = model[name] + off[k]
    frame = scene.make_scene(50000)[:20000]
    write_pcd(tmp_path / "m_ascii.pcd", model, False)
    write_pcd(tmp_path / "m_bin.pcd", model, True)
    model.tofile(tmp_path / "m.bin")
    frame.tofile(tmp_path / "frame.bin")
    outs = []
    for m in ("m_ascii.pcd", "m_bin.pcd", "m.bin"):
        r = subprocess.run([exe, str(tmp_path / m), str(tmp_path / "frame.bin"), "--seed", "5"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append([line for line in r.stdout.splitlines() if line.startswith("frame")])
        assert "downsampled:" in r.stderr
    assert len(outs[0]) == 1 and outs[0] == outs[1] == outs[2]  # %.9g round-trips float32
    # the same in Python: float centroid, re-centre, VoxelGrid(0.01) on the device, track, centroid of the moved model
    s = np.zeros(3, np.float32)
    for p in model:
        s[0] += p["x"]
        s[1] += p["y"]
        s[2] += p["z"]
    c = s / np.float32(len(model))
    ref = model.copy()
    for k, name in enumerate(("x", "y", "z")):
        ref[name] = ref[name] - c[k]
    g = filters.VoxelGrid()
    g.setLeafSize(0.01)
    g.setInputCloud(ref)
    down = g.filter()
    assert len(down) < len(ref)
    trans = np.eye(4, dtype=np.float32)
    trans[:3, 3] = c
    t = tracker.make_reference_tracker(particle_num=400, seed=5)
    t.setReferenceCloud(down)
    t.setTrans(trans)
    t.setInputCloud(frame)
    t.compute()
    res = t.getResult()
    line = outs[0][0]
    got = list(map(float, line.split("pose")[1].split("t =")[0].split()))
    np.testing.assert_allclose(got, [float(res[k]) for k in ("x", "y", "z", "roll", "pitch", "yaw")], atol=2e-6)
    T = t.toEigenMatrix(res)
    xyz = np.stack([ref["x"], ref["y"], ref["z"]], 1).astype(np.float64)
    want_c = (xyz @ np.asarray(T, np.float64)[:3, :3].T + np.asarray(T, np.float64)[:3, 3]).mean(0)
    got_c = list(map(float, line.split("centroid = [")[1].rstrip("]").split()))
    np.testing.assert_allclose(got_c, want_c, atol=1e-4)
