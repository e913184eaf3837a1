"""The C++ host side: pcl_tracking_amd/include/pft/*.hpp (mirror of the PCL classes the reference drives, plus the two
pcl::common functions and the PCD reader its app-level steps need) and the two drivers built on it,
examples/auto_tracking_amd.cpp (any number of objects, /root/reference/src/auto_tracking.cpp:199-257, :688-697) and
examples/dist_tracking_amd.cpp (one object sharded over the GPUs, pft_dist_* phases + RCCL collectives).

CPU: they compile and link; the PCD reader (ascii / binary / binary_compressed) against known clouds.
GPU: the drivers' output against the ORACLE -- model preparation (removeZeroPoints, centroid, re-centre, gridSample),
tracking (oracle tracker in its test-only device-arithmetic modes: bit-equal poses) and the result consumer (centroid of
the moved full-resolution model, bit-equal).  PARITY UNPINNED: the oracle restates PCL 1.8.0 (oracle/pft_oracle.h).
"""
import os
import struct
import subprocess

import numpy as np
import pytest

from pcl_tracking_amd import scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("x", "y", "z", "roll", "pitch", "yaw")


# ---- helpers --------------------------------------------------------------------------------------------------------
def cluster_of(model, with_junk=False, seed=0):
    """the segmented object cluster as create_model.cpp hands it over: in the camera frame; optionally with the invalid
    points a real cluster carries (NaN depth, points at the sensor origin) for removeZeroPoints to drop"""
    c = model.copy()
    off = np.array(scene.model_gt_pose()[:3], np.float32)
    for k, name in enumerate(("x", "y", "z")):
        c[name] = c[name] + off[k]
    if with_junk:
        rng = np.random.default_rng(seed)
        junk = np.zeros(40, scene.POINT_DTYPE)
        junk["w"] = 1.0
        junk["x"][:20] = rng.uniform(-0.009, 0.009, 20)  # within 1 cm of the origin on all axes
        junk["y"][:20] = rng.uniform(-0.009, 0.009, 20)
        junk["z"][:20] = rng.uniform(-0.009, 0.009, 20)
        junk["x"][20:30] = np.nan
        junk["x"][30:] = 0.5
        junk["z"][30:] = np.nan
        c = np.concatenate([c, junk])
        c = c[rng.permutation(len(c))]
    return c


def parse(stdout):
    """'frame F object K pose x y z roll pitch yaw  centroid cx cy cz' -> {(F, K): (pose float32[6], centroid float32[3])}"""
    out = {}
    for line in stdout.splitlines():
        if not line.startswith("frame"):
            continue
        t = line.split()
        f, k = int(t[1]), int(t[3])
        pose = np.array(list(map(float, t[5:11])), np.float32)
        cen = np.array(list(map(float, t[12:15])), np.float32)
        out[(f, k)] = (pose, cen)
    return out


def oracle_prepare(orc, cluster, leaf):
    """the 'set object to track' step (:646-677) by the oracle: -> (tracked model, trans, full-resolution re-centred model)"""
    nz = orc.remove_zero_points(cluster)
    c, n = orc.compute_3d_centroid(nz)
    assert n == len(nz)
    ref_full, trans = orc.recentre_model(nz, c)
    ref = orc.voxel_grid(ref_full, leaf) if leaf > 0 else ref_full
    return ref, trans, ref_full


def oracle_tracker(orc, ref, trans, particles, seed, kld=False):
    o = orc.Tracker(orc.default_config(particle_num=particles, seed=seed, threads=0, emulate_pcl_alloc=0,
                                       kld_adaptive=1 if kld else 0))
    o.set_trig_mode(1)  # the device's arithmetic, so that the comparison is bit for bit (tests/test_gpu_longrun.py)
    o.set_sum_mode(1)
    o.set_reference(ref)
    o.set_trans(trans)
    return o


def pose_array(r):
    return np.array([r[k] for k in KEYS], np.float32)


def as_particle(pose):
    p = np.zeros(1, scene.PARTICLE_DTYPE)
    for k, name in enumerate(KEYS):
        p[name] = pose[k]
    p["w"] = 1.0
    return p


def write_pcd(path, cloud, mode):
    """PCD v0.7 with FIELDS x y z rgba, as pcl::PCDWriter lays it out; mode = ascii | binary | binary_compressed"""
    n = len(cloud)
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgba\nSIZE 4 4 4 4\nTYPE F F F U\n"
           "COUNT 1 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n" % (n, n, mode))
    with open(path, "wb") as f:
        f.write(hdr.encode())
        if mode == "binary":
            rec = np.zeros(n, np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgba", "<u4")]))
            for k in ("x", "y", "z", "rgba"):
                rec[k] = cloud[k]
            f.write(rec.tobytes())
        elif mode == "binary_compressed":  # the fields one after the other, LZF-compressed (writeBinaryCompressed)
            raw = b"".join(np.ascontiguousarray(cloud[k]).astype("<f4" if k != "rgba" else "<u4").tobytes()
                           for k in ("x", "y", "z", "rgba"))
            comp = lzf_compress(raw)
            f.write(struct.pack("<II", len(comp), len(raw)))
            f.write(comp)
        else:
            for p in cloud:
                f.write(("%.9g %.9g %.9g %d\n" % (p["x"], p["y"], p["z"], p["rgba"])).encode())


def lzf_compress(data):
    """a greedy LZF encoder (liblzf stream format), enough to produce literal runs, short and long matches"""
    out, lit, table, i, n = bytearray(), bytearray(), {}, 0, len(data)

    def flush():
        for j in range(0, len(lit), 32):
            run = lit[j:j + 32]
            out.append(len(run) - 1)
            out.extend(run)
        lit.clear()

    while i < n:
        if i + 2 < n:
            key = bytes(data[i:i + 3])
            ref = table.get(key)
            table[key] = i
            if ref is not None and i - ref <= 8192:
                ln = 3
                while i + ln < n and ln < 264 and data[ref + ln] == data[i + ln]:
                    ln += 1
                flush()
                dist, code = i - ref - 1, ln - 2
                if code < 7:
                    out.append((code << 5) | (dist >> 8))
                else:
                    out.append((7 << 5) | (dist >> 8))
                    out.append(code - 7)
                out.append(dist & 0xFF)
                i += ln
                continue
        lit.append(data[i])
        i += 1
    flush()
    return bytes(out)


def lzf_decompress(data, n_out):
    out, ip = bytearray(), 0
    while ip < len(data):
        ctrl = data[ip]
        ip += 1
        if ctrl < 32:
            out.extend(data[ip:ip + ctrl + 1])
            ip += ctrl + 1
        else:
            ln = ctrl >> 5
            if ln == 7:
                ln += data[ip]
                ip += 1
            dist = ((ctrl & 0x1F) << 8) + data[ip] + 1
            ip += 1
            for _ in range(ln + 2):
                out.append(out[-dist])
    assert len(out) == n_out
    return bytes(out)


def run(exe, args, timeout=600, env=None):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return r


# ---- CPU ------------------------------------------------------------------------------------------------------------
def test_cpp_hosts_compile_and_link():
    from pcl_tracking_amd import build

    for exe in (build.build_example(), build.build_dist_example()):
        assert os.path.exists(exe) and os.access(exe, os.X_OK)
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "usage" in r.stderr


def test_cpp_mirror_uses_the_reference_call_names():
    """every tracker/coherence member the reference calls exists in the mirror"""
    hdr = open(os.path.join(ROOT, "pcl_tracking_amd", "include", "pft", "particle_filter_tracker.hpp")).read()
    for name in ("setTrans", "setStepNoiseCovariance", "setInitialNoiseCovariance", "setInitialNoiseMean",
                 "setIterationNum", "setParticleNum", "setResampleLikelihoodThr", "setUseNormal", "setCloudCoherence",
                 "getParticles", "getResult", "toEigenMatrix", "setReferenceCloud", "setMinIndices", "setInputCloud",
                 "compute", "addPointCoherence", "setWeight", "setSearchMethod", "setMaximumDistance",
                 "ParticleFilterOMPTracker", "ApproxNearestPairPointCloudCoherence", "DistanceCoherence",
                 "HSVColorCoherence", "KLDAdaptiveParticleFilterOMPTracker", "setMaximumParticleNum", "setDelta",
                 "setEpsilon", "setBinSize", "NearestPairPointCloudCoherence"):
        assert name in hdr, name
    # the filter classes of cloud_cb's front end (auto_tracking.cpp:536-575)
    fh = open(os.path.join(ROOT, "pcl_tracking_amd", "include", "pft", "filters.hpp")).read()
    for name in ("PassThrough", "ApproximateVoxelGrid", "VoxelGrid", "setFilterFieldName", "setFilterLimits",
                 "setKeepOrganized", "setLeafSize", "setInputCloud", "filter"):
        assert name in fh, name
    # the pcl::common functions of the model preparation / result consumer (auto_tracking.cpp:316, :433, :663, :668)
    ch = open(os.path.join(ROOT, "pcl_tracking_amd", "include", "pft", "common.hpp")).read()
    for name in ("compute3DCentroid", "transformPointCloud"):
        assert name in ch, name
    # the app keeps the reference's per-object dictionaries and its failure handling around compute()
    app = open(os.path.join(ROOT, "pcl_tracking_amd", "examples", "tracking_app.hpp")).read()
    drv = open(os.path.join(ROOT, "pcl_tracking_amd", "examples", "auto_tracking_amd.cpp")).read()
    for name in ("tracker_dict", "ref_cloud_dict", "reference_dict", "tracked_cloud_dict", "removeZeroPoints"):
        assert name in app, name
    assert "catch (int" in drv


def test_lzf_test_encoder_round_trips():
    rng = np.random.default_rng(1)
    for data in (b"", b"a", b"abcabcabcabcabcabcabcabc" * 40, bytes(rng.integers(0, 4, 5000, dtype=np.uint8)),
                 bytes(rng.integers(0, 256, 3000, dtype=np.uint8)), b"\x00" * 10000):
        assert lzf_decompress(lzf_compress(data), len(data)) == data


@pytest.fixture(scope="module")
def pcd_dump(tmp_path_factory):
    """a host-only program over pft/pcd_io.hpp: PCD in, raw 32-byte points out (no GPU, no library)"""
    d = tmp_path_factory.mktemp("pcd")
    src = d / "pcd_dump.cpp"
    src.write_text('#include <cstdio>\n#include "pft/pcd_io.hpp"\nint main(int c, char** v) {\n  pft::PointCloud<pft::PointXYZRGBA> cl;\n'
                   '  if (c < 3 || pft::io::loadPCDFile(v[1], cl) != 0) return 1;\n  FILE* f = std::fopen(v[2], "wb");\n'
                   '  std::fwrite(cl.points.data(), 32, cl.points.size(), f);\n  std::fclose(f);\n'
                   '  std::printf("%zu %u %u %d\\n", cl.points.size(), cl.width, cl.height, (int)cl.is_dense);\n  return 0;\n}\n')
    exe = d / "pcd_dump"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I",
                    os.path.join(ROOT, "pcl_tracking_amd", "include"), str(src), "-o", str(exe)], check=True)
    return str(exe)


@pytest.mark.parametrize("mode", ["ascii", "binary", "binary_compressed"])
def test_pcd_reader(tmp_path, pcd_dump, mode):
    cloud = scene.make_model(700, seed=3)
    cloud["x"][5] = np.nan  # an invalid point: is_dense must come out false, the bits must survive
    write_pcd(tmp_path / "c.pcd", cloud, mode)
    r = subprocess.run([pcd_dump, str(tmp_path / "c.pcd"), str(tmp_path / "c.bin")], capture_output=True, text=True)
    assert r.returncode == 0
    assert r.stdout.split() == [str(len(cloud)), str(len(cloud)), "1", "0"]
    got = np.fromfile(tmp_path / "c.bin", scene.POINT_DTYPE)
    for k in ("x", "y", "z"):
        np.testing.assert_array_equal(got[k].view(np.uint32), cloud[k].view(np.uint32))
    np.testing.assert_array_equal(got["rgba"], cloud["rgba"])


def test_pcd_reader_rejects_damaged_compressed_data(tmp_path, pcd_dump):
    cloud = scene.make_model(300, seed=4)
    write_pcd(tmp_path / "c.pcd", cloud, "binary_compressed")
    raw = bytearray(open(tmp_path / "c.pcd", "rb").read())
    open(tmp_path / "short.pcd", "wb").write(raw[:-40])
    assert subprocess.run([pcd_dump, str(tmp_path / "short.pcd"), str(tmp_path / "o.bin")]).returncode == 1


# ---- GPU ------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_cpp_driver_against_the_oracle(tmp_path, orc):
    """model cluster with invalid points -> removeZeroPoints -> centroid -> re-centre -> track 3 frames -> position:
    every pose bit-equal to the oracle tracker's, every published centroid bit-equal to the oracle's result consumer"""
    from pcl_tracking_amd import build

    exe = build.build_example()
    cluster = cluster_of(scene.make_model(512), with_junk=True, seed=5)
    frames = [scene.make_scene(50000, obj_pose=scene.advance_pose(scene.GT_POSE, 3 * f))[:15000] for f in range(3)]
    cluster.tofile(tmp_path / "model.bin")
    paths = []
    for i, fr in enumerate(frames):
        fr.tofile(tmp_path / ("frame%d.bin" % i))
        paths.append(tmp_path / ("frame%d.bin" % i))
    r = run(exe, [tmp_path / "model.bin"] + paths + ["--particles", 1000, "--seed", 6, "--model-leaf", 0])
    got = parse(r.stdout)
    assert sorted(got) == [(1, 0), (2, 0), (3, 0)]
    assert "nonzero_ref: 512" in r.stderr and "ref_cloud: %d" % len(cluster) in r.stderr
    ref, trans, ref_full = oracle_prepare(orc, cluster, 0.0)
    o = oracle_tracker(orc, ref, trans, 1000, 6)
    for f in range(3):
        o.set_input(frames[f])
        assert o.compute() == 0
        pose, cen = got[(f + 1, 0)]
        np.testing.assert_array_equal(pose.view(np.uint32), pose_array(o.get_result()).view(np.uint32), err_msg="frame %d" % f)
        _, want_c = orc.object_position(ref_full, as_particle(pose))
        np.testing.assert_array_equal(cen.view(np.uint32), want_c[:3].view(np.uint32))
        # the published position is the moved model's centroid, 5 mm towards the camera (:313): near the pose's translation
        assert abs(float(cen[2]) - (float(pose[2]) - 0.005)) < 0.02


@pytest.mark.gpu
def test_cpp_driver_tracks_several_objects(tmp_path, orc):
    """nb_objects trackers (auto_tracking.cpp:199-257), one loop over tracker_dict per frame (:688-697): each object's
    output is what the oracle gives for that object alone (object k is seeded seed + k)"""
    from pcl_tracking_amd import build

    exe = build.build_example()
    clusters = [cluster_of(scene.make_model(400 + 100 * k, seed=scene.MODEL_SEED + k), with_junk=(k == 1), seed=k) for k in range(3)]
    frames = [scene.make_scene(50000, obj_pose=scene.advance_pose(scene.GT_POSE, 2 * f))[:12000] for f in range(2)]
    margs = []
    for k, c in enumerate(clusters):
        c.tofile(tmp_path / ("m%d.bin" % k))
        margs.append(tmp_path / ("m%d.bin" % k))
    fargs = []
    for i, fr in enumerate(frames):
        fr.tofile(tmp_path / ("f%d.bin" % i))
        fargs.append(tmp_path / ("f%d.bin" % i))
    r = run(exe, margs + ["--frames"] + fargs + ["--particles", 600, "--seed", 40, "--model-leaf", 0])
    got = parse(r.stdout)
    assert sorted(got) == [(f, k) for f in (1, 2) for k in range(3)]
    for k, c in enumerate(clusters):
        ref, trans, ref_full = oracle_prepare(orc, c, 0.0)
        o = oracle_tracker(orc, ref, trans, 600, 40 + k)
        for f in range(2):
            o.set_input(frames[f])
            assert o.compute() == 0
            pose, cen = got[(f + 1, k)]
            np.testing.assert_array_equal(pose.view(np.uint32), pose_array(o.get_result()).view(np.uint32), err_msg="object %d frame %d" % (k, f))
            _, want_c = orc.object_position(ref_full, as_particle(pose))
            np.testing.assert_array_equal(cen.view(np.uint32), want_c[:3].view(np.uint32))


@pytest.mark.gpu
def test_cpp_driver_raw_frames_go_through_the_device_front_end(tmp_path):
    """--raw: sensor frame -> PassThrough + ApproximateVoxelGrid on the device -> tracker, all in HBM; the poses
    equal those of the driver fed the frame the Python binding filtered"""
    from pcl_tracking_amd import build, filters

    exe = build.build_example()
    cluster = cluster_of(scene.make_model(512))
    raw = scene.make_depth_frame(480, 270)
    f = filters.make_reference_input_filter()
    f.setInputCloud(raw)
    down = f.filter()
    cluster.tofile(tmp_path / "model.bin")
    raw.tofile(tmp_path / "raw.bin")
    down.tofile(tmp_path / "down.bin")
    outs, errs = [], []
    for args in ([tmp_path / "raw.bin", "--raw"], [tmp_path / "down.bin"]):
        r = run(exe, [tmp_path / "model.bin"] + args + ["--particles", 600, "--seed", 2, "--model-leaf", 0])
        outs.append([line for line in r.stdout.splitlines() if line.startswith("frame")])
        errs.append(r.stderr)
    assert len(outs[0]) == 1 and outs[0] == outs[1]
    assert "after downsampled: %d data points" % len(down) in errs[0]


@pytest.mark.gpu
def test_cpp_driver_kld_branch_against_the_oracle(tmp_path, orc):
    """--kld: the use_fixed == false branch of initialize_trackers() (auto_tracking.cpp:207-222)"""
    from pcl_tracking_amd import build

    exe = build.build_example()
    cluster = cluster_of(scene.make_model(512))
    frame = scene.make_scene(50000)[:20000]
    cluster.tofile(tmp_path / "model.bin")
    frame.tofile(tmp_path / "frame.bin")
    r = run(exe, [tmp_path / "model.bin", tmp_path / "frame.bin", tmp_path / "frame.bin", "--kld", "--seed", 8, "--model-leaf", 0])
    got = parse(r.stdout)
    assert sorted(got) == [(1, 0), (2, 0)]
    ref, trans, _ = oracle_prepare(orc, cluster, 0.0)
    o = oracle_tracker(orc, ref, trans, 400, 8, kld=True)
    for f in range(2):
        o.set_input(frame)
        assert o.compute() == 0
        np.testing.assert_array_equal(got[(f + 1, 0)][0].view(np.uint32), pose_array(o.get_result()).view(np.uint32))


@pytest.mark.gpu
def test_cpp_driver_pcd_models_and_gridsample(tmp_path, orc):
    """the model as a PCD file (ascii, binary, binary_compressed), gridSample of the re-centred model
    (VoxelGrid on the device, :672) against the oracle's VoxelGrid, tracking, position"""
    from pcl_tracking_amd import build

    exe = build.build_example()
    cluster = cluster_of(scene.make_model(4000))
    frame = scene.make_scene(50000)[:20000]
    for mode, name in (("ascii", "m_ascii.pcd"), ("binary", "m_bin.pcd"), ("binary_compressed", "m_lzf.pcd")):
        write_pcd(tmp_path / name, cluster, mode)
    cluster.tofile(tmp_path / "m.bin")
    frame.tofile(tmp_path / "frame.bin")
    outs = []
    for m in ("m_ascii.pcd", "m_bin.pcd", "m_lzf.pcd", "m.bin"):
        r = run(exe, [tmp_path / m, tmp_path / "frame.bin", "--seed", 5])
        outs.append([line for line in r.stdout.splitlines() if line.startswith("frame")])
    assert len(outs[0]) == 1 and outs[0] == outs[1] == outs[2] == outs[3]  # %.9g round-trips float32
    ref, trans, ref_full = oracle_prepare(orc, cluster, 0.01)
    assert len(ref) < len(ref_full)
    assert "downsampled: %d" % len(ref) in r.stderr
    o = oracle_tracker(orc, ref, trans, 400, 5)
    o.set_input(frame)
    assert o.compute() == 0
    pose, cen = parse(r.stdout)[(1, 0)]
    np.testing.assert_array_equal(pose.view(np.uint32), pose_array(o.get_result()).view(np.uint32))
    _, want_c = orc.object_position(ref_full, as_particle(pose))
    np.testing.assert_array_equal(cen.view(np.uint32), want_c[:3].view(np.uint32))


@pytest.mark.gpu
def test_cpp_rccl_host_with_one_rank_equals_compute(tmp_path, orc):
    """examples/dist_tracking_amd.cpp, world size 1: pft_dist_phase_a -> ncclAllReduce(max) -> phase_b -> ncclAllGather ->
    phase_c on one stream must give, bit for bit, what pft_compute gives (auto_tracking_amd) and what the oracle gives"""
    from pcl_tracking_amd import build

    single, dist = build.build_example(), build.build_dist_example()
    cluster = cluster_of(scene.make_model(1024))
    frames = [scene.make_scene(50000, obj_pose=scene.advance_pose(scene.GT_POSE, 2 * f))[:20000] for f in range(3)]
    cluster.tofile(tmp_path / "model.bin")
    paths = []
    for i, fr in enumerate(frames):
        fr.tofile(tmp_path / ("frame%d.bin" % i))
        paths.append(tmp_path / ("frame%d.bin" % i))
    args = [tmp_path / "model.bin"] + paths + ["--particles", 2048, "--seed", 3, "--model-leaf", 0]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    a = [line for line in run(single, args).stdout.splitlines() if line.startswith("frame")]
    b = [line for line in run(dist, args, env=env).stdout.splitlines() if line.startswith("frame")]
    assert len(a) == 3 and a == b
    ref, trans, _ = oracle_prepare(orc, cluster, 0.0)
    o = oracle_tracker(orc, ref, trans, 2048, 3)
    got = parse("\n".join(b))
    for f in range(3):
        o.set_input(frames[f])
        assert o.compute() == 0
        np.testing.assert_array_equal(got[(f + 1, 0)][0].view(np.uint32), pose_array(o.get_result()).view(np.uint32))
