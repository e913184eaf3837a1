"""Generates tests/golden/*.npz from the CPU oracle on small seeded cases.

These fixtures pin the ORACLE RESTATEMENT, not PCL: PCL 1.8.0 cannot be built or imported here and the
reference ships no vectors for this path, so parity stays "unpinned" with respect to PCL.
They exist so that (a) a change to the oracle is visible as a diff of committed data and (b) the GPU path
is checked against fixed numbers as well as against a live oracle run.

    python tests/golden/make_golden.py        # rewrites the fixtures
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import oracle as orc  # noqa: E402

from pcl_tracking_amd import scene  # noqa: E402


def particles_around(pose, n, seed, sig_t=0.015, sig_r=0.09):
    rng = np.random.default_rng(seed)
    p = np.zeros(n, scene.PARTICLE_DTYPE)
    for k, name in enumerate(("x", "y", "z")):
        p[name] = pose[k] + rng.normal(0, sig_t, n)
    for k, name in enumerate(("roll", "pitch", "yaw")):
        p[name] = pose[3 + k] + rng.normal(0, sig_r, n)
    p["w"] = 1.0
    p["weight"] = 1.0 / n
    return p


def case_eval(name, M, N, P, seed):
    model = scene.make_model(M, seed=1000 + M)
    cloud = scene.make_scene(50000)[:N]
    t = orc.Tracker(orc.default_config(particle_num=P, seed=seed, threads=1, emulate_pcl_alloc=0))
    t.set_reference(model)
    t.set_trans(scene.initial_trans())
    t.set_input(cloud)
    p = particles_around(scene.model_gt_pose(), P, seed)
    mats = np.stack([orc.get_transformation(*[q[k] for k in ("x", "y", "z", "roll", "pitch", "yaw")])[:3] for q in p])
    ev = t.eval_weights(p, want_nn=True, mats=mats)
    w, fit = orc.normalize_weights(ev["raw"])
    a, q = orc.gen_alias_table(w)
    pw = p.copy()
    pw["weight"] = w
    mean = orc.weighted_mean(pw)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        model=model.view(np.uint8), cloud=cloud.view(np.uint8), particles=p.view(np.float32).reshape(-1, 8),
        mats=mats, raw=ev["raw"], nn_idx=ev["nn_idx"], nn_d2=ev["nn_d2"], crop_idx=ev["crop_idx"],
        bbox=ev["bbox"], octree_depth=ev["octree_depth"], octree_min=ev["octree_min"], octree_max=ev["octree_max"],
        weights=w, fit_ratio=fit, alias_a=a, alias_q=q, mean=np.frombuffer(mean.tobytes(), np.float32),
        scan=np.array([ev["scan_queries"], ev["scan_points"]], np.uint64))
    print(name, "crop", len(ev["crop_idx"]), "depth", ev["octree_depth"])


def case_track(name, M, N, P, frames, seed):
    model = scene.make_model(M, seed=2000 + M)
    cloud = scene.make_scene(50000)[:N]
    t = orc.Tracker(orc.default_config(particle_num=P, seed=seed, threads=1, emulate_pcl_alloc=0))
    t.set_reference(model)
    t.set_trans(scene.initial_trans())
    t.set_input(cloud)
    res = []
    for f in range(frames):
        assert t.compute() == 0
        res.append(np.frombuffer(t.get_result().tobytes(), np.float32).copy())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), model=model.view(np.uint8), cloud=cloud.view(np.uint8),
                        results=np.stack(res), particles=t.get_particles().view(np.float32).reshape(-1, 8),
                        P=P, seed=seed)
    print(name, "final", res[-1][:3])


def case_filters(name, w, h):
    """the input front end on a small synthetic sensor frame"""
    frame = scene.make_depth_frame(w, h)
    leaf = 0.03  # the frame is small and sparse: a 3 cm leaf makes voxels hold several points
    idx = orc.pass_through(frame, "z", 0.0, 10.0)
    approx512 = orc.approx_voxel_grid(frame[idx], leaf, 512)
    approx64 = orc.approx_voxel_grid(frame[idx], leaf, 64)
    exact = orc.voxel_grid(frame[idx], leaf)

    def pack(c):  # x, y, z, rgba: the other 16 bytes of a point are constant (data[3] = 1, padding 0)
        return np.stack([c["x"].view(np.uint32), c["y"].view(np.uint32), c["z"].view(np.uint32), c["rgba"]], 1)

    np.savez_compressed(os.path.join(HERE, name + ".npz"), frame=pack(frame), pass_idx=idx, leaf=np.float32(leaf),
                        approx512=pack(approx512), approx64=pack(approx64), exact=pack(exact))
    print(name, len(frame), "->", len(idx), "->", len(approx512), len(approx64), len(exact))


def case_kld(name):
    """the KLD-adaptive resample: population + alias table in, new particle set / bins / k out"""
    rng = np.random.default_rng(42)
    gt = scene.model_gt_pose()
    old = np.zeros(220, scene.PARTICLE_DTYPE)
    for k, nm in enumerate(("x", "y", "z", "roll", "pitch", "yaw")):
        old[nm] = gt[k] + rng.normal(0, 0.004, len(old))  # tight population: the KL bound stops the loop early
    old["w"] = 1.0
    w = rng.random(len(old)).astype(np.float32) ** 3
    old["weight"] = w / w.sum()
    a, q = orc.gen_alias_table(old["weight"])
    motion = np.zeros(1, scene.PARTICLE_DTYPE)
    motion["x"], motion["pitch"] = 0.003, 0.01
    cfg = orc.default_config(kld_adaptive=1, seed=21)
    out = {}
    for epoch in (0, 5):
        p, bins, k = orc.kld_resample(cfg, old, a, q, motion, epoch)
        out["particles_%d" % epoch] = p.view(np.float32).reshape(-1, 8)
        out["bins_%d" % epoch] = bins
        out["k_%d" % epoch] = k
    np.savez_compressed(os.path.join(HERE, name + ".npz"), old=old.view(np.float32).reshape(-1, 8), alias_a=a, alias_q=q,
                        motion=motion.view(np.float32).reshape(-1, 8), seed=21, **out)
    print(name, [len(out["particles_%d" % e]) for e in (0, 5)], [out["k_%d" % e] for e in (0, 5)])


if __name__ == "__main__":
    if "kld" in sys.argv[1:]:
        case_kld("kld_small")
        sys.exit(0)
    if "filters" in sys.argv[1:]:  # only the front-end fixture (leaves the tracker fixtures untouched)
        case_filters("filters_small", 96, 54)
        sys.exit(0)
    case_eval("eval_small", M=96, N=1500, P=24, seed=7)
    case_eval("eval_ragged", M=257, N=4001, P=33, seed=8)
    case_track("track_small", M=200, N=5000, P=200, frames=4, seed=12)
