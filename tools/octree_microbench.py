"""Builder time on FIXED crops of growing size (HIP events around the build only, ~7 us event overhead included):
particles around the ground truth with a growing spread -> growing crop box.  Run with PFT_FORCE_BUILDER=single|sorted.
usage: python tools/octree_microbench.py [N]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcl_tracking_amd import scene, tracker  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
P = 2048
model = scene.make_model(2048)
cloud = scene.make_scene(N)
t = tracker.make_reference_tracker(particle_num=P, seed=1)
t.setReferenceCloud(model)
t.setTrans(scene.initial_trans())
t.setInputCloud(cloud)
t.compute()
gt = scene.model_gt_pose()
rng = np.random.default_rng(0)
t.profileEnable(True)
for spread in (0.0, 0.02, 0.05, 0.1, 0.15, 0.2, 0.3, 0.5):
    p = np.zeros(P, scene.PARTICLE_DTYPE)
    for k, name in enumerate(("x", "y", "z")):
        p[name] = gt[k] + rng.uniform(-spread, spread, P)
    for k, name in enumerate(("roll", "pitch", "yaw")):
        p[name] = gt[3 + k]
    p["w"] = 1.0
    p["weight"] = 1.0 / P
    st = t.evalWeights(p)
    t.evalWeights(p)  # the builder choice follows the previous crop size
    t.profileReset()
    for r in range(20):
        t.evalWeights(p)
    pr = t.profileGet()
    print("spread %.2f crop %6d depth %2d  octree %.1f us  likelihood %.1f us" % (
        spread, len(st["crop_idx"]), st["octree_depth"], pr["octree"][0] / pr["octree"][1] * 1e3,
        pr["likelihood"][0] / pr["likelihood"][1] * 1e3))
