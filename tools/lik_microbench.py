"""Stage shares of k_likelihood on a FIXED particle set (steady state of BASELINE configs[1]): the tracker
runs normally, then pft_eval_weights is timed on its particles with stages ablated (timing only).  The ablation mask
exists only in the diagnostic variant library: run as
    python tools/build_variant.py diag -DPFT_DIAG            (here: hipcc cross-compiles)
    PFT_LIB_PATH=$PWD/pcl_tracking_amd/_build/var_diag.so python tools/lik_microbench.py     (GPU box)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcl_tracking_amd import scene, tracker  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 60
reps = 10
N = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
model = scene.make_model(2048)
cloud = scene.make_scene(N, mode="organized" if N == 307200 else "voxel")
t = tracker.make_reference_tracker(particle_num=P, seed=1)
if not hasattr(t._L, "pft_debug_set_ablate"):
    sys.exit("this library has no stage ablation: build the diagnostic variant (see the docstring) and set PFT_LIB_PATH")
t.setReferenceCloud(model)
t.setTrans(scene.initial_trans())
t.setInputCloud(cloud)
for i in range(frames):
    t.compute()
p = t.getParticles()
t.profileEnable(True)
for mask, name in ((0, "full"), (1, "no generic levels"), (2, "no leaf scan"), (4, "no coherence"), (6, "descent only"),
                   (7, "transform+key+fast only")):
    t._L.pft_debug_set_ablate(mask)
    t.evalWeights(p)
    t.profileReset()
    for r in range(reps):
        t.evalWeights(p)
    pr = t.profileGet()
    print("%-26s likelihood %.1f us   octree %.1f us   aabb %.1f us" % (
        name, pr["likelihood"][0] / pr["likelihood"][1] * 1e3, pr["octree"][0] / pr["octree"][1] * 1e3,
        pr["aabb"][0] / pr["aabb"][1] * 1e3))
t._L.pft_debug_set_ablate(0)
