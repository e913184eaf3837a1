"""In-kernel timestamps (wall_clock64, 100 MHz) of the two single-workgroup kernels at the bench workload: octree build
phases and population-stage phases.  Usage: python tools/ticks.py [particles] [frames]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcl_tracking_amd import scene, tracker  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
FRAMES = int(sys.argv[2]) if len(sys.argv) > 2 else 10  # bench.py loops one frame: after ~200 the crop has grown to ~13 500 points
model, cloud = scene.make_model(2048), scene.make_scene(50000)
t = tracker.make_reference_tracker(particle_num=P, seed=1)
t.setReferenceCloud(model)
t.setTrans(scene.initial_trans())
t.setInputCloud(cloud)
for i in range(FRAMES):
    t.compute()
t.synchronize()
p = t.getParticles()
t.profileEnable(True)
for i in range(5):
    st = t.evalWeights(p)  # the same crop every time: event timing and in-kernel stamps of one build
pr = t.profileGet()
print("crop %d depth %d; octree kernel between events: %.1f us" % (len(st["crop_idx"]), st["octree_depth"], pr["octree"][0] / pr["octree"][1] * 1e3))
tk = np.zeros(32, np.uint64)
t._check(t._L.pft_debug_get_ticks(t._h, tk.ctypes.data_as(C.c_void_p)))
o = tk[:9].astype(np.int64)
names = ["replay", "keys", "levels", "leaf counts", "leaf list", "leaf rank+copy", "write-out", "tables"]
print("octree us:", dict(zip(names[: len(o) - 1], ((o[1:] - o[:-1]) / 100.0).round(2))), "total", (o[8] - o[0]) / 100.0)
print("  levels: point pass %.1f, popcount %.1f, scan %.1f, bases %.1f us" % tuple(tk[9:13].astype(np.int64) / 100.0))
p = tk[16:22].astype(np.int64)
print("population us:", ((p[1:] - p[:-1]) / 100.0).round(2), "total", (p[5] - p[0]) / 100.0)
