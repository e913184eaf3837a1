"""Phase breakdown of the two single-workgroup kernels (octree build, population) from the in-kernel
wall-clock stamps (100 MHz).  The stamps are compiled into the diagnostic variant only (they cost several microseconds on
the one-workgroup kernels' critical path):
    python tools/build_variant.py diag -DPFT_DIAG                              (here)
    PFT_LIB_PATH=$PWD/pcl_tracking_amd/_build/var_diag.so python tools/phase_ticks.py [P] [N]     (GPU box)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcl_tracking_amd import scene, tracker  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
model = scene.make_model(2048)
cloud = scene.make_scene(N, mode="organized" if N == 307200 else "voxel")
t = tracker.make_reference_tracker(particle_num=P, seed=1)
t.setReferenceCloud(model)
t.setTrans(scene.initial_trans())
t.setInputCloud(cloud)
for i in range(int(sys.argv[3]) if len(sys.argv) > 3 else 10):
    t.compute()
t.synchronize()
tk = np.zeros(32, np.uint64)
t._check(t._L.pft_debug_get_ticks(t._h, tk.ctypes.data_as(C.c_void_p)))
o = tk[:9].astype(np.int64)
names = ["init", "replay", "keys", "levels", "leaf-count+scan", "leaf-scatter", "leaf-rank+gather", "flush", "tables"]
print("octree phases (us):", dict(zip(names[1:], ((o[1:] - o[:-1]) / 100.0).round(2))), "total", (o[8] - o[0]) / 100.0)
print("  replay split (us): start->tail-aabb done %.2f, start->head done %.2f, whole replay %.2f" % ((int(tk[13]) - int(tk[0])) / 100.0, (int(tk[14]) - int(tk[0])) / 100.0, (int(tk[1]) - int(tk[0])) / 100.0))
print("  levels split (us): point pass %.2f  count %.2f  scan %.2f  write+zero %.2f" % tuple(tk[9:13].astype(np.float64) / 100.0))
p = tk[16:22].astype(np.int64)
names = ["load", "normalize", "mean", "alias-pass1", "alias-scan+pass2"]
print("population phases (us):", dict(zip(names, ((p[1:] - p[:-1]) / 100.0).round(2))), "total", (p[5] - p[0]) / 100.0)
