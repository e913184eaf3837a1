"""Which LDS layout the likelihood kernel gets per frame at the bench workload (depth, words, bytes needed against the 80 KiB
share) beside the launch time."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcl_tracking_amd import scene, tracker
model, cloud = scene.make_model(2048), scene.make_scene(50000)
t = tracker.make_reference_tracker(particle_num=8192, seed=1)
t.setReferenceCloud(model); t.setTrans(scene.initial_trans()); t.setInputCloud(cloud)
t.profileEnable(True)
rows = []
for f in range(240):
    t.profileReset()
    t.compute(); t.synchronize()
    pr = t.profileGet()
    depth, nl, nn = C.c_int32(), C.c_uint32(), C.c_uint32()
    mn, mx = np.zeros(3), np.zeros(3)
    t._check(t._L.pft_debug_get_octree(t._h, C.byref(depth), mn.ctypes.data_as(C.c_void_p), mx.ctypes.data_as(C.c_void_p), C.byref(nl), C.byref(nn)))
    D = depth.value
    leaf_start = nn.value - nl.value - 1
    need = 2048 + 3 * (2 << D) * 4 + (2 << 12) + leaf_start * 4 + (((nl.value + 1) * 2 + 3) & ~3)
    rows.append((f, D, nn.value, nl.value, need, pr["likelihood"][0] / pr["likelihood"][1] * 1e3))
lim = 80 * 1024
import collections
by = collections.defaultdict(list)
for r in rows[20:]:
    by[(r[1], r[4] <= lim)].append(r[5])
for k in sorted(by):
    print("depth %d fits-with-jump %s: %3d frames, likelihood %.1f us (events)" % (k[0], k[1], len(by[k]), np.mean(by[k])))
print("need bytes: min %d max %d" % (min(r[4] for r in rows[20:]), max(r[4] for r in rows[20:])))
