"""host time of pft_set_reference with the hull-shell computation (pft_hull.hip), and the subset sizes"""
import time, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from pcl_tracking_amd import scene, tracker
rng = np.random.default_rng(1)
def cloud(xyz):
    m = np.zeros(len(xyz), scene.POINT_DTYPE); m["w"] = 1.0
    xyz = xyz.astype(np.float32); m["x"], m["y"], m["z"] = xyz[:,0], xyz[:,1], xyz[:,2]; return m
t = tracker.make_reference_tracker(particle_num=256, seed=1)
t.setReferenceCloud(scene.make_model(2048))
t.setTrans(scene.initial_trans()); t.setInputCloud(scene.make_scene(5000)); t.compute(); t.synchronize()
for name, m in (("scan 2048", scene.make_model(2048)), ("scan 8192", scene.make_model(8192)),
                ("sphere shell 8192", cloud((lambda v: v/np.linalg.norm(v,axis=1,keepdims=True))(rng.normal(0,1,(8192,3))))),
                ("uniform ball 8192", cloud(rng.uniform(-1,1,(8192,3))))):
    t0 = time.perf_counter(); t.setReferenceCloud(m); dt = time.perf_counter() - t0
    import ctypes as C
    dbg = np.zeros(32, np.uint64); t._check(t._L.pft_debug_get_descent_stats(t._h, dbg.ctypes.data_as(C.c_void_p)))
    print("%-20s set_reference %.1f ms, box over %d of %d" % (name, dt*1e3, dbg[30], dbg[31]))
