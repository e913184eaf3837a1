"""per-kernel time of the reference's own operating point (400 particles fixed / KLD) from HIP events"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcl_tracking_amd import scene, tracker
P = int(sys.argv[1]) if len(sys.argv) > 1 else 400
kld = len(sys.argv) > 2 and sys.argv[2] == "kld"
model, cloud = scene.make_model(2048), scene.make_scene(50000)
t = tracker.make_reference_tracker(particle_num=P, seed=1, kld=kld)
t.setReferenceCloud(model); t.setTrans(scene.initial_trans()); t.setInputCloud(cloud)
for _ in range(30): t.compute()
t.synchronize()
t0 = time.perf_counter()
for _ in range(300): t.compute()
t.synchronize()
print("P=%d kld=%s: %.1f us/frame, crop %d" % (P, kld, (time.perf_counter() - t0) / 300 * 1e6, t.debugHostStat()[0]))
t.profileEnable(True); t.profileReset()
for _ in range(100): t.compute()
pr = t.profileGet()
print({k: round(v[0] / 100 * 1e3, 1) for k, v in pr.items()})
