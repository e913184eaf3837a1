"""The reference's own operating point (auto_tracking.cpp:201-254): 400 fixed particles / KLD-adaptive <= 500, same model
and cloud as bench.py; prints per-stage event timings.  Usage: python tools/refpoint_bench.py [fixed|kld] [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcl_tracking_amd import scene, tracker  # noqa: E402

kld = (sys.argv[1] if len(sys.argv) > 1 else "kld") == "kld"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
model, cloud = scene.make_model(2048), scene.make_scene(50000)
t = tracker.make_reference_tracker(particle_num=400, seed=1, kld=kld)
t.setReferenceCloud(model)
t.setTrans(scene.initial_trans())
t.setInputCloud(cloud)
for _ in range(20):
    t.compute()
t.synchronize()
t0 = time.perf_counter()
for _ in range(frames):
    t.compute()
t.synchronize()
print("%s: %.3f ms per frame, %d particles" % ("kld" if kld else "fixed", (time.perf_counter() - t0) / frames * 1e3, len(t.getParticles())))
t.profileEnable(True)
for _ in range(50):
    t.compute()
t.synchronize()
pr = t.profileGet()
print({k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in pr.items()}, "us per launch (events add ~7 us each)")
