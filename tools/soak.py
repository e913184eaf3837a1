"""Soak run: many frames through one handle (and the front end), watching device memory and result sanity.
usage: python tools/soak.py [frames]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcl_tracking_amd import filters, scene, tracker  # noqa: E402
import torch  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
model = scene.make_model(2048)
raw = scene.make_depth_frame(480, 270)
f = filters.make_reference_input_filter()
t = tracker.make_reference_tracker(particle_num=8192, seed=1)
t.setReferenceCloud(model)
t.setTrans(scene.initial_trans())
k = tracker.make_reference_tracker(particle_num=400, seed=2, kld=True)
k.setReferenceCloud(model)
k.setTrans(scene.initial_trans())
free0 = None
t0 = time.time()
for i in range(frames):
    f.setInputCloud(raw)
    ptr, n = f.filterDevice()
    t.setInputCloudDevice(ptr, n, keepalive=f)
    t.compute()
    k.setInputCloudDevice(ptr, n, keepalive=f)
    k.compute()
    if i % 2000 == 0 or i == frames - 1:
        r, rk = t.getResult(), k.getResult()
        free, total = torch.cuda.mem_get_info()
        if free0 is None and i > 0:
            free0 = free
        ok = all(np.isfinite(float(r[c])) and np.isfinite(float(rk[c])) for c in ("x", "y", "z", "roll", "pitch", "yaw"))
        ok = ok and (t.debugHostStat()[2:] == 0).all() and (k.debugHostStat()[2:] == 0).all()  # no device-side failure, ever
        print("frame %6d  %.1f s  free %.1f MiB  pose x=%.4f z=%.4f  kld particles %d  finite=%s" % (
            i, time.time() - t0, free / 2**20, float(r["x"]), float(r["z"]), len(k.getParticles()), ok), flush=True)
        assert ok
if free0 is not None:
    free, _ = torch.cuda.mem_get_info()
    print("device memory drift over the run: %.2f MiB" % ((free0 - free) / 2**20))
