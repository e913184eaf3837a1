"""diagnostic: per-frame differences GPU vs oracle over the long-run sequence (own trig / same trig)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from pcl_tracking_amd import scene, tracker
import test_gpu_longrun as T

KEYS = T.KEYS
P = int(sys.argv[1]); kld = int(sys.argv[2]); trig = int(sys.argv[3]); frames = int(sys.argv[4])
g, o = T.make_pair(tracker, oracle, P, seed=11, kld=bool(kld), trig_mode=trig, sum_mode=trig)
for f in range(frames):
    c = T.frame_cloud(f)
    g.setInputCloud(c); o.set_input(c)
    g.compute(); o.compute()
    rg, ro = g.getResult(), o.get_result()
    a = max(abs(float(rg[k]) - float(ro[k])) for k in KEYS)
    pg, po = g.getParticles(), o.get_particles()
    n = min(len(pg), len(po))
    d = np.zeros(n)
    for k in KEYS:
        d = np.maximum(d, np.abs(pg[k][:n].astype(np.float64) - po[k][:n]))
    hs = g.debugHostStat()
    wd = np.abs(pg["weight"][:n].astype(np.float64) - po["weight"][:n])
    print("f=%2d N=%6d crop=%6d D=%2d  pose diff %.3g  n=%d/%d  particles>1e-5: %d  >1e-7: %d  max %.3g  wdiff max %.3g rel %.3g  fit %.6g/%.6g" % (
        f, len(c), hs[0], hs[1], a, len(pg), len(po), int((d > 1e-5).sum()), int((d > 1e-7).sum()), d.max(),
        wd.max(), (wd / np.maximum(po["weight"][:n], 1e-30)).max(), g.getFitRatio(), o.fit_ratio()), flush=True)
