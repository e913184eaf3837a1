"""Which of the two arithmetic differences parts a long run first?  The device against the oracle
with its test-only modes switched one at a time: trig_mode (1 = the device's double sin / cos rounded to float; 0 = PCL's
cosf / sinf) and sum_mode (1 = the device's adjacent-pair trees in double; 0 = PCL's sequential sums, the mean in float).
Prints the first frame whose weighted-mean pose differs by >= 1e-4, per configuration and seed.  GPU box:
    python tools/diag/longrun_attribution.py [seeds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from pcl_tracking_amd import tracker  # noqa: E402
import test_gpu_longrun as T  # noqa: E402

seeds = [11, 21, 22, 23][: int(sys.argv[1]) if len(sys.argv) > 1 else 4]
for P, kld in ((8192, False), (400, False), (400, True)):
    for trig, summ in ((1, 1), (1, 0), (0, 1), (0, 0)):
        firsts = []
        for seed in seeds:
            g, o = T.make_pair(tracker, oracle, P, seed=seed, kld=kld, trig_mode=trig, sum_mode=summ)
            first = None
            for f in range(T.FRAMES):
                c = T.frame_cloud(f)
                g.setInputCloud(c)
                o.set_input(c)
                g.compute()
                o.compute()
                rg, ro = g.getResult(), o.get_result()
                a = max(abs(float(rg[k]) - float(ro[k])) for k in T.KEYS)
                if a >= 1e-4:
                    first = f
                    break
            firsts.append(first)
        print("P=%5d kld=%d  oracle trig_mode=%d sum_mode=%d  first frame over 1e-4 per seed %s: %s"
              % (P, int(kld), trig, summ, seeds, firsts), flush=True)
