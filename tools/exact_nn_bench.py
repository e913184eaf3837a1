"""Frame time of the NearestPairPointCloudCoherence mode (true nearest neighbour) beside the approximate one.
usage: python tools/exact_nn_bench.py [P]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcl_tracking_amd import scene, tracker  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
model, cloud = scene.make_model(2048), scene.make_scene(50000)
for exact in (False, True):
    t = tracker.ParticleFilterTracker(seed=1)
    t.setParticleNum(P)
    coh = tracker.NearestPairPointCloudCoherence() if exact else tracker.ApproxNearestPairPointCloudCoherence()
    coh.addPointCoherence(tracker.DistanceCoherence())
    hc = tracker.HSVColorCoherence()
    hc.setWeight(0.1)
    coh.addPointCoherence(hc)
    coh.setSearchMethod(tracker.OctreeSearch(0.01))
    coh.setMaximumDistance(0.1)
    t.setCloudCoherence(coh)
    t.setReferenceCloud(model)
    t.setTrans(scene.initial_trans())
    t.setInputCloud(cloud)
    for _ in range(10):
        t.compute()
    t.synchronize()
    t.profileEnable(True)
    t.profileReset()
    t0 = time.perf_counter()
    for _ in range(50):
        t.compute()
    t.synchronize()
    dt = (time.perf_counter() - t0) / 50
    pr = t.profileGet()
    print("%s: %.3f ms/frame (with events)  grid/octree %.1f us  likelihood %.1f us per launch" % (
        "exact NN " if exact else "approx NN", dt * 1e3, pr["octree"][0] / pr["octree"][1] * 1e3,
        pr["likelihood"][0] / pr["likelihood"][1] * 1e3))
    if exact:
        import ctypes as C
        import numpy as np
        dbg = np.zeros(32, np.uint64)
        t._check(t._L.pft_debug_get_descent_stats(t._h, dbg.ctypes.data_as(C.c_void_p)))
        print("  last iteration: %d cropped points in a grid of %d cells; %d cells hold queries, %d list entries (%.1f per cell); %d queries in %d blocks of 64"
              % (dbg[13], dbg[12], dbg[8], dbg[9], float(dbg[9]) / max(int(dbg[8]), 1), dbg[10], dbg[11]))
        if dbg[16:21].any():  # a -DPFT_EC_TIMING build: where k_ec_build's waves spend their time (us summed over the waves of ALL launches)
            print("  k_ec_build wave time (us): first shells %.0f, far sweeps %.0f, list walk %.0f, allot + copy %.0f, no-list exit %.0f"
                  % tuple(dbg[16:21].astype(float) / 100.0))
        os.environ["PFT_EXACT_PER_QUERY"] = "1"  # the list statistics below are counted by the per-query kernel's debug variant
        p = t.getParticles()
        t.evalWeights(p, want_nn=True)  # the debug variant counts list use (first call: counters start at zero?)
        dbg = np.zeros(32, np.uint64)
        t._check(t._L.pft_debug_get_descent_stats(t._h, dbg.ctypes.data_as(C.c_void_p)))
        q, served, walked, outside, waves, wmax, wfall = [int(v) for v in dbg[:7]]
        print("  queries %d, served by a list %.4f, mean list %.1f, outside the grid %d; per wave: max list %.1f, waves with a shell-search lane %.4f"
              % (q, served / max(q, 1), walked / max(served, 1), outside, wmax / max(waves - wfall, 1), wfall / max(waves, 1)))
