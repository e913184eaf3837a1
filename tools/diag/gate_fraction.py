"""fraction of the likelihood's queries whose (approximate) nearest neighbour is beyond the gate, and how far beyond:
what a conservative pre-test could skip (queries farther than the gate from EVERY cropped point contribute nothing)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcl_tracking_amd import scene, tracker
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
model, cloud = scene.make_model(2048), scene.make_scene(50000)
t = tracker.make_reference_tracker(particle_num=P, seed=1)
t.setReferenceCloud(model); t.setTrans(scene.initial_trans()); t.setInputCloud(cloud)
for _ in range(60): t.compute()
p = t.getParticles()
G = t.evalWeights(p, want_nn=True)
d = np.sqrt(G["nn_d2"].astype(np.float64))
print("queries %d, approx-NN beyond the 10 cm gate: %.4f; beyond 15 cm: %.4f; beyond 20 cm: %.4f; median %.3f m" % (d.size, (d >= 0.1).mean(), (d >= 0.15).mean(), (d >= 0.2).mean(), np.median(d)))
per = (d.reshape(P, -1) >= 0.1).mean(1)
print("per particle: share of queries beyond the gate: mean %.3f, particles with > 90 %% beyond: %.4f, with all beyond: %.4f" % (per.mean(), (per > 0.9).mean(), (per == 1).mean()))
print("weights: zero-weight particles %.4f" % (G["raw"] == 0).mean())
