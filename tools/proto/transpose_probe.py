"""Prototype (numpy, statistics only): is a wave of the likelihood kernel more homogeneous in the number of generic
descent levels when its 64 lanes are 64 PARTICLES at one reference point instead of 64 reference points of one particle?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0], "1024", sys.argv[1] if len(sys.argv) > 1 else "6"]
import order_probe as op  # runs the tracker, builds occupancy, defines descend / xf / G for 64 sampled particles

np.random.seed(0)
pall = op.pall
sel = pall[np.argsort(np.random.rand(len(pall)))[:256]]
G = np.stack([op.descend(op.xf(sel[i])) for i in range(len(sel))])  # (256 particles, 2048 points in Morton order)
print("mean gen %.3f" % G.mean())
print("(particle, 64 points)   wave-max mean %.3f" % G.reshape(256, 32, 64).max(2).mean())
print("(point, 64 particles)   wave-max mean %.3f" % G.T.reshape(2048, 4, 64).max(2).mean())
# particles sorted by pose similarity first (x, then yaw): neighbours in the population are not neighbours in space
o = np.lexsort((sel["yaw"], sel["x"]))
print("(point, 64 particles sorted by x)  %.3f" % G[o].T.reshape(2048, 4, 64).max(2).mean())
for (a, b) in ((8, 8), (4, 16), (16, 4), (2, 32), (32, 2)):
    g = G.reshape(256 // a, a, 2048 // b, b).transpose(0, 2, 1, 3).reshape(-1, a * b)
    print("tiles of %2d particles x %2d points: %.3f" % (a, b, g.max(1).mean()))
print("per-wave std of gen: by particle %.3f, by point %.3f" % (G.reshape(256, 32, 64).std(2).mean(), G.T.reshape(2048, 4, 64).std(2).mean()))
