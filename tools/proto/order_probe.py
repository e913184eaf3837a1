"""Prototype (numpy, statistics only): does ordering the reference points by the number of generic descent levels
they need at the mean pose make the waves of the likelihood kernel more homogeneous?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcl_tracking_amd import scene
from oracle import oracle as orc

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
model = scene.make_model(2048)
cloud = scene.make_scene(50000)
o = orc.Tracker(orc.default_config(particle_num=P, seed=1, threads=8, emulate_pcl_alloc=0))
o.set_reference(model); o.set_trans(scene.initial_trans()); o.set_input(cloud)
for f in range(frames):
    o.compute()
res_pose = o.get_result()
pall = o.get_particles()
R = o.eval_weights(pall)
p = pall[:: max(1, P // 64)]
D = R["octree_depth"]; omin = np.asarray(R["octree_min"], np.float64); res = 0.01
pts = np.stack([cloud["x"], cloud["y"], cloud["z"]], 1)[R["crop_idx"]].astype(np.float64)
keys = np.floor((pts - omin) / res).astype(np.int64)
occ = []
for l in range(D + 1):
    a = np.zeros((1 << l,) * 3, bool)
    k = keys >> (D - l)
    a[k[:, 0], k[:, 1], k[:, 2]] = True
    occ.append(a)

def descend(q):
    m = len(q); k = np.zeros((m, 3), np.int64); gen = np.zeros(m, np.int64); off = np.zeros(m, bool)
    for l in range(D):
        s = res * (1 << (D - l - 1))
        dist = np.full((m, 8), np.inf)
        for c in range(8):
            b = np.array([(c >> 2) & 1, (c >> 1) & 1, c & 1])
            ck = 2 * k + b
            ex = occ[l + 1][ck[:, 0], ck[:, 1], ck[:, 2]]
            cen = (ck + 0.5) * s + omin
            dist[:, c] = np.where(ex, ((cen - q) ** 2).sum(1), np.inf)
        bc = dist.argmin(1)
        bk = 2 * k + np.stack([(bc >> 2) & 1, (bc >> 1) & 1, bc & 1], 1)
        cont = np.floor((q - omin) / s).astype(np.int64)
        off |= ~(cont == bk).all(1)
        gen += off
        k = bk
    return gen

m = np.stack([model["x"], model["y"], model["z"]], 1).astype(np.float64)
# Morton order of the model (as the library stores it)
def morton(v):
    lo = v.min(0); span = (v.max(0) - lo).max() + 1e-9
    g = np.floor((v - lo) / span * 1023).astype(np.int64)
    code = np.zeros(len(v), np.int64)
    for b in range(10):
        for a in range(3):
            code |= ((g[:, a] >> b) & 1) << (3 * b + (2 - a))
    return np.argsort(code, kind="stable")
mo = morton(m)
mm = m[mo]
def xf(pp):
    M = np.asarray(orc.get_transformation(*[float(pp[f]) for f in ("x", "y", "z", "roll", "pitch", "yaw")]), np.float64)
    return mm @ M[:3, :3].T + M[:3, 3]
G = np.stack([descend(xf(p[i])) for i in range(len(p))])  # (particles, 2048) in Morton order
print("depth", D, "crop", len(pts), "particles", len(p), "mean gen %.2f" % G.mean())
def wave_cost(order):
    g = G[:, order].reshape(len(p), -1, 64)
    return g.max(2).mean()
print("Morton order: wave-max mean %.2f" % wave_cost(np.arange(2048)))
# mean pose classification
class R_: pass
rp = {f: float(np.average(pall[f], weights=pall["weight"])) for f in ("x", "y", "z", "roll", "pitch", "yaw")}
gm = descend(xf(rp))
print("gen at mean pose hist", np.bincount(gm, minlength=D + 1))
o1 = np.argsort(gm, kind="stable")
print("ordered by gen at the mean pose: wave-max mean %.2f" % wave_cost(o1))
o2 = np.argsort(G.mean(0), kind="stable")
print("ordered by per-point mean gen over particles (oracle bound): %.2f" % wave_cost(o2))
o3 = np.argsort(np.round(G.mean(0)), kind="stable")
print("ordered by rounded per-point mean gen: %.2f" % wave_cost(o3))
print("ideal (no divergence) %.2f" % G.mean())
