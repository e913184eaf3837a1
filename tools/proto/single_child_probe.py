"""Prototype (numpy, statistics only): how many of the generic descent levels are taken at nodes with a single existing
child (where the greedy choice needs no distance evaluation)?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcl_tracking_amd import scene
from oracle import oracle as orc

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = scene.make_model(2048)
cloud = scene.make_scene(50000)
o = orc.Tracker(orc.default_config(particle_num=P, seed=1, threads=8, emulate_pcl_alloc=0))
o.set_reference(model); o.set_trans(scene.initial_trans()); o.set_input(cloud)
for f in range(4):
    o.compute()
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
m = np.stack([model["x"], model["y"], model["z"]], 1).astype(np.float64)
lo = m.min(0); span = (m.max(0) - lo).max() + 1e-9
gq = np.floor((m - lo) / span * 1023).astype(np.int64)
code = np.zeros(len(m), np.int64)
for b in range(10):
    for a in range(3):
        code |= ((gq[:, a] >> b) & 1) << (3 * b + (2 - a))
mm = m[np.argsort(code, kind="stable")]

def descend(q):
    n = len(q); k = np.zeros((n, 3), np.int64); off = np.zeros(n, bool)
    gen = np.zeros(n, np.int64); multi = np.zeros(n, np.int64)
    seq = []  # per level: 0 fast, 1 generic single-child, 2 generic multi-child
    for l in range(D):
        s = res * (1 << (D - l - 1))
        dist = np.full((n, 8), np.inf)
        for c in range(8):
            b = np.array([(c >> 2) & 1, (c >> 1) & 1, c & 1])
            ck = 2 * k + b
            ex = occ[l + 1][ck[:, 0], ck[:, 1], ck[:, 2]]
            cen = (ck + 0.5) * s + omin
            dist[:, c] = np.where(ex, ((cen - q) ** 2).sum(1), np.inf)
        nch = np.isfinite(dist).sum(1)
        bc = dist.argmin(1)
        bk = 2 * k + np.stack([(bc >> 2) & 1, (bc >> 1) & 1, bc & 1], 1)
        cont = np.floor((q - omin) / s).astype(np.int64)
        off |= ~(cont == bk).all(1)
        gen += off
        multi += off & (nch > 1)
        seq.append(np.where(off, np.where(nch > 1, 2, 1), 0))
        k = bk
    return gen, multi, np.stack(seq, 1)

G, Mu, S = [], [], []
for i in range(len(p)):
    Mx = np.asarray(orc.get_transformation(*[float(p[f][i]) for f in ("x", "y", "z", "roll", "pitch", "yaw")]), np.float64)
    g_, m_, s_ = descend(mm @ Mx[:3, :3].T + Mx[:3, 3])
    G.append(g_); Mu.append(m_); S.append(s_)
G = np.stack(G); Mu = np.stack(Mu); S = np.stack(S)
print("depth", D, "generic levels mean %.2f, of which multi-child %.2f" % (G.mean(), Mu.mean()))
gw = G.reshape(len(p), -1, 64); mw = Mu.reshape(len(p), -1, 64)
print("per wave: max generic %.2f ; max multi-child %.2f" % (gw.max(2).mean(), mw.max(2).mean()))
# the loop "follow single-child chains cheaply, then one expensive step": cheap iterations per wave = sum over segments of max chain
Sw = S.reshape(len(p), -1, 64, D)
print("single-child generic steps: mean per query %.2f" % ((S == 1).sum(2).mean()))
