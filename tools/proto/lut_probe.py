"""Prototype (numpy, statistics only): how many queries of the bench workload would be answered by a per-leaf-cell
table of the greedy descent's result, if the table only holds cells over which the result is provably constant?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcl_tracking_amd import scene
from oracle import oracle as orc

P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
model = scene.make_model(2048)
cloud = scene.make_scene(50000)
o = orc.Tracker(orc.default_config(particle_num=P, seed=1, threads=8, emulate_pcl_alloc=0))
o.set_reference(model); o.set_trans(scene.initial_trans()); o.set_input(cloud)
for f in range(frames):
    o.compute()
p = o.get_particles()[:: max(1, P // 48)]
R = o.eval_weights(p, want_nn=True)
D = R["octree_depth"]; omin = np.asarray(R["octree_min"], np.float64); res = 0.01
print("depth", D, "crop", len(R["crop_idx"]))
pts = np.stack([cloud["x"], cloud["y"], cloud["z"]], 1)[R["crop_idx"]].astype(np.float64)
keys = np.floor((pts - omin) / res).astype(np.int64)
n = 1 << D
occ = []  # occ[l][x,y,z] at level l (cells of 2^(D-l) leaf cells)
for l in range(D + 1):
    a = np.zeros((1 << l,) * 3, bool)
    k = keys >> (D - l)
    a[k[:, 0], k[:, 1], k[:, 2]] = True
    occ.append(a)

def descend(q, shrink_h=None, margin_len=0.0):
    """vectorised greedy descent. q: (n,3) doubles. If shrink_h is given, also returns whether the result is constant
    over the cube of half-size shrink_h around q (every level's winner beats every other existing child by more than
    the variation of the distance difference over the cube + the rounding band)."""
    m = len(q)
    k = np.zeros((m, 3), np.int64)
    gen = np.zeros(m, np.int64)
    off = np.zeros(m, bool)
    const = np.ones(m, bool)
    clev = np.full(m, D, np.int64)
    for l in range(D):
        s = res * (1 << (D - l - 1))  # child size
        best = np.full(m, np.inf); bc = np.zeros(m, np.int64)
        dist = np.full((m, 8), np.inf)
        for c in range(8):
            b = np.array([(c >> 2) & 1, (c >> 1) & 1, c & 1])
            ck = 2 * k + b
            ex = occ[l + 1][ck[:, 0], ck[:, 1], ck[:, 2]]
            cen = (ck + 0.5) * s + omin
            dd = ((cen - q) ** 2).sum(1)
            dist[:, c] = np.where(ex, dd, np.inf)
        bc = dist.argmin(1)
        best = dist[np.arange(m), bc]
        # containing child?
        cont = np.floor((q - omin) / s).astype(np.int64)
        bk = 2 * k + np.stack([(bc >> 2) & 1, (bc >> 1) & 1, bc & 1], 1)
        is_cont = (cont == bk).all(1)
        off |= ~is_cont
        gen += off
        if shrink_h is not None:
            for c in range(8):
                dl1 = (((bc >> 2) & 1) != ((c >> 2) & 1)).astype(float) + (((bc >> 1) & 1) != ((c >> 1) & 1)) + ((bc & 1) != (c & 1))
                need = 2 * shrink_h * s * dl1 + 2 * s * margin_len
                ok = (c == bc) | (dist[:, c] - best > need)
                const &= ok
            clev = np.where(const, clev, np.minimum(clev, l))
        k = bk
    descend.clev = clev
    return k, gen, const

# queries
m = np.stack([model["x"], model["y"], model["z"]], 1).astype(np.float64)
qs = []
for i in range(len(p)):
    M = np.asarray(orc.get_transformation(*[float(p[f][i]) for f in ("x", "y", "z", "roll", "pitch", "yaw")]), np.float64)
    qs.append(m @ M[:3, :3].T + M[:3, 3])
q = np.concatenate(qs)
t = (q - omin) / res
inside = ((t >= 0) & (t < n)).all(1)
print("queries", len(q), "inside box", inside.mean())
kq, gen, _ = descend(q)
print("generic levels: mean %.2f" % gen.mean(), "hist", np.bincount(gen, minlength=D + 1) / len(gen))
g64 = gen[: len(gen) // 64 * 64].reshape(-1, 64)
print("wave max mean %.2f" % g64.max(1).mean(), " frac of waves with all-zero", (g64.max(1) == 0).mean())
# table over leaf cells
margin_cells = 2e-3
g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3)
cen = (g + 0.5) * res + omin
kc, genc, const = descend(cen, shrink_h=res * (0.5 - margin_cells), margin_len=margin_cells * res)
print("table cells", len(g), "constant", const.mean())
tab = const.reshape(n, n, n)
ki = np.clip(np.floor(t).astype(np.int64), 0, n - 1)
f = t - np.floor(t)
nearface = ((f < margin_cells) | (f > 1 - margin_cells)).any(1)
hit = inside & ~nearface & tab[ki[:, 0], ki[:, 1], ki[:, 2]]
# check the table's answer equals the query's own descent
leaf_of_cell = kc.reshape(n, n, n, 3)
agree = (leaf_of_cell[ki[:, 0], ki[:, 1], ki[:, 2]] == kq).all(1)
print("hit rate all %.3f ; among gen>0 %.3f ; table answer agrees on hits %.6f" % (hit.mean(), hit[gen > 0].mean(), agree[hit].mean()))
miss = ~hit
m64 = miss[: len(miss) // 64 * 64].reshape(-1, 64)
print("misses per wave mean %.1f ; misses needing generic: %.3f of all" % (m64.sum(1).mean(), (miss & (gen > 0)).mean()))
print("mean generic levels of misses %.2f" % gen[miss].mean())

clev = descend.clev.reshape(n, n, n)  # first level whose choice is not constant over the cell (D if all constant)
cl = clev[ki[:, 0], ki[:, 1], ki[:, 2]]
cl = np.where(inside & ~nearface, cl, 0)
first_gen = D - gen  # level of the first generic step of each query
rem = np.where(gen > 0, np.maximum(0, D - np.maximum(cl, first_gen)), 0)
# levels still to be walked generically when the table hands over at level cl (fast levels below cl are free)
rem2 = D - cl  # all levels after the hand-over are walked "generically or fast"; count only those at or after first_gen
print("remaining generic levels: mean %.2f" % rem.mean(), "hist", np.bincount(rem, minlength=D + 1) / len(rem))
r64 = rem[: len(rem) // 64 * 64].reshape(-1, 64)
print("wave max mean %.2f" % r64.max(1).mean())
print("constant prefix level hist over cells", np.bincount(clev.ravel(), minlength=D + 1) / clev.size)
