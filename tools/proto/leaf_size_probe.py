"""Leaf sizes the likelihood kernel's queries reach on BASELINE configs[2] (307 200-point organised cloud, no downsample),
and what a wave-cooperative scan of the long leaves would cost against the per-lane scan.  CPU only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O  # noqa: E402
from pcl_tracking_amd import scene  # noqa: E402
from refill_sim import morton_perm  # noqa: E402

model = scene.make_model(2048)
cloud = scene.make_scene(307200, mode="organized")
step = [0.015 * 0.015] * 3 + [0.015 * 0.015 * 40.0] * 3
t = O.Tracker(O.default_config(particle_num=512, iteration_num=2, step_cov=step, init_cov=[0.00001] * 6, init_mean=[0.0] * 6, seed=1, threads=8))
t.set_reference(model)
t.set_trans(scene.initial_trans())
t.set_input(cloud)
for _ in range(12):
    t.compute()
parts = t.get_particles()
st = t.eval_weights(parts)
sub = parts[::8][:64]
ev = t.eval_weights(sub, want_nn=True, bbox=st["bbox"])
D, omin = ev["octree_depth"], ev["octree_min"]
crop = cloud[ev["crop_idx"]]
ck = np.floor((np.stack([crop["x"], crop["y"], crop["z"]], 1).astype(np.float64) - omin) / 0.01).astype(np.int64)
key = ck[:, 0] << 40 | ck[:, 1] << 20 | ck[:, 2]
uk, cnt = np.unique(key, return_counts=True)
leafcnt = dict(zip(uk.tolist(), cnt.tolist()))
pos = {int(c): i for i, c in enumerate(ev["crop_idx"].tolist())}
perm = morton_perm(model)
LS = np.zeros((len(sub), 2048), np.int64)
for pi in range(len(sub)):
    idx = ev["nn_idx"][pi][perm]
    LS[pi] = [(leafcnt[int(key[pos[i]])] if i in pos else leafcnt[int(key[i])]) if i >= 0 else 0 for i in idx.tolist()]
print("crop", len(crop), "depth", D, "leaves", len(uk), "points per leaf mean %.2f" % cnt.mean())
print("candidates per query: mean %.2f  median %d  p90 %d  max %d" % (LS.mean(), np.median(LS), np.percentile(LS, 90), LS.max()))
W = LS.reshape(len(sub), -1, 64)
print("per wave round: max %.1f (mean of maxima), lane utilisation of the per-lane scan %.2f" % (W.max(2).mean(), (np.ceil(W / 2).sum(2) / (64 * np.ceil(W.max(2) / 2))).mean()))
C_PAIR = 28.0   # VALU instructions of one per-lane round of two candidates
C_COOP = 55.0   # one long leaf served by the whole wave: 7 readlane, load, distance, 6-step u64 DPP min, write-back
for T in (8, 12, 16, 20, 24, 32):
    serial = np.ceil(W.max(2) / 2) * C_PAIR
    capped = np.ceil(np.minimum(W, T).max(2) / 2) * C_PAIR
    nlong = (W > T).sum(2)
    coop = capped + nlong * C_COOP * np.ceil((W.max(2) - T).clip(0) / 64 + 1e-9).clip(1)
    print("threshold %2d: lanes over it per round %.1f (%.0f %%)  instructions per round: per-lane %.0f, cooperative %.0f (%.2f x)" % (
        T, nlong.mean(), 100 * (W > T).mean(), serial.mean(), coop.mean(), coop.mean() / serial.mean()))
