"""CPU model of the likelihood kernel's descent schedule (lane refill inside the work item).

The per-query work of the steady-state workload is taken from the oracle (no GPU): number of fast levels, number of
generic levels (= depth - deepest level at which the cell containing the query exists), size of the leaf reached, gate.
Schedules are then costed in SIMD cycles with the per-phase costs measured in round 2:
  lockstep   -- today's kernel: four rounds of 64 queries, every phase pays its worst lane
  refill     -- while-while: idle lanes take the item's next points (setup under a partial mask), generic levels run
                while enough lanes are in descent, leaf + coherence when enough lanes are parked
  sorted     -- setup full width for the whole item, then queries dealt to lanes by their (known) generic count
The model says what a schedule can gain before it is built; the kernel variants are then measured on the GPU.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O  # noqa: E402
from pcl_tracking_amd import scene  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
FRAMES = int(sys.argv[2]) if len(sys.argv) > 2 else 20
NPART = 128  # particles whose queries are analysed


def morton_perm(pts):
    xyz = np.stack([pts["x"], pts["y"], pts["z"]], 1).astype(np.float32)
    lo, hi = xyz.min(0), xyz.max(0)
    ext = np.float32((hi - lo).max())
    scale = np.float32(1023.0) / ext
    q = np.clip(((xyz - lo) * scale), 0, 1023).astype(np.uint32)
    m = np.zeros(len(xyz), np.uint64)
    for b in range(9, -1, -1):
        m = (m << np.uint64(3)) | (((q[:, 0] >> b) & 1) << 2 | ((q[:, 1] >> b) & 1) << 1 | ((q[:, 2] >> b) & 1)).astype(np.uint64)
    code = (m << np.uint64(32)) | np.arange(len(xyz), dtype=np.uint64)
    return np.argsort(code, kind="stable")


def workload():
    model = scene.make_model(2048)
    cloud = scene.make_scene(50000)
    step = [0.015 * 0.015] * 3 + [0.015 * 0.015 * 40.0] * 3
    cfg = O.default_config(particle_num=P, iteration_num=2, step_cov=step, init_cov=[0.00001] * 6, init_mean=[0.0] * 6,
                           seed=1, threads=8)
    t = O.Tracker(cfg)
    t.set_reference(model)
    t.set_trans(scene.initial_trans())
    t.set_input(cloud)
    for _ in range(FRAMES):
        t.compute()
    parts = t.get_particles()
    return t, model, cloud, parts


def per_query_work(t, model, cloud, parts):
    sub = parts[:: max(1, len(parts) // NPART)][:NPART]
    st = t.eval_weights(parts, want_nn=False)  # crop box of the whole population, as the tracker's own iteration
    ev = t.eval_weights(sub, want_nn=True, bbox=st["bbox"])
    D = ev["octree_depth"]
    omin = ev["octree_min"]
    res = 0.01
    crop = cloud[ev["crop_idx"]]
    cxyz = np.stack([crop["x"], crop["y"], crop["z"]], 1).astype(np.float64)
    ck = np.floor((cxyz - omin) / res).astype(np.int64)
    occ = []
    for l in range(D + 1):
        k = ck >> (D - l)
        occ.append(set((k[:, 0] << 40 | k[:, 1] << 20 | k[:, 2]).tolist()))
    leafkey = ck[:, 0] << 40 | ck[:, 1] << 20 | ck[:, 2]
    uk, cnt = np.unique(leafkey, return_counts=True)
    leafcnt = dict(zip(uk.tolist(), cnt.tolist()))
    perm = morton_perm(model)
    mxyz = np.stack([model["x"], model["y"], model["z"]], 1).astype(np.float32)[perm]
    G = np.zeros((len(sub), len(mxyz)), np.int32)
    F = np.zeros_like(G)
    LS = np.zeros_like(G)
    GATE = np.zeros(G.shape, bool)
    J = min(4, D)
    for pi, p in enumerate(sub):
        m = O.get_transformation(p["x"], p["y"], p["z"], p["roll"], p["pitch"], p["yaw"])
        q = (mxyz @ m[:3, :3].T + m[:3, 3]).astype(np.float32)
        k = np.floor((q.astype(np.float64) - omin) / res).astype(np.int64)
        inside = ((k >= 0) & (k < (1 << D))).all(1)
        L = np.zeros(len(q), np.int32)
        for l in range(1, D + 1):
            kk = k >> (D - l)
            key = (kk[:, 0] << 40 | kk[:, 1] << 20 | kk[:, 2])
            ex = np.fromiter((x in occ[l] for x in key.tolist()), bool, len(key)) & inside
            L = np.where(ex & (L == l - 1), l, L)
        G[pi] = D - L
        F[pi] = np.where(L >= J, L - J, L)
        idx = ev["nn_idx"][pi][perm]
        found = crop_index_to_leaf(idx, ev["crop_idx"], leafkey, leafcnt)
        LS[pi] = found
        GATE[pi] = ev["nn_d2"][pi][perm] < 0.01
    return D, G, F, LS, GATE, len(crop)


def crop_index_to_leaf(idx, crop_idx, leafkey, leafcnt):
    pos = {int(c): i for i, c in enumerate(crop_idx.tolist())}
    out = np.zeros(len(idx), np.int32)
    for n, i in enumerate(idx.tolist()):
        if i >= 0 and i in pos:
            out[n] = leafcnt[int(leafkey[pos[i]])]
        elif i >= 0:
            out[n] = leafcnt[int(leafkey[i])] if i < len(leafkey) else 1
    return out


# SIMD cycles per wave instruction block (round-2 measurements: 593 instructions / 1680 cycles per 64-query round)
C_SETUP = 210.0    # transform, key, in-box, margin pre-test, jump, path keys (straight line)
C_FAST = 52.0      # one key-following level
C_GEN = 230.0      # one exact 8-way level
C_LEAF0 = 30.0     # leaf start / end, reload of the winner
C_LEAF = 85.0      # one round of two candidates
C_COH = 300.0      # gate + both coherences + reciprocal (double)
C_ITEM = 60.0      # matrix load, wave sum, store, next-item atomic


def lockstep(G, F, LS, GATE, chunk=256):
    tot = 0.0
    Pn, M = G.shape
    for pi in range(Pn):
        for c0 in range(0, M, chunk):
            tot += C_ITEM
            for r0 in range(c0, c0 + chunk, 64):
                s = slice(r0, r0 + 64)
                tot += C_SETUP + C_FAST * F[pi, s].max() + C_GEN * G[pi, s].max()
                tot += C_LEAF0 + C_LEAF * np.ceil(LS[pi, s].max() / 2.0) + (C_COH if GATE[pi, s].any() else 0.0)
    return tot / (Pn * M / 64.0)


def refill(G, F, LS, GATE, chunk=256, t_gen=48, t_fill=16, t_leaf=48, c_ctl=12.0, c_park=20.0):
    """while-while: lanes hold one query; phases run under partial masks when their trigger fires"""
    tot = 0.0
    util_num = util_den = 0.0
    Pn, M = G.shape
    for pi in range(Pn):
        for c0 in range(0, M, chunk):
            tot += C_ITEM
            nxt, end = c0, c0 + chunk
            # lane state: -1 idle, else query id; rem generic levels; parked: query id awaiting leaf phase
            act = np.full(64, -1)
            rem = np.zeros(64, np.int32)
            parked = np.full(64, -1)
            while True:
                idle = (act < 0) & (parked < 0)
                n_idle = int(idle.sum())
                left = end - nxt
                in_desc = act >= 0
                n_desc = int(in_desc.sum())
                n_park = int((parked >= 0).sum())
                if left == 0 and n_desc == 0 and n_park == 0:
                    break
                tot += c_ctl
                # refill idle lanes
                if left > 0 and n_idle > 0 and (n_idle >= t_fill or n_desc == 0):
                    take = min(n_idle, left)
                    lanes = np.flatnonzero(idle)[:take]
                    ids = np.arange(nxt, nxt + take)
                    nxt += take
                    tot += C_SETUP + C_FAST * F[pi, ids].max()
                    act[lanes] = ids
                    rem[lanes] = G[pi, ids]
                    # queries with no generic level park at once
                    z = lanes[rem[lanes] == 0]
                    parked[z] = act[z]
                    act[z] = -1
                    tot += c_park
                    continue
                # leaf + coherence phase
                if n_park > 0 and (n_park >= t_leaf or (n_desc == 0 and (left == 0 or n_idle == 0)) or n_desc + n_park == 64 and n_desc < t_gen):
                    ids = parked[parked >= 0]
                    tot += C_LEAF0 + C_LEAF * np.ceil(LS[pi, ids].max() / 2.0) + (C_COH if GATE[pi, ids].any() else 0.0)
                    parked[:] = -1
                    continue
                if n_desc > 0:
                    tot += C_GEN
                    util_num += n_desc
                    util_den += 64
                    rem[in_desc] -= 1
                    done = in_desc & (rem == 0)
                    if done.any():
                        tot += c_park
                        # a lane whose parking slot is taken waits (stays in act with rem 0)
                        ok = done & (parked < 0)
                        parked[ok] = act[ok]
                        act[ok] = -1
                        blocked = done & ~ok
                        if blocked.any():  # force a leaf phase next
                            ids = parked[parked >= 0]
                            tot += C_LEAF0 + C_LEAF * np.ceil(LS[pi, ids].max() / 2.0) + (C_COH if GATE[pi, ids].any() else 0.0)
                            parked[:] = -1
                            parked[blocked] = act[blocked]
                            act[blocked] = -1
                    continue
                # nothing in descent, nothing to refill with the threshold: force
                if left > 0:
                    t_fill_now = 1
                    take = min(n_idle, left)
                    if take == 0:
                        ids = parked[parked >= 0]
                        tot += C_LEAF0 + C_LEAF * np.ceil(LS[pi, ids].max() / 2.0) + (C_COH if GATE[pi, ids].any() else 0.0)
                        parked[:] = -1
                        continue
                    lanes = np.flatnonzero(idle)[:take]
                    ids = np.arange(nxt, nxt + take)
                    nxt += take
                    tot += C_SETUP + C_FAST * F[pi, ids].max() + c_park
                    act[lanes] = ids
                    rem[lanes] = G[pi, ids]
                    z = lanes[rem[lanes] == 0]
                    parked[z] = act[z]
                    act[z] = -1
    return tot / (Pn * M / 64.0), util_num / max(util_den, 1)


def sorted_deal(G, F, LS, GATE, chunk=256, c_swap=25.0, c_redistribute=400.0):
    """full-width setup of the whole item; the queries are dealt to the lanes by their known generic count (largest
    first, snake order); every lane walks its own list; leaf + coherence full width afterwards"""
    tot = 0.0
    Pn, M = G.shape
    K = chunk // 64
    for pi in range(Pn):
        for c0 in range(0, M, chunk):
            tot += C_ITEM + c_redistribute
            g = G[pi, c0:c0 + chunk]
            for r0 in range(c0, c0 + chunk, 64):
                s = slice(r0, r0 + 64)
                tot += C_SETUP + C_FAST * F[pi, s].max()
                tot += C_LEAF0 + C_LEAF * np.ceil(LS[pi, s].max() / 2.0) + (C_COH if GATE[pi, s].any() else 0.0)
            order = np.argsort(-g, kind="stable")
            load = np.zeros(64, np.int32)
            for r in range(K):
                seg = order[r * 64:(r + 1) * 64]
                lanes = np.arange(64) if r % 2 == 0 else np.arange(63, -1, -1)
                load[lanes] += g[seg]
            steps = load.max()
            tot += steps * (C_GEN + c_swap)
    return tot / (Pn * M / 64.0)


def ideal(G, F, LS, GATE):
    Pn, M = G.shape
    per_round = C_ITEM / 4 + C_SETUP + C_FAST * F.mean() + C_GEN * G.mean() + C_LEAF0 + C_LEAF * np.ceil(LS / 2.0).mean() + C_COH * GATE.mean()
    return per_round


if __name__ == "__main__":
    t, model, cloud, parts = workload()
    D, G, F, LS, GATE, ncrop = per_query_work(t, model, cloud, parts)
    print("crop", ncrop, "depth", D, "generic mean %.2f" % G.mean(), "hist", np.bincount(G.ravel(), minlength=8)[:8] / G.size)
    wm = G.reshape(G.shape[0], -1, 64).max(2).mean()
    print("wave max generic %.2f  fast mean %.2f wave max %.2f  leaf mean %.2f wave max %.2f  gate %.3f" % (
        wm, F.mean(), F.reshape(F.shape[0], -1, 64).max(2).mean(), LS.mean(), LS.reshape(LS.shape[0], -1, 64).max(2).mean(), GATE.mean()))
    base = lockstep(G, F, LS, GATE)
    print("lockstep            cycles/round %.0f" % base)
    print("ideal (no divergence) %.0f  (%.3f)" % (ideal(G, F, LS, GATE), ideal(G, F, LS, GATE) / base))
    for chunk in (256, 512, 2048):
        for tg, tf, tl in ((48, 16, 48), (40, 24, 40), (32, 32, 32), (56, 8, 56), (48, 32, 32)):
            c, u = refill(G, F, LS, GATE, chunk=chunk, t_gen=tg, t_fill=tf, t_leaf=tl)
            print("refill chunk %4d gen>=%d fill>=%d leaf>=%d: %.0f (%.3f)  generic lane util %.2f" % (chunk, tg, tf, tl, c, c / base, u))
    for chunk in (256, 512):
        c = sorted_deal(G, F, LS, GATE, chunk=chunk)
        print("sorted deal chunk %d: %.0f (%.3f)" % (chunk, c, c / base))


def inlane_bound(G, chunk=256):
    """steps if every lane walks its own K queries back to back (no exchange): max over lanes of the lane's sum,
    against the lockstep sum of round maxima"""
    Pn, M = G.shape
    a = G.reshape(Pn, M // chunk, chunk // 64, 64)
    lock = a.max(3).sum(2).mean()
    own = a.sum(2).max(2).mean()
    perfect = np.ceil(a.sum((2, 3)) / 64.0).mean()
    return lock, own, perfect


if __name__ == "__main__":
    for chunk in (128, 256, 512, 2048):
        print("chunk %d: generic steps lockstep %.2f  in-lane queues %.2f  perfect %.2f" % ((chunk,) + inlane_bound(G, chunk)))
    wl = LS.reshape(LS.shape[0], -1, 64).max(2)
    print("wave max leaf size hist", np.bincount(wl.ravel(), minlength=12)[:12] / wl.size)
    wf = F.reshape(F.shape[0], -1, 64).max(2)
    print("wave max fast hist", np.bincount(wf.ravel(), minlength=8)[:8] / wf.size)
    wg = G.reshape(G.shape[0], -1, 64)
    mx = wg.max(2)
    print("lanes attaining the wave max: mean %.1f; rounds where <=4 lanes attain it: %.3f, <=8: %.3f, <=16: %.3f" % (
        (wg == mx[..., None]).sum(2).mean(), ((wg == mx[..., None]).sum(2) <= 4).mean(), ((wg == mx[..., None]).sum(2) <= 8).mean(), ((wg == mx[..., None]).sum(2) <= 16).mean()))


def low_lane_iterations(X, step=1):
    """iterations of a per-lane loop (trip count X per lane, `step` units per iteration) per 64-query round, and how many of
    them run with 1..8 active lanes (tools/micro/exec_mask_test.hip: such instructions cost 4-5 x on gfx950)"""
    W = X.reshape(X.shape[0], -1, 64)
    tot = low = 0
    for t in range(0, int(W.max()) + step, step):
        act = (W > t).sum(2)
        tot += (act > 0).sum()
        low += ((act > 0) & (act <= 8)).sum()
    n = W.shape[0] * W.shape[1]
    return tot / n, low / n


if __name__ == "__main__":
    for name, X, step in (("generic levels", G, 1), ("fast levels", F, 1), ("leaf rounds (2 candidates)", LS, 2)):
        tot, low = low_lane_iterations(X, step)
        print("%-28s iterations per round %.2f, of which with <= 8 active lanes %.2f" % (name, tot, low))
    print("coherence (gate) lanes active per round: mean %.1f; rounds with 1..8 lanes: %.3f" % (
        GATE.reshape(GATE.shape[0], -1, 64).sum(2).mean(), ((GATE.reshape(GATE.shape[0], -1, 64).sum(2) <= 8) & (GATE.reshape(GATE.shape[0], -1, 64).sum(2) > 0)).mean()))
