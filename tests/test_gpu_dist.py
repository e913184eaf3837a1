"""GPU checks of the particle-sharded path (pft_dist_* phases + pcl_tracking_amd/dist.py).

The 8-GPU run belongs to the driver; what can be verified on one MI355X:
  * world_size 1: the phase API reproduces pft_compute exactly;
  * world_size 2 rehearsal: two processes share cuda:0 and exchange through gloo (RCCL refuses two ranks on
    one device), so the sharded kernels run with id_offset != 0, P_local != P_total and a gathered
    population buffer; result must equal the single-handle tracker bit for bit, and the oracle within 1e-4.
"""
import os
import socket
import sys

import numpy as np
import pytest

from pcl_tracking_amd import scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("x", "y", "z", "roll", "pitch", "yaw")


def _data():
    return scene.make_model(1024), scene.make_scene(50000)[:20000]


def _single(P, frames, seed):
    from pcl_tracking_amd import tracker

    model, cloud = _data()
    t = tracker.make_reference_tracker(particle_num=P, seed=seed)
    t.setReferenceCloud(model)
    t.setTrans(scene.initial_trans())
    t.setInputCloud(cloud)
    out = []
    for f in range(frames):
        t.compute()
        out.append(t.getResult().copy())
    return out, t.getParticles()


def test_phase_api_world_size_one_equals_compute():
    import torch

    from pcl_tracking_amd.dist import HipPhases, ShardedFilter

    P, frames = 2048, 3
    want, want_p = _single(P, frames, 5)
    model, cloud = _data()
    ph = HipPhases(P, 0, 1, torch.device("cuda", 0), seed=5)
    ph.set_reference(model)
    ph.set_trans(scene.initial_trans())
    ph.set_input(cloud)
    sf = ShardedFilter(ph)
    for f in range(frames):
        sf.compute()
        assert sf.getResult().tobytes() == want[f].tobytes()
    np.testing.assert_array_equal(sf.getParticles().view(np.float32), want_p.view(np.float32))


def _worker(rank, world, port, outdir, P, frames, seed):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pcl_tracking_amd.dist import HipPhases, ShardedFilter

    model, cloud = _data()
    ph = HipPhases(P, rank, world, torch.device("cuda", 0), seed=seed)
    ph.set_reference(model)
    ph.set_trans(scene.initial_trans())
    ph.set_input(cloud)
    sf = ShardedFilter(ph)
    res = []
    for f in range(frames):
        sf.compute()
        res.append(np.frombuffer(sf.getResult().tobytes(), np.float32).copy())
    np.save(os.path.join(outdir, "res%d.npy" % rank), np.stack(res))
    np.save(os.path.join(outdir, "part%d.npy" % rank), sf.getParticles().view(np.float32).reshape(-1, 8))
    dist.barrier()
    dist.destroy_process_group()


def test_config3_world8_65536_particles_rehearsed_in_one_process(orc):
    """BASELINE configs[3] at full size -- 65 536 particles sharded 8 192 per rank over 8 ranks, 50 000-point cloud --
    on ONE GPU.  The box admits at most 6 processes on its card, so the eight ranks are eight handles of this one
    process (pft_config.rank = 0..7, world_size = 8) and the two exchange steps are done by hand on the device
    (element-wise max of the eight bbox6 buffers, concatenation of the eight shards): exactly what all-reduce(MAX) and
    all-gather deliver.  Every rank must reproduce the single 65 536-particle handle bit for bit; raw weights of the
    gathered population are spot-checked against the oracle inside the global crop box."""
    import ctypes as C

    import torch

    from pcl_tracking_amd import tracker
    from pcl_tracking_amd.dist import HipPhases

    P, world, frames, seed = 65536, 8, 2, 1
    model, cloud = scene.make_model(2048), scene.make_scene(50000)
    dev = torch.device("cuda", 0)
    single = tracker.make_reference_tracker(particle_num=P, seed=seed)
    single.setReferenceCloud(model)
    single.setTrans(scene.initial_trans())
    phs = [HipPhases(P, r, world, dev, seed=seed) for r in range(world)]
    for ph in phs:
        ph.set_reference(model)
        ph.set_trans(scene.initial_trans())
    raw_gathered = None
    for f in range(frames):
        single.setInputCloud(cloud)
        single.compute()
        want = single.getResult().tobytes()
        for ph in phs:
            ph.set_input(cloud)
            ph.begin_frame()
        for it in range(2):
            for ph in phs:
                ph.phase_a(it)
            bb = torch.stack([ph.bbox6 for ph in phs]).max(0).values  # all-reduce(MAX)
            for ph in phs:
                ph.bbox6.copy_(bb)
                ph.phase_b()
            g = torch.cat([ph.shard for ph in phs])  # all-gather, rank order
            raw_gathered = g.clone()
            for ph in phs:
                ph.gathered.copy_(g)
                ph.phase_c()
        for r, ph in enumerate(phs):
            assert ph.get_result().tobytes() == want, (f, r)
    want_p = single.getParticles().view(np.float32).reshape(-1, 8)
    for r, ph in enumerate(phs):
        np.testing.assert_array_equal(ph.get_particles().view(np.float32).reshape(-1, 8), want_p, err_msg="rank %d" % r)
    # oracle spot check: raw likelihoods of the last iteration's gathered population (before normalisation), a few
    # particles from every shard, inside the crop box of the whole population
    pop = raw_gathered.cpu().numpy().view(scene.PARTICLE_DTYPE)
    bbox = np.zeros(6, np.float32)
    t0 = phs[0].t
    t0._check(t0._L.pft_debug_get_bbox(t0._h, bbox.ctypes.data_as(C.c_void_p)))
    pick = np.concatenate([r * (P // world) + np.random.default_rng(r).choice(P // world, 3, replace=False)
                           for r in range(world)])
    o = orc.Tracker(orc.default_config(particle_num=len(pick), threads=0, emulate_pcl_alloc=0))
    o.set_reference(model)
    o.set_trans(scene.initial_trans())
    o.set_input(cloud)
    O = o.eval_weights(pop[pick], want_nn=False, mats=t0.debugPoseToMatrix(pop[pick]), bbox=bbox.astype(np.float64))
    a = np.ascontiguousarray(pop["weight"][pick]).view(np.int32).astype(np.int64)
    b = O["raw"].view(np.int32).astype(np.int64)
    assert (O["raw"] < 0).any() and np.abs(a - b).max() <= 1, (pop["weight"][pick], O["raw"])


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_rehearsal_on_one_gpu(tmp_path, orc, world):
    import torch.multiprocessing as mp

    P, frames, seed = 2048, 3, 9
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path), P, frames, seed), nprocs=world, join=True)
    want, want_p = _single(P, frames, seed)
    for r in range(world):
        got = np.load(tmp_path / ("res%d.npy" % r))
        for f in range(frames):
            assert got[f].tobytes() == want[f].tobytes(), (r, f)
        np.testing.assert_array_equal(np.load(tmp_path / ("part%d.npy" % r)),
                                      want_p.view(np.float32).reshape(-1, 8))
    # and against the CPU oracle
    model, cloud = _data()
    o = orc.Tracker(orc.default_config(particle_num=P, seed=seed, threads=0, emulate_pcl_alloc=0))
    o.set_reference(model)
    o.set_trans(scene.initial_trans())
    o.set_input(cloud)
    for f in range(frames):
        o.compute()
        ro = o.get_result()
        for k in KEYS:
            assert abs(float(want[f][k]) - float(ro[k])) < 1e-4


@pytest.mark.timeout(900)
def test_bench_launches_its_ranks_itself(tmp_path):
    """`python bench.py --gpus N` as the driver invokes it, without a launcher: it must start N ranks itself or fail
    loudly, never run one rank and call it N.  On this one-GPU box: --gpus 2 is an error exit; with the rehearsal switch
    (both ranks on cuda:0, gloo) the whole N-rank control flow runs -- self-spawn through torch.distributed.run, sharded
    frames with their two collectives, the replayed checkpoint, max-over-ranks timing -- and reports the ranks that took
    part."""
    import json
    import subprocess

    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PFT_BENCH_SHARE_GPU"):
        env.pop(k, None)
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "3", "--warmup", "1"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr, (r.returncode, r.stderr[-500:])
    assert "n_gpus" not in r.stdout
    # a launcher that disagrees with --gpus is an error as well
    env1 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], capture_output=True, text=True, env=env1, timeout=300)
    assert r.returncode == 2
    env["PFT_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "4", "--warmup", "2"], capture_output=True, text=True,
                       env=env, timeout=800)
    assert r.returncode == 0, r.stderr[-1500:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["particles_total"] == 16384
    assert "sharded x2" in d["config"]["parallelism"] and "REHEARSAL" in d["data"]
    assert d["ms_per_step"] > 0 and d["cropped_points"] > 1000
