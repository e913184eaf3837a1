"""Host logic of the particle-sharded path (pcl_tracking_amd/dist.py ShardedFilter) under gloo on CPU,
world_size 2 and 4: sharding by global particle id, the MAX all-reduce of the AABB, the all-gather of
(particle, raw weight) shards in rank order, RNG keyed by global id.

There is no GPU here, so the per-rank stages are played by a stand-in built from the oracle's stage
functions (test infrastructure; the product's stages are the HIP kernels behind pft_dist_*).  The check is
exact: two ranks must reproduce, bit for bit, what the single-process oracle tracker computes."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OraclePhases:
    """CPU stand-in for HipPhases with the same attributes and phase methods"""

    def __init__(self, orc, cfg, model, cloud, trans, rank, world):
        from pcl_tracking_amd import scene

        self.orc, self.cfg, self.scene = orc, cfg, scene
        self.P = cfg.particle_num
        self.P_local = self.P // world
        self.offset = rank * self.P_local
        self.iteration_num = cfg.iteration_num
        self.tr = orc.Tracker(cfg)
        self.tr.set_reference(model)
        self.tr.set_trans(trans)
        self.tr.set_input(cloud)
        self.trans = trans
        self.bbox6 = torch.zeros(6, dtype=torch.float32)
        self.shard = torch.zeros(self.P_local * 8, dtype=torch.float32)
        self.gathered = torch.zeros(self.P * 8, dtype=torch.float32)
        self.local = None
        self.changed = False
        self.epoch = 0

    def begin_frame(self):
        if self.local is None:
            rep = np.zeros(1, self.scene.PARTICLE_DTYPE)
            rep[0] = self.orc.to_state(self.trans)
            rep["weight"] = np.float32(1.0) / np.float32(self.P)
            self.rep = rep
            self.local = self.orc.init_particles(self.cfg, rep, self.offset, self.P_local)

    def phase_a(self, it):
        if self.changed:
            self.local = self.orc.resample(self.cfg, self.all, self.a, self.q, self.rep, self.epoch, self.offset,
                                           self.P_local)
            self.epoch += 1
        b = self.tr.bbox_of(self.local)  # x_min,x_max,y_min,y_max,z_min,z_max (float values held in doubles)
        self.bbox6.copy_(torch.tensor([-b[0], -b[2], -b[4], b[1], b[3], b[5]], dtype=torch.float32))

    def phase_b(self):
        g = self.bbox6.numpy().astype(np.float64)
        box = np.array([-g[0], g[3], -g[1], g[4], -g[2], g[5]])
        ev = self.tr.eval_weights(self.local, bbox=box)
        self.local["weight"] = ev["raw"]
        self.shard.copy_(torch.from_numpy(self.local.view(np.float32).reshape(-1).copy()))

    def phase_c(self):
        allp = self.gathered.numpy().copy().view(self.scene.PARTICLE_DTYPE)
        w, self.fit = self.orc.normalize_weights(allp["weight"])
        allp["weight"] = w
        rep = np.zeros(1, self.scene.PARTICLE_DTYPE)
        rep[0] = self.orc.weighted_mean(allp)
        self.rep = rep
        self.a, self.q = self.orc.gen_alias_table(w)
        self.all = allp
        self.changed = True

    def get_result(self):
        return self.rep[0]

    def get_particles(self):
        return self.all


def _worker(rank, world, port, outdir, P, frames):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as orc

    from pcl_tracking_amd import scene
    from pcl_tracking_amd.dist import ShardedFilter

    model = scene.make_model(256)
    cloud = scene.make_scene(50000)[:6000]
    cfg = orc.default_config(particle_num=P, seed=11, threads=1, emulate_pcl_alloc=0)
    ph = OraclePhases(orc, cfg, model, cloud, scene.initial_trans(), rank, world)
    sf = ShardedFilter(ph)
    res = []
    for f in range(frames):
        sf.compute()
        res.append(np.array(sf.getResult().tolist(), np.float32))
    np.save(os.path.join(outdir, "res%d.npy" % rank), np.stack(res))
    np.save(os.path.join(outdir, "part%d.npy" % rank), sf.getParticles().view(np.float32).reshape(-1, 8))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4])
def test_ranks_reproduce_the_single_process_tracker(tmp_path, orc, world):
    from pcl_tracking_amd import scene

    P, frames = 64, 3
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), P, frames), nprocs=world, join=True)
    r0 = np.load(tmp_path / "res0.npy")
    for r in range(1, world):  # replicated population stages agree on every rank
        np.testing.assert_array_equal(r0, np.load(tmp_path / ("res%d.npy" % r)))
        np.testing.assert_array_equal(np.load(tmp_path / "part0.npy"), np.load(tmp_path / ("part%d.npy" % r)))
    # single process, same seed
    model = scene.make_model(256)
    cloud = scene.make_scene(50000)[:6000]
    t = orc.Tracker(orc.default_config(particle_num=P, seed=11, threads=1, emulate_pcl_alloc=0))
    t.set_reference(model)
    t.set_trans(scene.initial_trans())
    t.set_input(cloud)
    for f in range(frames):
        assert t.compute() == 0
        want = np.array(t.get_result().tolist(), np.float32)
        np.testing.assert_array_equal(r0[f], want)
    np.testing.assert_array_equal(np.load(tmp_path / "part0.npy"),
                                  t.get_particles().view(np.float32).reshape(-1, 8))


def test_sharded_filter_world_size_one(orc):
    """without a process group the same driver runs a single shard (used by the GPU dist-vs-compute test)"""
    from pcl_tracking_amd import scene
    from pcl_tracking_amd.dist import ShardedFilter

    model = scene.make_model(128)
    cloud = scene.make_scene(50000)[:3000]
    cfg = orc.default_config(particle_num=32, seed=3, threads=1, emulate_pcl_alloc=0)
    ph = OraclePhases(orc, cfg, model, cloud, scene.initial_trans(), 0, 1)
    sf = ShardedFilter(ph)
    t = orc.Tracker(cfg)
    t.set_reference(model)
    t.set_trans(scene.initial_trans())
    t.set_input(cloud)
    for f in range(2):
        sf.compute()
        t.compute()
        assert sf.getResult().tobytes() == t.get_result().tobytes()
