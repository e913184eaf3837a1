"""Particle-sharded tracking across the GPUs of one node (one process per GPU, torch.distributed;
backend "nccl" is RCCL on ROCm).

The reference has no distributed path: particles are independent in the transform and
likelihood stages, so rank r owns global particle ids [r*P/W, (r+1)*P/W).  Per iteration there are
exactly two exchange steps, both tiny and latency-bound:
  1. all-reduce(MAX) of 6 floats {-min xyz, max xyz}: the crop box is the AABB over ALL particles'
     transformed reference clouds, and every rank must build the identical octree from it;
  2. all-gather of the shard (32 B per particle, raw likelihood in .weight): normalisation, the
     weighted mean and the alias table need the whole population and are recomputed redundantly
     (bit-identically) on every rank; the counter-based RNG is keyed by the GLOBAL particle id, so the
     result does not depend on the number of ranks.
`ShardedFilter` is the host logic; the per-rank compute is behind `phases` (HipPhases = the C ABI;
tests inject a CPU stand-in to exercise the collectives under gloo)."""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from ._lib import PftError
from .scene import PARTICLE_DTYPE, POINT_DTYPE
from .tracker import make_reference_tracker


class HipPhases:
    """per-rank stages of include/pft.h's pft_dist_* API, enqueued on torch's current HIP stream"""

    def __init__(self, particle_num_total, rank, world_size, device, seed=1, iteration_num=2):
        assert particle_num_total % world_size == 0
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.P, self.rank, self.world = particle_num_total, rank, world_size
        self.P_local = particle_num_total // world_size
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.t = make_reference_tracker(particle_num=particle_num_total, seed=seed, device_id=self.device.index or 0,
                                        stream=stream, rank=rank, world_size=world_size)
        self.t.setIterationNum(iteration_num)
        self.iteration_num = iteration_num
        # exchange buffers are torch tensors so the collectives see ordinary device memory
        self.bbox6 = torch.zeros(6, dtype=torch.float32, device=self.device)
        self.shard = torch.zeros(self.P_local * 8, dtype=torch.float32, device=self.device)
        self.gathered = torch.zeros(self.P * 8, dtype=torch.float32, device=self.device)
        self._bound = False

    def _L(self):
        return self.t._L

    def _bind(self):
        if not self._bound:
            self.t._ensure()
            self.t._check(self._L().pft_dist_bind(self.t._h, C.c_void_p(self.bbox6.data_ptr()),
                                                  C.c_void_p(self.shard.data_ptr()),
                                                  C.c_void_p(self.gathered.data_ptr())))
            self._bound = True

    def set_reference(self, cloud):
        self.t.setReferenceCloud(cloud)

    def set_trans(self, m):
        self.t.setTrans(m)

    def set_input(self, cloud):
        self.t.setInputCloud(cloud)

    def set_input_device(self, tensor, n):
        self.t.setInputCloudDevice(tensor.data_ptr(), n, keepalive=tensor)

    def begin_frame(self):
        self._bind()
        self.t._check(self._L().pft_dist_begin_frame(self.t._h))

    def phase_a(self, it):
        self.t._check(self._L().pft_dist_phase_a(self.t._h, it))

    def phase_b(self):
        self.t._check(self._L().pft_dist_phase_b(self.t._h))

    def phase_c(self):
        self.t._check(self._L().pft_dist_phase_c(self.t._h))

    def get_result(self):
        return self.t.getResult()

    def get_particles(self):
        torch.cuda.synchronize(self.device)
        return self.gathered.cpu().numpy().view(PARTICLE_DTYPE).copy()


class ShardedFilter:
    """A12 schedule with the two collectives in place; `phases` supplies the per-rank stages and the
    exchange buffers (bbox6, shard, gathered: torch tensors on the phases' device)."""

    def __init__(self, phases, group=None):
        self.p = phases
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # rehearsal on one GPU: run the collectives even with a single rank (exercises the RCCL calls in stream order)
        import os

        self.force = dist.is_initialized() and os.environ.get("PFT_DIST_FORCE_COLLECTIVES") == "1"

        # RCCL ("nccl") takes device tensors directly.  Under gloo (CPU tests, or several ranks sharing one
        # GPU in the single-GPU rehearsal of tests/test_gpu_dist.py) device tensors are staged through the host.
        self.stage = dist.is_initialized() and dist.get_backend(group) == "gloo" and phases.bbox6.is_cuda

    def _all_reduce_max(self, t):
        if self.stage:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)

    def _all_gather(self, out, shard):
        if self.stage:
            ho, hs = torch.empty(out.shape, dtype=out.dtype), shard.cpu()
            dist.all_gather_into_tensor(ho, hs, group=self.group)
            out.copy_(ho)
        else:
            dist.all_gather_into_tensor(out, shard, group=self.group)

    def compute(self):
        p = self.p
        p.begin_frame()
        for it in range(p.iteration_num):
            p.phase_a(it)
            if self.world > 1 or self.force:
                self._all_reduce_max(p.bbox6)
            p.phase_b()
            if self.world > 1 or self.force:
                self._all_gather(p.gathered, p.shard)
            else:
                p.gathered.copy_(p.shard)
            p.phase_c()

    def getResult(self):
        return self.p.get_result()

    def getParticles(self):
        return self.p.get_particles()
