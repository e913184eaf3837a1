#!/usr/bin/env python3
"""bench.py -- headline benchmark of the tracking hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one tracked frame = one pft_compute(): iteration_num x [resample, weight, update] over one
batch of synthetic input (BASELINE.json configs[1]: 2 048-point model vs 50 000-point cloud, 8 192
particles per GPU, single frame looped, filter left running).  Inputs are resident in HBM when the
timed region starts.  Rank 0 prints ONE JSON line.  value = P_total * N_points / t_frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

P_PER_GPU = 8192
M_MODEL = 2048
N_CLOUD = 50000
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles-per-gpu", type=int, default=P_PER_GPU)
    ap.add_argument("--model-points", type=int, default=M_MODEL)
    ap.add_argument("--cloud-points", type=int, default=N_CLOUD)
    ap.add_argument("--organized", action="store_true", help="N = w*h depth image, no downsample (config 3)")
    ap.add_argument("--objects", type=int, default=1, help="independent trackers, one HIP stream each (configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frontend", action="store_true", help="skip the PassThrough + ApproximateVoxelGrid side measurement")
    ap.add_argument("--cpu-frames", type=int, default=3)
    return ap.parse_args()


def cpu_baseline(model, cloud, trans, P, threads, pcl_alloc=1):
    """the oracle (CPU restatement of the PCL OMP path, PCL-structured: per-particle clouds materialised,
    pointer octree rebuilt per iteration, per-query index-vector allocation) timed on the host cores.
    Test infrastructure used here only as the reported baseline."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle

    cfg = oracle.default_config(particle_num=P, threads=threads, emulate_pcl_alloc=pcl_alloc, seed=1)
    tr = oracle.Tracker(cfg)
    tr.set_reference(model)
    tr.set_trans(trans)
    tr.set_input(cloud)
    tr.compute()  # warm-up frame (includes initParticles)
    times = []
    stages = None
    for _ in range(max(1, ARGS.cpu_frames)):
        t0 = time.perf_counter()
        tr.compute()
        times.append(time.perf_counter() - t0)
        stages = tr.stage_times()
    return min(times), sorted(times)[len(times) // 2], stages


def frontend_measurement(dev, with_cpu):
    """SURVEY 8f row 1, reported beside the headline (never part of `value`): the reference's per-frame input
    filters (auto_tracking.cpp:637 filterPassThrough, :683 gridSampleApprox) over a synthetic Kinect2 qhd frame
    (960x540, :775) resident in HBM; CPU figure = the oracle's sequential restatement, one core."""
    import numpy as np
    import torch

    from pcl_tracking_amd import filters, scene

    frame = scene.make_depth_frame(960, 540)
    d = torch.from_numpy(frame.view(np.uint8).reshape(-1).copy()).to(dev)
    f = filters.make_reference_input_filter(device_id=dev.index or 0)
    f.setInputCloudDevice(d.data_ptr(), len(frame), keepalive=d)
    f.filterDevice()
    ms, wall = [], []
    for _ in range(50):
        t0 = time.perf_counter()
        f.filterDevice()
        wall.append((time.perf_counter() - t0) * 1e3)
        ms.append(f.lastMilliseconds())
    n_pass, n_out = f.counts()
    out = {"workload": "PassThrough z in [0,10] + ApproximateVoxelGrid(0.01, 512-entry table) on a 960x540 frame",
           "points_in": len(frame), "points_pass": n_pass, "points_out": n_out,
           "gpu_ms": float(np.median(ms)), "wall_ms_incl_sync": float(np.median(wall)),
           "points_per_s": len(frame) / (float(np.median(ms)) * 1e-3)}
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle

        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            oracle.approx_voxel_grid(frame[oracle.pass_through(frame, "z", 0.0, 10.0)], 0.01, 512)
            ts.append(time.perf_counter() - t0)
        out["cpu_port_ms"] = min(ts) * 1e3
    return out


def reference_operating_point(model, cloud, trans, dev, with_cpu):
    """BASELINE configs[0], reported beside the headline: the reference's own settings (auto_tracking.cpp:201-254) --
    400 particles fixed (use_fixed) and the KLD-adaptive tracker it runs by default (<= 500 particles) -- on the same
    model / cloud; CPU figure = the oracle with the reference's 16 OpenMP threads (:845)."""
    from pcl_tracking_amd import tracker

    out = {}
    for name, kld in (("fixed_400", False), ("kld_adaptive_500", True)):
        t = tracker.make_reference_tracker(particle_num=400, seed=1, kld=kld, device_id=dev.index or 0)
        t.setReferenceCloud(model)
        t.setTrans(trans)
        t.setInputCloud(cloud)
        for _ in range(20):
            t.compute()
        t.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            t.compute()
        t.synchronize()
        ms = (time.perf_counter() - t0) / 300 * 1e3
        out[name] = {"gpu_ms_per_frame": ms, "gpu_frames_per_s": 1e3 / ms, "particles_after": int(len(t.getParticles()))}
        if with_cpu:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle

            o = oracle.Tracker(oracle.default_config(particle_num=400, threads=16, emulate_pcl_alloc=1, seed=1,
                                                     kld_adaptive=int(kld)))
            o.set_reference(model)
            o.set_trans(trans)
            o.set_input(cloud)
            o.compute()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                o.compute()
                ts.append(time.perf_counter() - t0)
            out[name]["cpu_port_ms_per_frame_16_threads"] = min(ts) * 1e3
    return out


def main():
    import numpy as np
    import torch

    from pcl_tracking_amd import build as _hip_build
    from pcl_tracking_amd import scene

    if int(os.environ.get("WORLD_SIZE", "1")) == 1:
        try:  # no-op when pcl_tracking_amd/_build/libpft_hip.so is current (it ships with the snapshot); multi-rank
            _hip_build.build()  # launches rely on the shipped library (no rebuild under the other ranks' feet)
        except Exception as e:  # a shipped library is still usable if the rebuild is not possible here
            sys.stderr.write("bench: rebuild skipped (%s)\n" % e)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if ARGS.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (ARGS.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = os.environ.get("PFT_DIST_FORCE_COLLECTIVES") == "1" and "RANK" in os.environ  # 1-rank rehearsal
    if world > 1 or force_dist:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=dev)

    P_local = ARGS.particles_per_gpu
    P_total = P_local * world
    M, N = ARGS.model_points, ARGS.cloud_points
    model = scene.make_model(M)
    cloud = scene.make_scene(N, mode="organized" if ARGS.organized else "voxel")
    trans = scene.initial_trans()
    cloud_dev = torch.from_numpy(cloud.view(np.uint8).reshape(-1).copy()).to(dev)  # PCL 32-B layout in HBM

    def sync():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if world == 1 and not force_dist:
        from pcl_tracking_amd import tracker

        # independent objects are independent handles, each on its own HIP stream (the reference tracks them in
        # a sequential loop, auto_tracking.cpp:688-697)
        ts = []
        for k in range(ARGS.objects):
            t = tracker.make_reference_tracker(particle_num=P_total, seed=1 + k, device_id=local_rank)
            t.setReferenceCloud(model)
            t.setTrans(trans)
            ts.append(t)

        def step():
            for t in ts:
                t.setInputCloudDevice(cloud_dev.data_ptr(), N, keepalive=cloud_dev)
                t.compute()

        trk = ts[0]
    else:
        from pcl_tracking_amd.dist import HipPhases, ShardedFilter

        ph = HipPhases(P_total, rank, world, dev, seed=1)
        ph.set_reference(model)
        ph.set_trans(trans)
        sf = ShardedFilter(ph)

        def step():
            ph.set_input_device(cloud_dev, N)
            sf.compute()

        trk = ph.t

    sync()
    tc = time.perf_counter()
    step()  # cold: buffer growth, initParticles, first octree
    sync()
    cold_ms = (time.perf_counter() - tc) * 1e3
    for _ in range(ARGS.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(ARGS.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    ms_per_step = dt / ARGS.steps * 1e3

    # ---- per-kernel durations: HIP events on the kernels' own stream, same loop again ----
    trk.profileEnable(True)
    trk.profileReset()
    sync()
    t1 = time.perf_counter()
    for _ in range(ARGS.steps):
        step()
    sync()
    dt_prof = time.perf_counter() - t1
    prof = trk.profileGet()
    trk.profileEnable(False)
    lik_ms, lik_n = prof["likelihood"]
    lik_avg_s = lik_ms / max(1, lik_n) * 1e-3

    out = None
    if rank == 0:
        # mean leaf occupancy of the steady-state workload (counted on the device, outside the timed region)
        pcur = trk.getParticles()[:P_local]
        st = trk.evalWeights(pcur, want_nn=True)
        kbar = st["scan_points"] / max(1, st["scan_queries"])
        # algorithmic bytes: per pair-eval 16 B reference point + 16 B per candidate scanned
        bytes_per_launch = P_local * M * 16.0 * (1.0 + kbar)
        achieved = bytes_per_launch / lik_avg_s / 1e9 if lik_avg_s > 0 else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                key = "P%d_M%d_N%d" % (P_local, M, N)
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        pips = ARGS.objects * P_total * N / (dt / ARGS.steps)
        out = {
            "metric": "particles x input-points / sec per frame (tracked frames/sec @8192 particles per GPU)",
            "value": pips,
            "unit": "particle-points/s",
            "n_gpus": world,
            "steps": ARGS.steps,
            "warmup": ARGS.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: %d-pt model vs %d-pt %s cloud, %d particles/GPU, 2 iterations/frame, "
                            "single frame looped" % (M, N, "organized" if ARGS.organized else "voxel-downsampled", P_local),
                "particles_total": P_total, "model_points": M, "cloud_points": N, "iterations_per_frame": 2,
                "parallelism": "particles sharded x%d" % world, "objects": ARGS.objects,
            },
            "frames_per_s": ARGS.steps / dt,
            "cold_first_frame_ms": cold_ms,
            "pair_evals_per_s": 2.0 * P_total * M / (dt / ARGS.steps),
            "cropped_points": int(len(st["crop_idx"])), "octree_depth": int(st["octree_depth"]),
            "mean_leaf_occupancy": kbar,
            "ms_per_step_with_events": dt_prof / ARGS.steps * 1e3,
            "kernel_ms_per_frame": {k: round(v[0] / ARGS.steps, 5) for k, v in prof.items()},
            "roofline": {
                "bound": "hbm", "kernel": "k_likelihood", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_us": lik_avg_s * 1e6,
                "launches": int(lik_n),
            },
        }
        if P_local == P_PER_GPU and M == M_MODEL and N == N_CLOUD and not ARGS.organized:
            # interpretation aid (SURVEY 8d "secondary (VALU) roof"): the kernel's working set is LDS/L2-resident and it
            # is VALU-issue bound.  Instruction count per launch from the committed PMC pass of this very workload
            # (profiles/r01_pmc_sq_counters.txt: SQ_INSTS_VALU of k_likelihood<false>); the duration is the live one.
            valu_insts = 1.572e8
            # Measured on this part (tools/micro/valu_rate_test.hip): a wave64 VALU instruction occupies its SIMD for
            # 2.5-2.7 cycles if it is a plain add / sub / mul / and / or, 4.1-4.4 cycles otherwise (compares, selects,
            # shifts, bit-field, fused and three-operand forms: two thirds of this kernel's child selection).  The
            # figure below is the SIMD time the launch spends per instruction: between those two costs = the VALU
            # pipes are busy for essentially the whole launch.
            out["roofline"]["valu_issue"] = {
                "wave_insts_per_launch": valu_insts, "wave_insts_per_s": valu_insts / lik_avg_s,
                "simd_cycles_per_wave_inst": lik_avg_s * 2.4e9 * 256 * 4 / valu_insts,
                "measured_cost_cycles": {"add_sub_mul_and_or": 2.5, "compare_select_shift_bitfield_fma": 4.2},
                "source": "profiles/r01_pmc_sq_counters.txt (SQ_INSTS_VALU of this workload); duration measured live"}
        if world == 1 and not ARGS.no_cpu_baseline:
            cores = os.cpu_count() or 1
            tmin, tmed, stages = cpu_baseline(model, cloud, trans, P_total, cores)
            out["cpu_baseline"] = {
                "value": P_total * N / tmin, "unit": "particle-points/s", "cores": cores, "kind": "port",
                "sample": "%d frames of the same workload (P=%d, M=%d, N=%d, 2 iterations) after 1 warm-up frame; "
                          "min frame time %.3f s, median %.3f s" % (ARGS.cpu_frames, P_total, M, N, tmin, tmed),
                "stage_seconds": {k: round(float(v), 4) for k, v in zip(
                    ("transform", "bbox_crop", "octree", "coherence", "normalize", "resample", "update"), stages)},
            }
            out["speedup_vs_cpu"] = pips / out["cpu_baseline"]["value"]
            # the same port without PCL's per-query index-vector allocation ("optimised CPU"): so that
            # the ratio is not credited to de-pessimising the allocator behaviour alone
            fmin, fmed, _ = cpu_baseline(model, cloud, trans, P_total, cores, pcl_alloc=0)
            out["cpu_optimised"] = {"value": P_total * N / fmin, "unit": "particle-points/s", "cores": cores,
                                    "kind": "port", "sample": "same, emulate_pcl_alloc=0; min frame time %.3f s" % fmin}
            out["speedup_vs_cpu_optimised"] = pips / out["cpu_optimised"]["value"]
        if world == 1 and not ARGS.no_frontend:
            out["frontend"] = frontend_measurement(dev, not ARGS.no_cpu_baseline)
            out["reference_operating_point"] = reference_operating_point(model, cloud, trans, dev, not ARGS.no_cpu_baseline)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    ARGS = parse()
    main()
