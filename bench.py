#!/usr/bin/env python3
"""bench.py -- headline benchmark of the tracking hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Under a launcher (RANK / WORLD_SIZE in the environment, e.g.
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) this process is one rank; invoked
directly with --gpus N > 1 it starts the N ranks itself (torch.distributed.run as a child process, before anything
here touches a GPU), relays their one JSON line and exits with their code.  It never falls back to fewer ranks:
a mismatch between --gpus, WORLD_SIZE and the visible GPUs is an error exit.

A "step" is one tracked frame = one pft_compute(): iteration_num x [resample, weight, update] over one batch of
synthetic input (BASELINE.json configs[1]: 2 048-point model vs 50 000-point cloud, 8 192 particles per GPU, single
frame looped).  The headline workload is STATIONARY: after the warm-up frames the filter state is checkpointed in
HBM and every timed step replays the same frame from it (one small restore kernel per step, inside the timed
region), so the number does not depend on --steps / --warmup; the free-running filter is timed beside it
("running").  Inputs are resident in HBM when the timed region starts.  Rank 0 prints ONE JSON line.
value = P_total * N_points / t_frame.
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
# same guide, "CU" section + constants table: 256 CUs x 4 SIMD-32; a wave64 VALU instruction issues over 2 cycles at
# 2.4 GHz => 1024 x 2.4e9 / 2 wave-instructions per second, chip-wide
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 2.0


PREROLL_FRAMES = 20  # age of the filter (frames) at which the replayed frame is checkpointed, at least


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
    ap.add_argument("--cpu-frames", type=int, default=20, help="CPU baseline: timed frames wanted (SURVEY 8d: >= 20)")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="CPU baseline: stop early after this much timed work")
    ap.add_argument("--running", action="store_true", help="headline = the free-running filter instead of the replayed frame")
    ap.add_argument("--repeats", type=int, default=7, help="the timed loop of exactly --steps steps is run this many times; "
                                                           "ms_per_step is the median repetition")
    return ap.parse_args()


def workload_label(P_local, M, N, organized, objects, world):
    """which BASELINE.json configuration this invocation is (the headline metric is quoted on configs[1])"""
    shape = "%d-pt model vs %d-pt %s cloud, %d particles/GPU, 2 iterations/frame, single frame looped" % (
        M, N, "organized (no downsample)" if organized else "voxel-downsampled", P_local)
    if objects == 1 and M == M_MODEL and not organized and N == N_CLOUD and P_local == P_PER_GPU:
        if world == 1:
            return "BASELINE configs[1]: " + shape
        if world == 8:
            return "BASELINE configs[3]: %d particles sharded %d/GPU over %d GPUs, " % (P_local * world, P_local, world) + shape
        return "BASELINE configs[3] shape at %d GPUs (%d particles sharded %d/GPU): " % (world, P_local * world, P_local) + shape
    if objects == 1 and organized and N == 307200 and P_local == 16384 and world == 1:
        return "BASELINE configs[2]: " + shape
    if objects == 4 and N == 200000 and P_local == P_PER_GPU and not organized:
        return "BASELINE configs[4]%s: 4 independent model clouds, one handle + HIP stream each, shared cloud; %s" % (
            "" if world == 8 else " shape on %d GPU(s)" % world, shape)
    return "custom (not a BASELINE configuration): %d object(s), %s" % (objects, shape)


def visible_gpu_count():
    """GPUs this process may use, WITHOUT loading the HIP runtime or torch (the launcher parent must not touch the GPU
    stack): the KFD topology in sysfs lists one node per agent (CPUs have simd_count 0), narrowed by the
    *_VISIBLE_DEVICES variables.  Falls back to asking a child process when sysfs is not there."""
    n = None
    top = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(top):
            with open(os.path.join(top, node, "properties")) as f:
                props = dict(l.split()[:2] for l in f if len(l.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except (OSError, ValueError):
        n = None
    if n is None:
        import subprocess

        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                           capture_output=True, text=True)
        try:
            return int(r.stdout.strip().splitlines()[-1])
        except (ValueError, IndexError):
            return 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def usable_cpus():
    """(threads this process can really run at once, cpu_count, affinity, cgroup quota in CPUs or None).  os.cpu_count()
    ignores both the affinity mask and the cgroup CPU quota: 256 OpenMP threads on the 8 CPUs a container is granted
    is what round 2's baseline measured."""
    import math

    cpu_count = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = cpu_count
    quota = None
    try:  # cgroup v2
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:  # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            quota = None
    usable = affinity if quota is None else max(1, min(affinity, int(math.ceil(quota))))
    return usable, cpu_count, affinity, quota


def spawn_ranks_if_needed():
    """`python bench.py --gpus N` with N > 1 and no launcher: start the N ranks as a child torch.distributed.run
    BEFORE this process touches a GPU, relay its output, exit with its code.  Never fewer ranks than asked for."""
    import subprocess

    world_env = os.environ.get("WORLD_SIZE")
    if "RANK" in os.environ or world_env is not None:
        if int(world_env or "1") != ARGS.gpus:
            sys.stderr.write("bench.py: --gpus %d but the launcher set WORLD_SIZE=%s: refusing to run\n" % (ARGS.gpus, world_env))
            sys.exit(2)
        return
    if ARGS.gpus <= 1:
        return
    ndev = visible_gpu_count()  # sysfs: the launcher parent never loads torch or the HIP runtime
    if ndev < ARGS.gpus and not (os.environ.get("PFT_BENCH_SHARE_GPU") == "1" and ndev >= 1):
        sys.stderr.write("bench.py: --gpus %d but %d GPU(s) are visible on this machine: refusing to run fewer ranks "
                         "and report them as %d\n" % (ARGS.gpus, ndev, ARGS.gpus))
        sys.exit(2)
    try:  # the ranks rely on the shipped / freshly built library (no rebuild under each other's feet)
        from pcl_tracking_amd import build as _hip_build

        _hip_build.build()
    except Exception as e:
        sys.stderr.write("bench: rebuild skipped (%s)\n" % e)
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ARGS.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def cpu_baseline(model, cloud, trans, P, threads, pcl_alloc=1, frames=20, seconds=30.0, warm=3):
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
    for _ in range(warm):  # warm-up frames (the first includes initParticles)
        tr.compute()
    times = []
    stages = None
    while len(times) < max(1, frames) and (len(times) < 3 or sum(times) < seconds):
        t0 = time.perf_counter()
        tr.compute()
        times.append(time.perf_counter() - t0)
        stages = tr.stage_times()
    return min(times), sorted(times)[len(times) // 2], stages, len(times)


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
            cores16 = min(16, usable_cpus()[0])
            npart = out[name]["particles_after"]
            out[name]["cpu_cores_behind_the_16_threads"] = cores16
            out[name]["cpu_pair_evals_per_s_per_core"] = 2.0 * npart * len(model) / min(ts) / cores16
    return out


def main():
    spawn_ranks_if_needed()  # --gpus N > 1 without a launcher: becomes the parent of N ranks and never returns

    import numpy as np
    import torch

    from pcl_tracking_amd import build as _hip_build
    from pcl_tracking_amd import scene

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        try:  # no-op when pcl_tracking_amd/_build/libpft_hip.so is current (it ships with the snapshot); multi-rank
            _hip_build.build()  # launches rely on the library their parent / the snapshot provides
        except Exception as e:  # a shipped library is still usable if the rebuild is not possible here
            sys.stderr.write("bench: rebuild skipped (%s)\n" % e)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    # PFT_BENCH_SHARE_GPU=1: rehearsal of the N-rank control flow on a one-GPU box -- every rank on cuda:0, collectives
    # through gloo with host staging (RCCL refuses two ranks on one device).  The numbers of such a run mean nothing.
    share_gpu = os.environ.get("PFT_BENCH_SHARE_GPU") == "1" and world > 1
    if share_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: LOCAL_RANK %d but %d GPU(s) visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = os.environ.get("PFT_DIST_FORCE_COLLECTIVES") == "1" and "RANK" in os.environ  # 1-rank rehearsal
    use_dist = world > 1 or force_dist
    if use_dist:
        import torch.distributed as dist

        if share_gpu:
            dist.init_process_group("gloo")
        else:
            # RCCL prints a version banner to STDOUT when its first communicator comes up; stdout is reserved for the one
            # JSON line, so the communicator is brought up (one tiny all-reduce) with file descriptor 1 pointing at stderr
            os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")  # (and whatever it logs later)
            sys.stdout.flush()
            saved_stdout = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("nccl", device_id=dev)
                warm = torch.zeros(1, device=dev)
                dist.all_reduce(warm)
                torch.cuda.synchronize()
            finally:
                sys.stdout.flush()
                os.dup2(saved_stdout, 1)
                os.close(saved_stdout)

    P_local = ARGS.particles_per_gpu
    sharded = world > 1 and ARGS.objects == 1  # several objects are replicas (one handle each), never sharded
    P_total = P_local * world if sharded else P_local
    M, N = ARGS.model_points, ARGS.cloud_points
    cloud = scene.make_scene(N, mode="organized" if ARGS.organized else "voxel")
    trans = scene.initial_trans()
    cloud_dev = torch.from_numpy(cloud.view(np.uint8).reshape(-1).copy()).to(dev)  # PCL 32-B layout in HBM

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    my_objects = 1
    if ARGS.objects > 1 or not use_dist:
        from pcl_tracking_amd import tracker

        # independent objects are independent handles, each on its own HIP stream (the reference tracks them in a
        # sequential loop, auto_tracking.cpp:688-697); with several ranks the objects are dealt out round-robin
        # ("replicas only": no data-path collective)
        ts = []
        for k in range(ARGS.objects):
            if k % world != rank:
                continue
            t = tracker.make_reference_tracker(particle_num=P_local, seed=1 + k, device_id=local_rank)
            # one model cloud per object (BASELINE configs[4]: "4 independent model clouds")
            t.setReferenceCloud(scene.make_model(M) if k == 0 else scene.make_model(M, seed=scene.MODEL_SEED + k))
            t.setTrans(trans)
            ts.append(t)
        my_objects = len(ts)
        model = scene.make_model(M)

        def run_frame(restore):
            for t in ts:
                if restore:
                    t.debugStateRestore()
                t.setInputCloudDevice(cloud_dev.data_ptr(), N, keepalive=cloud_dev)
                t.compute()

        def save_state():
            for t in ts:
                t.debugStateSave()

        trk = ts[0] if ts else None
    else:
        from pcl_tracking_amd.dist import HipPhases, ShardedFilter

        model = scene.make_model(M)
        ph = HipPhases(P_total, rank, world, dev, seed=1)
        ph.set_reference(model)
        ph.set_trans(trans)
        sf = ShardedFilter(ph)

        def run_frame(restore):
            if restore:
                ph.t.debugStateRestore()
            ph.set_input_device(cloud_dev, N)
            sf.compute()

        def save_state():
            ph.t.debugStateSave()

        trk = ph.t

    sync()
    tc = time.perf_counter()
    run_frame(False)  # cold: buffer growth, initParticles, first octree
    sync()
    cold_ms = (time.perf_counter() - tc) * 1e3
    # The replayed frame is always the frame of the same filter age, whatever --warmup: the particle cloud (and with it the
    # cropped cloud, 3 400 points at age 5, 7 000 at age 20) is still spreading during the first frames, so a run with
    # fewer than PREROLL_FRAMES warm-up steps first lets the filter run on, untimed, as part of the set-up.
    for _ in range(max(0, PREROLL_FRAMES - ARGS.warmup)):
        run_frame(False)
    for _ in range(ARGS.warmup):
        run_frame(False)
    sync()
    save_state()  # the frame every timed step replays: frame `max(warmup, PREROLL_FRAMES) + 2` of the run

    def timed(restore):
        for _ in range(3):
            run_frame(restore)
        sync()
        t0 = time.perf_counter()
        for _ in range(ARGS.steps):
            run_frame(restore)
        sync()
        dt = time.perf_counter() - t0
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if share_gpu else dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    replay = not ARGS.running
    # exactly --steps steps between barrier + synchronize, max over ranks -- REPEATED: a 10 ms timed region is one sample
    # of a shared machine, the median of >= 7 is the number (minimum and maximum beside it)
    reps = sorted(timed(replay) for _ in range(max(1, ARGS.repeats)))
    dt = reps[len(reps) // 2]
    ms_per_step = dt / ARGS.steps * 1e3
    dt_other = sorted(timed(not replay) for _ in range(3))[1]  # the other mode, reported beside the headline

    # ranks that actually took part (never a constant: a launch that lost ranks must not report them)
    took_part = torch.ones(1, dtype=torch.int32, device="cpu" if share_gpu else dev)
    if world > 1:
        dist.all_reduce(took_part)
    n_ranks = int(took_part.item())
    if n_ranks != ARGS.gpus:
        raise SystemExit("bench.py: %d rank(s) took part but --gpus %d" % (n_ranks, ARGS.gpus))

    # ---- per-kernel durations: HIP events on the kernels' own stream, the headline loop again ----
    # (The event objects are created on first use, and a stream that has run dry stamps a scope's first event before the
    # host has submitted the kernel behind it: a few untimed frames first, so that the pool exists and the GPU has work
    # queued, and at least 100 frames in the pass -- with the driver's --steps 20 the first frames' host latency used to
    # weigh 10 % on the likelihood figure, 198 us where rocprofv3 and a 100-frame pass both say 179.)
    prof, lik_ms, lik_n, dt_prof = {}, 0.0, 0, 0.0
    prof_frames = max(ARGS.steps, 100)
    if trk is not None:
        trk.profileEnable(True)
        for _ in range(5):
            run_frame(replay)
        sync()
        trk.profileReset()
    sync()
    t1 = time.perf_counter()
    for _ in range(prof_frames):
        run_frame(replay)
    sync()
    dt_prof = time.perf_counter() - t1
    if trk is not None:
        prof = trk.profileGet()
        trk.profileEnable(False)
        lik_ms, lik_n = prof["likelihood"]
    lik_event_s = lik_ms / max(1, lik_n) * 1e-3
    # What an event pair adds.  The frame is a strict chain of launches, so without events its time T is the sum of the
    # launches' true times (plus the replay's restore kernel, which no scope times); the event-timed scopes sum to
    # S = T + n * c per frame with n scopes and c the pair's cost.  c = (S - T) / n is subtracted from the likelihood
    # scope: the kernel's duration without instrumentation, which is what rocprofv3's kernel trace reports
    # (profiles/).  T still holds the ~4 us restore kernel, so c is slightly under- and the launch time slightly
    # over-estimated: the conservative side for `frac`.
    n_scopes = sum(v[1] for v in prof.values()) / prof_frames if prof else 0.0
    s_events_ms = sum(v[0] for v in prof.values()) / prof_frames if prof else 0.0
    ev_cost_s = max(0.0, (s_events_ms - ms_per_step) / n_scopes * 1e-3) if n_scopes > 0 else 0.0
    lik_avg_s = max(lik_event_s - ev_cost_s, 0.5 * lik_event_s)

    # mean leaf occupancy of the replayed frame's second likelihood launch (counted on the device, outside the timed
    # region): the particles after a replayed step are exactly the ones that launch evaluated.  (Every rank replays: a
    # sharded frame has collectives in it.)
    if replay:
        run_frame(True)
        sync()
    out = None
    if rank == 0:
        pcur = trk.getParticles()
        if sharded:
            pcur = pcur[rank * P_local:(rank + 1) * P_local]
        st = trk.evalWeights(pcur[:P_local], want_nn=True)
        kbar = st["scan_points"] / max(1, st["scan_queries"])
        # algorithmic bytes: per pair-eval 16 B reference point + 16 B per candidate scanned
        bytes_per_launch = P_local * M * 16.0 * (1.0 + kbar)
        achieved = bytes_per_launch / lik_avg_s / 1e9 if lik_avg_s > 0 else 0.0
        key = "P%d_M%d_N%d%s" % (P_local, M, N, "_organized" if ARGS.organized else "")
        pmc = {}
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                pmc = json.load(open(tf)).get(key, {})
            except Exception:
                pmc = {}
        traffic = pmc.get("hbm_bytes_per_launch")
        n_obj = ARGS.objects
        pips = n_obj * P_total * N / (dt / ARGS.steps)
        iters = 2
        frame_bytes = iters * bytes_per_launch * (world if sharded else 1) * n_obj  # B_frame = I P M 16 (1 + k)
        frame_gbs = frame_bytes / (dt / ARGS.steps) / 1e9
        out = {
            "metric": "particles x input-points / sec per frame (tracked frames/sec @8192 particles per GPU)",
            "value": pips,
            "unit": "particle-points/s",
            "n_gpus": n_ranks,
            "steps": ARGS.steps,
            "warmup": ARGS.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_step_min": reps[0] / ARGS.steps * 1e3, "ms_per_step_max": reps[-1] / ARGS.steps * 1e3,
            "repeats": len(reps),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not share_gpu else "synthetic -- REHEARSAL: all ranks share one GPU through gloo, the numbers mean nothing",
            "config": {
                "workload": workload_label(P_local, M, N, ARGS.organized, n_obj, world),
                "mode": ("replayed frame (state checkpointed at filter age %d frames, restored before every step)" % max(ARGS.warmup, PREROLL_FRAMES)) if replay
                        else "free-running filter",
                "particles_total": P_total, "model_points": M, "cloud_points": N, "iterations_per_frame": iters,
                "parallelism": ("particles sharded x%d (RCCL world size %d)" % (world, world)) if sharded else
                               ("%d object replica(s) over %d rank(s), no collective" % (n_obj, world)),
                "objects": n_obj,
            },
            "frames_per_s": ARGS.steps / dt,
            "cold_first_frame_ms": cold_ms,
            ("ms_per_step_running" if replay else "ms_per_step_replayed"): dt_other / ARGS.steps * 1e3,
            "pair_evals_per_s": float(iters) * n_obj * P_total * M / (dt / ARGS.steps),
            "cropped_points": int(len(st["crop_idx"])), "octree_depth": int(st["octree_depth"]),
            "mean_leaf_occupancy": kbar,
            "ms_per_step_with_events": dt_prof / prof_frames * 1e3,
            "kernel_ms_per_frame": {k: round(v[0] / prof_frames, 5) for k, v in prof.items()},
            "kernel_timing_frames": prof_frames,
            "roofline": {
                "bound": "hbm", "kernel": "k_likelihood", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_us": lik_avg_s * 1e6,
                "avg_launch_us_between_events": lik_event_s * 1e6, "event_pair_cost_us": ev_cost_s * 1e6,
                "timed_scopes_per_frame": n_scopes,
                "launches": int(lik_n),
                "note": "algorithmic bytes (SURVEY 8d) over the live launch time (HIP events on the kernel's stream, less "
                        "the measured cost of an event pair: see event_pair_cost_us); the working set is LDS / L2-resident "
                        "(see traffic), so HBM is the yardstick BASELINE.json asks for, not the limiter: roofline_valu",
            },
            # the whole frame against the same roof (B_frame / t_compute): what the tracked-frames rate is worth
            "roofline_frame": {"bound": "hbm", "achieved": frame_gbs, "peak": HBM_PEAK_GBS * n_ranks, "unit": "GB/s",
                               "frac": frame_gbs / (HBM_PEAK_GBS * n_ranks), "algorithmic_bytes_per_frame": frame_bytes},
        }
        if pmc.get("valu_wave_insts_per_launch"):
            # what actually limits k_likelihood: VALU issue.  Instruction and lane-cycle counts per launch from the
            # committed PMC pass of this workload (profiles/, tools/pmc.sh); the duration is the live one; the peak is
            # the guide's 2 cycles per wave64 VALU instruction on each of the 1024 SIMD-32s at 2.4 GHz.
            vi = float(pmc["valu_wave_insts_per_launch"])
            out["roofline_valu"] = {
                "bound": "valu_issue", "kernel": "k_likelihood", "achieved": vi / lik_avg_s, "peak": VALU_ISSUE_PEAK,
                "unit": "wave-inst/s", "frac": vi / lik_avg_s / VALU_ISSUE_PEAK,
                "lane_util": pmc.get("valu_lane_utilisation"),
                "wave_insts_per_launch": vi, "valu_insts_per_wave_query_round": vi / (P_local * M / 64.0),
                "source": pmc.get("source"),
            }
        if out["roofline"]["frac"] > 1.0:
            # The HBM yardstick (algorithmic bytes over launch time) exceeds the HBM peak for this shape: the candidates a
            # query scans come from L2 (see traffic), so the yardstick says nothing here.  The line's `roofline` is then the
            # bound that does describe the kernel -- VALU issue -- and the yardstick is kept beside it under its own name.
            out["roofline_hbm_yardstick"] = dict(out["roofline"], note="exceeds the HBM peak: the working set is L2-resident; not a "
                                                 "meaningful roof for this shape")
            if "roofline_valu" in out:
                rv = out["roofline_valu"]
                out["roofline"] = {"bound": "valu_issue", "kernel": "k_likelihood", "achieved": rv["achieved"], "peak": rv["peak"],
                                   "unit": rv["unit"], "frac": rv["frac"], "traffic": traffic, "lane_util": rv.get("lane_util"),
                                   "avg_launch_us": lik_avg_s * 1e6, "launches": int(lik_n),
                                   "note": "VALU issue bound (PMC instruction count of this shape over the live launch time); the "
                                           "HBM yardstick is under roofline_hbm_yardstick"}
            else:
                out["roofline"]["frac"] = None
                out["roofline"]["note"] = "HBM yardstick exceeds the peak for this shape (L2-resident working set) and no PMC " \
                                          "instruction count is on file for it: no meaningful fraction"
        if world == 1 and n_obj == 1 and not ARGS.no_cpu_baseline:
            # threads: what this process can really run at once (affinity mask and cgroup quota, not os.cpu_count()),
            # half of it (SMT siblings), and the reference's own 16 (auto_tracking.cpp:845); a short probe of each, the
            # best one takes the full measurement
            usable, cpu_count, affinity, quota = usable_cpus()
            cands = sorted({max(1, usable), max(1, usable // 2), 16})
            probe = {}
            for th in cands:
                pmin, _, _, _ = cpu_baseline(model, cloud, trans, P_total, th, frames=2, seconds=ARGS.cpu_seconds / 10, warm=1)
                probe[th] = pmin
            threads = min(probe, key=probe.get)
            tmin, tmed, stages, nfr = cpu_baseline(model, cloud, trans, P_total, threads, frames=ARGS.cpu_frames,
                                                   seconds=ARGS.cpu_seconds * 0.6)
            pair_evals = float(iters) * P_total * M
            out["cpu_baseline"] = {
                "value": P_total * N / tmed, "unit": "particle-points/s", "cores": min(threads, usable), "kind": "port",
                "threads_used": threads, "cpu_count": cpu_count, "affinity": affinity, "cgroup_quota": quota,
                "thread_probe_s_per_frame": {str(k): round(v, 4) for k, v in probe.items()},
                "sample": "%d timed frames of the same workload (P=%d, M=%d, N=%d, 2 iterations; free-running filter) after "
                          "3 warm-up frames, bounded at %.0f s of timed work; median frame time %.3f s (value), min %.3f s; "
                          "%d OpenMP threads = the best of a 2-frame probe of %s threads on %d usable CPUs"
                          % (nfr, P_total, M, N, ARGS.cpu_seconds * 0.6, tmed, tmin, threads, cands, usable),
                "frames": nfr, "median_s": tmed, "min_s": tmin,
                "pair_evals_per_s_per_core": pair_evals / tmed / min(threads, usable),
                "stage_seconds": {k: round(float(v), 4) for k, v in zip(
                    ("transform", "bbox_crop", "octree", "coherence", "normalize", "resample", "update"), stages)},
            }
            out["speedup_vs_cpu"] = pips / out["cpu_baseline"]["value"]
            # the same port without PCL's per-query index-vector allocation ("optimised CPU"): so that
            # the ratio is not credited to de-pessimising the allocator behaviour alone
            fmin, fmed, _, nf2 = cpu_baseline(model, cloud, trans, P_total, threads, pcl_alloc=0, frames=5,
                                              seconds=ARGS.cpu_seconds / 4, warm=1)
            out["cpu_optimised"] = {"value": P_total * N / fmed, "unit": "particle-points/s", "cores": min(threads, usable),
                                    "threads_used": threads,
                                    "kind": "port", "sample": "same, emulate_pcl_alloc=0; %d frames, median %.3f s, min %.3f s"
                                                              % (nf2, fmed, fmin)}
            out["speedup_vs_cpu_optimised"] = pips / out["cpu_optimised"]["value"]
        if world == 1 and n_obj == 1 and not ARGS.no_frontend:
            out["frontend"] = frontend_measurement(dev, not ARGS.no_cpu_baseline)
            out["reference_operating_point"] = reference_operating_point(model, cloud, trans, dev, not ARGS.no_cpu_baseline)
            if "cpu_baseline" in out and "cpu_pair_evals_per_s_per_core" in out["reference_operating_point"].get("fixed_400", {}):
                a = out["cpu_baseline"]["pair_evals_per_s_per_core"]
                b = out["reference_operating_point"]["fixed_400"]["cpu_pair_evals_per_s_per_core"]
                out["cpu_baseline"]["per_core_rate_vs_400_particle_line"] = a / b
                if not 0.5 <= a / b <= 2.0:
                    out["cpu_baseline"]["per_core_rate_note"] = (
                        "the per-core pair-evaluation rates of the P=%d and the P=400 lines differ by more than 2x: at 400 "
                        "particles the per-iteration crop + octree build (serial) and the OpenMP fork/join are a larger share "
                        "of the frame, and 16 threads oversubscribe %d usable CPUs" % (P_total, usable_cpus()[0]))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    ARGS = parse()
    main()
