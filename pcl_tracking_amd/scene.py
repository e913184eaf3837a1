"""Seeded synthetic model + RGB-D-like scene generator.

The reference tracks a segmented object model (a cluster of 500-25000 RGBA points written by
/root/reference/src/create_model.cpp:219-222) in voxel-downsampled Kinect2 frames
(/root/reference/src/auto_tracking.cpp:683).  No recorded frame ships with the reference
(*.pcd is git-ignored), so tests and bench.py use this deterministic stand-in: a coloured box model
and a ray-cast scene (table, wall, clutter boxes, the object at a ground-truth pose).
Pure numpy; no GPU, no oracle.
"""
import numpy as np

POINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4"), ("rgba", "<u4"), ("pad", "<u4", (3,))]
)
PARTICLE_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4"),
     ("roll", "<f4"), ("pitch", "<f4"), ("yaw", "<f4"), ("weight", "<f4")]
)

SCENE_SEED = 20261004
MODEL_SEED = 20261005
MODEL_DIMS = (0.40, 0.30, 0.25)
GT_POSE = (0.1, -0.05, 0.9, 0.2, -0.1, 0.4)  # x y z roll pitch yaw

# per-face base colours (r, g, b): -x +x -y +y -z +z
FACE_RGB = np.array(
    [[200, 40, 40], [40, 180, 60], [50, 70, 200], [210, 190, 40], [180, 60, 190], [40, 190, 190]], np.int32
)


def pose_matrix(x, y, z, roll, pitch, yaw):
    """R = Rz(yaw) Ry(pitch) Rx(roll), float64 4x4 (same convention as pcl::getTransformation)."""
    A, B = np.cos(yaw), np.sin(yaw)
    Cc, D = np.cos(pitch), np.sin(pitch)
    E, F = np.cos(roll), np.sin(roll)
    return np.array(
        [
            [A * Cc, A * D * F - B * E, B * F + A * D * E, x],
            [B * Cc, A * E + B * D * F, B * D * E - A * F, y],
            [-D, Cc * F, Cc * E, z],
            [0, 0, 0, 1.0],
        ]
    )


def pack_rgba(r, g, b, a=255):
    return (
        (np.asarray(a, np.uint32) << 24)
        | (np.asarray(r, np.uint32) << 16)
        | (np.asarray(g, np.uint32) << 8)
        | np.asarray(b, np.uint32)
    )


def make_points(xyz, rgb):
    n = len(xyz)
    p = np.zeros(n, POINT_DTYPE)
    p["x"], p["y"], p["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    p["w"] = 1.0
    rgb = np.clip(rgb, 0, 255).astype(np.uint32)
    p["rgba"] = pack_rgba(rgb[:, 0], rgb[:, 1], rgb[:, 2])
    return p


def _box_surface_lattice(dims, step):
    """lattice points on the 6 faces of an origin-centred box; returns xyz, face id"""
    hx, hy, hz = dims[0] / 2, dims[1] / 2, dims[2] / 2
    out, fid = [], []
    ax = [np.arange(-h + step / 2, h, step) for h in (hx, hy, hz)]
    for f in range(6):
        a = f // 2
        s = -1.0 if f % 2 == 0 else 1.0
        o = [i for i in range(3) if i != a]
        u, v = np.meshgrid(ax[o[0]], ax[o[1]], indexing="ij")
        pts = np.zeros((u.size, 3))
        pts[:, a] = s * (hx, hy, hz)[a]
        pts[:, o[0]] = u.ravel()
        pts[:, o[1]] = v.ravel()
        out.append(pts)
        fid.append(np.full(u.size, f))
    return np.concatenate(out), np.concatenate(fid)


def _visible_faces(dims, pose):
    """faces of the box at `pose` whose outward normal points towards the camera at the origin"""
    T = pose_matrix(*pose)
    vis = []
    for f in range(6):
        n = np.zeros(3)
        n[f // 2] = -1.0 if f % 2 == 0 else 1.0
        c = n * (np.asarray(dims) / 2)
        if (T[:3, :3] @ n) @ (T[:3, :3] @ c + T[:3, 3]) < 0:
            vis.append(f)
    return vis


def make_model(M=2048, seed=MODEL_SEED, dims=MODEL_DIMS, view_pose=GT_POSE, return_offset=False):
    """Object model of exactly M points, centred on its centroid (the contract of
    /root/reference/src/auto_tracking.cpp:663-673: reference cloud re-centred, then setReferenceCloud).
    Like the clusters /root/reference/src/create_model.cpp segments from one camera view, it holds only
    the faces visible from the camera at `view_pose` (a full-surface model has no maximum of PCL's
    count-within-10-cm likelihood at the true pose once the object is deeper than the 10 cm gate)."""
    rng = np.random.default_rng(seed)
    vis = _visible_faces(dims, view_pose)
    step = 0.01
    while True:
        xyz, fid = _box_surface_lattice(dims, step)
        keep = np.isin(fid, vis)
        xyz, fid = xyz[keep], fid[keep]
        if len(xyz) >= M:
            break
        step *= 0.9
    xyz = xyz + rng.uniform(-0.002, 0.002, xyz.shape)
    sel = np.sort(rng.choice(len(xyz), M, replace=False))
    xyz, fid = xyz[sel], fid[sel]
    offset = xyz.mean(axis=0)
    xyz = xyz - offset
    rgb = FACE_RGB[fid] + rng.integers(-8, 9, (M, 3))
    pts = make_points(xyz.astype(np.float32), rgb)
    return (pts, offset) if return_offset else pts


def model_gt_pose(obj_pose=GT_POSE, M=2048, seed=MODEL_SEED, dims=MODEL_DIMS):
    """pose the tracker should estimate for the re-centred model: the box pose composed with the
    model's centroid offset"""
    _, off = make_model(M, seed, dims, GT_POSE, return_offset=True)
    T = pose_matrix(*obj_pose)
    t = T[:3, :3] @ off + T[:3, 3]
    return (t[0], t[1], t[2], obj_pose[3], obj_pose[4], obj_pose[5])


class _Box:
    def __init__(self, center, dims, rpy, face_rgb):
        self.T = pose_matrix(center[0], center[1], center[2], *rpy)
        self.R = self.T[:3, :3]
        self.c = np.asarray(center, float)
        self.h = np.asarray(dims, float) / 2
        self.face_rgb = np.asarray(face_rgb, np.int32)

    def intersect(self, d):
        """rays from the origin with directions d (n,3): returns t (inf if miss) and face id"""
        o = -(self.R.T @ self.c)  # origin in box frame
        db = d @ self.R  # directions in box frame
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = (-self.h - o) / db
            t2 = (self.h - o) / db
        tn = np.minimum(t1, t2)
        tf = np.maximum(t1, t2)
        tnear = tn.max(axis=1)
        tfar = tf.min(axis=1)
        hit = (tnear <= tfar) & (tnear > 1e-6)
        axis = tn.argmax(axis=1)
        sgn = np.take_along_axis(db, axis[:, None], 1)[:, 0] > 0  # entering through the -face if d>0
        face = axis * 2 + np.where(sgn, 0, 1)
        return np.where(hit, tnear, np.inf), face


def _scene_boxes(obj_pose, rng):
    boxes = []
    # object
    boxes.append(_Box(obj_pose[:3], MODEL_DIMS, obj_pose[3:], FACE_RGB))
    # table: thin slab tilted 30 deg about x
    table_rgb = np.tile(np.array([[150, 110, 70]]), (6, 1))
    boxes.append(_Box((0.0, 0.35, 1.25), (2.0, 0.02, 1.6), (np.deg2rad(-30.0), 0, 0), table_rgb))
    # wall
    wall_rgb = np.tile(np.array([[170, 170, 165]]), (6, 1))
    boxes.append(_Box((0.0, 0.0, 2.5), (6.0, 4.0, 0.02), (0, 0, 0), wall_rgb))
    # six clutter boxes
    for k in range(6):
        c = (rng.uniform(-0.7, 0.7), rng.uniform(-0.25, 0.25), rng.uniform(0.8, 1.6))
        dims = rng.uniform(0.08, 0.22, 3)
        rpy = rng.uniform(-0.6, 0.6, 3)
        base = rng.integers(30, 226, (1, 3))
        rgbs = np.clip(base + rng.integers(-40, 41, (6, 3)), 0, 255)
        b = _Box(c, dims, rpy, rgbs)
        # keep clutter away from the object so the tracker has an unambiguous target
        if np.linalg.norm(np.asarray(c) - np.asarray(obj_pose[:3])) > 0.35:
            boxes.append(b)
    return boxes


def _raycast(width, height, fx, boxes):
    u, v = np.meshgrid(np.arange(width), np.arange(height), indexing="xy")
    d = np.stack([(u.ravel() - (width - 1) / 2) / fx, (v.ravel() - (height - 1) / 2) / fx, np.ones(u.size)], 1)
    best_t = np.full(len(d), np.inf)
    best_rgb = np.zeros((len(d), 3), np.int32)
    for b in boxes:
        t, face = b.intersect(d)
        closer = t < best_t
        best_t = np.where(closer, t, best_t)
        best_rgb[closer] = b.face_rgb[face[closer]]
    ok = np.isfinite(best_t)
    return d[ok] * best_t[ok, None], best_rgb[ok]


def make_scene(N=50000, seed=SCENE_SEED, obj_pose=GT_POSE, mode="voxel", leaf=0.01):
    """Scene cloud of exactly N points in the camera frame (z forward).
    mode 'voxel': rays cast on a fine grid, depth noise, one point kept per `leaf` voxel (stand-in for
    ApproximateVoxelGrid, auto_tracking.cpp:563-575), random subset of N, seeded shuffle.
    mode 'organized': a width x height depth image with width*height == N (e.g. 640x480 = 307200),
    no downsample, row-major order (config 3 of BASELINE.json)."""
    rng = np.random.default_rng(seed)
    boxes = _scene_boxes(obj_pose, rng)
    if mode == "organized":
        w = int(round(np.sqrt(N * 4 / 3)))
        h = N // w
        assert w * h == N, "organized mode needs N = w*h with w:h = 4:3 (e.g. 307200)"
        xyz, rgb = _raycast(w, h, 0.82 * w, boxes)  # ~63 deg horizontal fov
        xyz = xyz * (1.0 + rng.normal(0, 0.0015, len(xyz)) / np.maximum(xyz[:, 2], 0.1))[:, None]
        rgb = rgb + rng.integers(-8, 9, rgb.shape)
        if len(xyz) < N:  # rays that hit nothing: pad by repeating wall points
            pad = rng.choice(len(xyz), N - len(xyz))
            xyz = np.concatenate([xyz, xyz[pad]])
            rgb = np.concatenate([rgb, rgb[pad]])
        return make_points(xyz.astype(np.float32), rgb)
    scale = 1
    while True:
        w, h = 1280 * scale, 960 * scale
        xyz, rgb = _raycast(w, h, 0.82 * w, boxes)
        xyz = xyz * (1.0 + rng.normal(0, 0.0015, len(xyz)) / np.maximum(xyz[:, 2], 0.1))[:, None]
        rgb = rgb + rng.integers(-8, 9, rgb.shape)
        key = np.floor(xyz / leaf).astype(np.int64)
        key = (key[:, 0] + 4096) * (1 << 26) + (key[:, 1] + 4096) * (1 << 13) + (key[:, 2] + 4096)
        _, first = np.unique(key, return_index=True)
        first.sort()
        if len(first) >= N or scale >= 4:
            break
        scale *= 2
    xyz, rgb = xyz[first], rgb[first]
    if len(xyz) >= N:
        sel = rng.choice(len(xyz), N, replace=False)
    else:
        sel = np.concatenate([np.arange(len(xyz)), rng.choice(len(xyz), N - len(xyz))])
    sel = rng.permutation(sel)
    return make_points(xyz[sel].astype(np.float32), rgb[sel])


def make_depth_frame(width=960, height=540, seed=SCENE_SEED, obj_pose=GT_POSE, dropout=0.03):
    """Raw organised sensor frame as the reference receives it before filterPassThrough / gridSampleApprox
    (auto_tracking.cpp:637, 683; Kinect2 'qhd' = 960x540, :775): width*height points in row-major order, rays
    that hit nothing and a random `dropout` fraction of pixels are NaN (invalid depth), and a strip of far
    background lies beyond the PassThrough limit z <= 10."""
    rng = np.random.default_rng(seed)
    boxes = _scene_boxes(obj_pose, rng)
    far_rgb = np.tile(np.array([[90, 100, 120]]), (6, 1))
    boxes[2] = _Box((1.0, 0.0, 2.5), (4.0, 4.0, 0.02), (0, 0, 0), boxes[2].face_rgb)  # wall leaves the left edge open
    boxes.append(_Box((-3.0, 0.0, 12.0), (30.0, 9.0, 0.02), (0, 0, 0), far_rgb))  # visible through the gap, z = 12
    u, v = np.meshgrid(np.arange(width), np.arange(height), indexing="xy")
    fx = 0.82 * width
    d = np.stack([(u.ravel() - (width - 1) / 2) / fx, (v.ravel() - (height - 1) / 2) / fx, np.ones(u.size)], 1)
    best_t = np.full(len(d), np.inf)
    best_rgb = np.zeros((len(d), 3), np.int32)
    for b in boxes:
        t, face = b.intersect(d)
        closer = t < best_t
        best_t = np.where(closer, t, best_t)
        best_rgb[closer] = b.face_rgb[face[closer]]
    ok = np.isfinite(best_t) & (rng.random(len(d)) >= dropout)
    t = np.where(ok, best_t, 1.0)
    xyz = d * t[:, None]
    xyz = xyz * (1.0 + rng.normal(0, 0.0015, len(xyz)) / np.maximum(xyz[:, 2], 0.1))[:, None]
    rgb = best_rgb + rng.integers(-8, 9, best_rgb.shape)
    pts = make_points(xyz.astype(np.float32), rgb)
    for k in ("x", "y", "z"):
        pts[k][~ok] = np.nan
    return pts


def advance_pose(pose, frame):
    """ground-truth motion for multi-frame runs: +1 mm in x and +0.5 deg yaw per frame"""
    p = list(pose)
    p[0] += 0.001 * frame
    p[5] += np.deg2rad(0.5) * frame
    return tuple(p)


def initial_trans(obj_pose=GT_POSE, offset=(0.01, 0.01, 0.01)):
    """trans_ handed to setTrans: ground-truth translation of the model centroid + offset, identity
    rotation (the reference passes the model centroid, auto_tracking.cpp:663-674)"""
    g = model_gt_pose(obj_pose)
    m = np.eye(4, dtype=np.float32)
    m[0, 3] = g[0] + offset[0]
    m[1, 3] = g[1] + offset[1]
    m[2, 3] = g[2] + offset[2]
    return m
