"""Deterministic synthetic inputs: point clouds, colours, labels, EE crops, key points.

No reference data ships (sample frames are missing blobs), so every parity test and the
benchmark use these generators.  Pure numpy; identical on the build container and on the GPU box.
"""
import numpy as np


def gen_room(n: int, L: float = 2.4, seed: int = 0, sigma: float = 0.002):
    """n points on the six faces of an obliquely rotated cube of side L, pushed to z ~ L.

    Returns (points float32[n,3], rgb float32[n,3] in [-0.5,0.5), labels int64[n] in {0,1,2}).
    Labels are the face-pair id (the axis the face is normal to) -> a 3-class ground truth.
    """
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(Q) < 0:
        Q[:, 0] = -Q[:, 0]
    face = rng.integers(0, 6, size=n)
    uv = rng.uniform(-L / 2, L / 2, size=(n, 2))
    normal = np.where(face % 2 == 0, L / 2, -L / 2) + rng.normal(0.0, sigma, size=n)
    axis = face // 2
    pts = np.empty((n, 3), dtype=np.float64)
    for a in range(3):
        m = axis == a
        other = [i for i in range(3) if i != a]
        pts[m, a] = normal[m]
        pts[m, other[0]] = uv[m, 0]
        pts[m, other[1]] = uv[m, 1]
    pts = pts @ Q.T + np.array([0.0, 0.0, L])
    rgb = rng.uniform(0.0, 1.0, size=(n, 3)) - 0.5
    return pts.astype(np.float32), rgb.astype(np.float32), axis.astype(np.int64)


# the six constant end-effector key points of the reference (app/inference_engine.py:128-137)
REFERENCE_KEY_POINTS = np.array([
    [0.01982731, 0.08085986, 0.00321919],
    [0.02171595, -0.08986182, 0.00388430],
    [0.01288678, 0.09103118, 0.06127814],
    [0.02079032, -0.09790908, 0.05609143],
    [-0.00185802, 0.04654205, 0.11564558],
    [0.00241113, -0.04262756, 0.11564558],
])


def random_pose(rng):
    """(x, y, z, qw, qx, qy, qz) with a uniformly random unit quaternion and a position in front of the camera."""
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return np.concatenate([rng.uniform(-0.5, 0.5, size=2), rng.uniform(0.6, 1.4, size=1), q])


def quat_to_matrix(q):
    w, x, y, z = q
    return np.array([
        [2 * (w * w + x * x) - 1, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 2 * (w * w + y * y) - 1, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 2 * (w * w + z * z) - 1],
    ])


def gen_ee_crop(seed, n=4096, kp_noise=0.001):
    """Cfg-3 end-effector crop: n points in a 0.10 x 0.22 x 0.13 m box at a seeded pose, and the six
    key points = REFERENCE_KEY_POINTS moved by that pose + N(0, 1 mm).
    Returns (points float32[n,3], rgb float32[n,3], pose float64[7], key_points float64[6,3])."""
    rng = np.random.default_rng(10_000 + seed)
    pose = random_pose(rng)
    R = quat_to_matrix(pose[3:])
    local = rng.uniform(-0.5, 0.5, size=(n, 3)) * np.array([0.10, 0.22, 0.13]) + np.array([0.0, 0.0, 0.06])
    pts = local @ R.T + pose[:3]
    rgb = rng.uniform(0.0, 1.0, size=(n, 3)) - 0.5
    kps = REFERENCE_KEY_POINTS @ R.T + pose[:3] + rng.normal(0.0, kp_noise, size=(6, 3))
    return pts.astype(np.float32), rgb.astype(np.float32), pose, kps


def gen_scene(seed, n_bg=40_000, n_arm=4_000, n_ee=4_096, room=2.4, keyed_colors=False):
    """A labelled robot scene for the evaluation harness (reference frame format, README.md:55-62): background room
    (label 0), a 4 cm-radius arm cylinder from a base point to the end effector (label 1) and the EE crop (label 2) at a
    seeded pose.  Returns dict(points, rgb in [0,1], segmentation, pose (x,y,z,qw,qx,qy,qz), key_points [6,3],
    ee2base_pose, position).  keyed_colors: the red channel is bright (>= 0.8) exactly on the end effector and the green one
    exactly on the arm (everything else <= 0.45), so that a network carrying `wire_color_keyed_labels` labels the scene
    by construction - an EE crop of known size whatever the other weights are."""
    rng = np.random.default_rng(20_000 + seed)
    bg, _, _ = gen_room(n_bg, room, seed)
    ee, _, pose, kps = gen_ee_crop(seed, n=n_ee)
    base = np.array([0.0, -0.6, room * 0.6])
    t = rng.uniform(0.0, 1.0, size=(n_arm, 1))
    axis = pose[:3] - base
    u = np.cross(axis, [0.0, 0.0, 1.0])
    u /= np.linalg.norm(u)
    v = np.cross(axis, u)
    v /= np.linalg.norm(v)
    ang = rng.uniform(0, 2 * np.pi, size=(n_arm, 1))
    arm = base + t * axis + 0.04 * (np.cos(ang) * u + np.sin(ang) * v)
    points = np.concatenate([bg, arm.astype(np.float32), ee]).astype(np.float32)
    seg = np.concatenate([np.zeros(len(bg), np.int64), np.ones(n_arm, np.int64), np.full(len(ee), 2, np.int64)])
    rgb = rng.uniform(0.0, 1.0, size=(len(points), 3)).astype(np.float32)
    if keyed_colors:
        rgb[:, :2] *= 0.45
        rgb[seg == 2, 0] = 0.8 + 0.2 * rng.uniform(size=int((seg == 2).sum())).astype(np.float32)
        rgb[seg == 1, 1] = 0.8 + 0.2 * rng.uniform(size=int((seg == 1).sum())).astype(np.float32)
    perm = rng.permutation(len(points))
    ee2base = np.concatenate([rng.uniform(-0.3, 0.3, size=3), random_pose(rng)[3:]])
    return {"points": points[perm], "rgb": rgb[perm], "segmentation": seg[perm], "pose": pose, "key_points": kps,
            "ee2base_pose": ee2base, "position": f"p{seed % 3 + 1}"}


def wire_color_keyed_labels(model, gain=10.0):
    """Make a (randomly initialised) RobotNetSegmentation label by INPUT COLOUR, without taking work out of the network:
    two channels are wired from the input to the logits - red -> channel 0, green -> channel 1 through conv0's centre
    offset, the level-0 skip connection, block8's residual path, `final` and `regression.0` (every weight that would mix
    another value into those two channels is zeroed) - and `regression.2` reads only them: class 2 (end effector) where the
    voxel's mean red exceeds ~0.1 above mid-grey, class 1 (arm) where green does, class 0 elsewhere.  All other weights
    keep their random values, so every layer still multiplies full-size random operands (the kernels, their work and the
    clocks they sustain are those of a trained network), while `gen_scene(keyed_colors=True)` frames come out with their
    ground-truth labels by construction: tests and benchmarks of the stages BEHIND the segmentation (cluster rule, EE crop,
    pose networks, key points, Kabsch) no longer depend on what random weights happen to predict.  BatchNorm layers must be
    at their default running statistics on the wired channels (fresh models are).  In place; returns the model."""
    import torch

    with torch.no_grad():
        c0 = model.conv0p1s1.kernel  # [27, 3, 32]
        c0[:, :, 0:2] = 0.0
        c0[13, 0, 0] = 1.0
        c0[13, 1, 1] = 1.0
        n = model.N_LEVELS
        last_block = getattr(model, f"block{2 * n}")
        up_planes = getattr(model, model._up_names(2 * n - 1)[0]).out_channels  # columns of the transposed conv in the cat
        b0 = last_block[0]
        if b0.downsample is None:
            raise ValueError("the last decoder block has no 1x1 downsample branch to wire through")
        ds = b0.downsample[0].kernel  # [416, 384]
        ds[:, 0:2] = 0.0
        ds[up_planes + 0, 0] = 1.0
        ds[up_planes + 1, 1] = 1.0
        for blk in last_block:
            blk.conv2.kernel[:, :, 0:2] = 0.0
        fin = model.final.kernel  # [384, 256]
        fin[:, 0:2] = 0.0
        fin[0, 0] = 1.0
        fin[1, 1] = 1.0
        model.final.bias[0, 0:2] = 0.0
        r0 = model.regression[0].linear  # weight [1024, 256]
        r0.weight[0:2, :] = 0.0
        r0.weight[0, 0] = 1.0
        r0.weight[1, 1] = 1.0
        r0.bias[0:2] = 0.0
        r2 = model.regression[2].linear  # weight [classes, 1024]
        r2.weight.zero_()
        r2.weight[2, 0] = gain
        r2.weight[1, 1] = gain
        r2.bias.copy_(torch.tensor([0.0, -1.0, -1.0])[: r2.bias.numel()])
    return model
