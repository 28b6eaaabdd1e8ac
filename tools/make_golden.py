"""Generate tests/golden/*.npz by running the REFERENCE's own functions (imported from /root/reference) on seeded inputs.

Runs only in the build container (the reference never travels to the GPU box); the fixtures it writes are data:
inputs and the reference's outputs.  Two dev-only imports the reference drags in are satisfied with empty modules:
`ipdb` (utils/transformation.py:4, a debugger that is never called) and `turtle` (utils/calibration.py:1, an unused
`from turtle import pos`).  The reference's utils/output.py does `import MinkowskiEngine as ME` (for a type annotation):
MinkowskiEngine is not installable here, so this build's own ME-shaped namespace is registered under that name
(mrcc_amd.install_as_minkowski_engine() - the drop-in boundary doing its job); the two functions taken from that file,
get_pred_center and get_key_point_predictions, are pure torch/numpy and never touch ME.

    python tools/make_golden.py          # rewrites tests/golden/{kabsch,quat_avg,add,fps,ball_query,preprocess}.npz
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

sys.modules.setdefault("ipdb", types.ModuleType("ipdb"))
_turtle = types.ModuleType("turtle")
_turtle.pos = None
sys.modules.setdefault("turtle", _turtle)
sys.path.insert(0, REF)

sys.path.insert(1, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrcc_amd  # noqa: E402

mrcc_amd.install_as_minkowski_engine()

from utils import transformation as T  # noqa: E402
from utils import output as Out  # noqa: E402
from utils import calibration as Cal  # noqa: E402
from utils import metrics as Mx  # noqa: E402
from utils import preprocess as Pre  # noqa: E402
from utils import data as Dat  # noqa: E402
import torch  # noqa: E402
from model import pointnet2_utils as P2  # noqa: E402

def read_reference_key_points():
    import re

    src = open(os.path.join(REF, "app", "inference_engine.py")).read()
    m = re.search(r"self\.reference_key_points\s*=\s*np\.array\(\s*\[(.*?)\]\s*,?\s*(dtype=[^)]*)?\)", src, re.S)
    rows = re.findall(r"\[\s*([-\d.eE+]+)\s*,\s*([-\d.eE+]+)\s*,\s*([-\d.eE+]+)\s*\]", m.group(1))
    return np.array(rows, dtype=np.float64)


def rand_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def gen_kabsch(rng):
    kp = read_reference_key_points()
    B, Kmax = 256, 6
    ref = np.zeros((B, Kmax, 3))
    tgt = np.zeros((B, Kmax, 3))
    K = np.zeros(B, dtype=np.int32)
    R = np.zeros((B, 3, 3))
    t = np.zeros((B, 3))
    q = np.zeros((B, 4))
    kind = np.zeros(B, dtype=np.int32)
    for b in range(B):
        k = int(rng.integers(4, 7))
        cls = np.sort(rng.choice(6, size=k, replace=False))
        a = kp[cls].copy()
        mode = b % 8
        Rgt = T.get_quaternion_rotation_matrix(rand_quat(rng), switch_w=False)
        tgt_pts = (Rgt @ a.T).T + rng.uniform(-1, 1, size=3)
        if mode == 1:  # noisy key points (1 mm)
            tgt_pts += rng.normal(0, 1e-3, size=tgt_pts.shape)
        elif mode == 2:  # heavy noise (2 cm)
            tgt_pts += rng.normal(0, 2e-2, size=tgt_pts.shape)
        elif mode == 3:  # mirrored target -> the SVD solution is a reflection, fixed by Vt[2] *= -1
            tgt_pts = tgt_pts * np.array([1.0, 1.0, -1.0]) + rng.normal(0, 1e-3, size=tgt_pts.shape)
        elif mode == 4:  # random (non key-point) reference sets
            a = rng.uniform(-0.2, 0.2, size=(k, 3))
            tgt_pts = (Rgt @ a.T).T + rng.uniform(-1, 1, size=3) + rng.normal(0, 1e-3, size=(k, 3))
        elif mode == 5:  # coplanar reference (z = 0) with noise on the target
            a = rng.uniform(-0.2, 0.2, size=(k, 3))
            a[:, 2] = 0.0
            tgt_pts = (Rgt @ a.T).T + rng.normal(0, 1e-3, size=(k, 3))
        elif mode == 6:  # large offsets (metres), tiny object
            a = a + 3.0
            tgt_pts = (Rgt @ a.T).T + 5.0
        Rr, tr = T.get_rigid_transform_3D(a, tgt_pts)
        qr = T.get_q_from_matrix(Rr)
        ref[b, :k], tgt[b, :k], K[b] = a, tgt_pts, k
        R[b], t[b], q[b], kind[b] = Rr, tr, qr, mode
    return dict(ref=ref, tgt=tgt, K=K, R=R, t=t, q=q, kind=kind, reference_key_points=kp)


def gen_quat_avg(rng):
    B, Mmax = 64, 20
    Q = np.zeros((B, Mmax, 4))
    W = np.zeros((B, Mmax))
    M = np.zeros(B, dtype=np.int32)
    out = np.zeros((B, 4))
    poses = np.zeros((B, Mmax, 7))
    pose_avg = np.zeros((B, 7))
    for b in range(B):
        m = int(rng.integers(2, Mmax + 1))
        base = rand_quat(rng)
        for i in range(m):
            qi = base + rng.normal(0, 0.05 if b % 2 else 0.3, size=4)
            qi /= np.linalg.norm(qi)
            if rng.random() < 0.3:
                qi = -qi
            Q[b, i] = qi
        W[b, :m] = rng.uniform(0.1, 1.0, size=m) if b % 3 else 1.0
        M[b] = m
        out[b] = Cal.compute_quaternions_weighted_average(Q[b, :m], W[b, :m])
        poses[b, :m, :3] = rng.uniform(-1, 1, size=(m, 3))
        poses[b, :m, 3:] = Q[b, :m]
        pose_avg[b] = Cal.compute_poses_average(poses[b, :m], W[b, :m])
    return dict(Q=Q, W=W, M=M, out=out, poses=poses, pose_avg=pose_avg)


def gen_add(rng):
    B, Pmax = 32, 512
    pts = np.zeros((B, Pmax, 3))
    P = np.zeros(B, dtype=np.int32)
    gt = np.zeros((B, 7))
    pr = np.zeros((B, 7))
    add = np.zeros(B)
    for b in range(B):
        p = int(rng.integers(16, Pmax + 1))
        pts[b, :p] = rng.uniform(-0.1, 0.1, size=(p, 3))
        gt[b, :3] = rng.uniform(-1, 1, size=3)
        gt[b, 3:] = rand_quat(rng)
        pr[b, :3] = gt[b, :3] + rng.normal(0, 0.01, size=3)
        qn = gt[b, 3:] + rng.normal(0, 0.02, size=4)
        pr[b, 3:] = qn / np.linalg.norm(qn)
        P[b] = p
        add[b] = Mx.compute_ADD_np(pts[b, :p], gt[b], pr[b])
    return dict(points=pts, P=P, gt=gt, pred=pr, add=add)


def gen_fps(rng):
    # numpy FPS (utils/data.py:13-34) on an EE-crop-like cloud; the random first index is recovered from the output
    n_np, s_np = 4096, 2048
    cloud = (rng.uniform(-0.5, 0.5, size=(n_np, 3)) * np.array([0.10, 0.22, 0.13])).astype(np.float32)
    np.random.seed(7)
    idx_np = Dat.get_farthest_point_sample_idx(cloud, s_np)
    # torch FPS (model/pointnet2_utils.py:65-86), batch of 3 clouds
    B, N, S = 3, 1024, 256
    xyz = rng.uniform(-1, 1, size=(B, N, 3)).astype(np.float32)
    torch.manual_seed(11)
    idx_t = P2.farthest_point_sample(torch.from_numpy(xyz), S).numpy()
    return dict(np_cloud=cloud, np_idx=idx_np.astype(np.int64), np_start=np.int64(idx_np[0]), t_xyz=xyz,
                t_idx=idx_t.astype(np.int64), t_start=idx_t[:, 0].astype(np.int64))


def gen_ball_query(rng):
    B, N, S, nsample, radius = 2, 1024, 128, 32, 0.2
    xyz = rng.uniform(-1, 1, size=(B, N, 3)).astype(np.float32)
    torch.manual_seed(3)
    t = torch.from_numpy(xyz)
    fps = P2.farthest_point_sample(t, S)
    new_xyz = P2.index_points(t, fps)
    idx = P2.query_ball_point(radius, nsample, t, new_xyz).numpy()
    # margin of every point to the ball surface, so a test can skip borderline points (matmul rounding is unspecified)
    d = P2.square_distance(new_xyz, t).numpy()
    return dict(xyz=xyz, new_xyz=new_xyz.numpy(), idx=idx.astype(np.int64), radius=np.float64(radius),
                nsample=np.int64(nsample), min_margin=np.float64(np.abs(d - np.float32(radius ** 2)).min()))


def gen_metrics(rng):
    n, B = 4000, 12
    gts = rng.integers(0, 3, size=(B, n))
    preds = gts.copy()
    acc = np.zeros(B); prec = np.zeros(B); rec = np.zeros(B); cls = np.zeros((B, 3, 3))
    for b in range(B):
        flip = rng.random(n) < (0.02 * b)
        preds[b, flip] = rng.integers(0, 3, size=flip.sum())
        if b == 3:
            preds[b, preds[b] == 2] = 1  # a class that is never predicted
        if b == 5:
            gts[b, gts[b] == 0] = 1  # a class absent from the ground truth
        r = Mx.compute_segmentation_metrics(gts[b], preds[b])
        acc[b], prec[b], rec[b] = r["accuracy"], r["precision"], r["recall"]
        for ci, cn in enumerate(["background", "arm", "ee"]):
            c = r["class_results"][cn]
            cls[b, ci] = [c["accuracy"], float(c["precision"]), float(c["recall"])]
    P = 64
    gt_pose = np.zeros((P, 7)); pr_pose = np.zeros((P, 7)); dpos = np.zeros(P); dang = np.zeros(P)
    for i in range(P):
        gt_pose[i, :3] = rng.uniform(-1, 1, 3); gt_pose[i, 3:] = rand_quat(rng)
        pr_pose[i, :3] = gt_pose[i, :3] + rng.normal(0, 0.05, 3)
        q = gt_pose[i, 3:] + rng.normal(0, 0.1 * (1 + i % 5), 4)
        pr_pose[i, 3:] = (q / np.linalg.norm(q)) * (1 if i % 2 else -1) * (1.0 if i % 3 else 2.5)  # sign / scale
        r = Mx.compute_pose_metrics(gt_pose[i], pr_pose[i])
        dpos[i], dang[i] = r["dist_position"], r["angle_diff"]
    kp_gt = rng.uniform(-0.1, 0.1, size=(6, 3)); kp_cls = np.array([0, 2, 3, 5]); kp_pred = kp_gt[kp_cls] + rng.normal(0, 0.01, (4, 3))
    return dict(seg_gt=gts, seg_pred=preds, seg_accuracy=acc, seg_precision=prec, seg_recall=rec, seg_class=cls,
                gt_pose=gt_pose, pred_pose=pr_pose, dist_position=dpos, angle_diff=dang, kp_gt=kp_gt, kp_cls=kp_cls,
                kp_pred=kp_pred, kp_error=np.float64(Mx.compute_kp_error(kp_gt, kp_pred, kp_cls)))


def gen_preprocess(rng):
    pts = rng.normal(size=(1000, 3)).astype(np.float32)
    centred, off = Pre.center_at_origin(pts)
    rgb255 = rng.integers(0, 256, size=(500, 3)).astype(np.float32)
    rgb01 = rng.uniform(0, 1, size=(500, 3)).astype(np.float32)
    return dict(points=pts, centred=centred, offset=off, rgb255=rgb255, rgb255_out=Pre.normalize_colors(rgb255),
                rgb01=rgb01, rgb01_out=Pre.normalize_colors(rgb01), norm_points=Pre.normalize_points(pts))


def gen_calib_chain(rng):
    """utils/transformation.py:225-266 (get_base2cam_pose, transform_pose2pose), :63-101 (matrix <-> pose helpers):
    the chain InferenceEngine.calibrate runs per frame (app/inference_engine.py:152-244)."""
    B = 48
    ee2cam = np.zeros((B, 7)); ee2robot = np.zeros((B, 7)); base2cam = np.zeros((B, 7)); p2p = np.zeros((B, 7))
    mat = np.zeros((B, 4, 4)); mat_inv = np.zeros((B, 4, 4)); pose_back = np.zeros((B, 7))
    for b in range(B):
        ee2cam[b, :3] = rng.uniform(-1, 1, 3); ee2cam[b, 3:] = rand_quat(rng)
        ee2robot[b, :3] = rng.uniform(-1, 1, 3); ee2robot[b, 3:] = rand_quat(rng)
        if b % 5 == 0:  # near-180-degree rotations exercise every branch of the matrix -> quaternion conversion
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
            ang = np.pi - 1e-3 * (b // 5)
            ee2cam[b, 3:] = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
        base2cam[b] = T.get_base2cam_pose(ee2cam[b], ee2robot[b])
        p2p[b] = T.transform_pose2pose(ee2cam[b], ee2robot[b])
        mat[b] = T.get_transformation_matrix(ee2cam[b], switch_w=False)
        mat_inv[b] = T.get_transformation_matrix_inverse(mat[b])
        pose_back[b] = T.get_pose_from_matrix(mat[b])
    return dict(ee2cam=ee2cam, ee2robot=ee2robot, base2cam=base2cam, pose2pose=p2p, matrix=mat, matrix_inverse=mat_inv,
                pose_from_matrix=pose_back)


def gen_output_ops(rng):
    """utils/output.py:45-64 get_pred_center (top-8 vote mean, optional quaternion offset) and :81-87
    get_key_point_predictions (softmax over classes, max over points per class, threshold)."""
    n = 3000
    votes = rng.normal(size=(n, 2)).astype(np.float32)
    coords = rng.uniform(-0.5, 0.5, size=(n, 3)).astype(np.float32)
    centre = Out.get_pred_center(torch.from_numpy(votes), coords)
    q = rand_quat(rng).astype(np.float32)
    centre_q = Out.get_pred_center(torch.from_numpy(votes), coords.copy(), ee_r=0.03, q=q)
    m = 2500
    logits = (rng.normal(size=(m, 6)) * 2).astype(np.float32)
    for c, row in zip((0, 2, 3, 5), (17, 400, 1234, 2499)):  # four confident key points, two classes left uncertain
        logits[row, c] += 25.0
    idx, classes, probs = Out.get_key_point_predictions(torch.from_numpy(logits))
    idx9, classes9, probs9 = Out.get_key_point_predictions(torch.from_numpy(logits), conf_th=0.5)
    seg_logits = rng.normal(size=(2000, 3)).astype(np.float32)

    class _Field:
        features = torch.from_numpy(seg_logits)

    preds, conf = Out.get_segmentations_from_tensor_field(_Field())
    return dict(votes=votes, coords=coords, centre=np.asarray(centre, np.float64), q=q,
                centre_q=np.asarray(centre_q, np.float64), kp_logits=logits, kp_idx=np.asarray(idx, np.int64),
                kp_classes=np.asarray(classes, np.int64), kp_probs=np.asarray(probs, np.float32),
                kp_idx_th05=np.asarray(idx9, np.int64), kp_classes_th05=np.asarray(classes9, np.int64),
                kp_probs_th05=np.asarray(probs9, np.float32), seg_logits=seg_logits, seg_preds=preds.astype(np.int64),
                seg_conf=conf.astype(np.float32))


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, fn, seed in [("kabsch", gen_kabsch, 100), ("quat_avg", gen_quat_avg, 101), ("add", gen_add, 102),
                           ("fps", gen_fps, 103), ("ball_query", gen_ball_query, 104),
                           ("preprocess", gen_preprocess, 105), ("metrics", gen_metrics, 106),
                           ("calib_chain", gen_calib_chain, 107), ("output_ops", gen_output_ops, 108)]:
        data = fn(np.random.default_rng(seed))
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **data)
        print(f"{path}: {os.path.getsize(path) / 1024:.1f} KB")


if __name__ == "__main__":
    main()
