"""A9/A10/A12 + A8 parity on the GPU against golden vectors produced by the reference's own functions
(tools/make_golden.py).  Tolerance 1e-4 on pose floats is the north_star's; the solves actually agree to ~1e-12."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_kabsch_vs_reference_golden(gpu, golden):
    from mrcc_amd.utils import transformation as T

    g = golden("kabsch")
    R, t, q = T.get_rigid_transform_3D_batched(g["ref"], g["tgt"], g["K"], device=gpu)
    assert np.abs(R - g["R"]).max() < 1e-4 and np.abs(t - g["t"]).max() < 1e-4
    sign = np.sign((q * g["q"]).sum(axis=1, keepdims=True))
    assert np.abs(q * sign - g["q"]).max() < 1e-4
    # much tighter in practice (float64 Jacobi vs LAPACK)
    assert np.abs(R - g["R"]).max() < 1e-9 and np.abs(t - g["t"]).max() < 1e-9
    assert np.allclose(np.linalg.det(R), 1.0, atol=1e-12)
    # single-problem API with the reference's signature
    k = int(g["K"][3])
    R1, t1 = T.get_rigid_transform_3D(g["ref"][3, :k], g["tgt"][3, :k])
    assert np.abs(R1 - g["R"][3]).max() < 1e-9 and np.abs(t1 - g["t"][3]).max() < 1e-9
    q1 = T.get_q_from_matrix(R1)
    assert min(np.abs(q1 - g["q"][3]).max(), np.abs(q1 + g["q"][3]).max()) < 1e-9


def test_kabsch_round_trip_large_batch(gpu):
    """size-independent property at BASELINE batch sizes: recover a known rigid motion for 512 problems."""
    from mrcc_amd.utils import transformation as T

    rng = np.random.default_rng(0)
    B, K = 512, 6
    ref = rng.uniform(-0.1, 0.1, size=(B, K, 3))
    q = rng.normal(size=(B, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    Rgt = np.stack([T.get_quaternion_rotation_matrix(qi, switch_w=False) for qi in q])
    tgt_t = rng.uniform(-1, 1, size=(B, 3))
    tgt = np.einsum("bij,bkj->bki", Rgt, ref) + tgt_t[:, None, :]
    R, t, qo = T.get_rigid_transform_3D_batched(ref, tgt, np.full(B, K, np.int32), device=gpu)
    assert np.abs(R - Rgt).max() < 1e-10 and np.abs(t - tgt_t).max() < 1e-10
    sign = np.sign((qo * q).sum(axis=1, keepdims=True))
    assert np.abs(qo * sign - q).max() < 1e-10
    # device tensors in, device tensors out (no download, no host synchronisation): the same numbers
    Rd, td, qd = T.get_rigid_transform_3D_batched(torch.from_numpy(ref).to(gpu), torch.from_numpy(tgt).to(gpu), device=gpu,
                                                  as_tensors=True)
    assert Rd.is_cuda and Rd.dtype == torch.float64
    assert np.array_equal(Rd.cpu().numpy(), R) and np.array_equal(td.cpu().numpy(), t) and np.array_equal(qd.cpu().numpy(), qo)


def test_quaternion_average_vs_reference_golden(gpu, golden):
    from mrcc_amd.utils import calibration as C

    g = golden("quat_avg")
    out = C.compute_quaternions_weighted_average_batched(g["Q"], g["W"], g["M"], device=gpu)
    sign = np.sign((out * g["out"]).sum(axis=1, keepdims=True))
    assert np.abs(out * sign - g["out"]).max() < 1e-9
    b = 5
    m = int(g["M"][b])
    pose = C.compute_poses_average(g["poses"][b, :m], g["W"][b, :m])
    assert np.abs(pose[:3] - g["pose_avg"][b, :3]).max() < 1e-12
    assert min(np.abs(pose[3:] - g["pose_avg"][b, 3:]).max(), np.abs(pose[3:] + g["pose_avg"][b, 3:]).max()) < 1e-9


def test_add_metric_vs_reference_golden(gpu, golden):
    from mrcc_amd.utils import metrics as M

    g = golden("add")
    add = M.compute_ADD_batched(g["points"], g["P"], g["gt"], g["pred"], device=gpu)
    assert np.abs(add - g["add"]).max() < 1e-12
    p = int(g["P"][0])
    assert abs(M.compute_ADD_np(g["points"][0, :p], g["gt"][0], g["pred"][0]) - g["add"][0]) < 1e-12


def test_fps_vs_reference_golden(gpu, golden):
    from mrcc_amd.model import pointnet2_utils as P2
    from mrcc_amd.utils import data as D

    g = golden("fps")
    idx = D.get_farthest_point_sample_idx(g["np_cloud"], len(g["np_idx"]), start=int(g["np_start"]))
    assert np.array_equal(idx, g["np_idx"])  # bit-exact index sequence, 2048 of 4096
    xyz = torch.from_numpy(g["t_xyz"]).to(gpu)
    got = P2.farthest_point_sample(xyz, g["t_idx"].shape[1], start=torch.from_numpy(g["t_start"]).to(gpu))
    assert np.array_equal(got.cpu().numpy(), g["t_idx"])


@pytest.mark.parametrize("N,S", [(700, 64), (4096, 256), (8192, 300), (16384, 128), (20000, 96)])
def test_fps_every_kernel_variant_matches_oracle(gpu, oracle, N, S):
    """Register-resident FPS (4 / 8 / 16 points per thread) and the LDS fallback (N > 16384) give the oracle's index
    sequence, ties included (duplicated points force equal distances)."""
    from mrcc_amd.model import pointnet2_utils as P2

    rng = np.random.default_rng(N)
    xyz = rng.uniform(-1, 1, size=(3, N, 3)).astype(np.float32)
    xyz[:, N // 2:N // 2 + 50] = xyz[:, :50]  # exact duplicates -> argmax ties
    start = np.array([0, N - 1, N // 3], dtype=np.int64)
    got = P2.farthest_point_sample(torch.from_numpy(xyz).to(gpu), S, start=torch.from_numpy(start).to(gpu))
    want = oracle.farthest_point_sample(xyz, S, start)
    assert np.array_equal(got.cpu().numpy(), want)


def test_ball_query_vs_reference_golden(gpu, golden):
    from mrcc_amd.model import pointnet2_utils as P2

    g = golden("ball_query")
    xyz = torch.from_numpy(g["xyz"]).to(gpu)
    new_xyz = torch.from_numpy(g["new_xyz"]).to(gpu)
    got = P2.query_ball_point(float(g["radius"]), int(g["nsample"]), xyz, new_xyz).cpu().numpy()
    # the golden's closest point to a sphere surface is 1.8e-5 away in squared distance (its `min_margin`), far above the
    # float32 rounding of any distance formulation, so every row must be identical
    assert float(g["min_margin"]) > 1e-5
    assert np.array_equal(got, g["idx"]), f"{(got != g['idx']).any(axis=2).sum()} query rows differ"


def test_post_ops_on_cuda_tensors_match_reference(gpu, golden):
    """utils/output.py:45-87 on DEVICE tensors against the reference-generated vectors: get_key_point_predictions through
    sv_key_point_predictions (softmax -> per-class max -> threshold in one pass), get_pred_center and the row-max /
    sigmoid post-op on CUDA inputs."""
    from mrcc_amd.utils import output as Out

    g = golden("output_ops")
    for suffix, th in (("", 0.999), ("_th05", 0.5)):
        idx, classes, probs = Out.get_key_point_predictions(torch.from_numpy(g["kp_logits"]).to(gpu), conf_th=th)
        assert np.array_equal(idx, g["kp_idx" + suffix]) and np.array_equal(classes, g["kp_classes" + suffix])
        assert np.allclose(np.asarray(probs), g["kp_probs" + suffix], atol=1e-6)
    # a strided view (column slice of a wider logits buffer) and ties: the lowest point index wins
    wide = torch.zeros(500, 9, device=gpu)
    wide[:, 2:8] = torch.from_numpy(g["kp_logits"][:500]).to(gpu)
    wide[7, 2:8] = wide[3, 2:8]
    i1, c1, p1 = Out.get_key_point_predictions(wide[:, 2:8], conf_th=0.0)
    ref = torch.from_numpy(g["kp_logits"][:500].copy())
    ref[7] = ref[3]
    sm = ref.softmax(1)
    assert list(c1) == list(range(6)) and np.allclose(np.asarray(p1), sm.max(0)[0].numpy(), atol=1e-6)
    for c in range(6):
        assert i1[c] == int(np.flatnonzero(sm[:, c].numpy() >= sm[:, c].max().item() - 1e-7)[0]) or i1[c] == int(sm[:, c].argmax())
    e_i, e_c, e_p = Out.get_key_point_predictions(torch.zeros(0, 6, device=gpu), conf_th=0.5)
    assert len(e_i) == 0 and len(e_c) == 0
    # get_pred_center: top-8 votes on the device
    c = Out.get_pred_center(torch.from_numpy(g["votes"]).to(gpu), g["coords"])
    assert np.array_equal(np.asarray(c, np.float64), g["centre"])
    cq = Out.get_pred_center(torch.from_numpy(g["votes"]).to(gpu), g["coords"].copy(), ee_r=0.03, q=g["q"])
    assert np.abs(np.asarray(cq, np.float64) - g["centre_q"]).max() < 1e-7

    class _Field:
        features = torch.from_numpy(g["seg_logits"]).to(gpu)

    preds, conf = Out.get_segmentations_from_tensor_field(_Field())
    assert np.array_equal(preds, g["seg_preds"]) and np.allclose(conf, g["seg_conf"], atol=1e-6)


def _ee_model(n, seed):
    """Points on the surface of a 0.10 x 0.22 x 0.13 m box with an asymmetric ridge (a stand-in CAD model)."""
    rng = np.random.default_rng(seed)
    p = rng.uniform(-0.5, 0.5, size=(n, 3))
    ax = rng.integers(0, 3, size=n)
    p[np.arange(n), ax] = np.where(rng.random(n) < 0.5, -0.5, 0.5)
    p *= np.array([0.10, 0.22, 0.13])
    ridge = (p[:, 1] > 0.05) & (p[:, 2] > 0.06)
    p[ridge, 0] += 0.03
    return p.astype(np.float32)


def test_icp_matches_oracle_and_recovers_pose(gpu, oracle):
    from mrcc_amd.utils import icp as I
    from mrcc_amd.utils.transformation import get_quaternion_rotation_matrix

    rng = np.random.default_rng(1)
    cad = _ee_model(3000, 0)
    q = np.array([0.9, 0.2, -0.3, 0.1])
    q /= np.linalg.norm(q)
    R = get_quaternion_rotation_matrix(q, switch_w=False)
    t = np.array([0.3, -0.1, 0.9])
    crop = (_ee_model(2500, 5).astype(np.float64) @ R.T + t + rng.normal(0, 5e-4, size=(2500, 3))).astype(np.float32)
    # initial guess: 6 degrees / 1.5 cm off
    dq = np.array([1.0, 0.03, -0.04, 0.02])
    dq /= np.linalg.norm(dq)
    T0 = np.eye(4)
    T0[:3, :3] = get_quaternion_rotation_matrix(dq, switch_w=False) @ R
    T0[:3, 3] = t + np.array([0.015, -0.01, 0.005])
    T, fit, rmse, iters = I.icp_point2point(cad, crop, T0, device=gpu)
    To, fo, ro, io = oracle.icp_point2point(cad, crop, T0)
    assert iters == io and abs(fit - fo) < 1e-9 and abs(rmse - ro) < 1e-7
    assert np.abs(T - To).max() < 1e-7
    assert np.abs(T[:3, 3] - t).max() < 3e-3 and np.abs(T[:3, :3] - R).max() < 2e-2 and fit > 0.99
    assert abs(np.linalg.det(T[:3, :3]) - 1) < 1e-12
    # max_iterations = 0 only evaluates; the reference-shaped matcher returns a pose
    T1, f1, r1, it1 = I.icp_point2point(cad, crop, T0, max_iterations=0, device=gpu)
    assert it1 == 0 and np.array_equal(T1, T0) and r1 > rmse
    match = I.get_point2point_matcher(cad, device=gpu)
    pose0 = np.concatenate([T0[:3, 3], [1.0, 0, 0, 0]])
    assert match(None, pose0) is pose0
    pose = match(crop, np.concatenate([T0[:3, 3], oracle.get_q_from_matrix(T0[:3, :3])]))
    assert np.abs(pose[:3] - t).max() < 3e-3


def test_calibration_chain_vs_reference_golden(gpu, golden):
    """N2: get_base2cam_pose / transform_pose2pose / get_pose_from_matrix (utils/transformation.py:87-101, 225-266;
    callers app/inference_engine.py:152-244) with the matrix -> quaternion step on the device, against vectors
    produced by the reference's own functions.  1e-4 is the north_star pose tolerance; observed < 1e-9."""
    from mrcc_amd.utils import transformation as T

    def close(a, b):
        q = min(np.abs(a[3:] - b[3:]).max(), np.abs(a[3:] + b[3:]).max())  # q and -q are the same rotation
        return max(np.abs(a[:3] - b[:3]).max(), q)

    g = golden("calib_chain")
    worst = 0.0
    for b in range(len(g["ee2cam"])):
        worst = max(worst, close(T.get_base2cam_pose(g["ee2cam"][b], g["ee2robot"][b]), g["base2cam"][b]))
        worst = max(worst, close(T.transform_pose2pose(g["ee2cam"][b], g["ee2robot"][b]), g["pose2pose"][b]))
        worst = max(worst, close(T.get_pose_from_matrix(g["matrix"][b]), g["pose_from_matrix"][b]))
    assert worst < 1e-9, worst


def test_preprocess_device_path_vs_reference_golden(gpu, golden):
    """A11: utils/preprocess.py:8-56 on rows that already live in HBM (sv_col_stats + sv_center_scale) against vectors
    produced by the reference's own functions.  center/base shifts are exact float32 operations -> bit-exact; /255 and
    the -0.5 shift likewise; the unit-sphere normalisation sums in float64 (the reference: numpy pairwise float32) -> 1e-6."""
    from mrcc_amd.utils import preprocess as P

    g = golden("preprocess")
    t = lambda a: torch.from_numpy(a).to(gpu)
    c, off = P.center_at_origin(t(g["points"]))
    assert c.is_cuda and np.array_equal(c.cpu().numpy(), g["centred"]) and np.array_equal(off.cpu().numpy(), g["offset"])
    b, boff = P.base_at_origin(t(g["points"]))
    assert np.array_equal(b.cpu().numpy(), g["points"] - g["points"].min(axis=0)) and np.array_equal(
        boff.cpu().numpy(), g["points"].min(axis=0))
    assert np.array_equal(P.normalize_colors(t(g["rgb255"])).cpu().numpy(), g["rgb255_out"])
    assert np.array_equal(P.normalize_colors(t(g["rgb01"])).cpu().numpy(), g["rgb01_out"])
    assert np.allclose(P.normalize_points(t(g["points"])).cpu().numpy(), g["norm_points"], atol=1e-6, rtol=0)
    # the min-max branch (negative colour values, utils/preprocess.py:28-32) against the host mirror of the same function
    rng = np.random.default_rng(1)
    neg = rng.uniform(-0.3, 0.9, size=(5000, 3)).astype(np.float32)
    assert np.allclose(P.normalize_colors(t(neg)).cpu().numpy(), P.normalize_colors(neg), atol=2e-7, rtol=0)
    # strided view (a column slice of a wider buffer) and a large input: 2M rows reduce in 489 slabs, deterministically
    wide = rng.normal(size=(2_000_000, 6)).astype(np.float32)
    c1, o1 = P.center_at_origin(t(wide)[:, 1:4])
    want_c, want_o = (lambda p: (p - (p.max(axis=0) + p.min(axis=0)) / 2, (p.max(axis=0) + p.min(axis=0)) / 2))(wide[:, 1:4])
    assert np.array_equal(c1.cpu().numpy(), want_c) and np.array_equal(o1.cpu().numpy(), want_o)
    c2, _ = P.center_at_origin(t(wide)[:, 1:4])
    assert torch.equal(c1, c2)
    # NaN inputs: numpy's min / max propagate NaN, so must the device statistics - a NaN point gives a NaN offset, and a
    # NaN colour steers the data-dependent branches of normalize_colors exactly like the host path
    bad = g["points"].copy()
    bad[17, 1] = np.nan
    cb, ob = P.center_at_origin(t(bad))
    hb, hob = P.center_at_origin(bad)
    assert np.array_equal(ob.cpu().numpy(), hob, equal_nan=True) and np.isnan(ob.cpu().numpy()[1])
    assert np.array_equal(cb.cpu().numpy(), hb, equal_nan=True)
    badc = g["rgb255"].copy()
    badc[5, 2] = np.nan
    assert np.array_equal(P.normalize_colors(t(badc)).cpu().numpy(), P.normalize_colors(badc), equal_nan=True)
    # four columns: only the first three take the min-max branch (utils/preprocess.py:28-30)
    neg4 = rng.uniform(-0.3, 0.9, size=(3000, 4)).astype(np.float32)
    assert np.allclose(P.normalize_colors(t(neg4)).cpu().numpy(), P.normalize_colors(neg4), atol=2e-7, rtol=0)


@pytest.mark.parametrize("B,N,S,C", [(2, 500, 64, 38), (1, 2048, 1024, 256), (3, 70, 3, 5), (1, 64, 300, 1)])
def test_three_nn_interpolation_matches_reference_formulation(gpu, B, N, S, C):
    """A8: PointNetFeaturePropagation's interpolation (model/pointnet2_utils.py:298-305: full distance matrix, sort,
    first three, 1 / (d + 1e-8) weights) on sv_three_nn_interpolate.  Indices exact wherever the three nearest
    distances are separated; values within 1e-5 (torch.matmul's rounding order inside square_distance is unspecified)."""
    from mrcc_amd.model import pointnet2_utils as P2

    g = torch.Generator().manual_seed(B * 1000 + N + S + C)
    xyz1 = torch.rand(B, N, 3, generator=g) - 0.5
    xyz2 = torch.rand(B, S, 3, generator=g) - 0.5
    pts2 = torch.randn(B, S, C, generator=g)
    got = P2.three_nn_interpolate(xyz1.to(gpu), xyz2.to(gpu), pts2.to(gpu)).cpu()
    d, idx = P2.square_distance(xyz1.double(), xyz2.double()).sort(dim=-1)
    d, idx = d[:, :, :3].float(), idx[:, :, :3]
    w = 1.0 / (d + 1e-8)
    w = w / w.sum(dim=2, keepdim=True)
    want = torch.sum(P2.index_points(pts2, idx) * w.view(B, N, 3, 1), dim=2)
    assert got.shape == want.shape == (B, N, C)
    assert (got - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())


def test_topk_indices_selection_equals_a_stable_descending_sort(gpu):
    """sv_topk_indices (utils/output.py:45-64 get_pred_center: `out[:, 1].sort(descending=True)[1][:8]` as a two-stage
    selection): the k largest entries of a strided column, largest first, ties to the lower row - against a stable descending
    sort on the host, for sizes around the 8 192-value chunk of the first stage, many ties, negative values, infinities, fewer
    rows than k, and k up to 64."""
    from mrcc_amd.utils.output import topk_indices

    rng = np.random.default_rng(7)
    for n, k, kind in ((200_000, 8, "normal"), (8192, 8, "ties"), (8193, 16, "ties"), (50_000, 64, "normal"), (5, 8, "normal"),
                       (1, 8, "normal"), (20_000, 8, "neg"), (3000, 8, "inf")):
        if kind == "ties":
            col = rng.integers(0, 5, size=n).astype(np.float32)
        elif kind == "neg":
            col = -np.abs(rng.normal(size=n)).astype(np.float32) - 1.0
        else:
            col = rng.normal(size=n).astype(np.float32)
        if kind == "inf":
            col[[17, 900]] = np.inf
            col[[3, 2999]] = -np.inf
        votes = np.zeros((n, 4), np.float32)
        votes[:, 1] = col
        got = topk_indices(torch.from_numpy(votes).to(gpu)[:, 1], k).cpu().numpy()
        want = np.argsort(-col.astype(np.float64), kind="stable")[:k]  # descending, ties in ascending row order
        assert got.shape == want.shape and np.array_equal(got, want), (n, k, kind)
