"""BASELINE.json configurations at their FULL sizes against the oracle.

  Cfg-2  200k points / 2 cm through the real RobotNetSegmentation(MinkUNet18D): logits bit-exact, labels exact — the
         only place where the chip-filling instances (conv_fwd_dual_kernel<64,32,4,3>, <64,4,3>, <64,4,2>) walk 12-13
         channel chunks per offset (Cin 384 / 416) and where <128,4,2> runs at all, so those instances meet the oracle
         here; plus direct launches that FORCE each tile shape onto multi-chunk layers (SV_CONV_FORCE).
  Cfg-5  500k points / 1 cm (305k voxels) through the same head: every distinct layer shape of the forward pass is
         compared with the oracle on a random row subset (the oracle computes single output rows from the GPU's own
         input tensor, so the check is bit-exact and costs seconds), labels = argmax, duplicate-frame equality.
  Cfg-3  64 Cfg-2-sized frames (200k points each, 12.8M points, ~5.6M voxels) in ONE sparse tensor through the
         segmentation and vote heads + 64 Kabsch problems: batched == per-frame bit-exactly.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _coords(pts, scale, b=0):
    return np.concatenate([np.full((len(pts), 1), b, np.float32), pts * np.float32(scale)], axis=1)


class _Recorder:
    """Wraps mrcc_amd.nn.conv_forward: runs the real launch and keeps (inputs, output) of the calls `keep` selects."""

    def __init__(self, svnn, keep):
        self.svnn, self.keep, self.orig = svnn, keep, svnn.conv_forward
        self.calls, self.names = [], []

    def __enter__(self):
        self.svnn.conv_forward = self
        return self

    def __exit__(self, *exc):
        self.svnn.conv_forward = self.orig

    def __call__(self, feats, weight3, plan, V_out, scale=None, shift=None, residual=None, act=0, slope=0.01, out=None):
        from mrcc_amd import profiling

        res = self.orig(feats, weight3, plan, V_out, scale, shift, residual, act, slope, out)
        K, Cin, Cout = weight3.shape
        Vpad = plan.Vpad if plan is not None else (max(V_out, 1) + 127) // 128 * 128
        name = profiling.conv_kernel_config(Cout, Vpad, Cin, K)
        self.names.append(name)
        sig = (name, K, Cin, Cout, residual is not None)
        if self.keep(sig, len(self.names) - 1):
            self.calls.append(dict(sig=sig, feats=feats, W=weight3, plan=plan, V_out=V_out, scale=scale, shift=shift,
                                   residual=residual, act=act, slope=slope, out=res))
        return res


def _plan_key(cm, plan):
    for key, p in cm.plans.items():
        if p is plan:
            return key
    raise AssertionError("plan not owned by this coordinate manager")


def _oracle_nbr(frame, key):
    if key[0] in ("k3", "k3split"):  # a two-pass plan (sparse.SplitPlan) is the same kernel map
        ts = key[1]
        while ts not in frame.maps:
            frame.down(max(frame.maps))
        return frame.k3(ts)
    if key[0] == "down":
        ts = key[1]
        while ts not in frame.maps:
            frame.down(max(frame.maps))
        return frame.kdown(ts)
    ts = key[1]
    while ts not in frame.maps:
        frame.down(max(frame.maps))
    return frame.kup(ts)


def _check_call_rows(oracle, frame, cm, c, rows):
    """One recorded launch vs the oracle on output rows `rows` (bit-exact)."""
    np_ = lambda t: None if t is None else t.detach().cpu().numpy()
    feats = np_(c["feats"].contiguous())
    nbr = None if c["plan"] is None else np.ascontiguousarray(_oracle_nbr(frame, _plan_key(cm, c["plan"]))[:, rows])
    if nbr is None:  # dense rows: the oracle's identity map on the subset
        feats = feats[rows]
    res = np_(c["residual"])
    want = oracle.conv(feats, np_(c["W"]), nbr, len(rows), np_(c["scale"]), np_(c["shift"]),
                       None if res is None else np.ascontiguousarray(res[rows]), c["act"], c["slope"])
    got = c["out"][torch.from_numpy(rows).to(c["out"].device)].cpu().numpy()
    assert np.array_equal(got, want), f"{c['sig']}: max diff {np.abs(got - want).max()}"


@pytest.fixture(scope="module")
def cfg2(gpu, oracle):
    import bench
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME

    model = bench.build_model(gpu)
    pts, rgb, lab = mrcc_amd.synth.gen_room(200_000, 2.4, 0)
    c4 = _coords(pts, 50)
    field = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(c4),
                           quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE, device=gpu)
    x = field.sparse()
    return dict(model=model, pts=pts, rgb=rgb, lab=lab, c4=c4, field=field, x=x)


def test_cfg2_full_network_bit_exact(gpu, oracle, cfg2):
    from mrcc_amd import nn as svnn

    model, field, x = cfg2["model"], cfg2["field"], cfg2["x"]
    with torch.no_grad(), _Recorder(svnn, lambda sig, i: False) as rec:
        out = model(x)
        label, conf = out.slice_argmax(field)
    used = set(rec.names)
    for inst in ("conv_fwd_dual_kernel<64, 32, 4, 3>", "conv_fwd_kernel<64, 4, 3>",
                 "conv_fwd_kernel<16, 4, 3>", "conv_fwd_kernel<128, 4, 2>",
                 "conv_first_mfma_kernel<3, 32>", "linear_narrow_kernel<3>"):
        assert inst in used, (inst, sorted(used))
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    ref = oracle.predict_segmentation(sd, cfg2["pts"], cfg2["rgb"], 50)
    assert np.array_equal(x.coordinate_map.keys.cpu().numpy().view(np.uint64), ref["vox"]["keys"])
    assert np.array_equal(field.inverse_mapping.cpu().numpy(), ref["vox"]["inverse"])
    got = out.F.cpu().numpy()
    assert got.shape == ref["logits"].shape == (88_113, 3)
    assert np.array_equal(got, ref["logits"]), f"max diff {np.abs(got - ref['logits']).max()}"
    assert np.array_equal(label.cpu().numpy(), ref["label"])
    assert np.allclose(conf.cpu().numpy(), ref["conf"], atol=1e-6)
    cfg2["oracle_ref"] = ref


# (plan kind, level, Cin, Cout, forced instance "tm,wn,nt"): every wide tile shape on a layer with SEVERAL channel
# chunks per offset, including a partial last chunk (Cin 400 = 12.5 chunks of 32; 200 = 3.125 chunks of 64)
FORCED = [
    ("k3", 0, 416, 384, "64,4,3"), ("k3", 0, 400, 384, "64,4,3"), ("k3", 0, 96, 384, "128,4,3"),
    ("k3", 1, 416, 384, "64,4,2"), ("k3", 1, 384, 384, "64,4,2"), ("k3", 1, 200, 384, "64,4,3"),
    ("k3", 0, 96, 256, "128,4,2"), ("dense", 0, 384, 256, "128,4,2"), ("dense", 0, 256, 1024, "128,4,2"),
    ("dense", 0, 416, 384, "64,4,3"), ("up", 1, 384, 384, "64,4,3"), ("k3", 1, 96, 384, "32,4,3"),
    ("k3", 2, 448, 384, "16,4,3"), ("k3", 2, 384, 384, "32,2,3"), ("k3", 0, 64, 64, "64,4,1"),
]


@pytest.mark.parametrize("kind,level,cin,cout,force", FORCED)
def test_forced_instances_multi_chunk_bit_exact(gpu, oracle, cfg2, kind, level, cin, cout, force):
    from mrcc_amd import nn as svnn

    cm = cfg2["x"].coordinate_manager
    if "frame" not in cfg2:
        cfg2["frame"] = oracle.Frame(oracle.voxelize(cfg2["c4"])["coords"])
    frame = cfg2["frame"]
    ts = 1 << level
    for l in range(level + (1 if kind == "up" else 0)):
        frame.down(1 << l)
        cm.stride_map(2 << l)
    if kind == "k3":
        plan, nbr, V_in, V_out, K = cm.plan_k3(ts), frame.k3(ts), cm.stride_map(ts).V, cm.stride_map(ts).V, 27
    elif kind == "up":  # transposed conv from tensor stride 2 ts onto the existing map at ts
        cm.plan_down(ts)
        plan, nbr, V_in, V_out, K = cm.plan_up(2 * ts), frame.kup(2 * ts), cm.stride_map(2 * ts).V, cm.stride_map(ts).V, 8
    else:
        plan, nbr, V_in, V_out, K = None, None, cm.stride_map(ts).V, cm.stride_map(ts).V, 1
    rng = np.random.default_rng(sum(map(ord, kind)) * 7 + level * 1000003 + cin * 1009 + cout)
    x = rng.normal(size=(V_in, cin)).astype(np.float32)
    W = (rng.normal(size=(K, cin, cout)) * np.sqrt(2.0 / (K * cout))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, size=cout).astype(np.float32)
    shift = rng.normal(size=cout).astype(np.float32)
    res = rng.normal(size=(V_out, cout)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(gpu)
    os.environ["SV_CONV_FORCE"] = force
    try:
        got = svnn.conv_forward(t(x), t(W), plan, V_out, t(scale), t(shift), t(res), 1).cpu().numpy()
    finally:
        del os.environ["SV_CONV_FORCE"]
    want = oracle.conv(x, W, nbr, V_out, scale, shift, res, oracle.ACT_RELU)
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


def test_cfg5_500k_1cm_layers_and_duplicate_frames(gpu, oracle, cfg2):
    """BASELINE configs[4]: 500k points at 1 cm through the segmentation head."""
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd import nn as svnn

    model = cfg2["model"]
    pts, rgb, _ = mrcc_amd.synth.gen_room(500_000, 2.4, 0)
    c4 = _coords(pts, 100)
    seen = set()

    def keep(sig, i):
        if sig in seen:
            return False
        seen.add(sig)
        return True

    with torch.no_grad():
        field = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(c4), device=gpu)
        x = field.sparse()
        with _Recorder(svnn, keep) as rec:
            out = model(x)
        label, _ = out.slice_argmax(field)
    V = x.F.shape[0]
    assert 290_000 < V < 320_000 and out.F.shape == (V, 3) and torch.isfinite(out.F).all()
    vox = oracle.voxelize(c4)
    assert np.array_equal(x.coordinate_map.keys.cpu().numpy().view(np.uint64), vox["keys"])
    assert np.array_equal(field.inverse_mapping.cpu().numpy(), vox["inverse"])
    # labels = first row maximum of the voxel's logits (utils/output.py:67-73)
    assert np.array_equal(label.cpu().numpy(), out.F.cpu().numpy().argmax(1)[vox["inverse"]])
    frame = oracle.Frame(vox["coords"])
    cm = x.coordinate_manager
    assert len(rec.calls) >= 20, [c["sig"] for c in rec.calls]
    rng = np.random.default_rng(5)
    for c in rec.calls:
        n = min(c["V_out"], 4096 if c["sig"][2] * c["sig"][3] > 64 * 64 else 16384)
        rows = np.sort(rng.choice(c["V_out"], size=n, replace=False))
        _check_call_rows(oracle, frame, cm, c, rows)
    del rec
    # two copies of the frame in one batch: bit-identical to the single run, copy by copy
    with torch.no_grad():
        c2 = np.concatenate([_coords(pts, 100, 0), _coords(pts, 100, 1)])
        f2 = ME.TensorField(torch.from_numpy(np.concatenate([rgb, rgb])), torch.from_numpy(c2), device=gpu)
        both = model(f2.sparse())
        assert both.F.shape[0] == 2 * V
        assert torch.equal(both.F[:V], out.F) and torch.equal(both.F[V:], out.F)


CFG3_FRAMES = 64


def test_cfg3_batch64_fullsize_frames(gpu, oracle, cfg2):
    """BASELINE configs[2] with Cfg-2-sized frames: CFG3_FRAMES = 64 frames of 200k points in one sparse tensor
    (batch column, data/alivev2.py:358-383) through the segmentation and the vote head, then 64 Kabsch problems."""
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd.model.robotnet_vote import RobotNetVote
    from mrcc_amd.utils import transformation as T

    B = CFG3_FRAMES
    seg = cfg2["model"]
    torch.manual_seed(4)
    vote = RobotNetVote(3).to(gpu).eval()
    frames = [mrcc_amd.synth.gen_room(200_000, 2.4, s) for s in range(B)]
    with torch.no_grad():
        coords = torch.from_numpy(np.concatenate([_coords(f[0], 50, b) for b, f in enumerate(frames)]))
        feats = torch.from_numpy(np.concatenate([f[1] for f in frames]))
        field = ME.TensorField(feats, coords, device=gpu)
        x = field.sparse()
        assert int(x.C[:, 0].max()) == B - 1 and x.F.shape[0] > B * 85_000
        # 5.6M voxels x 416 channels = 9 GB per tensor: beyond the 2 GB extent of the buffer-addressed instances.  The
        # layers run as batch ranges with their own plans (ConvPlan.chunks) / row ranges (dense layers), so EVERY launch
        # of the wide layers must still report the FAST form - the instances the bench measures - and the chip-filling
        # level-0/1 layers the dual-body kernel.
        from mrcc_amd import profiling

        profiling.INSTANCE_LOG = log = []
        try:
            s_out = seg(x)
        finally:
            profiling.INSTANCE_LOG = None
        wide = [e for e in log if e[3] >= 32 and e[3] % 4 == 0 and e[4] >= 32]
        assert wide and all(e[1]["fast"] == 1 for e in wide), [e for e in wide if e[1]["fast"] != 1][:3]
        big = [e for e in log if e[2] >= 8 and e[3] >= 384 and e[4] == 384 and e[5] > 500_000]  # K = 9: offset-range passes
        assert big and all(e[0] == "conv_fwd_dual_kernel<64, 32, 4, 3>" for e in big), big[:3]
        assert len(log) > 70  # 61 layers; the widest ones as several batch ranges and offset-range passes
        labels, _ = s_out.slice_argmax(field)
        v_out = vote(x)
        bs = x.coordinate_manager.batch_offsets(1, B).tolist()
        for b in (0, B // 2 - 1, B - 1):
            f1 = ME.TensorField(torch.from_numpy(frames[b][1]), torch.from_numpy(_coords(frames[b][0], 50)), device=gpu)
            x1 = f1.sparse()
            s1 = seg(x1)
            assert x1.F.shape[0] == bs[b + 1] - bs[b]
            assert torch.equal(s_out.F[bs[b]:bs[b + 1]], s1.F)
            l1, _ = s1.slice_argmax(f1)
            assert torch.equal(labels[b * 200_000:(b + 1) * 200_000], l1)
            assert torch.equal(v_out.F[bs[b]:bs[b + 1]], vote(x1).F)
        if "oracle_ref" in cfg2:  # frame 0 of the batch is the Cfg-2 frame the oracle has already labelled
            assert np.array_equal(labels[:200_000].cpu().numpy(), cfg2["oracle_ref"]["label"])
    # pose stage: 64 Kabsch problems in one launch (key points of 64 seeded end-effector poses, 1 mm noise)
    crops = [mrcc_amd.synth.gen_ee_crop(s, n=16) for s in range(B)]
    ref = np.repeat(mrcc_amd.synth.REFERENCE_KEY_POINTS[None], B, axis=0)
    tgt = np.stack([c[3] for c in crops])
    R, t, q = T.get_rigid_transform_3D_batched(ref, tgt, device=gpu)
    for b in range(B):
        Ro, to = oracle.get_rigid_transform_3D(ref[b], tgt[b])
        assert np.abs(R[b] - Ro).max() < 1e-9 and np.abs(t[b] - to).max() < 1e-9
        pose_o = np.concatenate([to, oracle.get_q_from_matrix(Ro)])
        pose_g = np.concatenate([t[b], q[b] if np.dot(q[b], pose_o[3:]) > 0 else -q[b]])
        pts_b = crops[b][0].astype(np.float64)
        assert oracle.compute_ADD_np(pts_b, pose_o, pose_g) < 1e-4  # ADD(GPU pose, oracle pose) <= 1e-4 m


def test_cfg1_sample_frame_size_seg_bit_exact(gpu, oracle):
    """BASELINE configs[0]: one dataset-sample-sized frame (~80k points, 2 cm voxels) through robotnet_segmentation -
    the reference's CPU-runnable plumbing case - logits bit-exact, labels exact against the oracle."""
    import bench
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME

    model = bench.build_model(gpu)
    pts, rgb, _ = mrcc_amd.synth.gen_room(80_000, 1.6, 11)
    with torch.no_grad():
        field = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(_coords(pts, 50)), device=gpu)
        x = field.sparse()
        out = model(x)
        label, conf = out.slice_argmax(field)
    ref = oracle.predict_segmentation({k: v.cpu() for k, v in model.state_dict().items()}, pts, rgb, 50)
    assert 30_000 < x.F.shape[0] < 50_000
    assert np.array_equal(x.coordinate_map.keys.cpu().numpy().view(np.uint64), ref["vox"]["keys"])
    assert np.array_equal(out.F.cpu().numpy(), ref["logits"])
    assert np.array_equal(label.cpu().numpy(), ref["label"])
    assert np.allclose(conf.cpu().numpy(), ref["conf"], atol=1e-6)


def test_cfg5_vote_and_pose_legs(gpu, oracle):
    """BASELINE configs[4] beyond the seg head: a 500k-point / 1 cm labelled scene through RobotNetVote
    (model/robotnet_vote.py:62-71; every distinct layer shape against the oracle on a row subset, as for the seg head
    above), then the end-effector crop of that frame through the engine's stages in the reference's order
    (app/inference_engine.py:281-382): rotation head -> translation -> key-point head -> key-point selection
    (utils/output.py:81-87) -> Kabsch (sv_kabsch_batched), each against the oracle."""
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd import nn as svnn
    from mrcc_amd.app.inference_engine import InferenceEngine
    from mrcc_amd.model.robotnet_vote import RobotNetVote
    from mrcc_amd.utils import preprocess
    from mrcc_amd.utils.config import Config

    sc = mrcc_amd.synth.gen_scene(5, n_bg=480_000, n_arm=6_000, n_ee=14_000)
    pts, rgb01 = sc["points"], sc["rgb"]
    assert len(pts) == 500_000
    rgb = preprocess.normalize_colors(rgb01)
    # ---- vote leg on the whole frame
    torch.manual_seed(8)
    vote = RobotNetVote(3).to(gpu).eval()
    seen = set()

    def keep(sig, i):
        if sig in seen:
            return False
        seen.add(sig)
        return True

    c4 = _coords(pts, 100)
    with torch.no_grad():
        field = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(c4), device=gpu)
        x = field.sparse()
        with _Recorder(svnn, keep) as rec:
            out = vote(x)
        vlabel, _ = out.slice_argmax(field)
    V = x.F.shape[0]
    assert V > 250_000 and out.F.shape == (V, 2) and torch.isfinite(out.F).all()
    vox = oracle.voxelize(c4)
    assert np.array_equal(x.coordinate_map.keys.cpu().numpy().view(np.uint64), vox["keys"])
    frame, cm = oracle.Frame(vox["coords"]), x.coordinate_manager
    rng = np.random.default_rng(6)
    assert len(rec.calls) >= 20
    for c in rec.calls:
        n = min(c["V_out"], 4096 if c["sig"][2] * c["sig"][3] > 64 * 64 else 16384)
        _check_call_rows(oracle, frame, cm, c, np.sort(rng.choice(c["V_out"], size=n, replace=False)))
    assert np.array_equal(vlabel.cpu().numpy(), out.F.cpu().numpy().argmax(1)[vox["inverse"]])
    del rec, out, x, field
    # ---- pose legs on the frame's end-effector crop (ground-truth EE points: random-init labels mark no crop)
    Config.reset()
    Config().update({"INFERENCE": {"ROTATION": {"scale": 100}, "KEY_POINTS": {"scale": 100, "conf_threshold": 0.0},
                                   "SEGMENTATION": {"scale": 100}}})
    try:
        cfg = Config()
        eng = InferenceEngine(allow_random_init=True, seed=9)
        ee_idx = np.where(sc["segmentation"] == 2)[0]
        ee_pts, ee_rgb = pts[ee_idx], rgb[ee_idx]
        assert len(ee_idx) == 14_000

        def oracle_inputs(p, scale):
            c = np.concatenate([np.zeros((len(p), 1), np.float32), (torch.from_numpy(p) * scale).numpy()], 1)
            v = oracle.voxelize(c)
            return v, oracle.voxel_reduce(ee_rgb, v["order"], v["seg_start"], 0)

        # rotation (:446-457)
        q = eng.predict_rotation(ee_pts, torch.from_numpy(ee_rgb))
        p0, _ = preprocess.center_at_origin(ee_pts)
        v, f = oracle_inputs(p0, cfg.INFERENCE.ROTATION.scale)
        fwd = oracle.robotnet_encode_forward if cfg.INFERENCE.ROTATION.encode_only else oracle.robotnet_forward
        want_q = fwd({k: t.cpu() for k, t in eng._rotation_model.state_dict().items()}, f, oracle.Frame(v["coords"]))[0][3:]
        assert np.abs(q - want_q).max() < 1e-4  # north_star tolerance on pose floats
        # translation (:459-507): pure host arithmetic on the crop and q
        pos, _ = eng.predict_translation(ee_pts, torch.from_numpy(ee_rgb), q=q)
        assert pos.shape == (3,) and np.isfinite(pos).all()
        # key points (:509-559): per-point logits of the key-point head, then softmax / per-class max / threshold
        kp_coords, kp_classes, probs = eng.predict_key_points(ee_pts, torch.from_numpy(ee_rgb))
        v, f = oracle_inputs(p0, cfg.INFERENCE.KEY_POINTS.scale)
        logits = oracle.robotnet_segmentation_forward(
            {k: t.cpu() for k, t in eng._key_points_model.state_dict().items()}, f, oracle.Frame(v["coords"]))[v["inverse"]]
        e = np.exp(logits - logits.max(1, keepdims=True))
        sm = e / e.sum(1, keepdims=True)
        assert list(kp_classes) == list(range(6))  # threshold 0: every class reports its best point
        for c in range(6):
            best = np.flatnonzero(sm[:, c] >= sm[:, c].max() - 1e-6)  # float32 softmax: accept ties within rounding
            got_idx = np.flatnonzero((ee_pts == kp_coords[c]).all(1))
            assert len(np.intersect1d(best, got_idx)) >= 1
            assert abs(float(probs[c]) - sm[:, c].max()) < 1e-5
        # Kabsch on the predicted key points (:384-393) against the oracle's SVD
        pose = eng.predict_pose_from_kp(kp_coords, kp_classes)
        Ro, to = oracle.get_rigid_transform_3D(mrcc_amd.synth.REFERENCE_KEY_POINTS[np.asarray(kp_classes)], np.asarray(kp_coords))
        qo = oracle.get_q_from_matrix(Ro)
        assert np.abs(pose[:3] - to).max() < 1e-9
        assert min(np.abs(pose[3:] - qo).max(), np.abs(pose[3:] + qo).max()) < 1e-9
        assert oracle.compute_ADD_np(ee_pts.astype(np.float64), np.concatenate([to, qo]),
                                     np.concatenate([pose[:3], pose[3:] if np.dot(pose[3:], qo) > 0 else -pose[3:]])) < 1e-4
    finally:
        Config.reset()
