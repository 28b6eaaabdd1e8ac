"""Engine-level behaviour on the GPU: the InferenceEngine mirror, PointNet2SSG on HIP sampling/grouping + MFMA MLPs,
batched frames (the training-format batch column, Cfg-3) and the two-stream frame pipeline."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _randomize_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)


def test_set_abstraction_and_propagation_match_torch(gpu):
    """eval path (HIP FPS / ball query, folded BN, MFMA rows) vs the plain torch ops the reference uses
    (model/pointnet2_utils.py:178-204, :278-317) on the same sampled groups.  fp32 tolerance 1e-4."""
    from mrcc_amd.model import pointnet2_utils as P2

    torch.manual_seed(0)
    sa = P2.PointNetSetAbstraction(64, 0.4, 16, 6 + 3, [16, 16, 32], False).to(gpu).eval()
    _randomize_bn(sa, 1)
    xyz = torch.rand(2, 3, 500, device=gpu) - 0.5
    pts = torch.randn(2, 6, 500, device=gpu)
    torch.manual_seed(5)
    new_xyz, new_pts = sa(xyz, pts)
    # torch restatement on identical groups (same FPS seed)
    torch.manual_seed(5)
    nx, grouped = P2.sample_and_group(64, 0.4, 16, xyz.permute(0, 2, 1), pts.permute(0, 2, 1))
    t = grouped.permute(0, 3, 2, 1)
    for conv, bn in zip(sa.mlp_convs, sa.mlp_bns):
        t = F.relu(bn(conv(t)))
    want = torch.max(t, 2)[0]
    assert torch.equal(new_xyz, nx.permute(0, 2, 1))
    assert (new_pts - want).abs().max().item() < 1e-4
    fp = P2.PointNetFeaturePropagation(32 + 6, [32, 16]).to(gpu).eval()
    _randomize_bn(fp, 2)
    got = fp(xyz, new_xyz, pts, new_pts)
    # reference formulation: full sort, first three
    x1, x2 = xyz.permute(0, 2, 1), new_xyz.permute(0, 2, 1)
    d, idx = P2.square_distance(x1, x2).sort(dim=-1)
    d, idx = d[:, :, :3], idx[:, :, :3]
    w = 1.0 / (d + 1e-8)
    w = w / w.sum(dim=2, keepdim=True)
    interp = torch.sum(P2.index_points(new_pts.permute(0, 2, 1), idx) * w.view(2, 500, 3, 1), dim=2)
    t = torch.cat([pts.permute(0, 2, 1), interp], dim=-1).permute(0, 2, 1)
    for conv, bn in zip(fp.mlp_convs, fp.mlp_bns):
        t = F.relu(bn(conv(t)))
    assert (got - t).abs().max().item() < 1e-4


def test_pointnet2_ssg_forward(gpu):
    from mrcc_amd.model.pointnet2 import PointNet2SSG

    torch.manual_seed(0)
    net = PointNet2SSG(num_classes=6, in_channels=6).to(gpu).eval()
    x = torch.cat([torch.rand(1, 3, 2048, device=gpu) * 0.2, torch.rand(1, 3, 2048, device=gpu) - 0.5], dim=1)
    with torch.no_grad():
        logits, l4 = net(x)
    assert logits.shape == (1, 2048, 6) and l4.shape == (1, 512, 16) and torch.isfinite(logits).all()
    assert "sa1.mlp_convs.0.weight" in net.state_dict() and "fp1.mlp_bns.2.running_var" in net.state_dict()


def test_batched_frames_equal_single_frames(gpu):
    """Cfg-3 format: B frames in one sparse tensor (batch index in column 0, data/alivev2.py:358-383).  Frames never
    interact, so each frame's logits must be BIT-identical to running it alone."""
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd.model.backbone.minkunet import MinkUNet14A

    torch.manual_seed(3)
    net = MinkUNet14A(3, 8).to(gpu).eval()
    _randomize_bn(net, 4)
    frames = [mrcc_amd.synth.gen_room(3000 + 400 * b, 0.5, 30 + b) for b in range(4)]
    with torch.no_grad():
        singles = []
        for pts, rgb, _ in frames:
            c = torch.from_numpy(np.concatenate([np.zeros((len(pts), 1), np.float32), pts * 50], axis=1))
            f = ME.TensorField(torch.from_numpy(rgb), c, device=gpu)
            singles.append(net(f.sparse()).slice(f).F)
        coords = ME.utils.batched_coordinates([torch.from_numpy(p * np.float32(50)) for p, _, _ in frames],
                                              dtype=torch.float32)
        feats = torch.from_numpy(np.concatenate([r for _, r, _ in frames]))
        fb = ME.TensorField(feats, coords, device=gpu)
        out = net(fb.sparse()).slice(fb).F
    start = 0
    for s in singles:
        assert torch.equal(out[start:start + len(s)], s)
        start += len(s)


def test_pipeline_groups_equal_single_frames(gpu):
    """FramePipeline.prepare_group (bench.py --group, the headline's default): frames that arrive one by one - each with its
    own batch column 0 - go through the segmentation network as one sparse tensor; labels, confidences and logits of every
    frame are the bits of its own single-frame pass through the same pipeline, on three compute streams, for group sizes
    that do and do not divide the number of frames."""
    import mrcc_amd
    from mrcc_amd.app.pipeline import FramePipeline
    from mrcc_amd.model.robotnet_segmentation import RobotNetSegmentation

    torch.manual_seed(5)
    net = RobotNetSegmentation(in_channels=3, num_classes=3).to(gpu).eval()
    _randomize_bn(net, 6)
    frames = []
    for b in range(5):
        pts, rgb, _ = mrcc_amd.synth.gen_room(20_000 + 3000 * b, 1.0, 40 + b)
        c = np.concatenate([np.zeros((len(pts), 1), np.float32), pts * np.float32(50)], axis=1)
        frames.append((torch.from_numpy(c).to(gpu), torch.from_numpy(rgb).to(gpu)))
    pipe = FramePipeline(gpu, levels=4, compute_streams=3)

    def fn(x, field):
        out = net(x)
        label, conf = out.slice_argmax(field)
        return label, conf, out.slice(field).F

    with torch.no_grad():
        singles = [pipe.run(pipe.prepare(c, f), fn) for c, f in frames]
        pipe.drain()
        for group in (2, 5):
            got = []
            for i in range(0, len(frames), group):
                prepared = pipe.prepare_group(frames[i:i + group])
                assert prepared.sizes == [int(c.shape[0]) for c, _ in frames[i:i + group]]
                label, conf, logits = pipe.run(prepared, fn)
                pipe.drain()
                got += [tuple(t[a:b] for t in (label, conf, logits))
                        for a, b in zip(np.cumsum([0] + prepared.sizes[:-1]), np.cumsum(prepared.sizes))]
            assert len(got) == len(singles)
            for (l1, c1, f1), (l2, c2, f2) in zip(singles, got):
                assert torch.equal(l1, l2) and torch.equal(c1.view(torch.int32), c2.view(torch.int32))
                assert torch.equal(f1.view(torch.int32), f2.view(torch.int32))
    assert all(float(c[0, 0]) == 0.0 for c, _ in frames)  # the callers' frames are not modified


def test_inference_engine_pipeline(gpu, oracle):
    import mrcc_amd
    from mrcc_amd.app.dto import PointCloudDTO, ResultDTO
    from mrcc_amd.app.inference_engine import InferenceEngine
    from mrcc_amd.utils.config import Config

    Config.reset()
    Config().update({"INFERENCE": {"SEGMENTATION": {"scale": 50}, "ROTATION": {"scale": 100},
                                   "KEY_POINTS": {"scale": 100, "conf_threshold": 0.0},
                                   "ee_point_counts_threshold": 64, "SANITY": {"min_num_of_ee_points": 64}}})
    try:
        eng = InferenceEngine(allow_random_init=True, seed=7)
        assert eng.pred_enabled
        _randomize_bn(eng._segmentation_model, 8)
        pts, rgb, _ = mrcc_amd.synth.gen_room(5000, 0.5, 11)
        rgb01 = rgb + 0.5
        seg = eng.predict_segmentation(pts, rgb)
        assert seg.shape == (5000,) and set(np.unique(seg)) <= {0, 1, 2}
        # same labels as the oracle graph + the same largest-cluster rule
        sd = {k: v.cpu() for k, v in eng._segmentation_model.state_dict().items()}
        ref = oracle.predict_segmentation(sd, pts, rgb, 50)["label"].copy()
        ee = np.where(ref == 2)[0]
        ref[ee] = 1
        if len(ee) > 1:
            ref[ee[eng.cluster_util.get_largest_cluster(pts[ee])]] = 2
        assert np.array_equal(seg, ref)
        # EE-crop stages on a synthetic crop
        ee_pts, ee_rgb, pose, kps = mrcc_amd.synth.gen_ee_crop(0, n=2048)
        q = eng.predict_rotation(ee_pts, torch.from_numpy(ee_rgb))
        assert q.shape == (4,) and abs(np.linalg.norm(q) - 1) < 1e-5
        pos, off = eng.predict_translation(ee_pts, None, q=q)
        assert pos.shape == (3,) and off.shape == (3,)
        with pytest.raises(ValueError):
            eng.predict_translation(ee_pts, None, q=None)
        kp_coords, kp_classes, probs = eng.predict_key_points(ee_pts, torch.from_numpy(ee_rgb))
        assert len(kp_coords) == len(kp_classes) <= 6
        # Kabsch from the ground-truth key points recovers the crop's pose (1 mm key-point noise)
        kp_pose = eng.predict_pose_from_kp(kps, np.arange(6))
        assert np.abs(kp_pose[:3] - pose[:3]).max() < 5e-3
        assert min(np.abs(kp_pose[3:] - pose[3:]).max(), np.abs(kp_pose[3:] + pose[3:]).max()) < 5e-2
        Ro, to = oracle.get_rigid_transform_3D(eng.reference_key_points, kps)
        assert np.abs(kp_pose[:3] - to).max() < 1e-9
        assert eng.predict_pose_from_kp(kps[:3], np.arange(3)) is None
        # full predict(): returns a ResultDTO whatever the (random-weight) segmentation says
        res = eng.predict(PointCloudDTO(points=pts, rgb=rgb01, ee2base_pose=np.array([0.1, 0, 0.5, 1.0, 0, 0, 0])))
        assert isinstance(res, ResultDTO) and res.segmentation.shape == (5000,)
        # calibration averaging vs the oracle's restatement of utils/calibration.py
        rng = np.random.default_rng(0)
        dtos = []
        for i in range(5):
            p = np.concatenate([pose[:3] + rng.normal(0, 0.01, 3), pose[3:] + rng.normal(0, 0.01, 4)])
            p[3:] /= np.linalg.norm(p[3:])
            dtos.append(ResultDTO(segmentation=None, ee_pose=p, base_pose=p, key_points_pose=p,
                                  key_points_base_pose=p, is_confident=True))
        cal = eng.calibrate({"pos1": dtos})
        want = oracle.compute_poses_average(np.array([d.base_pose for d in dtos], dtype=np.float32))
        assert np.abs(cal.base_pose[:3] - want[:3]).max() < 1e-6
        assert min(np.abs(cal.base_pose[3:] - want[3:]).max(), np.abs(cal.base_pose[3:] + want[3:]).max()) < 1e-6
        assert cal.pose_camera_link.shape == (7,)
    finally:
        Config.reset()


@pytest.mark.parametrize("streams", [1, 2, 4])
def test_frame_pipeline_matches_direct_path(gpu, streams):
    """prep stream + `streams` compute streams, consecutive frames on consecutive streams with their level-0 stages
    handed over by device-side events (FramePipeline._phase_hook): every frame's logits are BIT-identical to running it
    alone on the default stream, in whatever way the frames overlap."""
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd.app.pipeline import FramePipeline
    from mrcc_amd.model.backbone.minkunet import MinkUNet14A

    torch.manual_seed(1)
    net = MinkUNet14A(3, 4).to(gpu).eval()
    frames = []
    n = 9
    for s in range(n):
        pts, rgb, _ = mrcc_amd.synth.gen_room(4000 + 700 * (s % 4), 0.5, 40 + s)
        c = torch.from_numpy(np.concatenate([np.zeros((len(pts), 1), np.float32), pts * 50], axis=1)).to(gpu)
        frames.append((c, torch.from_numpy(rgb).to(gpu)))
    pipe = FramePipeline(gpu, levels=4, compute_streams=streams)
    assert pipe.stagger_level0
    with torch.no_grad():
        direct = []
        for c, f in frames:
            fld = ME.TensorField(f, c, device=gpu)
            out = net(fld.sparse())
            direct.append((out.F.clone(), out.slice_argmax(fld)[0].clone()))
        nxt = pipe.prepare(*frames[0])
        got = []
        for i in range(n):
            cur = nxt

            def fn(x, fld):
                out = net(x)
                return out.F, out.slice_argmax(fld)[0]
            got.append(pipe.run(cur, fn))
            if i + 1 < n:
                nxt = pipe.prepare(*frames[i + 1])
        pipe.drain()
    for (fa, la), (fb, lb) in zip(direct, got):
        assert torch.equal(fa, fb) and torch.equal(la, lb)


def test_evaluation_harness_end_to_end(gpu):
    """app/test.py recipe on synthetic labelled scenes with the real engine (random weights: the
    numbers are meaningless, the flow and the metric types are what is checked) and with ground-truth segmentation."""
    import mrcc_amd
    from mrcc_amd.app.evaluate import TestApp
    from mrcc_amd.app.inference_engine import InferenceEngine
    from mrcc_amd.utils.config import Config

    Config.reset()
    Config().update({"INFERENCE": {"SEGMENTATION": {"scale": 50}, "ROTATION": {"scale": 100},
                                   "KEY_POINTS": {"scale": 100, "conf_threshold": 0.0},
                                   "ee_point_counts_threshold": 64, "SANITY": {"min_num_of_ee_points": 64}}})
    try:
        eng = InferenceEngine(allow_random_init=True, seed=3)
        frames = [mrcc_amd.synth.gen_scene(s, n_bg=6000, n_arm=800, n_ee=1500) for s in range(3)]
        app = TestApp(eng, evaluate_segmentation=False)  # use the ground-truth segmentation for the EE crop
        out = app.run_tests(frames)
        assert len(out["instances"]) == 3
        for inst in out["instances"].values():
            assert np.isfinite(inst["nn_ADD"]) and 0 <= inst["nn_angle_diff"] <= np.pi + 1e-9
            assert "base_dist_position" in inst
        assert out["calibration"] is not None
        assert set(out["overall"]["nn_ADD"]) == {"mean", "min", "max", "median", "stdev"}
        # with predicted segmentation the per-frame segmentation metrics appear
        out2 = TestApp(eng, ee_point_counts_threshold=10 ** 9).run_tests(frames[:1])
        assert out2["instances"] == {}  # every frame fails the EE-count threshold, as in app/test.py:110-113
    finally:
        Config.reset()


def test_rccl_metrics_gather_single_rank(gpu):
    """The run's one collective on the real backend: RCCL (backend "nccl") all_gather of the float64 metrics record.
    One rank is all a 1-GPU box allows; the 2-rank exchange itself is covered on gloo in tests/test_dist_cpu.py."""
    import os
    import socket

    import torch.distributed as dist
    from mrcc_amd.app import sharding

    if dist.is_initialized():
        pytest.skip("process group already initialised")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        vec = torch.arange(13, dtype=torch.float64, device="cuda")
        parts = [torch.zeros_like(vec)]
        dist.all_gather(parts, vec)
        dist.barrier()
        assert torch.equal(parts[0], vec)
        agg = sharding.gather_metrics({"frames": 3, "elapsed": 0.5, "confusion": np.eye(3, dtype=np.int64), "seed_sum": 7},
                                      device="cuda")
        assert agg["frames"] == 3 and agg["elapsed_max"] == 0.5 and agg["seed_sum"] == 7
    finally:
        dist.destroy_process_group()


def test_engine_stages_match_the_oracle(gpu, oracle):
    """N1: the numbers the evaluation harness reports come from InferenceEngine's stages - pin each stage to the oracle
    on a labelled synthetic scene: segmentation labels (incl. the largest-cluster rule of app/inference_engine.py:419-433)
    exact, the NN rotation within 1e-4 (north_star tolerance), the key-point pose and its ADD against the oracle's SVD."""
    import mrcc_amd
    from mrcc_amd.app.inference_engine import InferenceEngine
    from mrcc_amd.utils import metrics as M
    from mrcc_amd.utils import preprocess
    from mrcc_amd.utils.config import Config

    Config.reset()
    Config().update({"INFERENCE": {"SEGMENTATION": {"scale": 50}, "ROTATION": {"scale": 100},
                                   "KEY_POINTS": {"scale": 100, "conf_threshold": 0.0},
                                   "ee_point_counts_threshold": 64, "SANITY": {"min_num_of_ee_points": 64}}})
    try:
        cfg = Config()
        eng = InferenceEngine(allow_random_init=True, seed=3)
        sc = mrcc_amd.synth.gen_scene(1, n_bg=6000, n_arm=800, n_ee=1500)
        pts = sc["points"]
        rgb = preprocess.normalize_colors(sc["rgb"])
        # ---- segmentation
        got = eng.predict_segmentation(pts, rgb)
        sd = {k: v.cpu() for k, v in eng._segmentation_model.state_dict().items()}
        ref = oracle.predict_segmentation(sd, pts, rgb, cfg.INFERENCE.SEGMENTATION.scale)
        want = ref["label"].copy()
        ee = np.where(want == 2)[0]
        want[ee] = 1
        if len(ee) > 1:
            want[ee[eng.cluster_util.get_largest_cluster(pts[ref["label"] == 2])]] = 2
        assert np.array_equal(got, want)
        # ---- rotation head on the ground-truth end-effector crop
        ee_idx = np.where(sc["segmentation"] == 2)[0]
        ee_pts, ee_rgb = pts[ee_idx], rgb[ee_idx]
        q = eng.predict_rotation(ee_pts, torch.from_numpy(ee_rgb))
        p = ee_pts
        if cfg.INFERENCE.ROTATION.center_at_origin:
            p, _ = preprocess.center_at_origin(p)
        c4 = np.concatenate([np.zeros((len(p), 1), np.float32), (torch.from_numpy(p) * cfg.INFERENCE.ROTATION.scale).numpy()], 1)
        vox = oracle.voxelize(c4)
        feats = oracle.voxel_reduce(ee_rgb, vox["order"], vox["seg_start"], 0)
        sdr = {k: v.cpu() for k, v in eng._rotation_model.state_dict().items()}
        fwd = oracle.robotnet_encode_forward if cfg.INFERENCE.ROTATION.encode_only else oracle.robotnet_forward
        want_q = fwd(sdr, feats, oracle.Frame(vox["coords"]))[0][3:]
        assert q.shape == want_q.shape and np.abs(q - want_q).max() < 1e-4
        # ---- key-point pose (Kabsch on the device) and the ADD the harness would report for it
        kp_classes = np.array([0, 1, 3, 4, 5])
        pose = eng.predict_pose_from_kp(sc["key_points"][kp_classes], kp_classes)
        Ro, to = oracle.get_rigid_transform_3D(mrcc_amd.synth.REFERENCE_KEY_POINTS[kp_classes], sc["key_points"][kp_classes])
        qo = oracle.get_q_from_matrix(Ro)
        assert np.abs(pose[:3] - to).max() < 1e-9 and min(np.abs(pose[3:] - qo).max(), np.abs(pose[3:] + qo).max()) < 1e-9
        local = (ee_pts.astype(np.float64) - sc["pose"][:3]) @ mrcc_amd.synth.quat_to_matrix(sc["pose"][3:])
        add_gpu = M.compute_ADD_np(local, sc["pose"], pose)
        add_ora = oracle.compute_ADD_np(local, sc["pose"], np.concatenate([to, qo]))
        assert abs(add_gpu - add_ora) < 1e-9 and add_gpu < 5e-3  # 1 mm key-point noise
        assert eng.predict_pose_from_kp(sc["key_points"][:3], np.arange(3)) is None  # fewer than 4 key points (:384-386)
    finally:
        Config.reset()


def _keyed_engine(seed=3):
    """engine whose segmentation network labels `gen_scene(keyed_colors=True)` frames by construction (two colour channels
    wired to the logits, every other weight random: synth.wire_color_keyed_labels)"""
    import mrcc_amd
    from mrcc_amd.app.inference_engine import InferenceEngine

    eng = InferenceEngine(allow_random_init=True, seed=seed)
    mrcc_amd.synth.wire_color_keyed_labels(eng._segmentation_model)
    return eng


def test_keyed_scene_labels_are_the_oracle_s_and_the_ground_truth(gpu, oracle):
    """synth.wire_color_keyed_labels does not bypass the network: the oracle, given the wired weights, runs the whole
    U-Net and predicts the same labels as the GPU path (bit-exact logits), and on a colour-keyed scene those labels are the
    ground truth except in the few voxels that mix end-effector / arm points with background ones."""
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd.utils import preprocess
    from mrcc_amd.utils.config import Config

    Config.reset()
    Config().update({"INFERENCE": {"SEGMENTATION": {"scale": 50}}})
    try:
        eng = _keyed_engine(seed=5)
        sc = mrcc_amd.synth.gen_scene(3, n_bg=7000, n_arm=900, n_ee=1500, keyed_colors=True)
        pts, rgb = sc["points"], preprocess.normalize_colors(sc["rgb"])
        got = eng.predict_segmentation(pts, rgb)
        sd = {k: v.cpu() for k, v in eng._segmentation_model.state_dict().items()}
        ref = oracle.predict_segmentation(sd, pts, rgb, 50)
        with torch.no_grad():
            field = eng._field(pts, rgb, 50)
            logits = eng._segmentation_model(field.sparse()).F.cpu().numpy()
        assert np.array_equal(logits, ref["logits"])  # full network, wired and random weights alike: bit-exact
        want = ref["label"].copy()
        ee = np.where(want == 2)[0]
        want[ee] = 1
        want[ee[eng.cluster_util.get_largest_cluster(pts[ee])]] = 2
        assert np.array_equal(got, want)
        gt = sc["segmentation"]
        assert (got == gt).mean() > 0.97 and ((got == 2) & (gt == 2)).sum() > 0.95 * (gt == 2).sum()
        # and the hidden layers are not idle: apart from the two wired channels the logits' inputs are random features
        h = eng._segmentation_model.forward_except_final(eng._field(pts, rgb, 50).sparse()).F
        assert (h[:, 2:].abs().mean() > 1e-3).item()
    finally:
        Config.reset()


def _same_result(o, r):
    assert np.array_equal(o.segmentation, r.segmentation)
    for name in ("ee_pose", "key_points_pose", "base_pose", "key_points_base_pose"):
        a, b = getattr(o, name), getattr(r, name)
        assert (a is None) == (b is None) and (a is None or np.array_equal(a, b)), name
    assert o.is_confident == r.is_confident
    assert (o.key_points is None) == (r.key_points is None)
    if o.key_points is not None:
        assert len(o.key_points) == len(r.key_points)
        for (ca, pa), (cb, pb) in zip(o.key_points, r.key_points):
            assert ca == cb and np.array_equal(pa, pb)


def test_engine_streaming_equals_per_frame_predict(gpu):
    """The streaming entry points (predict_segmentation_stream / predict_stream: pinned staging, frame i+1 prepared while
    frame i computes, frame i-1's cluster rule + label download finishing, the pose networks of GROUPS of frames as one
    sparse tensor each) return, in order, exactly what the per-frame calls of the reference's loop (app/main.py:432-456)
    return - for frames of different sizes, float64 points, frames without an end-effector crop, and more frames than the
    pipeline is deep.  The scenes' labels are fixed by construction (colour-keyed scenes + wired weights), so every frame
    with an end effector has its crop whatever the random weights are."""
    import mrcc_amd
    from mrcc_amd.app.dto import PointCloudDTO
    from mrcc_amd.utils import preprocess
    from mrcc_amd.utils.config import Config

    Config.reset()
    Config().update({"INFERENCE": {"SEGMENTATION": {"scale": 50}, "ROTATION": {"scale": 100},
                                   "KEY_POINTS": {"scale": 100, "conf_threshold": 0.0},
                                   "ee_point_counts_threshold": 64, "SANITY": {"min_num_of_ee_points": 64}}})
    try:
        eng = _keyed_engine()
        scenes = [mrcc_amd.synth.gen_scene(s, n_bg=5000 + 900 * s, n_arm=700, n_ee=(0 if s == 2 else 1200 + 50 * s),
                                           keyed_colors=True) for s in range(7)]
        frames = []
        for i, sc in enumerate(scenes):
            pts = sc["points"].astype(np.float64) if i % 3 == 1 else sc["points"]
            frames.append((pts, preprocess.normalize_colors(sc["rgb"])))
        want = [eng.predict_segmentation(p, c) for p, c in frames]
        for w, sc in zip(want, scenes):  # by construction: (nearly) the ground truth, the whole end effector one cluster
            n_ee = int((sc["segmentation"] == 2).sum())
            assert abs(int((w == 2).sum()) - n_ee) <= 0.05 * n_ee + 4 and (w == 1).sum() > 0
        for streams in (1, 3):
            got = list(eng.predict_segmentation_stream(iter(frames), compute_streams=streams))
            assert len(got) == len(want)
            for g, w in zip(got, want):
                assert g.dtype == w.dtype and np.array_equal(g, w)
        assert list(eng.predict_segmentation_stream(iter([]))) == []
        # groups of frames per sparse tensor (7 frames: groups of 3 + 3 + 1, of 4 + 3, one of 7): the same labels, in order
        for seg_group in (3, 4, 8):
            got = list(eng.predict_segmentation_stream(iter(frames), compute_streams=3, group=seg_group))
            assert len(got) == len(want)
            for g, w in zip(got, want):
                assert g.dtype == w.dtype and np.array_equal(g, w)
        assert list(eng.predict_segmentation_stream(iter([]), group=4)) == []
        # the whole predict() flow: per frame, streamed in groups of 4 (7 frames: a full and a partial group), of 1 and of 3
        dtos = [PointCloudDTO(points=sc["points"], rgb=sc["rgb"], ee2base_pose=(None if i == 4 else sc["ee2base_pose"]))
                for i, sc in enumerate(scenes)]
        ref = [eng.predict(d) for d in dtos]
        assert ref[2].ee_pose is None and sum(r.ee_pose is not None for r in ref) == 6
        assert all(r.key_points_pose is not None for i, r in enumerate(ref) if i != 2)  # conf_threshold 0: six key points
        assert ref[4].base_pose is None and ref[3].base_pose is not None
        for group, seg_group in ((4, 1), (1, 1), (3, 1), (4, 4), (2, 3)):
            out = list(eng.predict_stream(iter(dtos), group=group, seg_group=seg_group))
            assert len(out) == len(ref)
            for o, r in zip(out, ref):
                _same_result(o, r)
        # the public per-stage calls agree with what predict() used
        sc = scenes[0]
        rgb = preprocess.normalize_colors(sc["rgb"])
        ee = np.where(ref[0].segmentation == 2)[0]
        q = eng.predict_rotation(sc["points"][ee], torch.from_numpy(rgb[ee]))
        assert np.array_equal(q, ref[0].ee_pose[3:].astype(np.float32))
        kpc, kcl, _ = eng.predict_key_points(sc["points"][ee], torch.from_numpy(rgb[ee]))
        assert [int(c) for c in kcl] == [int(c) for c, _ in ref[0].key_points]
        assert all(np.array_equal(a, b) for a, (_, b) in zip(kpc, ref[0].key_points))
    finally:
        Config.reset()


@pytest.mark.perf
def test_engine_streaming_throughput_at_200k_points(gpu):
    """Wall-clock budget of the engine path (not selected by `-m gpu`... it carries the gpu mark through pytestmark, so it is
    deselected explicitly: run with `-m perf`).  The driver-visible figure is bench.py's `engine` block (`within_budget`)."""
    import time

    import mrcc_amd
    from mrcc_amd.app.inference_engine import InferenceEngine
    from mrcc_amd.utils.config import Config

    Config.reset()
    Config().update({"INFERENCE": {"SEGMENTATION": {"scale": 50}}})
    try:
        eng = InferenceEngine(allow_random_init=True, seed=1)
        pool = [mrcc_amd.synth.gen_room(200_000, 2.4, s)[:2] for s in range(4)]
        frames = [pool[i % 4] for i in range(24)]
        list(eng.predict_segmentation_stream(iter(frames[:6])))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = list(eng.predict_segmentation_stream(iter(frames)))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / len(frames) * 1e3
        assert len(got) == 24 and ms <= 20.0, f"{ms:.1f} ms per frame through the engine's streaming path"
    finally:
        Config.reset()


def test_engine_streaming_labels_at_200k_points(gpu):
    """200k-point frames through predict_segmentation_stream: labels equal to the per-frame call, repeated frames equal"""
    import mrcc_amd
    from mrcc_amd.app.inference_engine import InferenceEngine
    from mrcc_amd.utils.config import Config

    Config.reset()
    Config().update({"INFERENCE": {"SEGMENTATION": {"scale": 50}}})
    try:
        eng = InferenceEngine(allow_random_init=True, seed=1)
        pool = [mrcc_amd.synth.gen_room(200_000, 2.4, s)[:2] for s in range(4)]
        frames = [pool[i % 4] for i in range(10)]
        got = list(eng.predict_segmentation_stream(iter(frames)))
        assert np.array_equal(got[1], eng.predict_segmentation(*pool[1]))
        assert np.array_equal(got[5], got[1]) and np.array_equal(got[9], got[1]) and len(got) == 10
    finally:
        Config.reset()
