"""End-to-end parity of the heads on the GPU vs the oracle's independent restatement of the reference graph.

Same seeded weights (state_dict handed to the oracle), same synthetic cloud.  Logits are BIT-EXACT (identical fmaf
chains), hence labels are exact; pose floats within 1e-4 (north_star tolerance; pooled sums are order-sensitive)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _randomize_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)


def _cloud(n, L, seed, scale):
    import mrcc_amd

    pts, rgb, lab = mrcc_amd.synth.gen_room(n, L, seed)
    coords4 = np.concatenate([np.zeros((n, 1), np.float32), pts * np.float32(scale)], axis=1)
    return pts, rgb, lab, coords4


def test_segmentation_head_bit_exact(gpu, oracle):
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd.model.robotnet_segmentation import RobotNetSegmentation

    torch.manual_seed(1)
    model = RobotNetSegmentation(in_channels=3, num_classes=3)
    _randomize_bn(model, 2)
    model = model.to(gpu).eval()
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    pts, rgb, lab, coords4 = _cloud(6000, 0.5, 3, 50)
    with torch.no_grad():
        field = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(coords4),
                               quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE, device=gpu)
        sin = field.sparse()
        out = model(sin)
        logits_pts = out.slice(field).F.cpu().numpy()
        label, conf = out.slice_argmax(field)
    ref = oracle.predict_segmentation(sd, pts, rgb, 50)
    got_logits = out.F.cpu().numpy()
    assert got_logits.shape == ref["logits"].shape
    assert np.array_equal(got_logits, ref["logits"]), f"max diff {np.abs(got_logits - ref['logits']).max()}"
    assert np.array_equal(label.cpu().numpy(), ref["label"])
    assert np.array_equal(logits_pts, ref["logits"][ref["vox"]["inverse"]])
    assert np.allclose(conf.cpu().numpy(), ref["conf"], atol=1e-6)
    # reference post-op on the sliced field (utils/output.py:67-73) gives the same labels
    assert np.array_equal(logits_pts.argmax(1), ref["label"])


def test_unfused_module_api_matches_fused(gpu):
    """Calling the layers one by one the way the reference's forward does (conv -> bn -> relu) must give the same
    bits as the fused path."""
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd.model.backbone.minkunet import MinkUNet14A

    torch.manual_seed(5)
    net = MinkUNet14A(3, 16).to(gpu).eval()
    _randomize_bn(net, 6)
    pts, rgb, lab, coords4 = _cloud(4000, 0.5, 7, 50)
    with torch.no_grad():
        x = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(coords4), device=gpu).sparse()
        fused = net.conv0p1s1.forward_fused(x, bn=net.bn0, act=1)
        seq = net.relu(net.bn0(net.conv0p1s1(x)))
        assert torch.equal(fused.F, seq.F)
        d_f = net.conv1p1s2.forward_fused(fused, bn=net.bn1, act=1)
        d_s = net.relu(net.bn1(net.conv1p1s2(seq)))
        assert torch.equal(d_f.F, d_s.F) and d_f.tensor_stride == 2
        full = net(x)
        assert full.F.shape == (x.F.shape[0], 16) and torch.isfinite(full.F).all()


def test_pose_heads_within_tolerance(gpu, oracle):
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd.model.robotnet import RobotNet
    from mrcc_amd.model.robotnet_encode import RobotNetEncode

    parts = [_cloud(2500, 0.3, 20 + b, 100) for b in range(3)]
    rgb = np.concatenate([p[1] for p in parts])
    coords4 = np.concatenate([np.concatenate([np.full((len(p[0]), 1), b, np.float32), p[3][:, 1:]], axis=1)
                              for b, p in enumerate(parts)])
    vox = oracle.voxelize(coords4)
    feats = oracle.voxel_reduce(rgb, vox["order"], vox["seg_start"], 0)
    for cls, fwd in ((RobotNet, oracle.robotnet_forward), (RobotNetEncode, oracle.robotnet_encode_forward)):
        torch.manual_seed(9)
        model = cls(in_channels=3, out_channels=7)
        _randomize_bn(model, 10)
        model = model.to(gpu).eval()
        sd = {k: v.cpu() for k, v in model.state_dict().items()}
        with torch.no_grad():
            x = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(coords4), device=gpu).sparse()
            got = model(x).cpu().numpy()
        want = fwd(sd, feats, oracle.Frame(vox["coords"]))
        assert got.shape == (3, 7)
        assert np.allclose(got, want, atol=1e-4, rtol=0), f"max diff {np.abs(got - want).max()}"  # 1e-4: north_star
        assert np.allclose(np.linalg.norm(got[:, 3:7], axis=1), 1.0, atol=1e-5)
        # reference indexing of the result: rot_output[0][3:] (app/inference_engine.py:457)
        assert got[0][3:].shape == (4,)


def _oracle_inputs(oracle, rgb, coords4):
    vox = oracle.voxelize(coords4)
    return vox, oracle.voxel_reduce(rgb, vox["order"], vox["seg_start"], 0)


def test_alive_unet_bit_exact_and_pose_head(gpu, oracle):
    """A4: the 7-level fallback backbone with the reference's default STRUCTURE (m = 32, block_reps = 2,
    config/default.yaml:69-70) against the oracle's restatement of model/backbone/aliveunet.py:177-265, and the
    RobotNet head on top of it (model/robotnet.py:29-30 picks AliveUNet for every non-minkunet backbone name)."""
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd.model.backbone.aliveunet import make_alive_unet
    from mrcc_amd.model.robotnet import make_robotnet

    pts, rgb, lab, coords4 = _cloud(9000, 1.5, 11, 100)  # 150 voxels across: the stride-128 level has several voxels
    vox, feats = _oracle_inputs(oracle, rgb, coords4)
    torch.manual_seed(3)
    net = make_alive_unet(m=32, block_reps=2, bottleneck=False)(3, 8)
    _randomize_bn(net, 4)
    net = net.to(gpu).eval()
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        x = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(coords4), device=gpu).sparse()
        out = net(x)
        assert sorted(x.coordinate_manager.maps) == [1, 2, 4, 8, 16, 32, 64, 128]
    frame = oracle.Frame(vox["coords"])
    want = oracle.alive_unet_forward(sd, feats, frame)
    assert len(frame.maps[128]) > 1
    got = out.F.cpu().numpy()
    assert got.shape == want.shape == (len(vox["keys"]), 32)
    assert np.array_equal(got, want), f"max diff {np.abs(got - want).max()}"
    # pose head on the AliveUNet body
    from mrcc_amd.utils.config import Config

    Config.reset()
    Config().update({"STRUCTURE": {"bottleneck": False}})  # default.yaml:76 says true, which the reference's own
    torch.manual_seed(5)                                     # AliveUNet cannot run (declared 1088 vs concatenated 992)
    head = make_robotnet(backbone="aliveunet")(3, 7)
    Config.reset()
    assert head.convtr8.kernel.shape == (8, 192, 192)
    _randomize_bn(head, 6)
    head = head.to(gpu).eval()
    sdh = {k: v.cpu() for k, v in head.state_dict().items()}
    with torch.no_grad():
        got = head(ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(coords4), device=gpu).sparse()).cpu().numpy()
    want = oracle.robotnet_forward(sdh, feats, oracle.Frame(vox["coords"]), backbone=oracle.alive_unet_forward)
    assert got.shape == (1, 7) and np.allclose(got, want, atol=1e-4, rtol=0), np.abs(got - want).max()


@pytest.mark.parametrize("name", ["MinkUNet50", "MinkUNet101"])
def test_bottleneck_backbones_bit_exact(gpu, oracle, name):
    """A3: the Bottleneck (expansion 4) U-Nets of the heads' backbone tables (model/robotnet_segmentation.py:20-27,
    model/backbone/minkunet.py:204-211) against the oracle's bottleneck_block restatement."""
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd.model.backbone import minkunet

    pts, rgb, lab, coords4 = _cloud(5000, 0.5, 13, 50)
    vox, feats = _oracle_inputs(oracle, rgb, coords4)
    torch.manual_seed(7)
    net = getattr(minkunet, name)(3, 16)
    _randomize_bn(net, 8)
    net = net.to(gpu).eval()
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    assert "block4.5.conv3.kernel" in sd and sd["block5.0.conv1.kernel"].shape == (256 + 128 * 4, 256)
    with torch.no_grad():
        out = net(ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(coords4), device=gpu).sparse())
    want = oracle.minkunet_forward(sd, feats, oracle.Frame(vox["coords"]))
    got = out.F.cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want), f"max diff {np.abs(got - want).max()}"


def test_reference_shaped_model_code_on_the_installed_namespace(gpu):
    """The drop-in boundary used the way the reference uses it: `import MinkowskiEngine as ME` resolves to this build
    (mrcc_amd.install_as_minkowski_engine()), and a forward pass WRITTEN LIKE the reference's model files - explicit
    conv -> bn -> relu module calls, ME.cat, ME's BasicBlock recipe with `out += residual`
    (model/backbone/minkunet.py:125-187) - gives the same bits as the fused mirror."""
    import importlib
    import sys

    import mrcc_amd

    me = mrcc_amd.install_as_minkowski_engine()
    import MinkowskiEngine as ME
    from MinkowskiEngine.modules.resnet_block import BasicBlock, Bottleneck
    import MinkowskiEngine.MinkowskiOps as MEOps

    assert ME is me and sys.modules["MinkowskiEngine.utils"] is me.utils
    assert BasicBlock.expansion == 1 and Bottleneck.expansion == 4 and hasattr(MEOps, "MinkowskiLinear")
    assert ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE and ME.MinkowskiAlgorithm.SPEED_OPTIMIZED
    from mrcc_amd.model.backbone.minkunet import MinkUNet14A

    def block_forward(blk, x):  # MinkowskiEngine/modules/resnet_block.py BasicBlock.forward, as published
        residual = x
        out = blk.conv1(x)
        out = blk.norm1(out)
        out = blk.relu(out)
        out = blk.conv2(out)
        out = blk.norm2(out)
        if blk.downsample is not None:
            residual = blk.downsample(x)
        out += residual
        return blk.relu(out)

    def stack_forward(seq, x):
        for blk in seq:
            x = block_forward(blk, x)
        return x

    def reference_forward(net, x):  # the statement sequence of model/backbone/minkunet.py:125-187
        out = net.conv0p1s1(x)
        out = net.bn0(out)
        out_p1 = net.relu(out)
        skips = [out_p1]
        out = out_p1
        for conv, bn, block in (("conv1p1s2", "bn1", "block1"), ("conv2p2s2", "bn2", "block2"),
                                ("conv3p4s2", "bn3", "block3"), ("conv4p8s2", "bn4", "block4")):
            out = getattr(net, conv)(out)
            out = getattr(net, bn)(out)
            out = net.relu(out)
            out = stack_forward(getattr(net, block), out)
            skips.append(out)
        skips.pop()
        for conv, bn, block in (("convtr4p16s2", "bntr4", "block5"), ("convtr5p8s2", "bntr5", "block6"),
                                ("convtr6p4s2", "bntr6", "block7"), ("convtr7p2s2", "bntr7", "block8")):
            out = getattr(net, conv)(out)
            out = getattr(net, bn)(out)
            out = net.relu(out)
            out = ME.cat(out, skips.pop())
            out = stack_forward(getattr(net, block), out)
        return net.final(out)

    torch.manual_seed(15)
    net = MinkUNet14A(3, 12)
    _randomize_bn(net, 16)
    net = net.to(gpu).eval()
    pts, rgb, lab, coords4 = _cloud(7000, 0.6, 17, 50)
    with torch.no_grad():
        field = ME.TensorField(features=torch.from_numpy(rgb), coordinates=torch.from_numpy(coords4),
                               quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE,
                               minkowski_algorithm=ME.MinkowskiAlgorithm.SPEED_OPTIMIZED, device=gpu)
        x = field.sparse()
        fused = net(x)
        unfused = reference_forward(net, x)
        assert unfused.tensor_stride == 1 and torch.equal(fused.F, unfused.F)
        assert torch.equal(unfused.slice(field).F, fused.F[field.inverse_mapping])


def test_vote_head_bit_exact_vs_oracle(gpu, oracle):
    """A5: RobotNetVote (model/robotnet_vote.py:62-71) against the oracle's restatement of the head - not only against
    itself (batched vs single, tests/test_gpu_cfg.py).  2 classes (ee_seg data) and 4 classes (:39)."""
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd.model.robotnet_vote import RobotNetVote

    pts, rgb, lab, coords4 = _cloud(5000, 0.45, 31, 100)
    vox, feats = _oracle_inputs(oracle, rgb, coords4)
    for ncls, seed in ((None, 12), (4, 13)):
        torch.manual_seed(seed)
        model = RobotNetVote(3) if ncls is None else RobotNetVote(3, num_classes=ncls)
        _randomize_bn(model, seed + 1)
        model = model.to(gpu).eval()
        sd = {k: v.cpu() for k, v in model.state_dict().items()}
        with torch.no_grad():
            field = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(coords4), device=gpu)
            out = model(field.sparse())
            votes = out.slice(field).F.cpu().numpy()
        want = oracle.robotnet_segmentation_forward(sd, feats, oracle.Frame(vox["coords"]))
        got = out.F.cpu().numpy()
        assert got.shape == want.shape == (len(vox["keys"]), 2 if ncls is None else ncls)
        assert np.array_equal(got, want), f"max diff {np.abs(got - want).max()}"
        assert np.array_equal(votes, want[vox["inverse"]])


def test_pose_head_config_branches(gpu, oracle):
    """A6 branches of the pose heads that the default configuration never takes:
    STRUCTURE.use_joint_angles (model/robotnet.py:49-50,68-71: 9 joint angles behind the pooled features),
    STRUCTURE.compute_confidence (10 outputs, sigmoid on [:, 7:], :77) and DATA.voxelize_position
    (model/robotnet_encode.py:100-119: position * quantization_size in eval)."""
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd.model.robotnet import make_robotnet, make_robotnet_encode
    from mrcc_amd.utils.config import Config

    parts = [_cloud(2000, 0.3, 40 + b, 100) for b in range(2)]
    rgb = np.concatenate([p[1] for p in parts])
    coords4 = np.concatenate([np.concatenate([np.full((len(p[0]), 1), b, np.float32), p[3][:, 1:]], axis=1)
                              for b, p in enumerate(parts)])
    vox, feats = _oracle_inputs(oracle, rgb, coords4)
    ja = np.random.default_rng(3).uniform(-3, 3, size=(2, 9)).astype(np.float32)
    Config.reset()
    Config().update({"STRUCTURE": {"use_joint_angles": True, "compute_confidence": True},
                     "DATA": {"voxelize_position": True, "quantization_size": 0.02}})
    try:
        for make, fwd, kw in ((make_robotnet, oracle.robotnet_forward, {}),
                              (make_robotnet_encode, oracle.robotnet_encode_forward, {"quantization_size": 0.02})):
            torch.manual_seed(21)
            model = make("minkunet")(in_channels=3, out_channels=10)
            assert model.pose_regression[0].in_features == model.pose_regression_input_size
            assert model.pose_regression_input_size % 128 == 9  # 384 + 9 / 256 + 9
            _randomize_bn(model, 22)
            model = model.to(gpu).eval()
            sd = {k: v.cpu() for k, v in model.state_dict().items()}
            with torch.no_grad():
                x = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(coords4), device=gpu).sparse()
                got = model((x, torch.from_numpy(ja).to(gpu))).cpu().numpy()
            want = fwd(sd, feats, oracle.Frame(vox["coords"]), joint_angles=ja, **kw)
            assert got.shape == want.shape == (2, 10)
            assert np.allclose(got, want, atol=1e-4, rtol=0), f"max diff {np.abs(got - want).max()}"
            assert ((got[:, 7:] > 0) & (got[:, 7:] < 1)).all()  # confidences went through the sigmoid
            assert np.allclose(np.linalg.norm(got[:, 3:7], axis=1), 1.0, atol=1e-5)
            if kw:  # the position really was scaled: undo it and compare with the unscaled oracle
                raw = fwd(sd, feats, oracle.Frame(vox["coords"]), joint_angles=ja)
                assert np.allclose(got[:, :3], raw[:, :3] * np.float32(0.02), atol=1e-5)
            # joint angles matter: a different vector moves the output
            with torch.no_grad():
                other = model((x, torch.from_numpy(ja[::-1].copy()).to(gpu))).cpu().numpy()
            assert np.abs(other - got).max() > 1e-6
    finally:
        Config.reset()
