"""Host-side logic that needs no GPU: model construction / state_dict layout, config, kernel dispatch mirror."""
import numpy as np
import torch


def test_state_dict_layout_matches_minkowski_conventions():
    from mrcc_amd.model.robotnet_segmentation import RobotNetSegmentation

    torch.manual_seed(0)
    m = RobotNetSegmentation(in_channels=3, num_classes=3)
    sd = m.state_dict()
    assert sum(p.numel() for p in m.parameters()) == 80_222_179  # 80.2 M
    assert sd["conv0p1s1.kernel"].shape == (27, 3, 32)
    assert sd["conv1p1s2.kernel"].shape == (8, 32, 32)
    assert sd["convtr4p16s2.kernel"].shape == (8, 256, 384)
    assert sd["block5.0.conv1.kernel"].shape == (27, 512, 384)
    assert sd["block5.0.downsample.0.kernel"].shape == (512, 384)  # kernel_size 1 -> [Cin, Cout]
    assert sd["block8.0.conv1.kernel"].shape == (27, 416, 384)
    assert sd["final.kernel"].shape == (384, 256) and sd["final.bias"].shape == (1, 256)
    assert sd["regression.0.linear.weight"].shape == (1024, 256) and sd["regression.2.linear.bias"].shape == (3,)
    for k in ("bn0.bn.running_mean", "block1.0.norm1.bn.weight", "bntr7.bn.num_batches_tracked"):
        assert k in sd
    # kaiming-normal(fan_out) conv init and BN gamma=1, beta=0 (model/backbone/resnet.py:86-93)
    std = sd["block8.1.conv2.kernel"].std().item()
    assert abs(std - np.sqrt(2.0 / (27 * 384))) / std < 0.02
    assert torch.all(sd["bn0.bn.weight"] == 1) and torch.all(sd["bn0.bn.bias"] == 0)
    # a state_dict round trip through a fresh model (what utils/utils.py:87-126 checkpoint_restore does)
    m2 = RobotNetSegmentation(in_channels=3, num_classes=3)
    m2.load_state_dict(sd)


def test_heads_and_backbones_construct():
    from mrcc_amd.model.backbone import minkunet
    from mrcc_amd.model.backbone.aliveunet import make_alive_unet
    from mrcc_amd.model.robotnet import RobotNet
    from mrcc_amd.model.robotnet_encode import RobotNetEncode
    from mrcc_amd.model.robotnet_vote import RobotNetVote

    assert RobotNetVote(3).regression[2].linear.out_features == 2  # ee_seg -> 2 classes
    r = RobotNet(3, 7)
    assert r.pose_regression[0].in_features == 384 and "output_layer.0.bn.weight" in r.state_dict()
    assert RobotNetEncode(3, 7).pose_regression[0].in_features == 256
    n34 = minkunet.MinkUNet34C(3, 8)
    assert len(n34.block3) == 4 and len(n34.block4) == 6
    b = minkunet.MinkUNet50(3, 8)  # Bottleneck, expansion 4
    assert b.final.kernel.shape == (96 * 4, 8)
    alive = make_alive_unet(m=16, block_reps=1, bottleneck=False)(3, 8)
    assert hasattr(alive, "conv7p64s2") and hasattr(alive, "convtr13") and hasattr(alive, "block14")
    assert alive.block8[0].conv1.kernel.shape == (27, 16 * 7 + 16 * 6, 16 * 6)


# kernel shapes of AliveUNetBase with its default PLANES (32,64,96,128,160,192,224,224,192,160,128,96,64,32), LAYERS 1,
# transcribed from /root/reference/model/backbone/aliveunet.py:62-174 (conv{i}: inplanes->inplanes :76-114; block{i} =
# PLANES[i-1]; convtr7 -> PLANES[7] :116-119; inplanes = PLANES[j+1] + PLANES[13-j] :121,128,...; block{j+1} =
# PLANES[j+1]; convtr{j+1}: inplanes -> PLANES[j+1]; block14 = PLANES[13] on PLANES[13] + INIT_DIM :163-165)
ALIVE_KERNELS = {
    "conv0p1s1": (27, 3, 32), "conv1p1s2": (8, 32, 32), "conv2p2s2": (8, 32, 32), "conv3p4s2": (8, 64, 64),
    "conv4p8s2": (8, 96, 96), "conv5p16s2": (8, 128, 128), "conv6p32s2": (8, 160, 160), "conv7p64s2": (8, 192, 192),
    "block1.0.conv1": (27, 32, 32), "block2.0.conv1": (27, 32, 64), "block3.0.conv1": (27, 64, 96),
    "block4.0.conv1": (27, 96, 128), "block5.0.conv1": (27, 128, 160), "block6.0.conv1": (27, 160, 192),
    "block7.0.conv1": (27, 192, 224),
    "convtr7": (8, 224, 224), "block8.0.conv1": (27, 416, 192), "block8.0.downsample.0": (416, 192),
    "convtr8": (8, 192, 192), "block9.0.conv1": (27, 352, 160), "block9.0.downsample.0": (352, 160),
    "convtr9": (8, 160, 160), "block10.0.conv1": (27, 288, 128), "block10.0.downsample.0": (288, 128),
    "convtr10": (8, 128, 128), "block11.0.conv1": (27, 224, 96), "block11.0.downsample.0": (224, 96),
    "convtr11": (8, 96, 96), "block12.0.conv1": (27, 160, 64), "block12.0.downsample.0": (160, 64),
    "convtr12": (8, 64, 64), "block13.0.conv1": (27, 96, 32), "block13.0.downsample.0": (96, 32),
    "convtr13": (8, 32, 32), "block14.0.conv1": (27, 64, 32), "block14.0.conv2": (27, 32, 32),
    "block14.0.downsample.0": (64, 32), "final": (32, 8),
}


def test_alive_unet_state_dict_matches_reference_table():
    from mrcc_amd.model.backbone.aliveunet import AliveUNetBase, make_alive_unet

    sd = AliveUNetBase(3, 8).state_dict()
    for name, shape in ALIVE_KERNELS.items():
        assert tuple(sd[name + ".kernel"].shape) == shape, name
    assert sum(1 for k in sd if k.endswith(".kernel")) == 1 + 7 + 7 + 14 * 2 + 13 + 1  # 13 blocks change width
    for j, c in zip(range(7, 14), (224, 192, 160, 128, 96, 64, 32)):
        assert sd[f"bntr{j}.bn.weight"].shape == (c,)
    # STRUCTURE.m / block_reps variant (aliveunet.py:268-275): planes m*(1..7,7..1), block_reps blocks per stage
    sd = make_alive_unet(m=16, block_reps=2, bottleneck=False)(3, 8).state_dict()
    assert tuple(sd["block8.0.conv1.kernel"].shape) == (27, 16 * 6 + 16 * 7, 16 * 6)
    assert tuple(sd["convtr8.kernel"].shape) == (8, 96, 96) and tuple(sd["block8.1.conv2.kernel"].shape) == (27, 96, 96)
    assert tuple(sd["block14.1.conv1.kernel"].shape) == (27, 16, 16)


def test_config_override_and_backbone_selection():
    from mrcc_amd.model import _select
    from mrcc_amd.model.backbone import minkunet
    from mrcc_amd.utils.config import Config

    Config.reset()
    cfg = Config()
    assert cfg.INFERENCE.SEGMENTATION.scale == 200 and cfg.DATA.classes == 3
    assert _select.segmentation_backbone() is minkunet.MinkUNet18D
    cfg.update({"INFERENCE": {"SEGMENTATION": {"backbone": "minkunet34C"}}})
    assert _select.segmentation_backbone() is minkunet.MinkUNet34C
    assert cfg.INFERENCE.ROTATION.encode_only is True  # untouched sibling keys survive a recursive override
    Config.reset()


def test_kernel_dispatch_mirror():
    from mrcc_amd.profiling import conv_kernel_config as c

    assert c(384, 88192) == "conv_fwd_kernel<64, 4, 3>"
    # level 1 (thresholds x CONV_WANT_SCALE = 0.3: the taller tile wins earlier inside the two-stream pipeline)
    assert c(384, 26624) == "conv_fwd_kernel<64, 4, 3>"        # Cin unknown: plain launch of the 64-row tile
    assert c(384, 26624, 416, 27) == "conv_fwd_dual_kernel<64, 32, 4, 3>"
    assert c(384, 26624, 384, 27) == "conv_fwd_dual_kernel<64, 32, 4, 3>"
    assert c(384, 6912, 384, 27) == "conv_fwd_kernel<32, 4, 3>"  # level 2: 32-row tiles (chosen inside the pipeline)
    assert c(384, 6912, 448, 27) == "conv_fwd_kernel<32, 4, 3>"
    assert c(384, 1792, 384, 27) == "conv_fwd_kernel<16, 4, 3>"  # level 3: FULL form of the 16-row tile
    assert c(384, 88192, 384, 27) == "conv_fwd_dual_kernel<64, 32, 4, 3>"  # chip-filling: 64-row + 32-row tail tiles
    assert c(384, 88192, 416, 27) == "conv_fwd_dual_kernel<64, 32, 4, 3>"
    assert c(384, 88192, 416, 1) == "conv_fwd_kernel<64, 4, 3>"   # dense rows (no plan): plain launch
    assert c(384, 6912) == "conv_fwd_kernel<32, 4, 3>"
    assert c(192, 26624) == "conv_fwd_kernel<32, 4, 3>"
    assert c(384, 1792) == "conv_fwd_kernel<16, 4, 3>"
    assert c(384, 1792, 512, 27) == "conv_fwd_kernel<16, 4, 3>"
    assert c(192, 1792) == "conv_fwd_kernel<32, 2, 3>"
    assert c(256, 512) == "conv_fwd_kernel<16, 4, 1>"
    assert c(32, 26624) == "conv_fwd_kernel<32, 2, 1>"
    assert c(1024, 88192) == "conv_fwd_kernel<128, 4, 2>"


def test_segmentation_metrics_and_synth():
    import mrcc_amd
    from mrcc_amd.utils import metrics as M

    pts, rgb, lab = mrcc_amd.synth.gen_room(2000, 1.0, 0)
    assert pts.dtype == np.float32 and rgb.min() >= -0.5 and rgb.max() < 0.5 and set(np.unique(lab)) == {0, 1, 2}
    p2, _, _ = mrcc_amd.synth.gen_room(2000, 1.0, 0)
    assert np.array_equal(pts, p2)  # deterministic
    cm = M.confusion_matrix(lab, lab, 3)
    m = M.segmentation_metrics_from_confusion(cm)
    assert m["miou"] == 1.0 and m["accuracy"] == 1.0
    pred = lab.copy()
    pred[:100] = (pred[:100] + 1) % 3
    m = M.compute_segmentation_metrics(lab, pred)
    assert 0.8 < m["miou"] < 1.0 and 0.9 < m["accuracy"] < 1.0  # reference "accuracy" = (sensitivity + specificity)/2
    plain = M.segmentation_metrics_from_confusion(M.confusion_matrix(pred, lab, 3))
    assert abs(plain["accuracy"] - 0.95) < 1e-9 and abs(plain["miou"] - m["miou"]) < 1e-12


def test_kernel_offset_permutation_is_applied_when_loading():
    """nn.KERNEL_OFFSET_PERMUTATION: the hook for a checkpoint whose kernel-offset numbering differs (it was
    documented but never read)."""
    from mrcc_amd import nn as svnn
    from mrcc_amd.model.backbone.minkunet import MinkUNet14A

    torch.manual_seed(0)
    src = MinkUNet14A(3, 4)
    sd = src.state_dict()
    dst = MinkUNet14A(3, 4)
    rev27 = list(range(26, -1, -1))  # e.g. a z-fastest reflected numbering
    try:
        svnn.KERNEL_OFFSET_PERMUTATION = {27: rev27}
        dst.load_state_dict(sd)
    finally:
        svnn.KERNEL_OFFSET_PERMUTATION = None
    assert torch.equal(dst.conv0p1s1.kernel, sd["conv0p1s1.kernel"][rev27])
    assert torch.equal(dst.block8[0].conv2.kernel[3], sd["block8.0.conv2.kernel"][23])
    assert torch.equal(dst.conv1p1s2.kernel, sd["conv1p1s2.kernel"])  # volume 8: no entry -> untouched
    assert torch.equal(dst.final.kernel, sd["final.kernel"]) and torch.equal(sd["conv0p1s1.kernel"], src.conv0p1s1.kernel)
    dst.load_state_dict(sd)  # hook off again: plain copy
    assert torch.equal(dst.conv0p1s1.kernel, sd["conv0p1s1.kernel"])
    import pytest

    with pytest.raises(ValueError):
        try:
            svnn.KERNEL_OFFSET_PERMUTATION = {8: [0, 0, 1, 2, 3, 4, 5, 6]}
            dst.load_state_dict(sd)
        finally:
            svnn.KERNEL_OFFSET_PERMUTATION = None


def test_split_rules_parse_and_apply():
    """nn.SPLIT_RULES ("min_rows:cut[,cut...];..."): which 3x3x3 layers run as offset-range passes (sparse.SplitPlan)."""
    from mrcc_amd import nn as svnn

    rules = svnn._parse_split_rules("20000:9,18;60000:7,14,20;5000:14;")
    assert rules == [(60000, (7, 14, 20)), (20000, (9, 18)), (5000, 14)]  # biggest map first; one cut -> an int
    assert svnn._parse_split_rules("") == []
    old = svnn.SPLIT_RULES
    try:
        svnn.SPLIT_RULES = rules
        assert svnn.split_points_for(88_113) == (7, 14, 20) and svnn.split_points_for(26_552) == (9, 18)
        assert svnn.split_points_for(6_849) == 14 and svnn.split_points_for(1_732) is None
        svnn.SPLIT_RULES = []
        assert svnn.split_points_for(10 ** 9) is None
    finally:
        svnn.SPLIT_RULES = old
    # the default: three passes on maps of at least 20 000 voxels (tools/ab_split.sh)
    assert svnn._parse_split_rules("20000:9,18") == [(20000, (9, 18))]


def test_wired_colour_key_labels_on_the_oracle(oracle):
    """synth.wire_color_keyed_labels + synth.gen_scene(keyed_colors=True), without a GPU: the ORACLE runs the whole
    MinkUNet18D head with the wired weights (every other weight random) on a small colour-keyed scene and predicts the
    ground-truth labels except in the few voxels that mix classes - the labelled frames the engine tests and the bench's
    `predict_full` rely on are labelled by construction, by the network itself."""
    import mrcc_amd
    from mrcc_amd.model.robotnet_segmentation import RobotNetSegmentation
    from mrcc_amd.utils import preprocess

    torch.manual_seed(9)
    model = mrcc_amd.synth.wire_color_keyed_labels(RobotNetSegmentation(in_channels=3, num_classes=3))
    sc = mrcc_amd.synth.gen_scene(2, n_bg=2500, n_arm=400, n_ee=700, room=1.6, keyed_colors=True)
    rgb = preprocess.normalize_colors(sc["rgb"])
    assert rgb.min() >= -0.5 and rgb.max() <= 0.5
    ref = oracle.predict_segmentation({k: v for k, v in model.state_dict().items()}, sc["points"], rgb, 50)
    gt = sc["segmentation"]
    assert (ref["label"] == gt).mean() > 0.97
    assert ((ref["label"] == 2) & (gt == 2)).sum() > 0.95 * (gt == 2).sum() and ((ref["label"] == 1) & (gt == 1)).sum() > 0.9 * (gt == 1).sum()
    # an unkeyed scene under the same weights has no end effector: nothing is bright red
    sc2 = mrcc_amd.synth.gen_scene(2, n_bg=2500, n_arm=400, n_ee=700, room=1.6)
    rgb2 = np.clip(preprocess.normalize_colors(sc2["rgb"]), -0.5, 0.05).astype(np.float32)
    ref2 = oracle.predict_segmentation({k: v for k, v in model.state_dict().items()}, sc2["points"], rgb2, 50)
    assert (ref2["label"] == 0).all()


def test_bench_groups_consecutive_frames_and_keeps_the_remainder():
    """bench.run_frames(group=G): K steps stay K frames - consecutive frames of the resident pool go to
    FramePipeline.prepare_group G at a time, the last group of a region holds the remainder, one group is prepared ahead of
    the one that runs, and the voxel count / label histogram are sums over all frames (host logic only: stub pipeline)."""
    import importlib.util
    import os
    import types

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    frames = [(torch.full((10 + i, 4), float(i)), torch.zeros(10 + i, 3), None, None, None) for i in range(4)]
    events = []

    class Pipe:
        def prepare_group(self, members):
            ids = [int(c[0, 1]) for c, _ in members]
            events.append(("prepare", ids))
            return types.SimpleNamespace(x=types.SimpleNamespace(F=torch.zeros(sum(c.shape[0] for c, _ in members), 1)), ids=ids)

        def run(self, prepared, fn):
            events.append(("run", prepared.ids))
            return torch.tensor([len(prepared.ids), 0, 0])

        def drain(self):
            events.append(("drain",))

    hist = torch.zeros(3, dtype=torch.int64)
    voxels = bench.run_frames(None, Pipe(), frames, 10, hist, group=4)
    groups = [[0, 1, 2, 3], [0, 1, 2, 3], [0, 1]]
    assert [e[1] for e in events if e[0] == "run"] == groups and [e[1] for e in events if e[0] == "prepare"] == groups
    assert [e[0] for e in events] == ["prepare", "run", "prepare", "run", "prepare", "run", "drain"]  # one group ahead
    assert voxels == 2 * (10 + 11 + 12 + 13) + 10 + 11 and int(hist[0]) == 10
    events.clear()
    assert bench.run_frames(None, Pipe(), frames, 5, None, group=1) == 10 + 11 + 12 + 13 + 10
    assert [e[1] for e in events if e[0] == "run"] == [[0], [1], [2], [3], [0]]
