"""An INDEPENDENT end-to-end check of the sparse semantics this build defines (the sparse oracle is parity-unpinned:
MinkowskiEngine cannot be installed).

A whole MinkUNet14A is evaluated as a DENSE-GRID float64 torch network - conv3d / conv3d(stride 2) /
conv_transpose3d(stride 2) masked to the active voxel sets, BatchNorm(eval), ReLU, channel concatenation, residual adds,
1x1 convs - on a 32^3 grid that contains negative coordinates, and compared with the GPU path's logits (1e-4).  The
dense model shares NO code with oracle/sv_oracle.py or the HIP path: no hash, no kernel map, no stride map, no Morton
order.  What it pins is composition: which voxel is whose neighbour, floor on negative coordinates in the stride-2 maps,
the transposed convolution landing on the encoder's map, the order of ME.cat, skip wiring over four levels, batch
separation.  The graph below is written from /root/reference/model/backbone/minkunet.py:125-187 and resnet.py:95-127.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

G, OFF = 32, 16  # grid size and the shift that makes coordinates in [-16, 16) non-negative (a multiple of 2^4, so that
                 # floor(c / 2^l) * 2^l of the sparse side is plain integer division on the shifted dense side)


def _dense_weights3(kernel, transposed=False):
    """ME kernel [K, Cin, Cout], offset index x fastest -> torch conv3d weight on a [N, C, Z, Y, X] tensor."""
    K, cin, cout = kernel.shape
    k = round(K ** (1 / 3))
    w = kernel.double().reshape(k, k, k, cin, cout)  # [z, y, x, ci, co]
    return w.permute(3, 4, 0, 1, 2).contiguous() if transposed else w.permute(4, 3, 0, 1, 2).contiguous()


class DenseNet:
    def __init__(self, sd, masks):
        self.sd, self.masks = {k: v.detach().cpu() for k, v in sd.items()}, masks

    def bn(self, x, name):
        g = lambda s: self.sd[f"{name}.bn.{s}"].double().view(1, -1, 1, 1, 1)
        return (x - g("running_mean")) / torch.sqrt(g("running_var") + 1e-5) * g("weight") + g("bias")

    def conv(self, x, name, level):  # 3x3x3 (or 1x1x1) stride-1 convolution on the level's active set
        w = self.sd[name + ".kernel"]
        if w.dim() == 2:
            out = F.conv3d(x, w.double().t().reshape(w.shape[1], w.shape[0], 1, 1, 1))
        else:
            out = F.conv3d(x, _dense_weights3(w), padding=1)
        return out * self.masks[level]

    def down(self, x, name, level):  # kernel 2 stride 2: children -> parent at level + 1
        return F.conv3d(x, _dense_weights3(self.sd[name + ".kernel"]), stride=2) * self.masks[level + 1]

    def up(self, x, name, level):  # transposed kernel 2 stride 2 onto the EXISTING map of level - 1
        return F.conv_transpose3d(x, _dense_weights3(self.sd[name + ".kernel"], True), stride=2) * self.masks[level - 1]

    def block(self, x, name, level):
        m = self.masks[level]
        if name + ".conv3.kernel" in self.sd:  # Bottleneck (resnet_block.Bottleneck): 1x1 - 3x3 - 1x1, expansion 4
            out = F.relu(self.bn(self.conv(x, name + ".conv1", level), name + ".norm1")) * m
            out = F.relu(self.bn(self.conv(out, name + ".conv2", level), name + ".norm2")) * m
            out = self.bn(self.conv(out, name + ".conv3", level), name + ".norm3") * m
            res = x
            if name + ".downsample.0.kernel" in self.sd:
                res = self.bn(self.conv(x, name + ".downsample.0", level), name + ".downsample.1") * m
            return F.relu(out + res) * m
        # BasicBlock
        out = F.relu(self.bn(self.conv(x, name + ".conv1", level), name + ".norm1")) * m
        out = self.bn(self.conv(out, name + ".conv2", level), name + ".norm2") * m
        res = x
        if name + ".downsample.0.kernel" in self.sd:
            res = self.bn(self.conv(x, name + ".downsample.0", level), name + ".downsample.1") * m
        return F.relu(out + res) * m

    def stack(self, x, name, level):
        i = 0
        while f"{name}.{i}.conv1.kernel" in self.sd:
            x = self.block(x, f"{name}.{i}", level)
            i += 1
        return x

    def forward(self, x, n=4, alive=False, stop="full"):
        """n stride-2 levels; alive: AliveUNet's names / block widths (aliveunet.py:177-265), no `final`;
        stop = "encoder" (deepest encoder tensor), "except_final" (last decoder block) or "full" (the default)."""
        m = self.masks
        out_p1 = F.relu(self.bn(self.conv(x, "conv0p1s1", 0), "bn0")) * m[0]
        skips, out = [out_p1], out_p1
        for i in range(1, n + 1):
            out = F.relu(self.bn(self.down(out, f"conv{i}p{2 ** (i - 1)}s2", i - 1), f"bn{i}")) * m[i]
            out = self.stack(out, f"block{i}", i)
            skips.append(out)
        out = skips.pop()
        if stop == "encoder":
            return out
        for j in range(n, 2 * n):
            level = 2 * n - j  # input level of the transposed conv
            name = f"convtr{j}" if alive else f"convtr{j}p{2 ** level}s2"
            out = F.relu(self.bn(self.up(out, name, level), f"bntr{j}")) * m[level - 1]
            out = torch.cat([out, skips.pop()], dim=1)
            out = self.stack(out, f"block{j + 1}", level - 1)
        if alive or stop == "except_final":
            return out
        w, b = self.sd["final.kernel"].double(), self.sd["final.bias"].double().view(1, -1, 1, 1, 1)
        out = (F.conv3d(out, w.t().reshape(w.shape[1], w.shape[0], 1, 1, 1)) + b) * m[0]
        if "regression.0.linear.weight" not in self.sd:
            return out
        # classification head (model/robotnet_segmentation.py:55-64): LeakyReLU -> Linear 256->1024 -> LeakyReLU -> Linear
        lin = lambda t, n_: F.conv3d(t, self.sd[f"regression.{n_}.linear.weight"].double()[:, :, None, None, None],
                                     self.sd[f"regression.{n_}.linear.bias"].double())
        out = F.leaky_relu(out, 0.01)
        out = F.leaky_relu(lin(out, 0), 0.01)
        return lin(out, 2) * m[0]


def _cloud(seed, n):
    """integer voxel coordinates in [-16, 16)^3: a sphere shell, a slab and scattered single voxels."""
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    shell = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(10.5, 12.5, size=(n, 1)) + rng.uniform(-2, 2, 3)
    slab = np.concatenate([rng.uniform(-16, 16, size=(n // 2, 2)), rng.uniform(-3.2, -1.1, size=(n // 2, 1))], axis=1)
    lone = rng.uniform(-16, 16, size=(40, 3))
    c = np.floor(np.concatenate([shell, slab, lone])).astype(np.int64)
    c = np.unique(c[(np.abs(c + 0.5) < 16).all(axis=1)], axis=0)
    return c[rng.permutation(len(c))]


def _nets():
    from mrcc_amd.MinkowskiEngine.modules.resnet_block import Bottleneck
    from mrcc_amd.model.backbone.minkunet import MinkUNet14A, MinkUNetBase
    from mrcc_amd.model.robotnet_segmentation import _classification_head

    class BottleneckUNet(MinkUNetBase):
        """the MinkUNet50 / 101 graph (Bottleneck blocks, expansion 4: 1x1 downsample-with-concat widths, 1x1 - 3x3 - 1x1
        stacks) at widths a float64 dense grid can afford"""
        BLOCK = Bottleneck
        LAYERS = (1, 2, 1, 1, 1, 1, 2, 1)
        PLANES = (8, 16, 16, 32, 32, 16, 8, 8)
        INIT_DIM = 16

    return {"minkunet14a": lambda: MinkUNet14A(3, 20),
            "bottleneck_unet": lambda: BottleneckUNet(3, 20),
            # the segmentation head's dense layers (256 -> 1024 -> classes) behind a BasicBlock U-Net
            "seg_head": lambda: _classification_head(MinkUNet14A, lambda: 3, "SegHead14A")(3, num_classes=3)}


@pytest.mark.parametrize("which", ["minkunet14a", "bottleneck_unet", "seg_head", "minkunet14a as offset-range passes"])
def test_unet_matches_dense_grid_float64(gpu, which, monkeypatch):
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd import nn as svnn

    if which.endswith("passes"):
        # every 3x3x3 layer with >= 16 channels on every level as three offset-range passes (sparse.SplitPlan), whatever the
        # map's size: the independent dense network then also pins the accumulator hand-over between the passes
        monkeypatch.setattr(svnn, "SPLIT_RULES", [(0, (9, 18))])
        monkeypatch.setattr(svnn, "SPLIT_MIN_CHANNELS", 16)
        which = "minkunet14a"
    torch.manual_seed(21)
    net = _nets()[which]()
    n_out = 3 if which == "seg_head" else 20
    g = torch.Generator().manual_seed(22)
    with torch.no_grad():
        for mod in net.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
                mod.bias.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
                mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
                mod.running_var.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
    net = net.to(gpu).eval()
    clouds = [_cloud(1, 2500), _cloud(2, 900)]  # two frames in one batch
    assert all(c.min() < -10 and c.max() > 10 for c in clouds)
    rng = np.random.default_rng(3)
    feats = [rng.uniform(-0.5, 0.5, size=(len(c), 3)).astype(np.float32) for c in clouds]
    coords4 = np.concatenate([np.concatenate([np.full((len(c), 1), b, np.int64), c], axis=1)
                              for b, c in enumerate(clouds)])
    with torch.no_grad():
        x = ME.SparseTensor(torch.from_numpy(np.concatenate(feats)), coordinates=torch.from_numpy(coords4).int(),
                            device=gpu)
        out = net(x)
    if svnn.SPLIT_RULES and svnn.SPLIT_RULES[0][0] == 0:
        assert any(k[0] == "k3split" for k in x.coordinate_manager.plans), "the passes were not used"
    got = out.F.cpu().numpy().astype(np.float64)
    oc = out.C.cpu().numpy().astype(np.int64)  # (batch, x, y, z) of every output row
    # ---- the dense side: [N, C, Z, Y, X]
    dense = torch.zeros((2, 3, G, G, G), dtype=torch.float64)
    m0 = torch.zeros((2, 1, G, G, G), dtype=torch.float64)
    for b, (c, f) in enumerate(zip(clouds, feats)):
        s = c + OFF
        dense[b, :, s[:, 2], s[:, 1], s[:, 0]] = torch.from_numpy(f.astype(np.float64)).t()
        m0[b, 0, s[:, 2], s[:, 1], s[:, 0]] = 1.0
    masks = [m0]
    for _ in range(4):
        masks.append(F.max_pool3d(masks[-1], 2))  # a coarse voxel exists iff one of its 8 children does
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    want_dense = DenseNet(net.state_dict(), masks).forward(dense)
    want = want_dense[oc[:, 0], :, oc[:, 3] + OFF, oc[:, 2] + OFF, oc[:, 1] + OFF].numpy()
    assert got.shape == want.shape == (sum(len(c) for c in clouds), n_out)
    scale = np.abs(want).max()
    err = np.abs(got - want).max()
    assert scale > 0.05 and err < 1e-4 * max(1.0, scale), (err, scale)
    # the sparse output lives exactly on the input's voxel set, level by level
    cm = x.coordinate_manager
    for level in range(5):
        assert cm.stride_map(1 << level).V == int(masks[level].sum().item())


def _randomize_bn(net, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in net.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
                mod.bias.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
                mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
                mod.running_var.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)


def _dense_inputs(clouds, feats, grid, off, levels):
    """[N, C, Z, Y, X] float64 features and the active-set masks of every pyramid level"""
    C = feats[0].shape[1]
    dense = torch.zeros((len(clouds), C, grid, grid, grid), dtype=torch.float64)
    m0 = torch.zeros((len(clouds), 1, grid, grid, grid), dtype=torch.float64)
    for b, (c, f) in enumerate(zip(clouds, feats)):
        s = c + off
        dense[b, :, s[:, 2], s[:, 1], s[:, 0]] = torch.from_numpy(np.asarray(f, dtype=np.float64)).t()
        m0[b, 0, s[:, 2], s[:, 1], s[:, 0]] = 1.0
    masks = [m0]
    for _ in range(levels):
        masks.append(F.max_pool3d(masks[-1], 2))
    return dense, masks


def test_tensorfield_quantisation_mean_and_slice_match_a_dense_scatter(gpu):
    """Entering through ME.TensorField with FLOAT coordinates (negative ones, several points per voxel, two frames) and
    leaving through SparseTensor.slice - the reference's own way in and out (app/inference_engine.py:405-417) - against a
    float64 restatement that shares nothing with the build: voxel = floor(coordinate) per axis (numpy), voxel feature = the
    float64 mean of its points, the dense-grid network above, every point reading its voxel's row.  Pins floor on
    negative coordinates, UNWEIGHTED_AVERAGE, the inverse map and the voxel -> point broadcast."""
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd.model.backbone.minkunet import MinkUNet14A

    torch.manual_seed(31)
    net = MinkUNet14A(3, 12)
    _randomize_bn(net, 32)
    net = net.to(gpu).eval()
    rng = np.random.default_rng(33)
    pts, feats, batch = [], [], []
    for b, n in enumerate((9000, 4000)):
        d = rng.normal(size=(n, 3))
        shell = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(9.0, 12.0, size=(n, 1)) + rng.uniform(-2, 2, 3)
        slab = np.concatenate([rng.uniform(-15.9, 15.9, size=(n // 2, 2)), rng.uniform(-3.5, -0.2, size=(n // 2, 1))], axis=1)
        p = np.concatenate([shell, slab]).astype(np.float32)
        p = p[(np.abs(p) < 15.9).all(axis=1)]
        pts.append(p)
        feats.append(rng.uniform(-0.5, 0.5, size=(len(p), 3)).astype(np.float32))
        batch.append(np.full((len(p), 1), b, np.float32))
    coords4 = np.concatenate([np.concatenate([b_, p], axis=1) for b_, p in zip(batch, pts)])
    allf = np.concatenate(feats)
    assert (coords4[:, 1:] < 0).any() and len(np.unique(np.floor(pts[0]), axis=0)) < 0.6 * len(pts[0])  # several points per voxel
    with torch.no_grad():
        field = ME.TensorField(torch.from_numpy(allf), torch.from_numpy(coords4),
                               quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE, device=gpu)
        got = net(field.sparse()).slice(field).F.cpu().numpy().astype(np.float64)
    # ---- float64 side
    clouds, vfeat, vox_of_point = [], [], []
    for p, f in zip(pts, feats):
        v = np.floor(p.astype(np.float64)).astype(np.int64)
        uniq, inv = np.unique(v, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        acc = np.zeros((len(uniq), 3))
        np.add.at(acc, inv, f.astype(np.float64))
        clouds.append(uniq)
        vfeat.append(acc / np.bincount(inv, minlength=len(uniq))[:, None])
        vox_of_point.append(v)
    dense, masks = _dense_inputs(clouds, vfeat, G, OFF, 4)
    want_dense = DenseNet(net.state_dict(), masks).forward(dense)
    want = np.concatenate([want_dense[b, :, v[:, 2] + OFF, v[:, 1] + OFF, v[:, 0] + OFF].numpy().T
                           for b, v in enumerate(vox_of_point)])
    assert got.shape == want.shape == (len(allf), 12)
    scale = np.abs(want).max()
    assert scale > 0.05 and np.abs(got - want).max() < 1e-4 * max(1.0, scale)


def test_alive_unet_seven_levels_matches_dense_grid_float64(gpu):
    """The 7-level AliveUNet graph (aliveunet.py:177-265: strides to 128, plane table m * (1..7, 7..1), block(j+1) sized
    one table entry further than MinkUNet's, no `final`) as a dense float64 network on a 128^3 grid, at widths the CPU can
    afford (m = 2, INIT_DIM 4).  Coordinates are non-negative here (a 128-cell grid holds exactly one stride-128 voxel only
    when it starts at a multiple of 128; floor on negative coordinates is pinned by the 4-level tests above)."""
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd.model.backbone.aliveunet import make_alive_unet

    Alive = type("ThinAliveUNet", (make_alive_unet(m=2, block_reps=1, bottleneck=False),), {"INIT_DIM": 4})
    torch.manual_seed(41)
    net = Alive(3, 5)
    _randomize_bn(net, 42)
    net = net.to(gpu).eval()
    GA, OA = 128, 0
    rng = np.random.default_rng(43)
    clouds = []
    for n in (5000, 1500):
        d = rng.normal(size=(n, 3))
        shell = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(40.0, 47.0, size=(n, 1)) + 64 + rng.uniform(-8, 8, 3)
        slab = np.concatenate([rng.uniform(0, 128, size=(n // 2, 2)), rng.uniform(55.0, 59.0, size=(n // 2, 1))], axis=1)
        c = np.floor(np.concatenate([shell, slab, rng.uniform(0, 128, size=(60, 3))])).astype(np.int64)
        c = np.unique(c[((c >= 0) & (c < 128)).all(axis=1)], axis=0)
        clouds.append(c[rng.permutation(len(c))])
    feats = [rng.uniform(-0.5, 0.5, size=(len(c), 3)).astype(np.float32) for c in clouds]
    coords4 = np.concatenate([np.concatenate([np.full((len(c), 1), b, np.int64), c], axis=1) for b, c in enumerate(clouds)])
    with torch.no_grad():
        x = ME.SparseTensor(torch.from_numpy(np.concatenate(feats)), coordinates=torch.from_numpy(coords4).int(), device=gpu)
        out = net(x)
    got = out.F.cpu().numpy().astype(np.float64)
    oc = out.C.cpu().numpy().astype(np.int64)
    dense, masks = _dense_inputs(clouds, feats, GA, OA, 7)
    want_dense = DenseNet(net.state_dict(), masks).forward(dense, n=7, alive=True)
    want = want_dense[oc[:, 0], :, oc[:, 3] + OA, oc[:, 2] + OA, oc[:, 1] + OA].numpy()
    assert got.shape == want.shape and got.shape[1] == 2  # block14: PLANES[13] = m
    scale = np.abs(want).max()
    assert scale > 0.01 and np.abs(got - want).max() < 1e-4 * max(1.0, scale)
    cm = x.coordinate_manager
    for level in range(8):
        assert cm.stride_map(1 << level).V == int(masks[level].sum().item())


@pytest.mark.parametrize("encode_only", [False, True])
def test_pooled_pose_heads_match_order_free_float64(gpu, encode_only):
    """RobotNet (U-Net body -> BN + ReLU -> global MAX pool -> MLP) and RobotNetEncode (encoder half -> BN + ReLU -> global
    AVG pool -> MLP) per batch row (model/robotnet.py:62-83, model/robotnet_encode.py:68-119) against a float64
    restatement: dense-grid body, pooling as an ORDER-FREE float64 max / mean over each frame's active voxels, nn.Linear in
    float64, sigmoid on the confidences, L2-normalised quaternion.  1e-4 (north_star tolerance on pose floats)."""
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd.model.robotnet import make_robotnet, make_robotnet_encode
    from mrcc_amd.utils.config import Config

    Config.reset()
    Config().update({"STRUCTURE": {"compute_confidence": True}})
    try:
        torch.manual_seed(51)
        net = (make_robotnet_encode if encode_only else make_robotnet)("minkunet14A")(3, 10)
        _randomize_bn(net, 52)
        net = net.to(gpu).eval()
        clouds = [_cloud(5, 2500), _cloud(6, 900), _cloud(7, 1700)]
        rng = np.random.default_rng(53)
        feats = [rng.uniform(-0.5, 0.5, size=(len(c), 3)).astype(np.float32) for c in clouds]
        coords4 = np.concatenate([np.concatenate([np.full((len(c), 1), b, np.int64), c], axis=1) for b, c in enumerate(clouds)])
        with torch.no_grad():
            x = ME.SparseTensor(torch.from_numpy(np.concatenate(feats)), coordinates=torch.from_numpy(coords4).int(), device=gpu)
            got = net(x).cpu().numpy().astype(np.float64)
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        dense, masks = _dense_inputs(clouds, feats, G, OFF, 4)
        dn = DenseNet(sd, masks)
        body = dn.forward(dense, stop="encoder" if encode_only else "except_final")
        level = 4 if encode_only else 0
        act = F.relu(dn.bn(body, "output_layer.0")) * masks[level]
        pooled = []
        for b in range(len(clouds)):
            sel = masks[level][b, 0] > 0
            rows = act[b][:, sel]  # [C, voxels of frame b at that level]
            pooled.append(rows.mean(dim=1) if encode_only else rows.max(dim=1)[0])
        h = torch.stack(pooled)
        lin = lambda t, n_: t @ sd[f"pose_regression.{n_}.weight"].double().t() + sd[f"pose_regression.{n_}.bias"].double()
        o = lin(F.leaky_relu(lin(h, 0), 0.01), 2)
        o[:, 7:] = torch.sigmoid(o[:, 7:])
        o[:, 3:7] = o[:, 3:7] / o[:, 3:7].norm(dim=1, keepdim=True)
        want = o.numpy()
        assert got.shape == want.shape == (3, 10)
        assert np.abs(got - want).max() < 1e-4 * max(1.0, np.abs(want).max())
    finally:
        Config.reset()
