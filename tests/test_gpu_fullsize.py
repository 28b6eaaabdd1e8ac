"""Full BASELINE sizes (Cfg-2: 200k pts / 2 cm, Cfg-5: 500k pts / 1 cm) where the oracle is too slow: size-independent
properties — voxelisation laws, duplicate-frame equality (frames in a batch never interact), linearity of the conv,
plan invariants — plus the secondary backbones (AliveUNet, vote head)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _coords(pts, scale, b=0):
    return np.concatenate([np.full((len(pts), 1), b, np.float32), pts * np.float32(scale)], axis=1)


@pytest.mark.parametrize("n,scale,expect_v", [(200_000, 50, (85_000, 92_000)), (500_000, 100, (290_000, 320_000))])
def test_fullsize_voxelise_and_maps(gpu, n, scale, expect_v):
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME

    pts, rgb, _ = mrcc_amd.synth.gen_room(n, 2.4, 0)
    c4 = _coords(pts, scale)
    field = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(c4), device=gpu)
    x = field.sparse()
    V = x.F.shape[0]
    assert expect_v[0] < V < expect_v[1]  # 88 102 / 305 659 for its generator
    keys = x.coordinate_map.keys.cpu().numpy().view(np.uint64)
    assert np.all(keys[1:] > keys[:-1])  # canonical, unique
    inv = field.inverse_mapping.cpu().numpy()
    vc = x.C.cpu().numpy()
    assert np.array_equal(vc[inv][:, 1:], np.floor(c4[:, 1:]).astype(np.int32))  # every point in floor(coord)
    cnt = np.bincount(inv, minlength=V)
    assert cnt.min() >= 1 and cnt.sum() == n
    # feature mean against a float64 scatter-add
    sums = np.zeros((V, 3))
    np.add.at(sums, inv, rgb.astype(np.float64))
    assert np.abs(x.F.cpu().numpy() - sums / cnt[:, None]).max() < 1e-5
    # pyramid + plan invariants
    cm = x.coordinate_manager
    prev = V
    for level in range(4):
        ts = 2 ** level
        plan = cm.plan_k3(ts)
        perm = plan.perm.cpu().numpy()
        valid = perm >= 0
        assert valid.sum() == plan.V_out == prev and np.array_equal(np.sort(perm[valid]), np.arange(prev))
        nbr = plan.nbr_s.cpu().numpy()
        assert np.array_equal(nbr[13][valid], perm[valid])  # centre offset = the voxel itself
        assert nbr.max() < prev
        # symmetry of the 3x3x3 map: o --k--> i  implies  i --(26-k)--> o
        k = 5
        src = perm[valid]
        dst = nbr[k][valid]
        has = dst >= 0
        back = np.full(prev, -1, np.int64)
        back[src] = nbr[26 - k][valid]
        assert np.array_equal(back[dst[has]], src[has])
        down = cm.plan_down(ts)
        prev = down.V_out
        assert down.num_pairs() == plan.V_out  # every fine voxel has exactly one parent
        up = cm.plan_up(2 * ts)
        assert up.num_pairs() == plan.V_out and up.V_out == plan.V_out


def test_fullsize_duplicate_frames_and_linearity(gpu):
    """Cfg-2 cloud twice in one batch: bit-identical logits per copy, and equal to the single-frame run."""
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd import nn as svnn
    from mrcc_amd.model.backbone.minkunet import MinkUNet14A

    torch.manual_seed(0)
    net = MinkUNet14A(3, 16).to(gpu).eval()
    pts, rgb, _ = mrcc_amd.synth.gen_room(200_000, 2.4, 0)
    with torch.no_grad():
        f1 = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(_coords(pts, 50)), device=gpu)
        single = net(f1.sparse())
        c2 = np.concatenate([_coords(pts, 50, 0), _coords(pts, 50, 1)])
        f2 = ME.TensorField(torch.from_numpy(np.concatenate([rgb, rgb])), torch.from_numpy(c2), device=gpu)
        x2 = f2.sparse()
        both = net(x2)
        V = single.F.shape[0]
        assert both.F.shape[0] == 2 * V
        assert torch.equal(both.F[:V], single.F) and torch.equal(both.F[V:], single.F)
        l1, _ = single.slice_argmax(f1)
        l2, _ = both.slice_argmax(f2)
        assert torch.equal(l2[:200_000], l1) and torch.equal(l2[200_000:], l1)
        # linearity of a bias-free sparse conv at full size: conv(2x - 3y) == 2 conv(x) - 3 conv(y) (fp32 tolerance)
        cm = single.coordinate_manager
        plan = cm.plan_k3(1)
        a = torch.randn(V, 32, device=gpu)
        b = torch.randn(V, 32, device=gpu)
        W = torch.randn(27, 32, 64, device=gpu) * 0.1
        lhs = svnn.conv_forward(2 * a - 3 * b, W, plan, V)
        rhs = 2 * svnn.conv_forward(a, W, plan, V) - 3 * svnn.conv_forward(b, W, plan, V)
        assert (lhs - rhs).abs().max().item() < 1e-3


def test_fullsize_translation_equivariance(gpu):
    """A sparse conv net commutes with translations of the voxel grid by multiples of its coarsest tensor stride (16 for the
    4-level MinkUNet).  The shifted Cfg-2 frame gets a different canonical row order (rows are sorted by a Morton key of the
    biased coordinates), different hash tables, tiles and plans - and every voxel's logits must still be the same bits,
    because an output element's fma chain is ordered by (kernel offset, channel) only.  Full size, no oracle needed."""
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd.model.backbone.minkunet import MinkUNet14A

    def by_coordinate(c):
        c = c.cpu().numpy()
        return np.lexsort((c[:, 3], c[:, 2], c[:, 1], c[:, 0]))

    torch.manual_seed(0)
    net = MinkUNet14A(3, 16).to(gpu).eval()
    pts, rgb, _ = mrcc_amd.synth.gen_room(200_000, 2.4, 0)
    with torch.no_grad():
        x = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(_coords(pts, 50)), device=gpu).sparse()
        base = net(x)
        ia = by_coordinate(x.C)  # a translation keeps the lexicographic order
        ref_bits = base.F.view(torch.int32).cpu().numpy()[ia]
        for shift in ((32, -48, 16), (-1024, 2048, -16)):
            c = x.C.clone()
            c[:, 1:] += torch.tensor(shift, dtype=c.dtype, device=c.device)
            xs = ME.SparseTensor(x.F.clone(), coordinates=c, device=gpu)
            ib = by_coordinate(xs.C)
            assert np.array_equal(xs.C.cpu().numpy()[ib], c.cpu().numpy()[ia])
            assert not torch.equal(xs.C, c)  # the canonical order really differs: the check below is not vacuous
            assert torch.equal(xs.F.cpu()[ib], x.F.cpu()[ia])
            out = net(xs)
            assert np.array_equal(out.F.view(torch.int32).cpu().numpy()[ib], ref_bits), shift
            for ts in (2, 4, 8, 16):
                assert x.coordinate_manager.stride_map(ts).V == xs.coordinate_manager.stride_map(ts).V


def test_alive_unet_and_vote_head(gpu):
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd.model.backbone.aliveunet import make_alive_unet
    from mrcc_amd.model.robotnet_vote import RobotNetVote

    pts, rgb, _ = mrcc_amd.synth.gen_room(30_000, 1.2, 3)
    with torch.no_grad():
        f = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(_coords(pts, 100)), device=gpu)
        x = f.sparse()
        torch.manual_seed(0)
        alive = make_alive_unet(m=16, block_reps=1, bottleneck=False)(3, 8).to(gpu).eval()
        out = alive(x)  # 7 levels down to tensor stride 128 and back; returns block14's output (no final conv)
        assert out.tensor_stride == 1 and out.F.shape == (x.F.shape[0], 16) and torch.isfinite(out.F).all()
        assert sorted(x.coordinate_manager.maps) == [1, 2, 4, 8, 16, 32, 64, 128]
        vote = RobotNetVote(3).to(gpu).eval()
        v = vote(x)
        assert v.F.shape == (x.F.shape[0], 2)
        # reference post-op for votes: top-8 mean (utils/output.py:45-64)
        from mrcc_amd.utils.output import get_pred_center

        pf = v.slice(f).F
        centre = get_pred_center(pf, pts)
        assert centre.shape == (3,) and np.isfinite(centre).all()


def test_cfg3_batch64_seg_vote_and_pose(gpu, oracle):
    """BASELINE configs[2]: 64 frames in one sparse tensor (batch column), segmentation + vote heads, then 64 Kabsch
    problems in one launch.  Checks: batched == per-frame (bit-exact, spot-checked), per-frame vote centres, and the
    batched pose solve against the oracle's SVD restatement and the generating poses."""
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd.model.backbone.minkunet import MinkUNet14A
    from mrcc_amd.model.robotnet_segmentation import _classification_head
    from mrcc_amd.utils import transformation as T

    B = 64
    torch.manual_seed(0)
    Head = _classification_head(MinkUNet14A, lambda: 3, "SegSmall")
    Vote = _classification_head(MinkUNet14A, lambda: 2, "VoteSmall")
    seg = Head(3, num_classes=3).to(gpu).eval()
    vote = Vote(3, num_classes=2).to(gpu).eval()
    crops = [mrcc_amd.synth.gen_ee_crop(s, n=1500) for s in range(B)]
    with torch.no_grad():
        coords = ME.utils.batched_coordinates([torch.from_numpy(c[0] * np.float32(100)) for c in crops],
                                              dtype=torch.float32)
        feats = torch.from_numpy(np.concatenate([c[1] for c in crops]))
        field = ME.TensorField(feats, coords, device=gpu)
        x = field.sparse()
        assert int(x.C[:, 0].max()) == B - 1
        s_out = seg(x)
        v_out = vote(x).slice(field).F
        labels, _ = s_out.slice_argmax(field)
        assert labels.shape[0] == B * 1500
        # spot-check three frames against single-frame runs
        for b in (0, 31, 63):
            c1 = torch.from_numpy(np.concatenate([np.zeros((1500, 1), np.float32), crops[b][0] * np.float32(100)], 1))
            f1 = ME.TensorField(torch.from_numpy(crops[b][1]), c1, device=gpu)
            l1, _ = seg(f1.sparse()).slice_argmax(f1)
            assert torch.equal(labels[b * 1500:(b + 1) * 1500], l1)
    # vote centres per frame (utils/output.py:45-64: mean of the 8 highest-vote points)
    from mrcc_amd.utils.output import get_pred_center

    for b in (0, 63):
        centre = get_pred_center(v_out[b * 1500:(b + 1) * 1500], crops[b][0])
        assert np.isfinite(centre).all() and np.abs(centre - crops[b][2][:3]).max() < 0.25  # inside the crop
    # 64 Kabsch problems in one launch
    ref = np.repeat(mrcc_amd.synth.REFERENCE_KEY_POINTS[None], B, axis=0)
    tgt = np.stack([c[3] for c in crops])
    R, t, q = T.get_rigid_transform_3D_batched(ref, tgt, device=gpu)
    for b in range(B):
        Ro, to = oracle.get_rigid_transform_3D(ref[b], tgt[b])
        assert np.abs(R[b] - Ro).max() < 1e-9 and np.abs(t[b] - to).max() < 1e-9
        pose = crops[b][2]
        assert np.abs(t[b] - pose[:3]).max() < 5e-3  # 1 mm key-point noise
        assert min(np.abs(q[b] - pose[3:]).max(), np.abs(q[b] + pose[3:]).max()) < 5e-2
