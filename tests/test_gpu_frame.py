"""The frame composites (sv_frame_maps / sv_frame_plans: a frame's coordinate work as two host calls) build exactly the
arrays of the piecewise entry points (sv_voxelize, sv_stride_map, sv_hash_build, sv_kernel_map_*, sv_plan_build), and the
one-frame pipeline built on them returns the labels of the plain API path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _frame(n, seed, batch=1, L=0.8):
    import mrcc_amd

    cs, fs = [], []
    for b in range(batch):
        pts, rgb, _ = mrcc_amd.synth.gen_room(n + 37 * b, L, seed + b)
        cs.append(np.concatenate([np.full((len(pts), 1), b, np.float32), pts * np.float32(50)], axis=1))
        fs.append(rgb)
    return torch.from_numpy(np.concatenate(cs)), torch.from_numpy(np.concatenate(fs))


def _same_plan(a, b):
    assert (a.V_out, a.Vpad, a.K) == (b.V_out, b.Vpad, b.K)
    for name in ("perm", "nbr_s", "submask", "tile_order"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert torch.equal(a.raw[0], b.raw[0]) and a.raw[1] == b.raw[1] and torch.equal(a.raw[2].to(torch.int32), b.raw[2].to(torch.int32))
    assert (a.in_stride, a.out_stride) == (b.in_stride, b.out_stride)


@pytest.mark.parametrize("n,batch,levels,rules", [(3000, 1, 4, "200:9,18"), (9000, 2, 4, "1000:14;300:9,18"),
                                                   (700, 1, 6, ""), (20000, 1, 4, "2000:7,14,20")])
def test_frame_composites_equal_the_piecewise_calls(gpu, n, batch, levels, rules):
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd import nn as svnn

    c, f = _frame(n, 3, batch)
    parsed = svnn._parse_split_rules(rules)
    fa = ME.TensorField(f, c, device=gpu)
    xa = fa.sparse(pyramid_levels=levels)
    fb = ME.TensorField(f, c, device=gpu)
    xb = fb.sparse()
    cma, cmb = xa.coordinate_manager, xb.coordinate_manager
    cma.split_rules = cmb.split_rules = parsed
    assert torch.equal(xa.F, xb.F) and torch.equal(fa.inverse_mapping, fb.inverse_mapping)
    assert torch.equal(fa._order, fb._order) and torch.equal(fa._seg_start, fb._seg_start)
    cmb.stride_map(1 << levels)  # the piecewise manager builds its maps (and parent tables) on demand
    for l in range(levels + 1):
        ma, mb = cma.stride_map(1 << l), cmb.stride_map(1 << l)
        assert ma.V == mb.V and torch.equal(ma.keys, mb.keys) and torch.equal(ma.coords, mb.coords)
        if l < levels:
            for ta, tb in zip(cma.parents[1 << l], cmb.parents[1 << l]):
                assert torch.equal(ta, tb)
    # one call for everything / staged as the one-frame pipeline does it
    cma.build_plans(levels)
    staged = ME.TensorField(f, c, device=gpu).sparse(pyramid_levels=levels).coordinate_manager
    staged.split_rules = parsed
    staged.build_plans(levels, split=False)
    staged.build_plans(levels, k3=False, down=False, up=False, split=True)
    n_split = 0
    for l in range(levels + 1):
        ts = 1 << l
        for cm in (cma, staged):
            assert ("k3", ts, 1) in cm.plans
            _same_plan(cm.plans[("k3", ts, 1)], cmb.plan_k3(ts))
            ha, hb = cm.stride_map(ts)._hash, cmb.stride_map(ts).hash()
            # open addressing with atomic inserts: which of two colliding keys takes a slot first is a race, so the tables are
            # equal as key -> row maps, not slot by slot
            assert ha[2] == hb[2]
            for (ka, va), (kb, vb) in [((ha[0], ha[1]), (hb[0], hb[1]))]:
                oa, ob = torch.argsort(ka), torch.argsort(kb)
                used = (ka[oa] != -1)
                assert torch.equal(ka[oa], kb[ob]) and torch.equal(va[oa][used], vb[ob][used])
            if l < levels:
                _same_plan(cm.plans[("down", ts)], cmb.plan_down(ts))
                _same_plan(cm.plans[("up", 2 * ts)], cmb.plan_up(2 * ts))
            cuts = cm.split_cuts_for(cm.stride_map(ts).V)
            if cuts is not None:
                assert ("k3split", ts, cuts) in cm.plans, "the composite must have built the offset-range plans"
                sa, sb = cm.plans[("k3split", ts, cuts)], cmb.plan_k3_split(ts, cuts)
                assert [(p[0], p[1]) for p in sa.parts] == [(p[0], p[1]) for p in sb.parts]
                for pa, pb in zip(sa.parts, sb.parts):
                    _same_plan(pa[2], pb[2])
                n_split += 1
    assert (n_split > 0) == bool(rules)


def test_frame_composite_network_bits_and_encoder_only(gpu):
    """a network run on composite-built plans (offset-range passes included) has the bits of the lazily planned run"""
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd import nn as svnn
    from mrcc_amd.app.pipeline import FramePipeline
    from mrcc_amd.model.backbone.minkunet import MinkUNet14D

    torch.manual_seed(2)
    net = MinkUNet14D(3, 5).to(gpu).eval()
    c, f = _frame(12000, 9, 1, L=1.0)
    c, f = c.to(gpu), f.to(gpu)
    rules = svnn._parse_split_rules("1500:9,18")
    with torch.no_grad():
        fld = ME.TensorField(f, c, device=gpu)
        x = fld.sparse()
        x.coordinate_manager.split_rules = rules
        want = net(x).F.clone()
        for one_frame in (False, True):
            pipe = FramePipeline(gpu, levels=4, one_frame=one_frame)
            pipe.split_rules = rules
            prepared = pipe.prepare(c, f)
            got = pipe.run(prepared, lambda x_, f_: net(x_).F)
            pipe.drain()
            assert torch.equal(got, want), f"one_frame={one_frame}"
            assert any(k[0] == "k3split" for k in prepared.x.coordinate_manager.plans)
        enc = FramePipeline(gpu, levels=4, encoder_only=True)
        pe = enc.prepare(c, f)
        got = enc.run(pe, lambda x_, f_: net.encode(x_)[0].F)
        enc.drain()
        assert torch.equal(got, net.encode(x)[0].F)
        assert not any(k[0] in ("up", "k3split") for k in pe.x.coordinate_manager.plans)


def test_frame_maps_reports_out_of_range_points(gpu):
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME

    c = torch.zeros((10, 4))
    c[3, 1] = 2.0 ** 18
    with pytest.raises(mrcc_amd._lib.SvHipError, match="outside the key range"):
        ME.TensorField(torch.zeros((10, 3)), c, device=gpu).sparse(pyramid_levels=2)
