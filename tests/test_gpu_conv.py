"""A3/A5 parity: the fp32-MFMA sparse convolution vs the C oracle's fmaf chain — BIT-EXACT (same accumulation order)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(gpu, n=20000, L=1.0, scale=50, seed=0, batch=1):
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME

    parts, bcol = [], []
    for b in range(batch):
        p, _, _ = mrcc_amd.synth.gen_room(n, L, seed + b)
        parts.append(p - np.float32(L))
        bcol.append(np.full((len(p), 1), b, np.float32))
    pts = np.concatenate(parts)
    coords4 = np.concatenate([np.concatenate(bcol), pts * np.float32(scale)], axis=1)
    rgb = np.random.default_rng(seed).uniform(-0.5, 0.5, size=(len(pts), 3)).astype(np.float32)
    field = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(coords4), device=gpu)
    return ME, field, field.sparse(), coords4


def _same(a, b):
    """bitwise equality up to the sign of zero (a skipped neighbour is fma(0, w, acc) on the GPU)."""
    return np.array_equal(a, b)


@pytest.mark.parametrize("cin,cout", [(3, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (256, 256),
                                      (96, 384), (384, 384), (416, 384), (7, 5), (130, 200)])
def test_conv_k3_bit_exact(gpu, oracle, cin, cout):
    from mrcc_amd import nn as svnn

    ME, field, st, coords4 = _setup(gpu, n=12000, L=0.8)
    frame = oracle.Frame(oracle.voxelize(coords4)["coords"])
    V = st.F.shape[0]
    rng = np.random.default_rng(cin * 1000 + cout)
    x = rng.normal(size=(V, cin)).astype(np.float32)
    W = (rng.normal(size=(27, cin, cout)) * np.sqrt(2.0 / (27 * cout))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, size=cout).astype(np.float32)
    shift = rng.normal(size=cout).astype(np.float32)
    res = rng.normal(size=(V, cout)).astype(np.float32)
    plan = st.coordinate_manager.plan_k3(1)
    t = lambda a: torch.from_numpy(a).to(gpu)
    # plain conv
    got = svnn.conv_forward(t(x), t(W), plan, V).cpu().numpy()
    want = oracle.conv(x, W, frame.k3(1), V)
    assert _same(got, want), f"max abs diff {np.abs(got - want).max()}"
    # fused BN + residual + ReLU
    got = svnn.conv_forward(t(x), t(W), plan, V, t(scale), t(shift), t(res), 1).cpu().numpy()
    want = oracle.conv(x, W, frame.k3(1), V, scale, shift, res, oracle.ACT_RELU)
    assert _same(got, want), f"max abs diff {np.abs(got - want).max()}"


@pytest.mark.parametrize("cin,cout", [(32, 32), (128, 128), (256, 384), (384, 384)])
def test_conv_down_up_bit_exact(gpu, oracle, cin, cout):
    from mrcc_amd import nn as svnn

    ME, field, st, coords4 = _setup(gpu, n=15000, L=0.8, batch=2)
    cm = st.coordinate_manager
    frame = oracle.Frame(oracle.voxelize(coords4)["coords"])
    rng = np.random.default_rng(7)
    t = lambda a: torch.from_numpy(a).to(gpu)
    for ts in (1, 2):
        Vf, Vc = len(frame.maps[ts]), len(frame.down(ts))
        W = (rng.normal(size=(8, cin, cout)) * 0.1).astype(np.float32)
        xf = rng.normal(size=(Vf, cin)).astype(np.float32)
        got = svnn.conv_forward(t(xf), t(W), cm.plan_down(ts), Vc, act=1).cpu().numpy()
        want = oracle.conv(xf, W, frame.kdown(ts), Vc, act=oracle.ACT_RELU)
        assert _same(got, want)
        xc = rng.normal(size=(Vc, cin)).astype(np.float32)
        got = svnn.conv_forward(t(xc), t(W), cm.plan_up(2 * ts), Vf).cpu().numpy()
        want = oracle.conv(xc, W, frame.kup(2 * ts), Vf)
        assert _same(got, want)


@pytest.mark.parametrize("V,cin,cout", [(1, 4, 4), (127, 256, 1024), (129, 1024, 3), (5000, 384, 256), (300, 9, 7),
                                        (64, 2048, 7), (5000, 1024, 3), (777, 256, 2), (1000, 64, 4), (300, 100, 1)])
def test_dense_linear_bit_exact(gpu, oracle, V, cin, cout):
    from mrcc_amd import nn as svnn

    rng = np.random.default_rng(V + cin + cout)
    x = rng.normal(size=(V, cin)).astype(np.float32)
    W = (rng.normal(size=(1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(gpu)
    got = svnn.conv_forward(t(x), t(W), None, V, None, t(bias), None, 2, 0.01).cpu().numpy()
    want = oracle.conv(x, W, None, V, None, bias, None, oracle.ACT_LEAKY, 0.01)
    assert _same(got, want), f"max abs diff {np.abs(got - want).max()}"


def test_strided_input_and_affine(gpu, oracle):
    """row strides (ld) != channel count: a column slice of a wider buffer as conv input."""
    from mrcc_amd import nn as svnn

    rng = np.random.default_rng(0)
    V = 1000
    big = rng.normal(size=(V, 96)).astype(np.float32)
    W = rng.normal(size=(1, 64, 48)).astype(np.float32)
    tb = torch.from_numpy(big).to(gpu)
    got = svnn.conv_forward(tb[:, 32:], torch.from_numpy(W).to(gpu), None, V).cpu().numpy()
    assert _same(got, oracle.conv(big[:, 32:], W, None, V))
    s = rng.uniform(0.5, 2, size=96).astype(np.float32)
    b = rng.normal(size=96).astype(np.float32)
    got = svnn.affine_act(tb, torch.from_numpy(s).to(gpu), torch.from_numpy(b).to(gpu), act=2, slope=0.01)
    assert _same(got.cpu().numpy(), oracle.affine_act(big, s, b, None, oracle.ACT_LEAKY, 0.01))


@pytest.mark.parametrize("level,cin", [(0, 32), (1, 64), (1, 128), (2, 96), (3, 128), (3, 96)])
def test_conv_every_tile_shape_of_the_pyramid_bit_exact(gpu, oracle, level, cin):
    """The Cfg-2 pyramid (88k / 26k / 7k / 2k voxels) with 384 output channels walks the dispatch table of the wide
    layers: 64x192 (+ 32x192 tail tiles in one launch) on levels 0-1, 32x192 on level 2 and 16x192 (register-ring; FULL form
    when Cin is a multiple of 128) tiles on level 3; 64x128 is covered by the forced-instance cases of test_gpu_cfg.py.  Narrow inputs keep
    the oracle fast."""
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd import nn as svnn
    from mrcc_amd import profiling

    pts, rgb, _ = mrcc_amd.synth.gen_room(200_000, 2.4, 0)
    coords4 = np.concatenate([np.zeros((len(pts), 1), np.float32), pts * np.float32(50)], axis=1)
    st = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(coords4), device=gpu).sparse()
    cm = st.coordinate_manager
    ts = 1 << level
    plan = cm.plan_k3(ts)
    V = cm.stride_map(ts).V
    expect = {0: "conv_fwd_dual_kernel<64, 32, 4, 3>", 1: "conv_fwd_dual_kernel<64, 32, 4, 3>",
              2: "conv_fwd_kernel<32, 4, 3>", 3: "conv_fwd_kernel<16, 4, 3>"}[level]
    assert profiling.conv_kernel_config(384, plan.Vpad, cin, 27) == expect
    frame = oracle.Frame(oracle.voxelize(coords4)["coords"])
    for l in range(level):
        frame.down(1 << l)
    rng = np.random.default_rng(level)
    x = rng.normal(size=(V, cin)).astype(np.float32)
    W = (rng.normal(size=(27, cin, 384)) * 0.05).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, size=384).astype(np.float32)
    shift = rng.normal(size=384).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(gpu)
    got = svnn.conv_forward(t(x), t(W), plan, V, t(scale), t(shift), None, 1).cpu().numpy()
    want = oracle.conv(x, W, frame.k3(ts), V, scale, shift, None, oracle.ACT_RELU)
    assert _same(got, want), f"max abs diff {np.abs(got - want).max()}"


@pytest.mark.parametrize("cout", [1, 3, 4])
def test_narrow_output_stream_kernel_epilogue(gpu, oracle, cout):
    """The narrow-output dense layer (Cout <= 4) runs on the row-streaming kernel: same chain, same epilogue
    (folded BN, residual, ReLU) and a strided input view."""
    from mrcc_amd import nn as svnn

    V, cin = 1300, 200
    rng = np.random.default_rng(cout)
    xfull = rng.normal(size=(V, cin + 56)).astype(np.float32)
    x = xfull[:, :cin]
    W = (rng.normal(size=(1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, size=cout).astype(np.float32)
    shift = rng.normal(size=cout).astype(np.float32)
    res = rng.normal(size=(V, cout)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(gpu)
    xd = t(xfull)[:, :cin]  # row stride 256 floats, 200 used
    got = svnn.conv_forward(xd, t(W), None, V, t(scale), t(shift), t(res), 1).cpu().numpy()
    want = oracle.conv(np.ascontiguousarray(x), W, None, V, scale, shift, res, oracle.ACT_RELU)
    assert _same(got, want), f"max abs diff {np.abs(got - want).max()}"


def test_relu_epilogue_propagates_nan_like_torch(gpu, oracle):
    """torch.relu keeps NaN (the reference's MinkowskiReLU is torch.relu on the features); the fused epilogue, the
    stand-alone affine/activation kernel and the oracle must not turn an upstream NaN into 0."""
    from mrcc_amd import nn as svnn

    ME, field, st, coords4 = _setup(gpu, n=3000, L=0.4)
    frame = oracle.Frame(oracle.voxelize(coords4)["coords"])
    V = st.F.shape[0]
    rng = np.random.default_rng(0)
    x = rng.normal(size=(V, 32)).astype(np.float32)
    x[V // 2, 5] = np.nan
    W = (rng.normal(size=(27, 32, 64)) * 0.1).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(gpu)
    got = svnn.conv_forward(t(x), t(W), st.coordinate_manager.plan_k3(1), V, act=1).cpu().numpy()
    want = oracle.conv(x, W, frame.k3(1), V, act=oracle.ACT_RELU)
    assert np.isnan(want).any() and np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got[~np.isnan(want)], want[~np.isnan(want)])
    y = svnn.affine_act(t(x), act=1).cpu().numpy()
    assert np.array_equal(np.isnan(y), np.isnan(x)) and np.array_equal(y[~np.isnan(x)], np.maximum(x, 0)[~np.isnan(x)])
    assert np.array_equal(np.isnan(torch.relu(torch.from_numpy(x)).numpy()), np.isnan(y))


def test_thin_wave_per_subtile_kernel_strided_and_batched(gpu, oracle):
    """The thin 32 -> 32 kernels (one wave per 16-row sub-tile, direct register gathers, lane-group transposes;
    conv_thin_lds_kernel: the layer's weights resident in LDS, sub-tiles pulled from a per-workgroup queue;
    conv_thin_kernel: weights through L1, taken when the weight pointer is not 16-byte aligned): a column slice of a
    wider buffer as input AND as output, a two-frame batch, every epilogue option, and the 8-offset maps."""
    from mrcc_amd import nn as svnn
    from mrcc_amd import profiling

    ME, field, st, coords4 = _setup(gpu, n=9000, L=0.7, batch=2)
    cm = st.coordinate_manager
    frame = oracle.Frame(oracle.voxelize(coords4)["coords"])
    V = st.F.shape[0]
    plan = cm.plan_k3(1)
    import mrcc_amd

    assert profiling.conv_kernel_config(32, plan.Vpad, 32, 27) == "conv_thin_kernel<32, 32>"
    rng = np.random.default_rng(11)
    wide_in = rng.normal(size=(V, 80)).astype(np.float32)
    W = (rng.normal(size=(27, 32, 32)) * 0.1).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, size=32).astype(np.float32)
    shift = rng.normal(size=32).astype(np.float32)
    res = rng.normal(size=(V, 32)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(gpu)
    x = t(wide_in)[:, 16:48]  # row stride 80 floats, 16-byte aligned start
    out_buf = torch.full((V, 96), 7.0, device=gpu)
    want = oracle.conv(np.ascontiguousarray(wide_in[:, 16:48]), W, frame.k3(1), V, scale, shift, res, oracle.ACT_LEAKY, 0.05)
    # default dispatch (frames overlapped in a multi-stream pipeline): four-wave workgroups, weights through L1
    svnn.conv_forward(x, t(W), plan, V, t(scale), t(shift), t(res), 2, 0.05, out=out_buf[:, 32:64])
    assert mrcc_amd._lib.conv_last_instance()[0] == "conv_thin_kernel<32, 32>"
    assert np.array_equal(out_buf[:, 32:64].cpu().numpy(), want)
    assert (out_buf[:, :32] == 7.0).all() and (out_buf[:, 64:] == 7.0).all()  # neighbours of the slice untouched
    # the dispatch of a GPU that holds ONE frame (sv_conv_set_dispatch(1.0): the per-frame InferenceEngine.predict path):
    # the layer's weights resident in LDS, one 16-wave workgroup per CU, sub-tiles pulled from a per-workgroup queue
    with mrcc_amd._lib.conv_dispatch(1.0):
        out_buf.fill_(7.0)
        svnn.conv_forward(x, t(W), plan, V, t(scale), t(shift), t(res), 2, 0.05, out=out_buf[:, 32:64])
        assert mrcc_amd._lib.conv_last_instance()[0] == "conv_thin_lds_kernel<32, 32>"
        assert np.array_equal(out_buf[:, 32:64].cpu().numpy(), want)
        assert (out_buf[:, :32] == 7.0).all() and (out_buf[:, 64:] == 7.0).all()
        # a weight tensor that starts 4 bytes off a 16-byte boundary cannot be staged with float4 copies: the other kernel
        # takes the launch - same bits
        w_off = torch.zeros(27 * 32 * 32 + 1, device=gpu)
        w_off[1:] = t(W).reshape(-1)
        out2 = svnn.conv_forward(x, w_off[1:].view(27, 32, 32), plan, V, t(scale), t(shift), t(res), 2, 0.05)
        assert mrcc_amd._lib.conv_last_instance()[0] == "conv_thin_kernel<32, 32>"
        assert np.array_equal(out2.cpu().numpy(), want)

    # bias-only epilogue (no scale), no residual, no activation
    got = svnn.conv_forward(x, t(W), plan, V, None, t(shift)).cpu().numpy()
    assert np.array_equal(got, oracle.conv(np.ascontiguousarray(wide_in[:, 16:48]), W, frame.k3(1), V, None, shift))
    # stride-2 down and transposed up maps (8 offsets)
    Vc = len(frame.down(1))
    W8 = (rng.normal(size=(8, 32, 32)) * 0.2).astype(np.float32)
    xf = rng.normal(size=(V, 32)).astype(np.float32)
    got = svnn.conv_forward(t(xf), t(W8), cm.plan_down(1), Vc, act=1).cpu().numpy()
    assert np.array_equal(got, oracle.conv(xf, W8, frame.kdown(1), Vc, act=oracle.ACT_RELU))
    xc = rng.normal(size=(Vc, 32)).astype(np.float32)
    got = svnn.conv_forward(t(xc), t(W8), cm.plan_up(2), V).cpu().numpy()
    assert np.array_equal(got, oracle.conv(xc, W8, frame.kup(2), V))
    with mrcc_amd._lib.conv_dispatch(1.0):  # the 8-offset maps on the LDS-weights kernel
        got = svnn.conv_forward(t(xf), t(W8), cm.plan_down(1), Vc, act=1).cpu().numpy()
        assert mrcc_amd._lib.conv_last_instance()[0] == "conv_thin_lds_kernel<32, 32>"
        assert np.array_equal(got, oracle.conv(xf, W8, frame.kdown(1), Vc, act=oracle.ACT_RELU))
        got = svnn.conv_forward(t(xc), t(W8), cm.plan_up(2), V).cpu().numpy()
        assert np.array_equal(got, oracle.conv(xc, W8, frame.kup(2), V))


@pytest.mark.parametrize("cin,cout", [(64, 64), (96, 384)])
def test_epilogue_every_combination_same_bits(gpu, oracle, cin, cout):
    """BN scale / bias / residual present or absent x the three activations through the branch-free epilogue of the
    buffer-addressed instances: absent operands are replaced by values that must change no bit (fmaf(x, 1, -0),
    x + (-0)), so the comparison is on the BIT PATTERNS (signed zeros included), with zero rows and exact
    cancellations in the input to produce both zeros."""
    from mrcc_amd import nn as svnn

    ME, field, st, coords4 = _setup(gpu, n=6000, L=0.6, seed=3)
    frame = oracle.Frame(oracle.voxelize(coords4)["coords"])
    V = st.F.shape[0]
    rng = np.random.default_rng(cin + cout)
    x = rng.normal(size=(V, cin)).astype(np.float32)
    x[rng.random(V) < 0.2] = 0.0                      # rows of zeros: accumulators stay +0
    W = (rng.normal(size=(27, cin, cout)) * np.sqrt(2.0 / (27 * cout))).astype(np.float32)
    scale = rng.uniform(-1.5, 1.5, size=cout).astype(np.float32)   # negative scales: -0 products
    shift = rng.normal(size=cout).astype(np.float32)
    shift[::3] = 0.0
    res = rng.normal(size=(V, cout)).astype(np.float32)
    res[rng.random(V) < 0.3] = -0.0
    plan = st.coordinate_manager.plan_k3(1)
    t = lambda a: torch.from_numpy(a).to(gpu) if a is not None else None
    for use_scale in (False, True):
        for use_shift in (False, True):
            for use_res in (False, True):
                for act in (oracle.ACT_NONE, oracle.ACT_RELU, oracle.ACT_LEAKY):
                    sc, sh, rs = (scale if use_scale else None), (shift if use_shift else None), (res if use_res else None)
                    got = svnn.conv_forward(t(x), t(W), plan, V, t(sc), t(sh), t(rs), act, 0.01).cpu().numpy()
                    want = oracle.conv(x, W, frame.k3(1), V, sc, sh, rs, act, 0.01)
                    same = np.array_equal(got.view(np.int32), want.view(np.int32))
                    assert same, (use_scale, use_shift, use_res, act,
                                  int((got.view(np.int32) != want.view(np.int32)).sum()), np.abs(got - want).max())


def test_batch_range_plans_equal_whole_map_launches(gpu, monkeypatch):
    """ConvPlan.chunks (tensors beyond the 2 GB extent run as batch ranges with their own plans, sv_plan_build nbr_base):
    with the extent limit lowered so that a 7-frame batch of small frames has to be split - frames of different sizes, one
    EMPTY batch index in the middle, a frame count that is not a power of two - every kernel map kind (3x3x3, stride-2
    down, transposed up) gives the bits of the single whole-map launch, with BN, residual and a strided output."""
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd import nn as svnn
    from mrcc_amd import sparse

    parts = []
    for b, n in enumerate((3000, 500, 0, 4200, 1500, 2600, 900)):
        if n:
            pts, rgb, _ = mrcc_amd.synth.gen_room(n, 0.5, 50 + b)
            parts.append((np.concatenate([np.full((n, 1), b, np.float32), pts * np.float32(50)], axis=1), rgb))
    coords = torch.from_numpy(np.concatenate([p[0] for p in parts]))
    feats = torch.from_numpy(np.concatenate([p[1] for p in parts]))
    x = ME.TensorField(feats, coords, device=gpu).sparse()
    cm = x.coordinate_manager
    cm.plan_down(1)
    V0, V1 = cm.stride_map(1).V, cm.stride_map(2).V
    torch.manual_seed(3)
    split_seen = False
    cases = [("k3", cm.plan_k3(1), V0, V0, 27), ("k3 level 1", cm.plan_k3(2), V1, V1, 27), ("down", cm.plan_down(1), V0, V1, 8),
             ("up", cm.plan_up(2), V1, V0, 8)]
    for name, plan, V_in, V_out, K in cases:
        for cin, cout in ((64, 96), (32, 32), (3, 32)):
            f = torch.randn(V_in, cin, device=gpu)
            W = torch.randn(K, cin, cout, device=gpu) * 0.1
            sc, sh = torch.rand(cout, device=gpu) + 0.5, torch.randn(cout, device=gpu)
            res = torch.randn(V_out, cout, device=gpu)
            whole = svnn.conv_forward(f, W, plan, V_out, sc, sh, res, 1)
            assert plan.chunks(4 * cin, 4 * cout) is None  # fits: one launch
            bi, bo = cm.batch_bounds(plan.in_stride), cm.batch_bounds(plan.out_stride)
            biggest = max(max(bi[c + 1] - bi[c] for c in range(7)) * 4 * cin, max(bo[c + 1] - bo[c] for c in range(7)) * 4 * (cout + 5))
            for factor in (1.01, 2.2, 4.5):  # one frame per range, two, four
                monkeypatch.setattr(sparse, "BUF_LIMIT", int(biggest * factor))
                plan._chunked.clear()
                buf = torch.full((V_out, cout + 5), 7.0, device=gpu)  # strided output: a column slice of a wider buffer
                got = svnn.conv_forward(f, W, plan, V_out, sc, sh, res, 1, out=buf[:, 2:2 + cout])
                parts_ = plan.chunks(4 * cin, 4 * (cout + 5))
                fits = V_in * 4 * cin < sparse.BUF_LIMIT and V_out * 4 * (cout + 5) < sparse.BUF_LIMIT
                assert (parts_ is None) == fits, (name, factor)
                if parts_ is not None:
                    assert 2 <= len(parts_) <= 6 and sum(o1 - o0 for _, _, _, o0, o1 in parts_) == V_out
                    split_seen = True
                assert torch.equal(got, whole), (name, cin, cout, factor)
                assert (buf[:, :2] == 7.0).all() and (buf[:, 2 + cout:] == 7.0).all()
            monkeypatch.setattr(sparse, "BUF_LIMIT", 0x7fff0000 - 4096)
            plan._chunked.clear()
    assert split_seen
    # a single frame beyond the limit cannot be split: chunks() says so and the launch still runs (guarded form)
    monkeypatch.setattr(sparse, "BUF_LIMIT", 1000)
    plan = cm.plan_k3(1)
    plan._chunked.clear()
    assert plan.chunks(4 * 64, 4 * 64) is None


def test_two_pass_layers_have_the_bits_of_one_launch(gpu, oracle):
    """sparse.SplitPlan / sv_conv_fwd_acc: a 3x3x3 layer run as two passes over offsets [0, 14) and [14, 27), each with its
    own row order, the raw accumulators handed over through memory - bit-identical to the single launch and to the
    oracle, with BN, residual, ReLU, multi-chunk Cin with a partial last chunk, other split points, and a forced tile shape
    on every pass (so that the hand-over is exercised on the generic, the FAST and the dual-body instances)."""
    import os

    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd import nn as svnn
    from mrcc_amd.sparse import SplitPlan

    pts, rgb, _ = mrcc_amd.synth.gen_room(60_000, 1.3, 9)
    c4 = np.concatenate([np.zeros((len(pts), 1), np.float32), pts * np.float32(50)], axis=1)
    x = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(c4), device=gpu).sparse()
    cm = x.coordinate_manager
    V = cm.stride_map(1).V
    whole = cm.plan_k3(1)
    frame = oracle.Frame(oracle.voxelize(c4)["coords"])
    nbr = frame.k3(1)
    rng = np.random.default_rng(11)
    t = lambda a: torch.from_numpy(a).to(gpu)
    for cin, cout, split, force in ((416, 384, 14, None), (200, 384, 13, "64,4,3"), (96, 192, 9, "16,4,3"), (35, 50, 20, None),
                                    (384, 384, 14, "32,4,3")):
        f = rng.normal(size=(V, cin)).astype(np.float32)
        W = (rng.normal(size=(27, cin, cout)) * np.sqrt(2.0 / (27 * cout))).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, size=cout).astype(np.float32)
        sh = rng.normal(size=cout).astype(np.float32)
        res = rng.normal(size=(V, cout)).astype(np.float32)
        one = svnn.conv_forward(t(f), t(W), whole, V, t(sc), t(sh), t(res), 1)
        sp = cm.plan_k3_split(1, split)
        assert isinstance(sp, SplitPlan) and [(a, b) for a, b, _ in sp.parts] == [(0, split), (split, 27)]
        if force:
            os.environ["SV_CONV_FORCE"] = force
        try:
            two = svnn.conv_forward(t(f), t(W), sp, V, t(sc), t(sh), t(res), 1)
        finally:
            os.environ.pop("SV_CONV_FORCE", None)
        assert torch.equal(one, two), (cin, cout, split, force, (one - two).abs().max().item())
        if cin <= 200:  # the oracle once per shape class (seconds)
            want = oracle.conv(f, W, nbr, V, sc, sh, res, oracle.ACT_RELU)
            assert np.array_equal(two.cpu().numpy(), want)
    # row-slot efficiency really is what the split is for: fewer active (sub-tile, offset) slots than the single plan
    def slots(pl):
        return int(sum(bin(int(v) & 0xFFFFFFFF).count("1") for v in pl.submask.cpu().numpy().reshape(-1)))
    sp = cm.plan_k3_split(1, 14)
    assert sum(slots(pl) for _, _, pl in sp.parts) < 0.95 * slots(whole)
