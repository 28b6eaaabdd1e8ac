"""A1/A2 parity: HIP voxelisation / coordinate maps / kernel maps vs the CPU oracle — bit-exact (integer work)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _field(ME, pts, rgb, scale, dev, batch=None):
    b = np.zeros((len(pts), 1), np.float32) if batch is None else batch.reshape(-1, 1).astype(np.float32)
    coords = torch.from_numpy(np.concatenate([b, pts * np.float32(scale)], axis=1))
    return ME.TensorField(features=torch.from_numpy(rgb), coordinates=coords,
                          quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE,
                          minkowski_algorithm=ME.MinkowskiAlgorithm.SPEED_OPTIMIZED, device=dev), coords.numpy()


@pytest.mark.parametrize("n,L,scale,seed", [(20000, 1.0, 50, 0), (80000, 1.5, 50, 1), (5000, 0.5, 200, 2)])
def test_voxelize_matches_oracle(gpu, oracle, n, L, scale, seed):
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME

    pts, rgb, _ = mrcc_amd.synth.gen_room(n, L, seed)
    pts[: n // 2] -= np.float32(L)  # negative coordinates exercise floor toward -inf
    field, coords4 = _field(ME, pts, rgb, scale, gpu)
    st = field.sparse()
    ref = oracle.voxelize(coords4)
    cmap = st.coordinate_map
    assert cmap.V == len(ref["keys"])
    assert np.array_equal(cmap.keys.cpu().numpy().view(np.uint64), ref["keys"])
    assert np.array_equal(cmap.coords.cpu().numpy(), ref["coords"])
    assert np.array_equal(field.inverse_mapping.cpu().numpy(), ref["inverse"])
    feats = oracle.voxel_reduce(rgb, ref["order"], ref["seg_start"], 0)
    assert np.array_equal(st.F.cpu().numpy(), feats)  # sequential mean in point order: bit-exact


def test_voxelize_batched_and_int_coords(gpu, oracle):
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME

    rng = np.random.default_rng(3)
    parts, batch = [], []
    for b in range(5):
        p, _, _ = mrcc_amd.synth.gen_room(3000 + 500 * b, 0.8, 10 + b)
        parts.append(p)
        batch.append(np.full(len(p), b))
    pts, batch = np.concatenate(parts), np.concatenate(batch)
    rgb = rng.uniform(-0.5, 0.5, size=(len(pts), 3)).astype(np.float32)
    perm = rng.permutation(len(pts))  # interleave the frames: rows must still come out batch-major
    pts, batch, rgb = pts[perm], batch[perm], rgb[perm]
    field, coords4 = _field(ME, pts, rgb, 50, gpu, batch)
    st = field.sparse()
    ref = oracle.voxelize(coords4)
    assert np.array_equal(st.C.cpu().numpy(), ref["coords"])
    assert np.all(np.diff(ref["coords"][:, 0]) >= 0)
    # ME.SparseTensor(feats, coordinates=int coords) path (train_segmentation.py:78)
    ic = np.concatenate([batch.reshape(-1, 1), np.floor(pts * 50)], axis=1).astype(np.int32)
    st2 = ME.SparseTensor(torch.from_numpy(rgb), coordinates=torch.from_numpy(ic), device=gpu)
    ref2 = oracle.voxelize(ic, coords_are_int=True)
    assert np.array_equal(st2.C.cpu().numpy(), ref2["coords"])
    assert np.array_equal(st2.F.cpu().numpy(), oracle.voxel_reduce(rgb, ref2["order"], ref2["seg_start"], 1))
    dc = st.decomposed_coordinates
    assert len(dc) == 5 and sum(len(d) for d in dc) == len(ref["coords"])


def test_voxelize_edge_cases(gpu, oracle):
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME

    # single point, duplicated points, and out-of-range coordinates must be reported, not wrapped
    one = np.array([[0.013, -0.002, 0.4]], np.float32)
    field, c4 = _field(ME, one, np.ones((1, 3), np.float32), 50, gpu)
    st = field.sparse()
    assert st.F.shape == (1, 3) and np.array_equal(st.C.cpu().numpy(), oracle.voxelize(c4)["coords"])
    dup = np.repeat(one, 1000, axis=0)
    f = np.arange(3000, dtype=np.float32).reshape(1000, 3)
    field, c4 = _field(ME, dup, f, 50, gpu)
    st = field.sparse()
    ref = oracle.voxelize(c4)
    assert st.F.shape == (1, 3)
    assert np.array_equal(st.F.cpu().numpy(), oracle.voxel_reduce(f, ref["order"], ref["seg_start"], 0))
    far = np.array([[1e6, 0, 0]], np.float32)
    field, _ = _field(ME, far, np.ones((1, 3), np.float32), 50, gpu)
    with pytest.raises(mrcc_amd._lib.SvHipError):
        field.sparse()


@pytest.mark.parametrize("n,L,scale", [(30000, 1.0, 50), (8000, 0.4, 200)])
def test_stride_and_kernel_maps_match_oracle(gpu, oracle, n, L, scale):
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME

    pts, rgb, _ = mrcc_amd.synth.gen_room(n, L, 5)
    pts -= np.float32(L)  # straddle zero
    field, coords4 = _field(ME, pts, rgb, scale, gpu)
    st = field.sparse()
    cm = st.coordinate_manager
    frame = oracle.Frame(oracle.voxelize(coords4)["coords"])
    for level in range(4):
        ts = 2 ** level
        coarse = frame.down(ts)
        m = cm.stride_map(2 * ts)
        assert np.array_equal(m.coords.cpu().numpy(), coarse)
        parent, child_start = cm.parents[ts]
        assert np.array_equal(parent.cpu().numpy().astype(np.int64), frame.parent[ts])
        cs = child_start.cpu().numpy()
        assert cs[0] == 0 and cs[-1] == len(frame.maps[ts]) and np.all(np.diff(cs) >= 1)

    def unsort(plan):
        """Undo the mask sort: nbr in canonical output-row order."""
        perm = plan.perm.cpu().numpy()
        nbr_s = plan.nbr_s.cpu().numpy()
        valid = perm >= 0
        assert valid.sum() == plan.V_out and np.array_equal(np.sort(perm[valid]), np.arange(plan.V_out))
        out = np.full((plan.K, plan.V_out), -2, np.int32)
        out[:, perm[valid]] = nbr_s[:, valid]
        assert np.all(nbr_s[:, ~valid] == -1)
        # submask bit s of tile t for offset k <=> some row of that 16-row sub-tile has a neighbour at k
        sub = (nbr_s >= 0).reshape(plan.K, plan.Vpad // 128, 8, 16).any(axis=3)  # [K, tiles, 8]
        bits = (sub * (1 << np.arange(8))).sum(axis=2).T  # [tiles, K]
        assert np.array_equal(plan.submask.cpu().numpy().astype(np.int64), bits)
        # tile_order: a permutation of the plan tiles, work (active sub-tile slots) non-increasing
        order = plan.tile_order.cpu().numpy()
        assert np.array_equal(np.sort(order), np.arange(plan.Vpad // 128))
        work = sub.sum(axis=(0, 2))[order]
        assert np.all(np.diff(work) <= 0)
        return out

    for ts in (1, 2, 4):
        assert np.array_equal(unsort(cm.plan_k3(ts)), frame.k3(ts))
        assert np.array_equal(unsort(cm.plan_down(ts)), frame.kdown(ts))
        assert np.array_equal(unsort(cm.plan_up(2 * ts)), frame.kup(2 * ts))


def test_empty_and_single_point_inputs_run_through_a_network(gpu):
    """ragged / degenerate inputs: an empty cloud and a one-point cloud must flow through voxelise -> U-Net -> slice."""
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd.model.backbone.minkunet import MinkUNet14A

    torch.manual_seed(0)
    net = MinkUNet14A(3, 5).to(gpu).eval()
    with torch.no_grad():
        for n in (0, 1, 17):
            pts = np.random.default_rng(n).uniform(-0.3, 0.3, size=(n, 3)).astype(np.float32)
            rgb = np.zeros((n, 3), np.float32) + 0.25
            field, _ = _field(ME, pts, rgb, 50, gpu)
            x = field.sparse()
            assert x.F.shape[0] <= n
            out = net(x)
            assert out.F.shape == (x.F.shape[0], 5)
            label, conf = out.slice_argmax(field)
            assert label.shape == (n,) and (n == 0 or torch.isfinite(out.F).all())


@pytest.mark.parametrize("cols,as_torch", [(3, False), (4, False), (3, True)])
def test_sparse_quantize_matches_oracle(gpu, oracle, cols, as_torch):
    """A2: ME.utils.sparse_quantize (data/alivev2.py:290-296): first-occurrence representative per voxel, label
    collisions -> ignore_label, negative coordinates floor towards -inf, return_index / return_inverse."""
    from mrcc_amd import MinkowskiEngine as ME

    rng = np.random.default_rng(cols)
    n, qs = 20_000, 0.02
    pts = rng.uniform(-0.4, 0.4, size=(n, 3)).astype(np.float32)
    pts[:500] = pts[500:1000] + np.float32(1e-4)          # guaranteed multi-point voxels
    pts[1000:1010] = np.float32(-qs) * np.arange(10, dtype=np.float32)[:, None]  # exact negative multiples of the size
    coords = pts if cols == 3 else np.concatenate([rng.integers(0, 3, size=(n, 1)).astype(np.float32) * np.float32(qs),
                                                   pts], axis=1)
    feats = rng.normal(size=(n, 3)).astype(np.float32)
    labels = rng.integers(0, 3, size=n).astype(np.int32)
    labels[:500] = labels[500:1000]                       # some shared voxels agree on the label ...
    labels[100:200] = (labels[600:700] + 1) % 3           # ... and some collide
    wc, wf, wl, widx, winv = oracle.sparse_quantize(coords, feats, labels, qs, ignore_label=255)
    conv = (lambda a: torch.from_numpy(a)) if as_torch else (lambda a: a)
    gc, gf, gl, gidx, ginv = ME.utils.sparse_quantize(conv(coords), features=conv(feats), labels=conv(labels),
                                                      quantization_size=qs, ignore_label=255, return_index=True,
                                                      return_inverse=True)
    back = (lambda a: a.numpy()) if as_torch else (lambda a: a)
    assert isinstance(gc, torch.Tensor) == as_torch
    assert np.array_equal(back(gc), wc) and np.array_equal(back(gidx), widx) and np.array_equal(back(ginv), winv)
    assert np.array_equal(back(gf), wf) and np.array_equal(back(gl), wl) and back(gl).dtype == labels.dtype
    assert (wl == 255).sum() >= 50 and (wl != 255).sum() > 1000 and len(wc) < n - 400
    # representative = lowest original index of the voxel; every point maps to the voxel holding floor(coord / size)
    first_of = np.full(len(wc), n, np.int64)
    np.minimum.at(first_of, winv, np.arange(n))
    assert np.array_equal(first_of, back(gidx))
    assert np.array_equal(back(gc)[back(ginv)][:, -3:], np.floor(pts.astype(np.float64) / qs).astype(np.int32))
    # the three-value form the reference unpacks, and the maps-only form
    c3, f3, l3 = ME.utils.sparse_quantize(coordinates=coords, features=feats, labels=labels, quantization_size=qs,
                                          ignore_label=255)
    assert np.array_equal(c3, wc) and np.array_equal(f3, wf) and np.array_equal(l3, wl)
    m_idx, m_inv = ME.utils.sparse_quantize(coords, quantization_size=qs, return_maps_only=True, return_inverse=True)
    assert np.array_equal(m_idx, widx) and np.array_equal(m_inv, winv)


def _gray_key(mask, K):
    """sort key of sv_plan_build (csrc/sv_coords.hip iota_key_kernel): rarest offsets (corners, edges, faces, centre) as
    the most significant bits, then the rank in reflected-Gray order."""
    m = mask.astype(np.int64)
    if K == 27:
        pos, rank = {}, 0
        for cls in (3, 2, 1, 0):
            for k in range(27):
                if abs(k % 3 - 1) + abs((k // 3) % 3 - 1) + abs(k // 9 - 1) == cls:
                    pos[k] = 26 - rank
                    rank += 1
        m2 = np.zeros_like(m)
        for k in range(27):
            m2 |= ((m >> k) & 1) << pos[k]
        m = m2
    for s in (1, 2, 4, 8, 16):
        m ^= m >> s
    return m


@pytest.mark.parametrize("n,L", [(200_000, 2.4), (9_000, 0.6), (300, 0.2)])
def test_plan_order_is_the_stable_sort_of_its_key(gpu, n, L):
    """The hand-written radix sort (csrc/sv_sort.hip) behind the conv plans: multi-workgroup passes at the big levels,
    the one-launch single-workgroup sort at the small ones and for every tile order.  perm must be EXACTLY the stable
    sort of the rows by their Gray key, tile_order the stable sort of the plan tiles by work, descending."""
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME

    pts, rgb, _ = mrcc_amd.synth.gen_room(n, L, 5)
    c4 = np.concatenate([np.zeros((n, 1), np.float32), pts * np.float32(50)], axis=1)
    x = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(c4), device=gpu).sparse()
    cm = x.coordinate_manager
    plans = [(cm.plan_k3(1 << l), 27) for l in range(4)] + [(cm.plan_down(1), 8), (cm.plan_up(2), 8), (cm.plan_down(4), 8)]
    for plan, K in plans:
        V = plan.V_out
        perm = plan.perm.cpu().numpy()
        nbr_s = plan.nbr_s.cpu().numpy()
        assert np.array_equal(np.sort(perm[:V]), np.arange(V)) and (perm[V:] == -1).all()
        mask_sorted = np.zeros(plan.Vpad, np.int64)
        for k in range(K):
            mask_sorted |= (nbr_s[k] >= 0).astype(np.int64) << k
        mask = np.zeros(V, np.int64)
        mask[perm[:V]] = mask_sorted[:V]
        want = np.argsort(_gray_key(mask, K), kind="stable")
        assert np.array_equal(perm[:V], want)
        sub = plan.submask.cpu().numpy().astype(np.uint32)
        cost = np.array([[bin(int(v)).count("1") for v in row] for row in sub]).sum(axis=1)
        assert np.array_equal(plan.tile_order.cpu().numpy(), np.argsort(255 - np.minimum(cost, 255), kind="stable"))
