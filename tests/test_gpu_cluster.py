"""Device path of ClusterUtil.get_largest_cluster (csrc/sv_cluster.hip) against the oracle's flood fill
(oracle.single_linkage_roots, pinned against sklearn in tests/test_cluster_cpu.py) and the host k-d-tree path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _scene(rng, n_blob=600, dtype=np.float32):
    blob = lambda c, n, s=0.02: rng.normal(0, s, size=(n, 3)) + np.asarray(c)
    chain = np.stack([np.arange(80) * 0.055, np.zeros(80), np.zeros(80)], axis=1) + [1.0, 1.0, 0.0]
    pts = np.concatenate([blob([0, 0, 0], n_blob), blob([0.3, 0, 0], n_blob // 3), blob([0, 0.4, 0.1], 50), chain,
                          rng.uniform(-1, 1, size=(40, 3))])
    return pts[rng.permutation(len(pts))].astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_roots_and_largest_cluster_match_the_oracle(dtype):
    from mrcc_amd.utils.output import ClusterUtil, single_linkage_roots
    from oracle import sv_oracle as O

    rng = np.random.default_rng(0)
    pts = _scene(rng, dtype=dtype)
    want = O.single_linkage_roots(pts, 0.06)
    root, best = single_linkage_roots(torch.from_numpy(pts).to(_dev()), 0.06)
    torch.cuda.synchronize()
    assert np.array_equal(root.cpu().numpy(), want)  # root = smallest member: identical labelling, not just partition
    u, c = np.unique(want, return_counts=True)
    assert best.cpu().tolist() == [int(u[c.argmax()]), int(c.max())]
    got = ClusterUtil().get_largest_cluster(torch.from_numpy(pts).to(_dev()))
    assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), O.largest_cluster(pts, 0.06))
    assert np.array_equal(got.cpu().numpy(), ClusterUtil().get_largest_cluster(pts))  # host path of the same class


def test_threshold_is_strict_ties_go_to_the_lowest_index_and_tiny_inputs():
    from mrcc_amd.utils.output import ClusterUtil, single_linkage_roots

    d = _dev()
    # exactly dist apart: not connected (sklearn merges below the threshold); one ulp closer: connected
    pts = torch.tensor([[0.0, 0, 0], [0.5, 0, 0], [2.0, 0, 0], [2.0, np.nextafter(0.5, 0), 0]], dtype=torch.float64, device=d)
    root, best = single_linkage_roots(pts, 0.5)
    assert root.cpu().tolist() == [0, 1, 2, 2] and best.cpu().tolist() == [2, 2]
    # two clusters of three: the one with the lowest index wins, wherever its points sit in the array
    pts = torch.tensor([[5, 5, 5], [0, 0, 0], [5.01, 5, 5], [0.01, 0, 0], [0, 0.01, 0], [5, 5.01, 5]], dtype=torch.float32, device=d)
    assert ClusterUtil().get_largest_cluster(pts).cpu().tolist() == [0, 2, 5]
    assert ClusterUtil().get_largest_cluster(pts[1:]).cpu().tolist() == [0, 2, 3]
    assert ClusterUtil().get_largest_cluster(pts[:1]).cpu().tolist() == [0]
    assert ClusterUtil().get_largest_cluster(pts[:0]).numel() == 0
    # duplicates are at distance 0 < dist
    dup = torch.zeros(300, 3, device=d)
    assert ClusterUtil().get_largest_cluster(dup).cpu().tolist() == list(range(300))


def test_index_indirection_and_a_dense_blob_of_several_tiles():
    """idx selects rows of a bigger cloud (the engine's EE points inside the frame); 5 000 points inside one 6-cm ball
    neighbourhood structure = millions of unions racing on the same forest; a long chain = deep trees."""
    from mrcc_amd.utils.output import ClusterUtil, single_linkage_roots
    from oracle import sv_oracle as O

    rng = np.random.default_rng(3)
    cloud = np.concatenate([rng.uniform([-0.05, -0.11, -0.065], [0.05, 0.11, 0.065], size=(5000, 3)),
                            np.stack([np.arange(3000) * 0.05, np.full(3000, 3.0), np.zeros(3000)], axis=1),
                            rng.uniform(-2, 2, size=(2000, 3))]).astype(np.float32)
    cloud = cloud[rng.permutation(len(cloud))]
    idx = np.sort(rng.choice(len(cloud), 7000, replace=False))
    sub = cloud[idx]
    root, best = single_linkage_roots(torch.from_numpy(cloud).to(_dev()), 0.06, idx=torch.from_numpy(idx).to(_dev()))
    want = O.single_linkage_roots(sub, 0.06)
    assert np.array_equal(root.cpu().numpy(), want)
    got = ClusterUtil().get_largest_cluster(torch.from_numpy(cloud).to(_dev()), idx=torch.from_numpy(idx).to(_dev()))
    assert np.array_equal(got.cpu().numpy(), O.largest_cluster(sub, 0.06))
    # run to run: the unions race, the result does not
    for _ in range(3):
        r2, _ = single_linkage_roots(torch.from_numpy(cloud).to(_dev()), 0.06, idx=torch.from_numpy(idx).to(_dev()))
        assert torch.equal(r2, root)


def test_select_equal_is_np_where():
    from mrcc_amd.utils.output import select_equal

    rng = np.random.default_rng(5)
    for n, dt in ((0, np.int64), (1, np.int64), (1023, np.int32), (1024, np.int64), (200_003, np.int64), (70_000, np.int32)):
        v = rng.integers(0, 3, size=n).astype(dt)
        t = torch.from_numpy(v).to(_dev())
        for value in (2, 7):
            got = select_equal(t, value)
            assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), np.where(v == value)[0])
        if n:
            got = select_equal(t, torch.tensor([1], dtype=torch.int32, device=_dev()))
            assert np.array_equal(got.cpu().numpy(), np.where(v == 1)[0])
