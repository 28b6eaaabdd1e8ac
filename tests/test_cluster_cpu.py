"""ClusterUtil (host): connected components at distance < 0.06 == single-linkage agglomerative clustering with
distance_threshold=0.06 (utils/output.py:13-28), checked against sklearn itself on small clouds."""
import numpy as np


def test_largest_cluster_matches_sklearn_single_linkage():
    from sklearn.cluster import AgglomerativeClustering

    from mrcc_amd.utils.output import ClusterUtil

    rng = np.random.default_rng(0)
    blob = lambda c, n: rng.normal(0, 0.015, size=(n, 3)) + np.asarray(c)
    pts = np.concatenate([blob([0, 0, 0], 300), blob([0.3, 0, 0], 120), blob([0, 0.4, 0.1], 50),
                          rng.uniform(-1, 1, size=(20, 3))])
    cu = ClusterUtil()
    got = np.sort(cu.get_largest_cluster(pts))
    sk = AgglomerativeClustering(distance_threshold=0.06, n_clusters=None, linkage="single").fit(pts).labels_
    u, c = np.unique(sk, return_counts=True)
    want = np.sort(np.where(sk == u[c.argmax()])[0])
    assert np.array_equal(got, want)
    # the partition itself is identical up to label names
    mine = cu.labels(pts)
    assert len(np.unique(mine)) == len(u)
    for lab in np.unique(mine):
        assert len(np.unique(sk[mine == lab])) == 1


def test_oracle_clusters_match_sklearn_single_linkage():
    """oracle.single_linkage_roots / largest_cluster (the checker of the device path) against sklearn itself - the
    library call the reference's ClusterUtil makes (utils/output.py:13-28)."""
    from sklearn.cluster import AgglomerativeClustering

    from oracle import sv_oracle as O

    rng = np.random.default_rng(1)
    for case in range(4):
        blobs = [rng.normal(0, 0.02, size=(int(rng.integers(20, 200)), 3)) + rng.uniform(-0.5, 0.5, 3) for _ in range(4)]
        chain = np.stack([np.arange(30) * 0.055, np.zeros(30), np.zeros(30)], axis=1) + [1.0, 1.0, 0.0]
        pts = np.concatenate(blobs + [chain, rng.uniform(-1, 1, size=(15, 3))]).astype(np.float32 if case % 2 else np.float64)
        root = O.single_linkage_roots(pts, 0.06)
        sk = AgglomerativeClustering(distance_threshold=0.06, n_clusters=None, linkage="single").fit(pts).labels_
        assert len(np.unique(root)) == len(np.unique(sk))
        for r in np.unique(root):
            assert len(np.unique(sk[root == r])) == 1 and r == np.where(root == r)[0].min()
        u, c = np.unique(sk, return_counts=True)
        if (c == c.max()).sum() == 1:  # a unique largest cluster: the reference's answer is well defined
            assert np.array_equal(O.largest_cluster(pts, 0.06), np.where(sk == u[c.argmax()])[0])
