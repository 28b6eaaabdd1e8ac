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
