"""Post-ops of the reference (utils/output.py:45-87) on device tensors."""
import numpy as np
import torch


def get_segmentations_from_tensor_field(field):
    """row max -> label, sigmoid(max) -> confidence (utils/output.py:67-73).  Callers that still hold the sparse
    output should prefer SparseTensor.slice_argmax(field), which fuses slice + this into one kernel."""
    logits = field.features
    conf, preds = logits.max(1)
    return preds.cpu().numpy(), torch.sigmoid(conf).cpu().numpy()


def get_key_point_predictions(logits, conf_th=0.999):
    """softmax over classes, max over points per class, threshold (utils/output.py:81-87)."""
    softmax = logits.softmax(1).max(0)
    classes = np.where(softmax[0].cpu() > conf_th)[0]
    idx = softmax[1][classes].cpu().numpy()
    return idx, classes, softmax[0].cpu()[classes]


def get_pred_center(out, coords, ee_r=0.03, q=None):
    """mean of the 8 highest-vote points, optionally moved by (-ee_r, 0, 0) rotated by the quaternion q (w first)
    (utils/output.py:45-64)."""
    sel = out[:, 1].sort(descending=True)[1][:8]
    pred_center = coords[sel.cpu().numpy()].mean(axis=0)
    if q is not None:
        from .transformation import get_quaternion_rotation_matrix_torch

        if not isinstance(q, torch.Tensor):
            q = torch.tensor(q, dtype=torch.float32)
        rot_mat = get_quaternion_rotation_matrix_torch(q.view(1, -1))[0]
        offset_rotated = torch.matmul(rot_mat, torch.tensor([-ee_r, 0, 0]))
        if isinstance(pred_center, np.ndarray):
            offset_rotated = offset_rotated.cpu().numpy()
        pred_center += offset_rotated
    return pred_center


class ClusterUtil:
    """Largest single-linkage cluster of the end-effector points (utils/output.py:13-28: sklearn
    AgglomerativeClustering(distance_threshold=0.06, linkage='single')).  Single linkage with a distance threshold
    is exactly the connected components of the graph "distance < threshold"; computed on the host with a k-d tree
    instead of sklearn's O(n^2) linkage matrix (stays a CPU step)."""

    def __init__(self, dist=0.06, linkage="single"):
        if linkage != "single":
            raise NotImplementedError("only single linkage (what the reference uses)")
        self.dist = dist

    def labels(self, points):
        from scipy.sparse import coo_matrix
        from scipy.sparse.csgraph import connected_components
        from scipy.spatial import cKDTree

        pts = np.asarray(points, dtype=np.float64)
        n = len(pts)
        pairs = cKDTree(pts).query_pairs(self.dist, output_type="ndarray")
        if len(pairs):
            d = np.linalg.norm(pts[pairs[:, 0]] - pts[pairs[:, 1]], axis=1)
            pairs = pairs[d < self.dist]  # sklearn merges while the linkage distance is < threshold
        g = coo_matrix((np.ones(len(pairs)), (pairs[:, 0], pairs[:, 1])), shape=(n, n))
        return connected_components(g, directed=False)[1]

    def get_largest_cluster(self, points):
        labels = self.labels(points)
        unique, counts = np.unique(labels, return_counts=True)
        return np.where(labels == unique[counts.argmax()])[0]
