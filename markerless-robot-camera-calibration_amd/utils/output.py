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
    """softmax over classes, max over points per class, threshold (utils/output.py:81-87).  CUDA logits: one libsvhip
    pass (sv_key_point_predictions: softmax in registers, per-class packed (probability, index) max) and ONE read-back of
    the C per-class results; the reference's torch formulation leaves the device three times.  Ties take the lowest point
    index.  Host tensors (golden vectors of the reference on CPU) keep the reference formulation."""
    if torch.is_tensor(logits) and logits.is_cuda:
        from ctypes import c_float, c_int, c_int64, c_size_t

        from .._lib import call, ptr, stream_ptr

        x = logits.detach()
        if x.dtype != torch.float32 or x.stride(1) != 1:
            x = x.to(torch.float32).contiguous()
        N, C = x.shape
        # one int64 buffer: [0, C) packed best keys (workspace), [C, 2C) indices, then C float32 probs and C int32 flags
        buf = torch.empty(3 * C, dtype=torch.int64, device=x.device)
        prob = buf[2 * C:].view(torch.float32)[:C]
        sel = buf[2 * C:].view(torch.int32)[C:2 * C]
        call("sv_key_point_predictions", ptr(x), c_int64(x.stride(0)), c_int(C), c_int64(N), c_float(conf_th), ptr(buf),
             c_size_t(8 * C), ptr(prob), ptr(buf[C:2 * C]), ptr(sel), stream_ptr())
        host = buf.cpu()
        p = host[2 * C:].view(torch.float32)[:C]
        classes = np.where(host[2 * C:].view(torch.int32)[C:2 * C].numpy() != 0)[0]
        return host[C:2 * C].numpy()[classes], classes, p[classes]
    softmax = logits.softmax(1).max(0)
    classes = np.where(softmax[0].cpu() > conf_th)[0]
    idx = softmax[1][classes].cpu().numpy()
    return idx, classes, softmax[0].cpu()[classes]


def get_pred_center(out, coords, ee_r=0.03, q=None):
    """mean of the 8 highest-vote points, optionally moved by (-ee_r, 0, 0) rotated by the quaternion q (w first)
    (utils/output.py:45-64)."""
    if torch.is_tensor(out) and out.is_cuda:
        sel = topk_indices(out[:, 1], 8)  # one selection pass instead of a full sort of every point's vote
    else:
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


def topk_indices(column, k):
    """rows of the k largest entries of a float32 CUDA column (any stride), largest first, ties to the lower row - the
    `column.sort(descending=True)[1][:k]` of utils/output.py:47 as a selection (sv_topk_indices), int64 CUDA [min(k, n)]."""
    from ctypes import c_int, c_int64, c_size_t

    from .._lib import call, load, ptr, stream_ptr

    x = column.detach()
    if x.dtype != torch.float32:
        x = x.to(torch.float32)
    n = x.shape[0]
    nbytes = load().sv_topk_workspace_bytes(c_int64(n), c_int(k))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    idx = torch.empty(k, dtype=torch.int64, device=x.device)
    call("sv_topk_indices", ptr(x), c_int64(x.stride(0) if n > 1 else 1), c_int64(n), c_int(k), ptr(ws), c_size_t(nbytes), ptr(idx),
         stream_ptr())
    return idx[: min(k, n)]


def select_equal(values, value):
    """ascending positions i with values[i] == value, on the device (np.where(values == value)[0] of
    utils/output.py:24 / app/inference_engine.py:420 without the host round trip).  values: int32 / int64 CUDA tensor;
    value: python int or a 1-element int32 CUDA tensor."""
    from ctypes import c_int, c_int64

    from .._lib import call, ptr, stream_ptr

    v = values.contiguous()
    if v.dtype not in (torch.int32, torch.int64):
        raise TypeError("select_equal: int32 or int64 values")
    n = v.numel()
    out = torch.empty(n, dtype=torch.int64, device=v.device)
    count = torch.empty(1, dtype=torch.int64, device=v.device)
    on_dev = torch.is_tensor(value)
    call("sv_select_equal", ptr(v), c_int(v.element_size()), c_int64(n), c_int64(0 if on_dev else int(value)),
         ptr(value) if on_dev else None, ptr(out), ptr(count), stream_ptr())
    return out[: int(count.item())]


def single_linkage_roots(points, dist, idx=None):
    """Connected components of "distance < dist" among the rows idx (None = all) of a CUDA [*, >=3] float32 / float64
    tensor: (root int32[n] = smallest member position of each point's component, best int32[2] = root and size of the
    largest component) - sv_single_linkage_roots."""
    from ctypes import c_double, c_int, c_int64, c_size_t

    from .._lib import call, load, ptr, stream_ptr

    if points.dtype not in (torch.float32, torch.float64) or points.stride(1) != 1:
        points = points.to(torch.float32).contiguous()
    n = int(points.shape[0] if idx is None else idx.numel())
    if idx is not None:
        idx = idx.to(torch.int32).contiguous()
    root = torch.empty(n, dtype=torch.int32, device=points.device)
    best = torch.zeros(2, dtype=torch.int32, device=points.device)
    nbytes = load().sv_cluster_workspace_bytes(c_int64(n))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=points.device)
    call("sv_single_linkage_roots", ptr(points), c_int(points.element_size()), c_int64(points.stride(0)), ptr(idx),
         c_int64(n), c_double(float(dist)), ptr(ws), c_size_t(nbytes), ptr(root), ptr(best), stream_ptr())
    return root, best


class ClusterUtil:
    """Largest single-linkage cluster of the end-effector points (utils/output.py:13-28: sklearn
    AgglomerativeClustering(distance_threshold=0.06, linkage='single')).  Single linkage with a distance threshold
    is exactly the connected components of the graph "distance < threshold".  CUDA tensors take the device path
    (lock-free union-find over all pairs, csrc/sv_cluster.hip: 4 096 points in well under a millisecond where the
    reference's linkage - and this class's host path - take hundreds); numpy inputs keep the host path (k-d tree +
    scipy connected components), which the tests pin against sklearn itself.  Equal-size clusters: the reference takes
    whichever label np.unique lists first (sklearn's numbering, arbitrary); both paths here take the cluster that
    contains the lowest point index."""

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

    def get_largest_cluster(self, points, idx=None):
        """positions (ascending) of the largest cluster's members.  CUDA tensor in -> int64 CUDA tensor out; idx
        (device path only) restricts the points to those rows, positions then refer to idx."""
        if torch.is_tensor(points) and points.is_cuda:
            n = int(points.shape[0] if idx is None else idx.numel())
            if n == 0:
                return torch.empty(0, dtype=torch.int64, device=points.device)
            root, best = single_linkage_roots(points, self.dist, idx)
            return select_equal(root, best[:1])
        labels = self.labels(points)
        # scipy numbers components by their first member, so the first maximum is the cluster with the lowest index
        unique, counts = np.unique(labels, return_counts=True)
        return np.where(labels == unique[counts.argmax()])[0]
