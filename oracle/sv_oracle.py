"""CPU ORACLE for the sparse-voxel hot path — TEST INFRASTRUCTURE, never imported by the product package.

numpy restatement (integer/index work) + ctypes calls into oracle/sv_oracle.c (float arithmetic with a defined
order) of what the reference computes through MinkowskiEngine 0.5.4, numpy and scipy on the path
voxelise -> sparse U-Net -> slice/argmax -> Kabsch.  Each function cites the reference lines it follows.

PARITY STATUS
  * sparse path (voxelise, maps, conv graph, pooling, slice): **parity unpinned** — MinkowskiEngine is an un-vendored
    third-party dependency (requirements.txt:101), is not installable here and the reference holds no golden
    vectors for it.  The semantics restated are ME's public API semantics;
    self-consistency is checked against dense torch conv3d / BatchNorm1d / Linear in tests/.
  * ICP (icp_point2point): **parity unpinned** — open3d (utils/icp.py:5) is not installable here; restated from
    Open3D's published registration_icp loop.
  * dense solves, metrics, FPS, ball query: pinned by tests/golden/*.npz generated from the reference's own
    utils/transformation.py, utils/calibration.py, utils/metrics.py, utils/data.py, model/pointnet2_utils.py
    (tools/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os

import numpy as np

NUM_THREADS = None  # threads of or_conv_fwd; None = OpenMP's default (OMP_NUM_THREADS, which torchrun sets to 1 per rank)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsvoracle.so")
_lib = None

COORD_BIAS = 1 << 17
COORD_BITS = 18
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
POOL_MAX, POOL_AVG = 0, 1


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} missing: run `make -C oracle` (or __graft_entry__.build())")
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.or_num_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ------------------------------------------------------------------------------------------------------------------
# keys / voxelisation   (ME.TensorField(...).sparse(): app/inference_engine.py:405-415)
# ------------------------------------------------------------------------------------------------------------------
def _part1by2(v):
    v = v.astype(np.uint64) & np.uint64(0x3FFFF)
    out = np.zeros_like(v)
    for b in range(COORD_BITS):
        out |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
    return out


def make_keys(bxyz):
    """uint64 key = batch << 54 | morton3(x + 2^17, y + 2^17, z + 2^17), x in bit 3j (include/sv_hip.h)."""
    bxyz = np.asarray(bxyz, dtype=np.int64)
    b, x, y, z = (bxyz[:, i] for i in range(4))
    if ((b < 0) | (b >= 1024)).any() or (np.abs(bxyz[:, 1:]) >= COORD_BIAS).any():
        raise ValueError("coordinate outside the key range")
    return ((b.astype(np.uint64) << np.uint64(54)) | _part1by2(x + COORD_BIAS) | (_part1by2(y + COORD_BIAS) << np.uint64(1))
            | (_part1by2(z + COORD_BIAS) << np.uint64(2)))


def voxelize(coords4, coords_are_int=False):
    """coords4 [N,4] = (batch, x*scale, y*scale, z*scale): voxel = floor(coord) (ME quantisation), unique voxels in
    canonical key order.  Returns dict(keys, coords int32[V,4], inverse int64[N], order int32[N], seg_start)."""
    c = np.asarray(coords4)
    if coords_are_int:
        q = c.astype(np.int64)
    else:
        c = c.astype(np.float32)
        q = np.empty(c.shape, dtype=np.int64)
        q[:, 0] = c[:, 0].astype(np.int64)
        q[:, 1:] = np.floor(c[:, 1:]).astype(np.int64)
    keys = make_keys(q)
    order = np.argsort(keys, kind="stable")
    sk = keys[order]
    head = np.ones(len(sk), dtype=bool)
    head[1:] = sk[1:] != sk[:-1]
    rank = np.cumsum(head) - 1
    inverse = np.empty(len(sk), dtype=np.int64)
    inverse[order] = rank
    seg_start = np.concatenate([np.nonzero(head)[0], [len(sk)]]).astype(np.int32)
    ukeys = sk[head]
    ucoords = q[order][head].astype(np.int32)
    return dict(keys=ukeys, coords=ucoords, inverse=inverse, order=order.astype(np.int32), seg_start=seg_start)


def sparse_quantize(coordinates, features=None, labels=None, quantization_size=None, ignore_label=-100):
    """ME.utils.sparse_quantize as called by data/alivev2.py:290-296 (**parity unpinned**: ME 0.5.4's
    utils/quantization.py is not installable here; restated from its public contract): voxel = floor(coordinate /
    quantization_size); one row per occupied voxel, represented by its FIRST point in input order; a voxel whose points
    carry more than one distinct label gets ignore_label.  Row order = this build's canonical key order (ME's is
    unspecified).  Returns (int32 coords [V, 3 or 4], features of the representatives, labels, index, inverse)."""
    c = np.asarray(coordinates)
    if quantization_size is not None:
        c = np.floor(c.astype(np.float64) / quantization_size)
    elif np.issubdtype(c.dtype, np.floating):
        c = np.floor(c)
    c = c.astype(np.int64)
    c4 = c if c.shape[1] == 4 else np.concatenate([np.zeros((len(c), 1), np.int64), c], axis=1)
    keys = make_keys(c4)
    ukeys, first, inverse = np.unique(keys, return_index=True, return_inverse=True)  # first occurrence per key
    out = [c[first].astype(np.int32)]
    out.append(None if features is None else np.asarray(features)[first])
    if labels is not None:
        lab = np.asarray(labels).astype(np.int64)
        lo = np.full(len(ukeys), np.iinfo(np.int64).max)
        hi = np.full(len(ukeys), np.iinfo(np.int64).min)
        np.minimum.at(lo, inverse, lab)
        np.maximum.at(hi, inverse, lab)
        out.append(np.where(lo == hi, lab[first], ignore_label).astype(np.asarray(labels).dtype))
    else:
        out.append(None)
    out += [first.astype(np.int64), inverse.astype(np.int64)]
    return tuple(out)


def voxel_reduce(feats, order, seg_start, mode=0):
    feats = _f32(feats)
    V = len(seg_start) - 1
    out = np.empty((V, feats.shape[1]), dtype=np.float32)
    lib().or_voxel_reduce(_p(feats), ctypes.c_int(feats.shape[1]), _p(order), _p(seg_start), ctypes.c_int64(V),
                          ctypes.c_int(mode), _p(out))
    return out


# ------------------------------------------------------------------------------------------------------------------
# coordinate maps and kernel maps  (ME coordinate manager)
# ------------------------------------------------------------------------------------------------------------------
def stride_map(coords, stride_in):
    """kernel_size 2 / stride 2 output coordinates: unique floor(c / (2 ts)) * (2 ts), canonical order.
    Returns (coords_out int32[Vc,4], parent int64[V_in])."""
    ts2 = 2 * stride_in
    c = coords.astype(np.int64).copy()
    c[:, 1:] = np.floor_divide(c[:, 1:], ts2) * ts2
    keys = make_keys(c)
    ukeys, first, parent = np.unique(keys, return_index=True, return_inverse=True)
    return c[first].astype(np.int32), parent.astype(np.int64)


def _lookup(keys_sorted, query_keys, valid):
    pos = np.searchsorted(keys_sorted, query_keys)
    pos[pos >= len(keys_sorted)] = 0
    hit = valid & (keys_sorted[pos] == query_keys) if len(keys_sorted) else np.zeros(len(query_keys), bool)
    return np.where(hit, pos, -1).astype(np.int32)


def kernel_map_k3(coords, tensor_stride, dilation=1):
    """nbr[k][o] = row of coords[o] + offset_k * tensor_stride * dilation or -1; k = (dx+1) + 3(dy+1) + 9(dz+1)."""
    V = len(coords)
    keys = make_keys(coords)
    nbr = np.full((27, V), -1, dtype=np.int32)
    step = tensor_stride * dilation
    k = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                q = coords.astype(np.int64).copy()
                q[:, 1] += dx * step
                q[:, 2] += dy * step
                q[:, 3] += dz * step
                valid = (np.abs(q[:, 1:]) < COORD_BIAS).all(axis=1)
                q[~valid, 1:] = 0
                nbr[k] = _lookup(keys, make_keys(q), valid)
                k += 1
    return nbr


def _child_id(coords_fine, stride_fine):
    c = coords_fine.astype(np.int64)
    ts2 = 2 * stride_fine
    d = (c[:, 1:] - np.floor_divide(c[:, 1:], ts2) * ts2) // stride_fine  # 0/1 per axis
    return (d[:, 0] + 2 * d[:, 1] + 4 * d[:, 2]).astype(np.int64)


def kernel_map_down(coords_fine, stride_fine, parent, V_coarse):
    """kernel_size 2 stride 2: out rows = coarse voxels, offset k = dx + 2 dy + 4 dz of the child inside its parent."""
    nbr = np.full((8, V_coarse), -1, dtype=np.int32)
    cid = _child_id(coords_fine, stride_fine)
    nbr[cid, parent] = np.arange(len(coords_fine), dtype=np.int32)
    return nbr


def kernel_map_up(coords_fine, stride_fine, parent):
    """transposed kernel_size 2 stride 2 onto the existing fine map: fine voxel i reads its parent with weight k(i)."""
    V = len(coords_fine)
    nbr = np.full((8, V), -1, dtype=np.int32)
    cid = _child_id(coords_fine, stride_fine)
    nbr[cid, np.arange(V)] = parent.astype(np.int32)
    return nbr


# ------------------------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------------------------
def fold_bn(weight, bias, running_mean, running_var, eps=1e-5):
    """BatchNorm1d(eval) (ME.MinkowskiBatchNorm, Appendix B.6) as y = fmaf(x, scale, shift), float32 throughout."""
    w, b = _f32(weight), _f32(bias)
    mean, var = _f32(running_mean), _f32(running_var)
    scale = w / np.sqrt(var + np.float32(eps))
    shift = b - mean * scale
    return scale.astype(np.float32), shift.astype(np.float32)


def conv(feats, W, nbr, V_out, scale=None, shift=None, residual=None, act=ACT_NONE, slope=0.01, nthreads=None):
    """W [K,Cin,Cout]; nbr int32[K][V_out] or None (identity, K == 1)."""
    feats = _f32(feats)
    W = _f32(W)
    if W.ndim == 2:
        W = W[None]
    K, Cin, Cout = W.shape
    assert feats.shape[1] == Cin
    out = np.empty((V_out, Cout), dtype=np.float32)
    if nbr is not None:
        nbr = np.ascontiguousarray(nbr, dtype=np.int32)
        assert nbr.shape == (K, V_out)
    scale = None if scale is None else _f32(scale)
    shift = None if shift is None else _f32(shift).reshape(-1)
    residual = None if residual is None else _f32(residual)
    if nthreads is None:
        nthreads = NUM_THREADS or lib().or_num_threads()
    lib().or_conv_fwd(_p(feats), ctypes.c_int64(feats.shape[1]), ctypes.c_int(Cin), _p(W), ctypes.c_int(K),
                      ctypes.c_int(Cout), _p(nbr), ctypes.c_int64(V_out), ctypes.c_int64(V_out), _p(scale), _p(shift),
                      _p(residual), ctypes.c_int64(Cout), ctypes.c_int(act), ctypes.c_float(slope), _p(out),
                      ctypes.c_int64(Cout), ctypes.c_int(nthreads))
    return out


def affine_act(feats, scale=None, shift=None, residual=None, act=ACT_NONE, slope=0.01):
    feats = _f32(feats)
    V, C = feats.shape
    out = np.empty((V, C), dtype=np.float32)
    scale = None if scale is None else _f32(scale)
    shift = None if shift is None else _f32(shift)
    residual = None if residual is None else _f32(residual)
    lib().or_affine_act(_p(feats), ctypes.c_int64(C), ctypes.c_int(C), ctypes.c_int64(V), _p(scale), _p(shift),
                        _p(residual), ctypes.c_int64(C), ctypes.c_int(act), ctypes.c_float(slope), _p(out),
                        ctypes.c_int64(C))
    return out


def batch_offsets(coords, B):
    return np.searchsorted(coords[:, 0], np.arange(B + 1), side="left").astype(np.int32)


def global_pool(feats, coords, mode, B=None):
    feats = _f32(feats)
    if B is None:
        B = int(coords[:, 0].max()) + 1
    bs = batch_offsets(coords, B)
    out = np.empty((B, feats.shape[1]), dtype=np.float32)
    lib().or_global_pool(_p(feats), ctypes.c_int64(feats.shape[1]), ctypes.c_int(feats.shape[1]), _p(bs),
                         ctypes.c_int(B), ctypes.c_int(mode), _p(out))
    return out


def slice_argmax(feats, inverse):
    feats = _f32(feats)
    inverse = np.ascontiguousarray(inverse, dtype=np.int64)
    N = len(inverse)
    label = np.empty(N, dtype=np.int64)
    conf = np.empty(N, dtype=np.float32)
    lib().or_slice_argmax(_p(feats), ctypes.c_int64(feats.shape[1]), ctypes.c_int(feats.shape[1]), _p(inverse),
                          ctypes.c_int64(N), _p(label), _p(conf))
    return label, conf


# ------------------------------------------------------------------------------------------------------------------
# the sparse U-Net graph, restated from the reference's model files on top of the ops above
# ------------------------------------------------------------------------------------------------------------------
class Frame:
    """Coordinate manager of one voxelised input: maps per stride + cached kernel maps."""

    def __init__(self, coords):
        self.maps = {1: coords}
        self.parent = {}
        self.cache = {}

    def down(self, stride):
        if 2 * stride not in self.maps:
            c, p = stride_map(self.maps[stride], stride)
            self.maps[2 * stride] = c
            self.parent[stride] = p
        return self.maps[2 * stride]

    def k3(self, stride):
        key = ("k3", stride)
        if key not in self.cache:
            self.cache[key] = kernel_map_k3(self.maps[stride], stride)
        return self.cache[key]

    def kdown(self, stride):
        key = ("down", stride)
        if key not in self.cache:
            coarse = self.down(stride)
            self.cache[key] = kernel_map_down(self.maps[stride], stride, self.parent[stride], len(coarse))
        return self.cache[key]

    def kup(self, stride):
        key = ("up", stride)
        if key not in self.cache:
            self.cache[key] = kernel_map_up(self.maps[stride // 2], stride // 2, self.parent[stride // 2])
        return self.cache[key]


def _np(sd, key):
    v = sd[key]
    return v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)


def _bn(sd, prefix, eps=1e-5):
    return fold_bn(_np(sd, prefix + ".bn.weight"), _np(sd, prefix + ".bn.bias"), _np(sd, prefix + ".bn.running_mean"),
                   _np(sd, prefix + ".bn.running_var"), eps)


def _kernel3(sd, key):
    w = _np(sd, key)
    return w if w.ndim == 3 else w[None]


def basic_block(sd, prefix, x, frame, stride, has_downsample):
    """ME BasicBlock (resnet_block): conv3-BN-ReLU-conv3-BN-(+residual)-ReLU; residual path 1x1 conv + BN when the
    width changes (model/backbone/resnet.py:95-107)."""
    nbr = frame.k3(stride)
    V = x.shape[0]
    s1, b1 = _bn(sd, prefix + ".norm1")
    out = conv(x, _kernel3(sd, prefix + ".conv1.kernel"), nbr, V, s1, b1, None, ACT_RELU)
    if has_downsample:
        sd_, bd_ = _bn(sd, prefix + ".downsample.1")
        res = conv(x, _kernel3(sd, prefix + ".downsample.0.kernel"), None, V, sd_, bd_)
    else:
        res = x
    s2, b2 = _bn(sd, prefix + ".norm2")
    return conv(out, _kernel3(sd, prefix + ".conv2.kernel"), nbr, V, s2, b2, res, ACT_RELU)


def bottleneck_block(sd, prefix, x, frame, stride, has_downsample):
    """ME Bottleneck (resnet_block, expansion 4): 1x1-BN-ReLU, 3x3x3-BN-ReLU, 1x1-BN, (+residual), ReLU; residual
    path 1x1 conv + BN when the width changes (model/backbone/resnet.py:95-107 builds it the same way)."""
    V = x.shape[0]
    s1, b1 = _bn(sd, prefix + ".norm1")
    out = conv(x, _kernel3(sd, prefix + ".conv1.kernel"), None, V, s1, b1, None, ACT_RELU)
    s2, b2 = _bn(sd, prefix + ".norm2")
    out = conv(out, _kernel3(sd, prefix + ".conv2.kernel"), frame.k3(stride), V, s2, b2, None, ACT_RELU)
    if has_downsample:
        sd_, bd_ = _bn(sd, prefix + ".downsample.1")
        res = conv(x, _kernel3(sd, prefix + ".downsample.0.kernel"), None, V, sd_, bd_)
    else:
        res = x
    s3, b3 = _bn(sd, prefix + ".norm3")
    return conv(out, _kernel3(sd, prefix + ".conv3.kernel"), None, V, s3, b3, res, ACT_RELU)


def _block_stack(sd, name, x, frame, stride):
    i = 0
    while f"{name}.{i}.conv1.kernel" in sd:
        block = bottleneck_block if f"{name}.{i}.conv3.kernel" in sd else basic_block
        x = block(sd, f"{name}.{i}", x, frame, stride, f"{name}.{i}.downsample.0.kernel" in sd)
        i += 1
    return x


def minkunet_encoder(sd, feats, frame, levels=4):
    """model/backbone/minkunet.py:126-148 — conv0(k3)+BN+ReLU, then 4 x [conv k2 s2 + BN + ReLU, block]
    (model/backbone/aliveunet.py:178-216: the same with 7 stages)."""
    V0 = feats.shape[0]
    s, b = _bn(sd, "bn0")
    out_p1 = conv(feats, _kernel3(sd, "conv0p1s1.kernel"), frame.k3(1), V0, s, b, None, ACT_RELU)
    skips = [out_p1]
    out = out_p1
    for i in range(1, levels + 1):
        ts = 2 ** (i - 1)
        coarse = frame.down(ts)
        s, b = _bn(sd, f"bn{i}")
        out = conv(out, _kernel3(sd, f"conv{i}p{ts}s2.kernel"), frame.kdown(ts), len(coarse), s, b, None, ACT_RELU)
        out = _block_stack(sd, f"block{i}", out, frame, 2 * ts)
        skips.append(out)
    return skips


def minkunet_forward_except_final(sd, feats, frame):
    """model/backbone/minkunet.py:125-183."""
    skips = minkunet_encoder(sd, feats, frame)
    out = skips.pop()
    for j in range(4, 8):
        ts = 2 ** (8 - j)
        s, b = _bn(sd, f"bntr{j}")
        fine = frame.maps[ts // 2]
        out = conv(out, _kernel3(sd, f"convtr{j}p{ts}s2.kernel"), frame.kup(ts), len(fine), s, b, None, ACT_RELU)
        out = np.concatenate([out, skips.pop()], axis=1)  # ME.cat
        out = _block_stack(sd, f"block{j + 1}", out, frame, ts // 2)
    return out


def alive_unet_forward(sd, feats, frame):
    """model/backbone/aliveunet.py:177-265: 7 stages down to tensor stride 128, then convtr{j} + BN + ReLU,
    ME.cat with the encoder tensor of that stride, block{j+1} for j = 7..13; returns block14's output (`final` is
    constructed but never applied, :264-265)."""
    skips = minkunet_encoder(sd, feats, frame, levels=7)
    out = skips.pop()
    for j in range(7, 14):
        ts = 2 ** (14 - j)
        s, b = _bn(sd, f"bntr{j}")
        fine = frame.maps[ts // 2]
        out = conv(out, _kernel3(sd, f"convtr{j}.kernel"), frame.kup(ts), len(fine), s, b, None, ACT_RELU)
        out = np.concatenate([out, skips.pop()], axis=1)  # ME.cat(out, out_b{.}p{ts/2})
        out = _block_stack(sd, f"block{j + 1}", out, frame, ts // 2)
    return out


def minkunet_forward(sd, feats, frame, act=ACT_NONE, slope=0.01):
    """model/backbone/minkunet.py:185-187: final 1x1 conv with bias."""
    out = minkunet_forward_except_final(sd, feats, frame)
    return conv(out, _kernel3(sd, "final.kernel"), None, out.shape[0], None, _np(sd, "final.bias").reshape(-1), None,
                act, slope)


def robotnet_segmentation_forward(sd, feats, frame):
    """model/robotnet_segmentation.py:55-64 (and robotnet_vote.py:62-71): U-Net -> LeakyReLU -> Linear 256->1024 ->
    LeakyReLU -> Linear 1024->classes.  ME.MinkowskiLinear = nn.Linear on feature rows."""
    out = minkunet_forward(sd, feats, frame, ACT_LEAKY, 0.01)
    V = out.shape[0]
    out = conv(out, _np(sd, "regression.0.linear.weight").T[None], None, V, None, _np(sd, "regression.0.linear.bias"),
               None, ACT_LEAKY, 0.01)
    return conv(out, _np(sd, "regression.2.linear.weight").T[None], None, V, None, _np(sd, "regression.2.linear.bias"))


def _pose_mlp(sd, pooled, training=False):
    h = conv(pooled, _np(sd, "pose_regression.0.weight").T[None], None, pooled.shape[0], None,
             _np(sd, "pose_regression.0.bias"), None, ACT_LEAKY, 0.01)
    out = conv(h, _np(sd, "pose_regression.2.weight").T[None], None, pooled.shape[0], None,
               _np(sd, "pose_regression.2.bias"))
    out[:, 7:] = 1.0 / (1.0 + np.exp(-out[:, 7:]))
    if not training:
        n = np.maximum(np.linalg.norm(out[:, 3:7], axis=1, keepdims=True), 1e-12)
        out[:, 3:7] = out[:, 3:7] / n
    return out


def _with_joint_angles(pooled, joint_angles):
    """model/robotnet.py:68-71: STRUCTURE.use_joint_angles concatenates the 9 joint angles behind the pooled features."""
    if joint_angles is None:
        return pooled
    return np.ascontiguousarray(np.concatenate([pooled, np.asarray(joint_angles, np.float32)], axis=1))


def robotnet_forward(sd, feats, frame, backbone=None, joint_angles=None):
    """model/robotnet.py:62-83: forward_except_final -> BN+ReLU -> global max pool -> [cat joint angles] -> MLP
    -> sigmoid on [:, 7:] (the three confidences of STRUCTURE.compute_confidence: out_channels = 10) -> normalise quaternion.
    backbone: the U-Net body (default MinkUNet's forward_except_final; alive_unet_forward for the fallback backbone,
    model/robotnet.py:29-30)."""
    out = (backbone or minkunet_forward_except_final)(sd, feats, frame)
    s, b = _bn(sd, "output_layer.0")
    out = affine_act(out, s, b, None, ACT_RELU)
    pooled = global_pool(out, frame.maps[1], POOL_MAX)
    return _pose_mlp(sd, _with_joint_angles(pooled, joint_angles))


def robotnet_encode_forward(sd, feats, frame, joint_angles=None, quantization_size=None):
    """model/robotnet_encode.py:68-119: encoder to stride 16 -> BN+ReLU -> global avg pool -> [cat joint angles] -> MLP;
    DATA.voxelize_position (eval only, :114-117): position *= quantization_size."""
    out = minkunet_encoder(sd, feats, frame)[-1]
    s, b = _bn(sd, "output_layer.0")
    out = affine_act(out, s, b, None, ACT_RELU)
    pooled = global_pool(out, frame.maps[16], POOL_AVG)
    res = _pose_mlp(sd, _with_joint_angles(pooled, joint_angles))
    if quantization_size is not None:
        res[:, :3] *= np.float32(quantization_size)
    return res


def predict_segmentation(sd, points, rgb, scale):
    """app/inference_engine.py:395-419 up to the labels: voxelise raw points * scale (F8a: the centred copy is
    discarded), U-Net, slice, row max."""
    coords4 = np.concatenate([np.zeros((len(points), 1), np.float32), np.asarray(points, np.float32) * np.float32(scale)],
                             axis=1)
    vox = voxelize(coords4)
    feats = voxel_reduce(rgb, vox["order"], vox["seg_start"], 0)
    frame = Frame(vox["coords"])
    logits = robotnet_segmentation_forward(sd, feats, frame)
    label, conf = slice_argmax(logits, vox["inverse"])
    return dict(vox=vox, frame=frame, logits=logits, label=label, conf=conf)


# ------------------------------------------------------------------------------------------------------------------
# dense solves and metrics (pinned by tests/golden)
# ------------------------------------------------------------------------------------------------------------------
def get_rigid_transform_3D(reference, target):
    """utils/transformation.py:178-222: Kabsch via SVD with the reflection fix on Vt[2]."""
    A = np.asarray(reference, dtype=np.float64).T
    B = np.asarray(target, dtype=np.float64).T
    cA = A.mean(axis=1).reshape(-1, 1)
    cB = B.mean(axis=1).reshape(-1, 1)
    H = (A - cA) @ (B - cB).T
    U, S, Vt = np.linalg.svd(H)
    R = Vt.T @ U.T
    if np.linalg.det(R) < 0:
        Vt[2, :] *= -1
        R = Vt.T @ U.T
    t = -R @ cA + cB
    return R, t.reshape(-1)


def get_q_from_matrix(rot_mat):
    """utils/transformation.py:80-84: scipy Rotation.from_matrix(R).as_quat() (xyzw) reordered to wxyz."""
    from scipy.spatial.transform import Rotation

    q = Rotation.from_matrix(np.array(rot_mat, copy=True)).as_quat()
    return np.insert(q[:3], 0, q[-1])


def get_quaternion_rotation_matrix(Q):
    """utils/transformation.py:16-60 with switch_w=False: Q = (w, x, y, z)."""
    q0, q1, q2, q3 = Q
    return np.array([
        [2 * (q0 * q0 + q1 * q1) - 1, 2 * (q1 * q2 - q0 * q3), 2 * (q1 * q3 + q0 * q2)],
        [2 * (q1 * q2 + q0 * q3), 2 * (q0 * q0 + q2 * q2) - 1, 2 * (q2 * q3 - q0 * q1)],
        [2 * (q1 * q3 - q0 * q2), 2 * (q2 * q3 + q0 * q1), 2 * (q0 * q0 + q3 * q3) - 1],
    ])


def compute_quaternions_weighted_average(Q, w):
    """utils/calibration.py:69-95: principal eigenvector of sum w_i q_i q_i^T / sum w_i (np.linalg.eig, real part)."""
    Q = np.asarray(Q, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    A = np.zeros((4, 4))
    for i in range(Q.shape[0]):
        A = w[i] * np.outer(Q[i], Q[i]) + A
    A = (1.0 / w.sum()) * A
    vals, vecs = np.linalg.eig(A)
    vecs = vecs[:, vals.argsort()[::-1]]
    return np.real(vecs[:, 0])


def compute_poses_average(poses, weights=None):
    """utils/calibration.py:108-139."""
    poses = np.asarray(poses, dtype=np.float64)
    if len(poses) == 1:
        return poses[0]
    if weights is None or len(weights) != len(poses):
        weights = np.ones(len(poses))
    weights = np.asarray(weights, dtype=np.float64)
    out = np.zeros(7)
    out[:3] = np.sum(poses[:, :3] * weights.reshape(-1, 1), axis=0) / weights.sum()
    out[3:] = compute_quaternions_weighted_average(poses[:, 3:], weights)
    return out


def compute_ADD_np(points, gt_pose, pred_pose):
    """utils/metrics.py:139-150."""
    Rg = get_quaternion_rotation_matrix(gt_pose[3:])
    Rp = get_quaternion_rotation_matrix(pred_pose[3:])
    g = (Rg @ points.T) + gt_pose[:3].reshape(3, 1)
    p = (Rp @ points.T) + pred_pose[:3].reshape(3, 1)
    return np.linalg.norm(g - p, axis=0).mean()


# ------------------------------------------------------------------------------------------------------------------
# PointNet++ sampling / grouping
# ------------------------------------------------------------------------------------------------------------------
def farthest_point_sample(xyz, npoint, start):
    """model/pointnet2_utils.py:65-86 / utils/data.py:13-34 with the random first index passed in.  float32
    distances ((dx^2 + dy^2) + dz^2), argmax = first maximum."""
    xyz = np.asarray(xyz, dtype=np.float32)
    B, N, _ = xyz.shape
    out = np.zeros((B, npoint), dtype=np.int64)
    for b in range(B):
        dist = np.full(N, 1e10, dtype=np.float32)
        far = int(start[b])
        for i in range(npoint):
            out[b, i] = far
            d = xyz[b] - xyz[b, far]
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            m = d2 < dist
            dist[m] = d2[m]
            far = int(np.argmax(dist))
    return out


def query_ball_point(radius, nsample, xyz, new_xyz):
    """model/pointnet2_utils.py:89-109: first nsample indices (ascending) with dist <= r^2, padded with the first;
    dist in the reference's expanded float32 form -2 q.p + |q|^2 + |p|^2."""
    xyz = np.asarray(xyz, dtype=np.float32)
    new_xyz = np.asarray(new_xyz, dtype=np.float32)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    r2 = np.float32(radius * radius)
    out = np.empty((B, S, nsample), dtype=np.int64)
    for b in range(B):
        p = xyz[b]
        pp = (p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]) + p[:, 2] * p[:, 2]
        for s in range(S):
            q = new_xyz[b, s]
            qq = (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]
            dot = (q[0] * p[:, 0] + q[1] * p[:, 1]) + q[2] * p[:, 2]
            d = (np.float32(-2.0) * dot + qq) + pp
            idx = np.nonzero(~(d > r2))[0]
            row = np.full(nsample, idx[0] if len(idx) else N, dtype=np.int64)
            row[: min(nsample, len(idx))] = idx[:nsample]
            out[b, s] = row
    return out


# ------------------------------------------------------------------------------------------------------------------
# ICP (parity unpinned: open3d is not installable here; restated from Open3D's published registration_icp loop with
# TransformationEstimationPointToPoint, the call made by utils/icp.py:66-72)
# ------------------------------------------------------------------------------------------------------------------
def icp_point2point(src, tgt, init_T=None, max_distance=0.1, max_iterations=30, rel_fitness=1e-6, rel_rmse=1e-6):
    """Returns (T 4x4, fitness, inlier rmse, updates applied).  Nearest neighbour by brute force in float32 (first
    minimum), inliers d <= max_distance, update = Kabsch on the inlier pairs, stop when fitness and rmse both move by
    less than the tolerances between two evaluations."""
    src = np.asarray(src, dtype=np.float32)
    tgt = np.asarray(tgt, dtype=np.float32)
    T = np.eye(4) if init_T is None else np.array(init_T, dtype=np.float64)
    max_d2 = np.float32(max_distance * max_distance)
    prev = None
    updates = 0
    for it in range(max_iterations + 1):
        p = (src.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
        nn = np.empty(len(p), dtype=np.int64)
        d2 = np.empty(len(p), dtype=np.float32)
        for s in range(0, len(p), 512):  # blocked brute force
            d = p[s:s + 512, None, :] - tgt[None, :, :]
            dd = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
            nn[s:s + 512] = dd.argmin(axis=1)
            d2[s:s + 512] = dd[np.arange(dd.shape[0]), nn[s:s + 512]]
        inl = d2 <= max_d2
        n = int(inl.sum())
        fitness = n / len(p)
        rmse = float(np.sqrt(d2[inl].astype(np.float64).sum() / n)) if n else 0.0
        if prev is not None and abs(prev[0] - fitness) < rel_fitness and abs(prev[1] - rmse) < rel_rmse:
            break
        prev = (fitness, rmse)
        if n < 3 or it == max_iterations:
            break
        pt = src[inl].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
        R, t = get_rigid_transform_3D(pt, tgt[nn[inl]].astype(np.float64))
        U = np.eye(4)
        U[:3, :3], U[:3, 3] = R, t
        T = U @ T
        updates += 1
    return T, fitness, rmse, updates


# ------------------------------------------------------------------------------------------------------------------
# N4: largest single-linkage cluster (utils/output.py:13-28 ClusterUtil.get_largest_cluster; app/inference_engine.py:422-433)
# ------------------------------------------------------------------------------------------------------------------
def single_linkage_roots(points, dist=0.06):
    """sklearn AgglomerativeClustering(linkage="single", distance_threshold=dist) merges while the linkage distance is
    below dist, i.e. its clusters are the connected components of the graph "distance < dist".  Restated as a plain
    O(n^2) flood fill in float64 (distance = sqrt((dx*dx + dy*dy) + dz*dz)); root[i] = the smallest member of i's
    component.  Pinned against sklearn itself in tests/test_cluster_cpu.py."""
    pts = np.asarray(points, dtype=np.float64)
    n = len(pts)
    root = np.full(n, -1, dtype=np.int64)
    for s in range(n):
        if root[s] >= 0:
            continue
        root[s] = s
        stack = [s]
        while stack:
            i = stack.pop()
            d = pts - pts[i]
            near = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]) < dist
            new = np.where(near & (root < 0))[0]
            root[new] = s
            stack.extend(new.tolist())
    return root


def largest_cluster(points, dist=0.06):
    """positions (ascending) of the largest component; equal sizes: the one containing the lowest index (the reference
    takes whichever label sklearn happens to number first - arbitrary)."""
    root = single_linkage_roots(points, dist)
    if len(root) == 0:
        return np.zeros(0, dtype=np.int64)
    u, c = np.unique(root, return_counts=True)
    return np.where(root == u[c.argmax()])[0]

