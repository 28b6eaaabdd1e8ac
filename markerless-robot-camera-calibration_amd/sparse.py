"""Sparse-tensor runtime behind the MinkowskiEngine names the reference uses.

  TensorField(features, coordinates, quantization_mode, minkowski_algorithm, device).sparse()
  SparseTensor(features, coordinates=..., device=...)  / .F .C .slice(field) .decomposed_coordinates
  CoordinateManager: per-frame coordinate maps (stride 1,2,4,...), hash tables, kernel maps and conv plans,
  shared by every tensor derived from one input (what ME calls the coordinate manager).

Reference call sites: app/inference_engine.py:405-417,446-454,540-551; test_segmentation.py:62-72;
train_segmentation.py:78; data/alivev2.py:363.  Everything numeric is a libsvhip.so call (include/sv_hip.h).
"""
import threading
import weakref
from ctypes import c_float, c_int, c_int32, c_int64, c_size_t, c_void_p
from enum import Enum

import torch

from . import _lib
from ._lib import SV_TILE_ROWS, call, ptr, stream_ptr


class SparseTensorQuantizationMode(Enum):
    RANDOM_SUBSAMPLE = 0
    UNWEIGHTED_AVERAGE = 1
    UNWEIGHTED_SUM = 2
    NO_QUANTIZATION = 3


class MinkowskiAlgorithm(Enum):
    DEFAULT = 0
    MEMORY_EFFICIENT = 1
    SPEED_OPTIMIZED = 2


def _round_up(x, m):
    return (x + m - 1) // m * m


def _next_pow2(x):
    p = 1
    while p < x:
        p <<= 1
    return p


class CoordinateMap:
    """Canonical (key-sorted) voxel set at one tensor stride."""

    __slots__ = ("keys", "coords", "V", "stride", "_hash")

    def __init__(self, keys, coords, V, stride):
        self.keys = keys  # int64 tensor holding the uint64 keys, [V]
        self.coords = coords  # int32 [V,4] (batch,x,y,z)
        self.V = V
        self.stride = stride
        self._hash = None

    def hash(self):
        if self._hash is None:
            cap = _next_pow2(max(2 * self.V, 2))
            tkeys = torch.empty(cap, dtype=torch.int64, device=self.keys.device)
            tvals = torch.empty(cap, dtype=torch.int32, device=self.keys.device)
            call("sv_hash_build", ptr(self.keys), c_int64(self.V), ptr(tkeys), ptr(tvals), c_int64(cap), stream_ptr())
            self._hash = (tkeys, tvals, cap)
        return self._hash


BUF_LIMIT = 0x7fff0000 - 4096  # extent (bytes) the buffer-addressed conv instances take (csrc/sv_conv.hip BUF_LIMIT)


class ConvPlan:
    """Mask-sorted execution plan of one kernel map (include/sv_hip.h sv_plan_build)."""

    __slots__ = ("perm", "nbr_s", "submask", "tile_order", "V_out", "Vpad", "K", "pairs", "cm", "in_stride",
                 "out_stride", "raw", "_chunked")

    def __init__(self, perm, nbr_s, submask, tile_order, V_out, Vpad, K):
        self.perm, self.nbr_s, self.submask, self.tile_order = perm, nbr_s, submask, tile_order
        self.V_out, self.Vpad, self.K = V_out, Vpad, K
        self.pairs = None  # number of (in,out) pairs, filled lazily for roofline accounting
        # set by the coordinate manager for plans of a whole kernel map: what chunks() needs to re-plan a batch range
        self.cm = None
        self.in_stride = self.out_stride = None
        self.raw = None  # (nbr int32[K, ld], ld, mask) - the unsorted kernel map
        self._chunked = {}

    def pairs_device(self):
        """Number of (in, out) pairs of the kernel map as a device scalar (no host sync)."""
        if self.pairs is None:
            self.pairs = (self.nbr_s >= 0).sum()
        return self.pairs

    def num_pairs(self):
        return int(self.pairs_device().item())

    def chunks(self, in_row_bytes, out_row_bytes):
        """Batched tensors whose feature tables exceed the 2 GB extent of the buffer-addressed conv instances (64 Cfg-2
        frames x 416 channels = 9 GB): rows are sorted by batch and the frames of a batch share no neighbours
        (data/alivev2.py:358-383), so the layer splits into batch ranges, each with its own plan, input rows rebased to
        the range's first row - the same (offset, channel) chain per output element, i.e. the same bits, on the FAST
        instances.  Returns [(plan, in0, in1, out0, out1)] or None when the whole map fits (or cannot be split: one
        frame alone beyond the extent, which then takes the guarded 64-bit form)."""
        cm = self.cm() if self.cm is not None else None  # weak reference: a plan must not keep its manager (and through it
        if cm is None or self.raw is None:               # every tensor of the frame) alive in a reference cycle
            return None
        V_in = cm.stride_map(self.in_stride).V
        if V_in * in_row_bytes < BUF_LIMIT and self.V_out * out_row_bytes < BUF_LIMIT:
            return None
        bi, bo = cm.batch_bounds(self.in_stride), cm.batch_bounds(self.out_stride)
        B = len(bi) - 1
        f = 1
        while f < B:
            f *= 2
        fits = lambda f: all((bi[min(c + f, B)] - bi[c]) * in_row_bytes < BUF_LIMIT and  # noqa: E731
                             (bo[min(c + f, B)] - bo[c]) * out_row_bytes < BUF_LIMIT for c in range(0, B, f))
        while f >= 1 and not fits(f):
            f //= 2
        if f < 1 or f >= B:
            return None
        if f not in self._chunked:
            nbr, ld, mask = self.raw
            parts = []
            for c in range(0, B, f):
                i0, i1, o0, o1 = bi[c], bi[min(c + f, B)], bo[c], bo[min(c + f, B)]
                if o1 > o0:
                    parts.append((cm._build_plan(nbr[:, o0:], ld, mask[o0:], self.K, o1 - o0, nbr_base=i0), i0, i1, o0, o1))
            self._chunked[f] = parts
        return self._chunked[f]


class SplitPlan:
    """A kernel map run as PASSES over ascending offset ranges, each range with its own plan - its own mask-sorted row
    order.  Rows that share their neighbours among 13-14 offsets group far better into 16-row matrix-op sub-tiles than rows
    that must agree on all 27: useful row slots 0.872 -> 0.960 on the 2 cm room level (0.936 -> 0.969 one level up; CPU
    count, tools/tile_experiment.py).  The passes hand the raw accumulators over through memory (sv_conv_fwd_acc), so every
    output element stays ONE fma chain over (offset ascending, channel ascending).  parts: [(k0, k1, ConvPlan)]."""

    __slots__ = ("parts", "whole")

    def __init__(self, parts, whole):
        self.parts, self.whole = parts, whole

    # what callers read off a plan
    @property
    def V_out(self):
        return self.whole.V_out

    @property
    def Vpad(self):
        return self.whole.Vpad

    @property
    def out_stride(self):
        return self.whole.out_stride

    def pairs_device(self):
        return self.whole.pairs_device()

    def num_pairs(self):
        return self.whole.num_pairs()


class CoordinateManager:
    def __init__(self, device):
        self.device = device
        self.maps = {}  # stride -> CoordinateMap
        self.parents = {}  # fine stride -> (parent int32[V_fine], child_start int32[V_coarse+1])
        self.plans = {}
        self._batch_offsets = {}
        self._batch_bounds = {}
        self.num_batches = None
        self.phase_hook = None  # set per frame by app/pipeline.py FramePipeline.run (see model/backbone/minkunet.py)
        self.split_rules = None  # [(min_rows, cuts)] of this frame's wide 3x3x3 layers; None = nn.SPLIT_RULES
        self.split_ready = None  # event: the offset-range plans, built on another stream while the encoder runs, are there
        self._arenas = []  # the frame composites' arenas (every map / plan array built by them is a view of one)

    def _own(self, plan, nbr, ld, mask, in_stride, out_stride):
        plan.cm, plan.raw, plan.in_stride, plan.out_stride = weakref.ref(self), (nbr, ld, mask), in_stride, out_stride
        return plan

    # ---- coordinate maps -------------------------------------------------------------------
    def stride_map(self, stride):
        """Map at tensor stride `stride` (power of two); created from the next finer one on demand."""
        if stride in self.maps:
            return self.maps[stride]
        if stride <= 1 or (stride & (stride - 1)):
            raise _lib.SvHipError(f"no coordinate map at tensor stride {stride}")
        fine = self.stride_map(stride // 2)
        level = (stride // 2).bit_length() - 1
        dev = self.device
        V_in = fine.V
        ws_bytes = _lib.load().sv_stride_map_workspace_bytes(c_int64(V_in))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        keys = torch.empty(max(V_in, 1), dtype=torch.int64, device=dev)
        coords = torch.empty((max(V_in, 1), 4), dtype=torch.int32, device=dev)
        parent = torch.empty(max(V_in, 1), dtype=torch.int32, device=dev)
        child_start = torch.empty(V_in + 1, dtype=torch.int32, device=dev)
        counters = torch.empty(4, dtype=torch.int32, device=dev)
        call("sv_stride_map", ptr(fine.keys), c_int64(V_in), c_int(level), ptr(ws), c_size_t(ws_bytes), ptr(keys),
             ptr(coords), ptr(parent), ptr(child_start), ptr(counters), stream_ptr())
        V = int(counters[0].item())
        m = CoordinateMap(keys[:V], coords[:V], V, stride)
        self.maps[stride] = m
        self.parents[stride // 2] = (parent[:V_in], child_start[: V + 1])
        return m

    # ---- plans -----------------------------------------------------------------------------
    def batch_bounds(self, stride):
        """host copy of the first row of every batch at tensor stride `stride` ([B + 1] python ints; one read-back per
        level, cached) - the ranges batched launches are split at (ConvPlan.chunks)"""
        if stride not in self._batch_bounds:
            B = self.num_batches
            if B is None:
                m1 = self.stride_map(1)
                B = int(m1.coords[:, 0].max().item()) + 1 if m1.V else 1
                self.num_batches = B
            self._batch_bounds[stride] = self.batch_offsets(stride, B).tolist()
        return self._batch_bounds[stride]

    def _build_plan(self, nbr, ld, mask, K, V_out, nbr_base=0):
        dev = self.device
        Vpad = _round_up(max(V_out, 1), SV_TILE_ROWS)
        ws_bytes = _lib.load().sv_plan_workspace_bytes(c_int64(V_out))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        perm = torch.empty(Vpad, dtype=torch.int32, device=dev)
        nbr_s = torch.empty((K, Vpad), dtype=torch.int32, device=dev)
        submask = torch.empty((Vpad // SV_TILE_ROWS, K), dtype=torch.int32, device=dev)
        tile_order = torch.empty(Vpad // SV_TILE_ROWS, dtype=torch.int32, device=dev)
        call("sv_plan_build", ptr(nbr), c_int64(ld), ptr(mask), c_int(K), c_int64(V_out), c_int64(nbr_base), ptr(ws),
             c_size_t(ws_bytes),
             ptr(perm), ptr(nbr_s), ptr(submask), ptr(tile_order), c_int64(Vpad), stream_ptr())
        return ConvPlan(perm, nbr_s, submask, tile_order, V_out, Vpad, K)

    def plan_k3(self, stride, dilation=1):
        key = ("k3", stride, dilation)
        if key not in self.plans:
            m = self.stride_map(stride)
            tkeys, tvals, cap = m.hash()
            V = m.V
            nbr = torch.empty((27, max(V, 1)), dtype=torch.int32, device=self.device)
            mask = torch.empty(max(V, 1), dtype=torch.int32, device=self.device)
            call("sv_kernel_map_k3", ptr(m.coords), c_int64(V), c_int(stride), c_int(dilation), ptr(tkeys), ptr(tvals),
                 c_int64(cap), ptr(nbr), c_int64(max(V, 1)), ptr(mask), stream_ptr())
            self.plans[key] = self._own(self._build_plan(nbr, max(V, 1), mask, 27, V), nbr, max(V, 1), mask, stride, stride)
        return self.plans[key]

    def plan_k3_split(self, stride, split=14):
        """The 3x3x3 map of `stride` as passes over ascending offset ranges (SplitPlan): split = 14 -> [0, 14) and [14, 27)
        (the centre offset in the first pass; the best two-pass split on the room levels, 13 equal); a tuple of split
        points gives more passes."""
        cuts = (split,) if isinstance(split, int) else tuple(split)
        key = ("k3split", stride, split)
        if key not in self.plans:
            whole = self.plan_k3(stride)
            nbr, ld, mask = whole.raw
            parts = []
            bounds = (0,) + cuts + (27,)
            for k0, k1 in zip(bounds[:-1], bounds[1:]):
                sub = ((mask >> k0) & ((1 << (k1 - k0)) - 1)).to(torch.int32)
                plan = self._build_plan(nbr[k0:k1], ld, sub, k1 - k0, whole.V_out)
                parts.append((k0, k1, self._own(plan, nbr[k0:k1], ld, sub, stride, stride)))
            self.plans[key] = SplitPlan(parts, whole)
        return self.plans[key]

    def plan_down(self, stride):
        """kernel_size 2, stride 2 convolution from tensor stride `stride` to 2*stride."""
        key = ("down", stride)
        if key not in self.plans:
            fine = self.stride_map(stride)
            coarse = self.stride_map(stride * 2)
            parent, _ = self.parents[stride]
            level = stride.bit_length() - 1
            Vc = coarse.V
            nbr = torch.empty((8, max(Vc, 1)), dtype=torch.int32, device=self.device)
            mask = torch.empty(max(Vc, 1), dtype=torch.int32, device=self.device)
            call("sv_kernel_map_down", ptr(fine.keys), ptr(parent), c_int64(fine.V), c_int(level), c_int64(Vc),
                 ptr(nbr), c_int64(max(Vc, 1)), ptr(mask), stream_ptr())
            self.plans[key] = self._own(self._build_plan(nbr, max(Vc, 1), mask, 8, Vc), nbr, max(Vc, 1), mask, stride,
                                        stride * 2)
        return self.plans[key]

    def plan_up(self, stride):
        """transposed kernel_size 2, stride 2 convolution from tensor stride `stride` to stride/2 (existing map)."""
        key = ("up", stride)
        if key not in self.plans:
            fs = stride // 2
            if fs not in self.maps or fs not in self.parents:
                raise _lib.SvHipError(
                    f"transposed conv to tensor stride {fs}: that coordinate map does not exist "
                    "(ME semantics: the output reuses the encoder's map, model/backbone/minkunet.py:152-156)")
            fine = self.maps[fs]
            parent, _ = self.parents[fs]
            level = fs.bit_length() - 1
            V = fine.V
            nbr = torch.empty((8, max(V, 1)), dtype=torch.int32, device=self.device)
            mask = torch.empty(max(V, 1), dtype=torch.int32, device=self.device)
            call("sv_kernel_map_up", ptr(fine.keys), ptr(parent), c_int64(V), c_int(level), ptr(nbr),
                 c_int64(max(V, 1)), ptr(mask), stream_ptr())
            self.plans[key] = self._own(self._build_plan(nbr, max(V, 1), mask, 8, V), nbr, max(V, 1), mask, stride, fs)
        return self.plans[key]

    # ---- frame composites (sv_frame_plans): everything a U-Net asks for in one or two host calls ------------
    def split_cuts_for(self, rows):
        """split points of the wide 3x3x3 layers on a map of `rows` voxels (None = one pass)"""
        if self.split_rules is None:
            from . import nn as svnn

            return svnn.split_points_for(rows)
        for min_rows, cuts in self.split_rules:
            if rows >= min_rows:
                return cuts
        return None

    def build_plans(self, levels, k3=True, down=True, up=True, split=True):
        """Hash tables, kernel maps and conv plans of pyramid levels 0..levels through ONE sv_frame_plans call (what
        plan_k3 / plan_down / plan_up / plan_k3_split build piece by piece; the same arrays).  The maps of every level
        must exist (TensorField.sparse(pyramid_levels=...) or stride_map)."""
        lib = _lib.load()
        L = levels
        ms = [self.stride_map(1 << l) for l in range(L + 1)]
        V = (c_int64 * (L + 1))(*[m.V for m in ms])
        flags = ((_lib.SV_FRAME_K3 if k3 else 0) | (_lib.SV_FRAME_DOWN if down else 0) | (_lib.SV_FRAME_UP if up else 0))
        cuts_of = {}
        cuts_arr = (c_int32 * ((L + 1) * _lib.SV_FRAME_MAX_CUTS))()
        if split:
            for l in range(L + 1):
                cuts = self.split_cuts_for(ms[l].V)
                if cuts is None or ("k3split", 1 << l, cuts) in self.plans:
                    continue
                tup = (cuts,) if isinstance(cuts, int) else tuple(cuts)
                if len(tup) > _lib.SV_FRAME_MAX_CUTS:
                    continue  # more passes than the composite takes: plan_k3_split builds them piecewise on demand
                cuts_of[l] = cuts
                for i, c in enumerate(tup):
                    cuts_arr[l * _lib.SV_FRAME_MAX_CUTS + i] = c
            if cuts_of:
                flags |= _lib.SV_FRAME_SPLIT
        if flags == 0:
            return
        pv = lambda ts: (c_void_p * (L + 1))(*[(t.data_ptr() if t is not None else None) for t in ts])  # noqa: E731
        keys = pv([m.keys for m in ms])
        coords = pv([m.coords for m in ms])
        parent = pv([self.parents[1 << l][0] if (1 << l) in self.parents else None for l in range(L + 1)])
        k3n = k3m = None
        if (flags & _lib.SV_FRAME_SPLIT) and not k3:
            whole = [self.plan_k3(1 << l) if l in cuts_of else None for l in range(L + 1)]
            k3n = pv([w.raw[0] if w is not None else None for w in whole])
            k3m = pv([w.raw[2] if w is not None else None for w in whole])
        nbytes = lib.sv_frame_plans_arena_bytes(V, c_int(L), c_int(flags), cuts_arr)
        arena = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        sbytes = lib.sv_frame_plans_scratch_bytes(V, c_int(L))
        scratch = torch.empty(sbytes, dtype=torch.uint8, device=self.device)
        max_rec = 8 * (L + 1) + 8
        layout = (c_int64 * (_lib.SV_FRAME_RECORD * (1 + max_rec)))()
        call("sv_frame_plans", keys, coords, parent, V, c_int(L), c_int(flags), cuts_arr, k3n, k3m, ptr(arena), c_size_t(nbytes),
             ptr(scratch), c_size_t(sbytes), layout, c_int(max_rec), stream_ptr())
        self._arenas.append(arena)
        i32 = lambda off, n: arena[off: off + 4 * n].view(torch.int32)  # noqa: E731
        pending_split = {}
        for r in range(layout[1]):
            rec = layout[_lib.SV_FRAME_RECORD * (1 + r): _lib.SV_FRAME_RECORD * (2 + r)]
            kind, l, k0, k1, o_nbr, o_mask, o_perm, o_nbrs, o_sub, o_tile, V_out, Vpad, K, ld, o_tk, cap = rec
            stride = 1 << l
            if kind == _lib.SV_FRAME_REC_HASH:
                ms[l]._hash = (arena[o_tk: o_tk + 8 * cap].view(torch.int64), i32(o_nbr, cap), cap)
                continue
            plan = ConvPlan(i32(o_perm, Vpad), i32(o_nbrs, K * Vpad).view(K, Vpad), i32(o_sub, (Vpad // SV_TILE_ROWS) * K).view(-1, K),
                            i32(o_tile, Vpad // SV_TILE_ROWS), V_out, Vpad, K)
            mask = i32(o_mask, ld)
            if kind == _lib.SV_FRAME_REC_SPLIT:
                whole = self.plans[("k3", stride, 1)]
                nbr = i32(o_nbr, K * ld).view(K, ld) if o_nbr >= 0 else whole.raw[0][k0:k1]
                pending_split.setdefault(l, []).append((k0, k1, self._own(plan, nbr, ld, mask, stride, stride)))
                continue
            nbr = i32(o_nbr, K * ld).view(K, ld)
            if kind == _lib.SV_FRAME_REC_K3:
                self.plans[("k3", stride, 1)] = self._own(plan, nbr, ld, mask, stride, stride)
            elif kind == _lib.SV_FRAME_REC_DOWN:
                self.plans[("down", stride)] = self._own(plan, nbr, ld, mask, stride, stride * 2)
            elif kind == _lib.SV_FRAME_REC_UP:
                self.plans[("up", stride * 2)] = self._own(plan, nbr, ld, mask, stride * 2, stride)
        for l, parts in pending_split.items():
            self.plans[("k3split", 1 << l, cuts_of[l])] = SplitPlan(parts, self.plans[("k3", 1 << l, 1)])

    def batch_offsets(self, stride, B):
        key = (stride, B)
        if key not in self._batch_offsets:
            m = self.stride_map(stride)
            bs = torch.empty(B + 1, dtype=torch.int32, device=self.device)
            call("sv_batch_offsets", ptr(m.keys), c_int64(m.V), c_int(B), ptr(bs), stream_ptr())
            self._batch_offsets[key] = bs
        return self._batch_offsets[key]


def _as_device(device):
    if device is None:
        device = "cuda"
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.SvHipError(
            f"device {device}: the MI355X-native path runs on the GPU only (no CPU fallback by design)")
    return device


def _voxelize(coords, device, coords_are_int):
    """Run sv_voxelize; returns (CoordinateMap at stride 1, inverse int64[N], order int32[N], seg_start int32[V+1])."""
    N = coords.shape[0]
    lib = _lib.load()
    ws_bytes = lib.sv_voxelize_workspace_bytes(c_int64(N))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    n1 = max(N, 1)
    keys = torch.empty(n1, dtype=torch.int64, device=device)
    vcoords = torch.empty((n1, 4), dtype=torch.int32, device=device)
    inverse = torch.empty(n1, dtype=torch.int64, device=device)
    order = torch.empty(n1, dtype=torch.int32, device=device)
    seg_start = torch.empty(n1 + 1, dtype=torch.int32, device=device)
    counters = torch.empty(4, dtype=torch.int32, device=device)
    call("sv_voxelize", ptr(coords), c_int(1 if coords_are_int else 0), c_int64(N), ptr(ws), c_size_t(ws_bytes),
         ptr(keys), ptr(vcoords), ptr(inverse), ptr(order), ptr(seg_start), ptr(counters), stream_ptr())
    cnt = counters.tolist()
    V, bad = cnt[0], cnt[1]
    if bad:
        raise _lib.SvHipError(
            f"{bad} points have coordinates outside the key range (|coord| < 2^17 voxels, 0 <= batch < 1024)")
    return CoordinateMap(keys[:V], vcoords[:V], V, 1), inverse[:N], order[:N], seg_start[: V + 1]


_tls = threading.local()


def _pinned_counters(n):
    """per-thread pinned int32 buffer the frame composite reads its level sizes back through (the call synchronises on
    its own kernels before it returns, so one buffer per host thread is enough)"""
    buf = getattr(_tls, "counters", None)
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n, 64), dtype=torch.int32).pin_memory()
        _tls.counters = buf
    return buf


def _voxelize_frame(coords, device, coords_are_int, levels):
    """sv_frame_maps: voxelisation and the `levels` stride-2 maps of a frame in one host call.  Returns (CoordinateManager
    with maps 1 .. 2^levels and their parent tables, inverse, order, seg_start) - the arrays sv_voxelize + sv_stride_map
    produce piece by piece."""
    N = coords.shape[0]
    lib = _lib.load()
    abytes = lib.sv_frame_maps_arena_bytes(c_int64(N), c_int(levels))
    sbytes = lib.sv_frame_maps_scratch_bytes(c_int64(N))
    arena = torch.empty(abytes, dtype=torch.uint8, device=device)
    scratch = torch.empty(sbytes, dtype=torch.uint8, device=device)
    counters = _pinned_counters(4 * (levels + 2))
    layout = (c_int64 * (8 + 6 * (levels + 1)))()
    call("sv_frame_maps", ptr(coords), c_int(1 if coords_are_int else 0), c_int64(N), c_int(levels), ptr(arena), c_size_t(abytes),
         ptr(scratch), c_size_t(sbytes), c_void_p(counters.data_ptr()), layout, stream_ptr())
    cm = CoordinateManager(device)
    cm._arenas.append(arena)
    view = lambda off, nbytes, dt: arena[off: off + nbytes].view(dt)  # noqa: E731
    Vs = [layout[8 + 6 * l] for l in range(levels + 1)]
    for l in range(levels + 1):
        V = Vs[l]
        o_keys, o_coords, o_parent, o_child = layout[9 + 6 * l: 13 + 6 * l]
        cm.maps[1 << l] = CoordinateMap(view(o_keys, 8 * V, torch.int64), view(o_coords, 16 * V, torch.int32).view(V, 4), V, 1 << l)
        if l < levels:
            cm.parents[1 << l] = (view(o_parent, 4 * V, torch.int32), view(o_child, 4 * (Vs[l + 1] + 1), torch.int32))
    V0 = Vs[0]
    return (cm, view(layout[3], 8 * N, torch.int64), view(layout[4], 4 * N, torch.int32), view(layout[5], 4 * (V0 + 1), torch.int32))


def _voxel_reduce(feats, order, seg_start, V, mode):
    C = feats.shape[1]
    out = torch.empty((V, C), dtype=torch.float32, device=feats.device)
    call("sv_voxel_reduce", ptr(feats), c_int(C), ptr(order), ptr(seg_start), c_int64(V), c_int(mode), ptr(out),
         stream_ptr())
    return out


class TensorField:
    """ME.TensorField: per-point features + continuous (pre-scaled) coordinates [N, 1+3] = (batch, x, y, z)."""

    def __init__(self, features, coordinates=None, quantization_mode=SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE,
                 minkowski_algorithm=MinkowskiAlgorithm.DEFAULT, device=None, coordinate_manager=None,
                 inverse_mapping=None):
        device = _as_device(device if device is not None else (features.device if features.is_cuda else None))
        self.device = device
        self._F = features.to(device=device, dtype=torch.float32).contiguous()
        self.quantization_mode = quantization_mode
        self.minkowski_algorithm = minkowski_algorithm
        self.coordinate_manager = coordinate_manager
        self.inverse_mapping = inverse_mapping
        if coordinates is not None:
            if coordinates.shape[0] != features.shape[0] or coordinates.shape[1] != 4:
                raise ValueError("coordinates must be [N, 4] = (batch, x, y, z) with one row per feature row")
            self._C = coordinates.to(device=device, dtype=torch.float32).contiguous()
        else:
            self._C = None
        self._order = None
        self._seg_start = None

    @property
    def F(self):
        return self._F

    features = F

    @property
    def C(self):
        return self._C

    coordinates = C

    def __len__(self):
        return self._F.shape[0]

    def sparse(self, pyramid_levels=None):
        """Voxelise: floor the coordinates, unique voxels in canonical order, per-voxel feature mean.
        pyramid_levels = n (this build's extension): the stride-2 maps of the next n pyramid levels are built in the same
        host call (sv_frame_maps) - what a U-Net's strided convolutions would otherwise ask for one by one, each with
        its own size read-back."""
        if self._C is None:
            raise ValueError("TensorField has no coordinates")
        if self.quantization_mode not in (SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE,
                                          SparseTensorQuantizationMode.RANDOM_SUBSAMPLE):
            raise NotImplementedError(f"quantization_mode {self.quantization_mode}")
        if pyramid_levels is not None and self._C.shape[0] > 0:
            cm, inverse, order, seg_start = _voxelize_frame(self._C, self.device, False, pyramid_levels)
            cmap = cm.maps[1]
        else:
            cmap, inverse, order, seg_start = _voxelize(self._C, self.device, coords_are_int=False)
            cm = CoordinateManager(self.device)
            cm.maps[1] = cmap
        self.coordinate_manager = cm
        self.inverse_mapping = inverse
        self._order, self._seg_start = order, seg_start
        mode = (_lib.SV_REDUCE_MEAN if self.quantization_mode == SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE
                else _lib.SV_REDUCE_FIRST)
        feats = _voxel_reduce(self._F, order, seg_start, cmap.V, mode)
        return SparseTensor(feats, coordinate_manager=cm, tensor_stride=1, _internal=True)


class SparseTensor:
    """ME.SparseTensor: features [V, C] on a coordinate map of a CoordinateManager."""

    def __init__(self, features, coordinates=None, tensor_stride=1, coordinate_manager=None, device=None,
                 quantization_mode=SparseTensorQuantizationMode.RANDOM_SUBSAMPLE, requires_grad=False,
                 _internal=False):
        if _internal:
            self._F = features
            self.coordinate_manager = coordinate_manager
            self.tensor_stride = tensor_stride
            self.device = features.device
            self.inverse_mapping = None
            return
        device = _as_device(device if device is not None else (features.device if features.is_cuda else None))
        self.device = device
        if coordinates is None:
            if coordinate_manager is None:
                raise ValueError("either coordinates or a coordinate_manager is required")
            self._F = features.to(device=device, dtype=torch.float32).contiguous()
            self.coordinate_manager = coordinate_manager
            self.tensor_stride = tensor_stride
            self.inverse_mapping = None
            return
        # ME.SparseTensor(feats, coordinates=int coords [N,4]) — train_segmentation.py:78: already-quantised rows
        coords = coordinates.to(device=device, dtype=torch.int32).contiguous()
        feats = features.to(device=device, dtype=torch.float32).contiguous()
        if coords.shape[0] != feats.shape[0] or coords.shape[1] != 4:
            raise ValueError("coordinates must be [N, 4] = (batch, x, y, z) with one row per feature row")
        cmap, inverse, order, seg_start = _voxelize(coords, device, coords_are_int=True)
        cm = CoordinateManager(device)
        cm.maps[1] = cmap
        mode = (_lib.SV_REDUCE_MEAN if quantization_mode == SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE
                else _lib.SV_REDUCE_FIRST)
        self._F = _voxel_reduce(feats, order, seg_start, cmap.V, mode)
        self.coordinate_manager = cm
        self.tensor_stride = 1
        self.inverse_mapping = inverse

    # ---- ME accessors ----------------------------------------------------------------------
    @property
    def F(self):
        return self._F

    features = F

    @property
    def coordinate_map(self):
        return self.coordinate_manager.stride_map(self.tensor_stride)

    @property
    def C(self):
        return self.coordinate_map.coords

    coordinates = C

    @property
    def shape(self):
        return self._F.shape

    def __len__(self):
        return self._F.shape[0]

    @property
    def decomposed_coordinates(self):
        m = self.coordinate_map
        B = int(m.coords[:, 0].max().item()) + 1 if m.V else 0
        bs = self.coordinate_manager.batch_offsets(self.tensor_stride, max(B, 1)).tolist()
        return [m.coords[bs[b]: bs[b + 1], 1:] for b in range(B)]

    @property
    def decomposed_features(self):
        m = self.coordinate_map
        B = int(m.coords[:, 0].max().item()) + 1 if m.V else 0
        bs = self.coordinate_manager.batch_offsets(self.tensor_stride, max(B, 1)).tolist()
        return [self._F[bs[b]: bs[b + 1]] for b in range(B)]

    def new(self, features, tensor_stride=None):
        return SparseTensor(features, coordinate_manager=self.coordinate_manager,
                            tensor_stride=self.tensor_stride if tensor_stride is None else tensor_stride,
                            _internal=True)

    def _same_map(self, other):
        if not isinstance(other, SparseTensor):
            return NotImplemented
        if other.coordinate_manager is not self.coordinate_manager or other.tensor_stride != self.tensor_stride:
            raise ValueError("sparse tensor arithmetic needs both operands on the same coordinate map")
        return other

    def __add__(self, other):
        """ME.SparseTensor `+` on a shared coordinate map (the residual add of ME's resnet blocks: `out += residual`)."""
        o = self._same_map(other)
        return o if o is NotImplemented else self.new(self._F + o._F)

    __iadd__ = __add__

    def __sub__(self, other):
        o = self._same_map(other)
        return o if o is NotImplemented else self.new(self._F - o._F)

    def slice(self, field):
        """Voxel -> point broadcast: TensorField whose rows are this tensor's rows at field.inverse_mapping."""
        if self.tensor_stride != 1:
            raise ValueError("slice needs a tensor at tensor stride 1")
        inv = field.inverse_mapping
        if inv is None or field.coordinate_manager is not self.coordinate_manager:
            raise ValueError("the field was not voxelised into this tensor's coordinate manager")
        N, C = inv.shape[0], self._F.shape[1]
        out = torch.empty((N, C), dtype=torch.float32, device=self._F.device)
        call("sv_slice_rows", ptr(self._F), c_int64(self._F.stride(0)), c_int(C), ptr(inv), c_int64(N), ptr(out),
             stream_ptr())
        return TensorField(out, coordinates=None, device=self._F.device, coordinate_manager=self.coordinate_manager,
                           inverse_mapping=inv)

    def slice_argmax(self, field, with_conf=True):
        """Fused slice + utils/output.py:67-73: per-point label (first row maximum) and sigmoid(max)."""
        inv = field.inverse_mapping
        if inv is None or field.coordinate_manager is not self.coordinate_manager:
            raise ValueError("the field was not voxelised into this tensor's coordinate manager")
        N, C = inv.shape[0], self._F.shape[1]
        label = torch.empty(N, dtype=torch.int64, device=self._F.device)
        conf = torch.empty(N, dtype=torch.float32, device=self._F.device) if with_conf else None
        call("sv_slice_argmax", ptr(self._F), c_int64(self._F.stride(0)), c_int(C), ptr(inv), c_int64(N), ptr(label),
             ptr(conf), stream_ptr())
        return label, conf

    # dense [B, C] results of global pooling are plain tensors in this build; `out[0][3:]` works on them directly


def cat(*tensors):
    """ME.cat: channel concatenation of tensors on the same coordinate map (model/backbone/minkunet.py:156)."""
    if len(tensors) == 1 and isinstance(tensors[0], (list, tuple)):
        tensors = tuple(tensors[0])
    first = tensors[0]
    for t in tensors[1:]:
        if t.coordinate_manager is not first.coordinate_manager or t.tensor_stride != first.tensor_stride:
            raise ValueError("ME.cat needs tensors on the same coordinate map")
    return first.new(torch.cat([t.F for t in tensors], dim=1))
