"""nn.Module layers with MinkowskiEngine's names, parameters and state_dict layout, running on
libsvhip.so.  Inference only: the forward pass needs eval() BatchNorm (running statistics), there is no backward.

state_dict keys match ME 0.5.4 so utils/utils.py:87-126 checkpoint_restore-style loading works unchanged:
  MinkowskiConvolution(.Transpose): `kernel` [K, Cin, Cout] ([Cin, Cout] when K == 1), `bias` [1, Cout]
  MinkowskiBatchNorm: `bn.weight bn.bias bn.running_mean bn.running_var bn.num_batches_tracked`
  MinkowskiLinear: `linear.weight` [Cout, Cin], `linear.bias` [Cout]
Kernel-offset order inside `kernel` is this build's definition (include/sv_hip.h); `KERNEL_OFFSET_PERMUTATION` is the
single hook for adapting a real ME checkpoint if its order turns out to differ.
"""
import math
from ctypes import c_float, c_int, c_int64

import numpy as np
import torch
import torch.nn as nn

from . import _lib, profiling
from ._lib import SV_ACT_LEAKY_RELU, SV_ACT_NONE, SV_ACT_RELU, call, ptr, stream_ptr
from .sparse import SparseTensor, SplitPlan

# Adapting a checkpoint written by a MinkowskiEngine build whose kernel-offset numbering differs from this build's
# (include/sv_hip.h: k = (dx+1) + 3 (dy+1) + 9 (dz+1), x fastest; k = dx + 2 dy + 4 dz for kernel_size 2):
# {kernel_volume: perm} with  this_build_kernel[k] = checkpoint_kernel[perm[k]].  Applied to every `kernel` of that
# volume inside load_state_dict (_ConvBase._load_from_state_dict); None = checkpoints already use this numbering.
KERNEL_OFFSET_PERMUTATION = None

# Wide 3x3x3 layers on big pyramid levels run as PASSES over ascending ranges of the kernel offsets (sparse.SplitPlan), each
# pass with its own row order: worth it where the matrix-op time saved by the better row grouping (~9 % of the slots at 88k
# voxels with two passes, ~12 % with three) exceeds the extra launches' fixed costs and the accumulator hand-over
# (measured inside the frame pipeline: tools/ab_split.sh).  MRCC_SPLIT_RULES = "min_rows:cut[,cut...];..." -
# the first rule whose min_rows the output map reaches applies; "" = never split.
import os as _os  # noqa: E402


def _parse_split_rules(text):
    rules = []
    for part in text.split(";"):
        if part.strip():
            rows, _, cuts = part.partition(":")
            cuts = tuple(int(v) for v in cuts.split(",") if v)
            rules.append((int(rows), cuts[0] if len(cuts) == 1 else cuts))
    return sorted(rules, key=lambda r: -r[0])


SPLIT_RULES = _parse_split_rules(_os.environ.get("MRCC_SPLIT_RULES", "20000:9,18"))
SPLIT_MIN_CHANNELS = int(_os.environ.get("MRCC_SPLIT_MIN_CHANNELS", "128"))


# one frame alone on the GPU (the per-frame InferenceEngine.predict call): level 1's launches are one round of workgroups
# already and LOSE with three passes when nothing else fills their tails (100.7 -> 95.5 TFLOP/s)
SPLIT_RULES_ONE_FRAME = _parse_split_rules(_os.environ.get("MRCC_SPLIT_RULES_ONE_FRAME", "50000:9,18"))


def split_points_for(rows):
    """split points of the 3x3x3 layers on an output map of `rows` voxels (None = one pass)"""
    for min_rows, cuts in SPLIT_RULES:
        if rows >= min_rows:
            return cuts
    return None


# ------------------------------------------------------------------------------------------------------------------
# functional layer: one libsvhip call per op
# ------------------------------------------------------------------------------------------------------------------
def conv_forward(feats, weight3, plan, V_out, scale=None, shift=None, residual=None, act=SV_ACT_NONE, slope=0.01,
                 out=None):
    """out[o] = act(BN(sum_k in[nbr_k(o)] @ W[k]) + residual[o]); plan None = dense rows (kernel_size 1 / Linear)."""
    K, Cin, Cout = weight3.shape
    if feats.shape[1] != Cin:
        raise ValueError(f"input has {feats.shape[1]} channels, kernel expects {Cin}")
    if feats.stride(1) != 1:
        feats = feats.contiguous()
    if out is None:
        out = torch.empty((V_out, Cout), dtype=torch.float32, device=feats.device)
    if isinstance(plan, SplitPlan):
        return _conv_forward_split(feats, weight3, plan, V_out, scale, shift, residual, act, slope, out)
    return _conv_forward_one(feats, weight3, plan, V_out, scale, shift, residual, act, slope, out, None)


def _conv_forward_split(feats, weight3, plan, V_out, scale, shift, residual, act, slope, out):
    """A layer as passes over ascending offset ranges (sparse.SplitPlan): every pass but the last writes the raw
    accumulators (no epilogue), the next one continues the chains from them (sv_conv_fwd_acc)."""
    K, Cin, Cout = weight3.shape
    timer = profiling.TIMER
    t0 = None
    if timer is not None:
        kname = profiling.conv_kernel_config(Cout, plan.Vpad, Cin, plan.parts[-1][1] - plan.parts[-1][0])
        if timer.want(kname):
            t0 = timer.start()
    if not weight3.is_contiguous():
        weight3 = weight3.contiguous()  # a pass takes the block W[k0:k1] by pointer
    acc = None
    for i, (k0, k1, sub) in enumerate(plan.parts):  # timed=False: the passes are ONE layer for the per-kernel table
        w = weight3[k0:k1]
        if i == len(plan.parts) - 1:
            _conv_forward_one(feats, w, sub, V_out, scale, shift, residual, act, slope, out, acc, timed=False)
        else:
            nxt = torch.empty((V_out, Cout), dtype=torch.float32, device=feats.device)
            _conv_forward_one(feats, w, sub, V_out, None, None, None, SV_ACT_NONE, slope, nxt, acc, timed=False)
            acc = nxt
    if t0 is not None:
        timer.stop(t0, _lib.conv_last_instance()[0], K, Cin, Cout, V_out, plan.pairs_device(), level=plan.out_stride,
                   passes=len(plan.parts))
    return out


def _conv_forward_one(feats, weight3, plan, V_out, scale, shift, residual, act, slope, out, acc_init, timed=True):
    K, Cin, Cout = weight3.shape
    if plan is None:
        Vpad = (max(V_out, 1) + _lib.SV_TILE_ROWS - 1) // _lib.SV_TILE_ROWS * _lib.SV_TILE_ROWS
    else:
        Vpad = plan.Vpad
    timer = profiling.TIMER if timed else None
    t0 = None
    if timer is not None:
        kname = profiling.conv_kernel_config(Cout, Vpad, Cin, K)
        if timer.want(kname):
            t0 = timer.start()
    res_ld = residual.stride(0) if residual is not None else 0

    log = profiling.INSTANCE_LOG

    acc_ld = acc_init.stride(0) if acc_init is not None else 0

    def launch(f, pl, o, r, v_out, v_pad, a=None):
        call("sv_conv_fwd_acc", ptr(f), c_int64(f.shape[0]), c_int64(f.stride(0)), c_int(Cin), ptr(weight3), c_int(K),
             c_int(Cout), ptr(pl.perm if pl else None), ptr(pl.nbr_s if pl else None), ptr(pl.submask if pl else None),
             ptr(pl.tile_order if pl else None), c_int64(v_out), c_int64(v_pad), ptr(a), c_int64(acc_ld), ptr(scale), ptr(shift),
             ptr(r), c_int64(res_ld), c_int(act), c_float(slope), ptr(o), c_int64(o.stride(0)), stream_ptr())
        if log is not None:  # what the library really launched (sv_conv_last_instance), not a re-derivation
            log.append((*_lib.conv_last_instance(), K, Cin, Cout, v_out))

    # batched tensors beyond the 2 GB extent of the buffer-addressed instances run as batch ranges (ConvPlan.chunks)
    parts = plan.chunks(4 * feats.stride(0), 4 * max(out.stride(0), res_ld, acc_ld)) if plan is not None else None
    if parts is None:
        launch(feats, plan, out, residual, V_out, Vpad, acc_init)
    else:
        for sub, i0, i1, o0, o1 in parts:
            launch(feats[i0:i1], sub, out[o0:o1], residual[o0:o1] if residual is not None else None, o1 - o0, sub.Vpad,
                   acc_init[o0:o1] if acc_init is not None else None)
    if t0 is not None:
        # recorded under the instance the library reports (the prediction above only decides whether to time at all)
        timer.stop(t0, _lib.conv_last_instance()[0], K, Cin, Cout, V_out,
                   plan.pairs_device() if plan is not None else None,
                   level=plan.out_stride if plan is not None else None)
    return out


def affine_act(feats, scale=None, shift=None, residual=None, act=SV_ACT_NONE, slope=0.01):
    V, C = feats.shape
    if feats.stride(1) != 1:
        feats = feats.contiguous()
    out = torch.empty((V, C), dtype=torch.float32, device=feats.device)
    call("sv_affine_act", ptr(feats), c_int64(feats.stride(0)), c_int(C), c_int64(V), ptr(scale), ptr(shift),
         ptr(residual), c_int64(residual.stride(0) if residual is not None else 0), c_int(act), c_float(slope),
         ptr(out), c_int64(C), stream_ptr())
    return out


def global_pool(x, mode):
    cm = x.coordinate_manager
    m = x.coordinate_map
    B = cm.num_batches
    if B is None:
        B = int(m.coords[:, 0].max().item()) + 1 if m.V else 1
        cm.num_batches = B
    bs = cm.batch_offsets(x.tensor_stride, B)
    F = x.F
    C = F.shape[1]
    out = torch.empty((B, C), dtype=torch.float32, device=F.device)
    call("sv_global_pool", ptr(F), c_int64(F.stride(0)), c_int(C), ptr(bs), c_int(B), c_int(mode), ptr(out),
         stream_ptr())
    return out


def fold_bn(bn):
    """BatchNorm1d(eval) as y = fmaf(x, scale, shift): scale = w / sqrt(var + eps), shift = b - mean * scale.
    Host-side, once per model, in numpy float32 (correctly rounded IEEE sqrt / divide; torch's vectorised CPU path
    is 1 ulp off), i.e. exactly the arithmetic oracle/sv_oracle.py:fold_bn defines."""
    n = bn.num_features
    w = bn.weight.detach().float().cpu().numpy() if bn.weight is not None else np.ones(n, np.float32)
    b = bn.bias.detach().float().cpu().numpy() if bn.bias is not None else np.zeros(n, np.float32)
    mean = bn.running_mean.detach().float().cpu().numpy()
    var = bn.running_var.detach().float().cpu().numpy()
    scale = (w / np.sqrt(var + np.float32(bn.eps))).astype(np.float32)
    shift = (b - mean * scale).astype(np.float32)
    dev = bn.running_mean.device
    return torch.from_numpy(scale).to(dev), torch.from_numpy(shift).to(dev)


def _tensor_versions(*ts):
    return tuple((t.data_ptr(), t._version) for t in ts if t is not None)


# ------------------------------------------------------------------------------------------------------------------
# modules
# ------------------------------------------------------------------------------------------------------------------
class _ConvBase(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=3,
                 transposed=False):
        super().__init__()
        if dimension != 3:
            raise NotImplementedError("only dimension=3 (the reference passes D=3 everywhere)")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation = kernel_size, stride, dilation
        self.dimension = dimension
        self.transposed = transposed
        self.kernel_volume = kernel_size ** dimension
        if self.kernel_volume > 1:
            self.kernel = nn.Parameter(torch.empty(self.kernel_volume, in_channels, out_channels))
        else:
            self.kernel = nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = nn.Parameter(torch.empty(1, out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # ME: uniform(-stdv, stdv), stdv = 1/sqrt(in_channels * kernel_volume) (out_channels for transposed)
        with torch.no_grad():
            n = (self.out_channels if self.transposed else self.in_channels) * self.kernel_volume
            stdv = 1.0 / math.sqrt(n)
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def weight3(self):
        w = self.kernel
        return w if w.dim() == 3 else w.unsqueeze(0)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        key = prefix + "kernel"
        perm_table = KERNEL_OFFSET_PERMUTATION
        if perm_table is not None and key in state_dict and state_dict[key].dim() == 3:
            perm = perm_table.get(state_dict[key].shape[0]) if isinstance(perm_table, dict) else perm_table
            if perm is not None and len(perm) == state_dict[key].shape[0]:
                if sorted(perm) != list(range(len(perm))):
                    raise ValueError(f"KERNEL_OFFSET_PERMUTATION for kernel volume {len(perm)} is not a permutation")
                state_dict[key] = state_dict[key][torch.as_tensor(list(perm), device=state_dict[key].device)]
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def _plan(self, x):
        cm, ts = x.coordinate_manager, x.tensor_stride
        ks, st = self.kernel_size, self.stride
        if not self.transposed:
            if ks == 1 and st == 1:
                return None, ts
            if ks == 3 and st == 1:
                if self.dilation == 1 and self.in_channels >= SPLIT_MIN_CHANNELS and self.out_channels >= SPLIT_MIN_CHANNELS:
                    cuts = cm.split_cuts_for(cm.stride_map(ts).V)  # the frame's own rules, else SPLIT_RULES
                    if cuts is not None:
                        if cm.split_ready is not None:
                            # the offset-range plans were built on the prep stream while this stream ran the encoder
                            torch.cuda.current_stream().wait_event(cm.split_ready)
                            cm.split_ready = None
                        return cm.plan_k3_split(ts, cuts), ts
                return cm.plan_k3(ts, self.dilation), ts
            if ks == 2 and st == 2:
                return cm.plan_down(ts), ts * 2
        else:
            if ks == 2 and st == 2:
                return cm.plan_up(ts), ts // 2
        raise NotImplementedError(
            f"kernel_size={ks} stride={st} transposed={self.transposed}: not used by the reference's U-Nets")

    def forward_fused(self, x, bn=None, residual=None, act=SV_ACT_NONE, slope=0.01, cat_with=None):
        """conv (+ folded BN / bias) (+ residual) (+ activation) in one launch.  cat_with: a SparseTensor on the
        output's coordinate map - the result is ME.cat(conv(x), cat_with) (model/backbone/minkunet.py:152-156), with
        the conv writing straight into the left columns of the concatenated buffer instead of being copied there."""
        plan, out_stride = self._plan(x)
        V_out = x.coordinate_manager.stride_map(out_stride).V
        scale = shift = None
        if bn is not None:
            scale, shift = bn.folded()
            if self.bias is not None:
                raise NotImplementedError("conv bias followed by a fused BatchNorm")
        elif self.bias is not None:
            shift = self.bias.detach().reshape(-1)
        res = residual.F if isinstance(residual, SparseTensor) else residual
        if cat_with is None:
            out = conv_forward(x.F, self.weight3().detach(), plan, V_out, scale, shift, res, act, slope)
            return x.new(out, tensor_stride=out_stride)
        if cat_with.coordinate_manager is not x.coordinate_manager or cat_with.tensor_stride != out_stride:
            raise ValueError("ME.cat needs tensors on the same coordinate map")
        skip = cat_with.F
        C = self.out_channels
        buf = torch.empty((V_out, C + skip.shape[1]), dtype=torch.float32, device=x.F.device)
        conv_forward(x.F, self.weight3().detach(), plan, V_out, scale, shift, res, act, slope, out=buf[:, :C])
        buf[:, C:].copy_(skip)
        return x.new(buf, tensor_stride=out_stride)

    def forward(self, x):
        return self.forward_fused(x)

    def extra_repr(self):
        return (f"in={self.in_channels}, out={self.out_channels}, kernel_size={self.kernel_size}, "
                f"stride={self.stride}, dilation={self.dilation}")


class MinkowskiConvolution(_ConvBase):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, convolution_mode=None, dimension=None):
        if dimension is None:
            raise ValueError("dimension is required")
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation, bias, dimension, transposed=False)


class MinkowskiConvolutionTranspose(_ConvBase):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, convolution_mode=None, dimension=None):
        if dimension is None:
            raise ValueError("dimension is required")
        if expand_coordinates:
            raise NotImplementedError("expand_coordinates=True (generative transposed conv) is not on the hot path")
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation, bias, dimension, transposed=True)


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)
        self._folded = None
        self._folded_ver = None

    def folded(self):
        if self.training:
            raise _lib.SvHipError("MinkowskiBatchNorm: training-mode statistics are out of scope (inference build); "
                                  "call model.eval()")
        ver = _tensor_versions(self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var)
        if self._folded is None or self._folded_ver != ver:
            self._folded = fold_bn(self.bn)
            self._folded_ver = ver
        return self._folded

    def forward(self, x):
        scale, shift = self.folded()
        return x.new(affine_act(x.F, scale, shift))


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        if isinstance(x, SparseTensor):
            return x.new(affine_act(x.F, act=SV_ACT_RELU))
        return affine_act(x, act=SV_ACT_RELU)


class MinkowskiLeakyReLU(nn.Module):
    def __init__(self, negative_slope=0.01, inplace=False):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, x):
        if isinstance(x, SparseTensor):
            return x.new(affine_act(x.F, act=SV_ACT_LEAKY_RELU, slope=self.negative_slope))
        return affine_act(x, act=SV_ACT_LEAKY_RELU, slope=self.negative_slope)


class MinkowskiSigmoid(nn.Module):
    # constructed by the heads (model/robotnet_segmentation.py:52) but never called in forward
    def forward(self, x):
        return x.new(torch.sigmoid(x.F))


class MinkowskiLinear(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)
        self._wt = None
        self._wt_ver = None

    def weight3(self):
        """[1, Cin, Cout] view of linear.weight^T, cached (the kernel wants W[k][c][n])."""
        ver = _tensor_versions(self.linear.weight)
        if self._wt is None or self._wt_ver != ver:
            self._wt = self.linear.weight.detach().t().contiguous().unsqueeze(0)
            self._wt_ver = ver
        return self._wt

    def forward_fused(self, x, act=SV_ACT_NONE, slope=0.01):
        F = x.F if isinstance(x, SparseTensor) else x
        shift = self.linear.bias.detach() if self.linear.bias is not None else None
        out = conv_forward(F, self.weight3(), None, F.shape[0], None, shift, None, act, slope)
        return x.new(out) if isinstance(x, SparseTensor) else out

    def forward(self, x):
        return self.forward_fused(x)


class MinkowskiGlobalMaxPooling(nn.Module):
    def forward(self, x):
        return PooledTensor(global_pool(x, _lib.SV_POOL_MAX))


class MinkowskiGlobalAvgPooling(nn.Module):
    def forward(self, x):
        return PooledTensor(global_pool(x, _lib.SV_POOL_AVG))


class PooledTensor:
    """Result of global pooling: one row per batch index; `.F/.features` as on ME's pooled SparseTensor."""

    def __init__(self, feats):
        self._F = feats

    @property
    def F(self):
        return self._F

    features = F


class BasicBlock(nn.Module):
    """MinkowskiEngine.modules.resnet_block.BasicBlock (expansion 1): conv3-BN-ReLU-conv3-BN-(+res)-ReLU."""
    expansion = 1
    NORM_TYPE = "BN"

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
        super().__init__()
        assert dimension > 0
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation,
                                          dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation,
                                          dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = self.conv1.forward_fused(x, bn=self.norm1, act=SV_ACT_RELU)
        if self.downsample is not None:
            ds_conv, ds_bn = self.downsample[0], self.downsample[1]
            residual = ds_conv.forward_fused(x, bn=ds_bn)
        else:
            residual = x
        return self.conv2.forward_fused(out, bn=self.norm2, residual=residual, act=SV_ACT_RELU)


class Bottleneck(nn.Module):
    """MinkowskiEngine.modules.resnet_block.Bottleneck (expansion 4): 1x1-BN-ReLU-3x3-BN-ReLU-1x1-BN-(+res)-ReLU."""
    expansion = 4
    NORM_TYPE = "BN"

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
        super().__init__()
        assert dimension > 0
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=1, dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=stride, dilation=dilation,
                                          dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv3 = MinkowskiConvolution(planes, planes * self.expansion, kernel_size=1, dimension=dimension)
        self.norm3 = MinkowskiBatchNorm(planes * self.expansion, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = self.conv1.forward_fused(x, bn=self.norm1, act=SV_ACT_RELU)
        out = self.conv2.forward_fused(out, bn=self.norm2, act=SV_ACT_RELU)
        if self.downsample is not None:
            residual = self.downsample[0].forward_fused(x, bn=self.downsample[1])
        else:
            residual = x
        return self.conv3.forward_fused(out, bn=self.norm3, residual=residual, act=SV_ACT_RELU)


def kaiming_normal_(tensor, mode="fan_out", nonlinearity="relu"):
    """ME.utils.kaiming_normal_ on a [K, Cin, Cout] kernel (model/backbone/resnet.py:89):
    fan_in = K * Cin, fan_out = K * Cout, std = sqrt(2 / fan)."""
    if tensor.dim() == 3:
        K, cin, cout = tensor.shape
    else:
        K, (cin, cout) = 1, tensor.shape
    fan = K * cout if mode == "fan_out" else K * cin
    gain = nn.init.calculate_gain(nonlinearity)
    std = gain / math.sqrt(fan)
    with torch.no_grad():
        return tensor.normal_(0, std)
