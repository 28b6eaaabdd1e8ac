"""MinkUNet backbones on the MI355X-native runtime.

Mirror of /root/reference/model/backbone/minkunet.py:37-251: same class names, PLANES/LAYERS tables, attribute names
(conv0p1s1, bn0, conv1p1s2 ... convtr7p2s2, bntr7, block1..8, final, relu) and therefore the same state_dict keys.
The graph is the reference's (`forward_except_final` :125-183, `forward` :185-187); what differs is HOW it runs: every
conv -> BN -> ReLU triple is ONE libsvhip call (BN folded into the conv epilogue), the residual add of a block is
fused into its second conv, and `final`'s bias (plus an optional trailing LeakyReLU of the heads) rides in its epilogue.
"""
from ... import MinkowskiEngine as ME
from ..._lib import SV_ACT_NONE, SV_ACT_RELU
from ...MinkowskiEngine.modules.resnet_block import BasicBlock, Bottleneck
from .resnet import ResNetBase


class MinkUNetBase(ResNetBase):
    BLOCK = None
    DILATIONS = (1, 1, 1, 1, 1, 1, 1, 1)
    LAYERS = (2, 2, 2, 2, 2, 2, 2, 2)
    PLANES = (32, 64, 128, 256, 256, 128, 96, 96)
    INIT_DIM = 32
    OUT_TENSOR_STRIDE = 1
    N_LEVELS = 4  # stride-2 steps down (and up)

    def __init__(self, in_channels, out_channels, D=3):
        ResNetBase.__init__(self, in_channels, out_channels, D)

    # names as in the reference: conv{i}p{stride_in}s2 / convtr{j}p{stride_in}s2
    @classmethod
    def _down_names(cls, i):
        return f"conv{i}p{2 ** (i - 1)}s2", f"bn{i}", f"block{i}"

    @classmethod
    def _up_names(cls, j):
        n = cls.N_LEVELS
        return f"convtr{j}p{2 ** (2 * n - j)}s2", f"bntr{j}", f"block{j + 1}"

    def _decoder_inplanes(self, j, skip_planes):
        # minkunet.py:91,98,105,112: width after ME.cat(convtr_j output, encoder skip)
        return self.PLANES[j] + skip_planes[2 * self.N_LEVELS - 1 - j]

    def _decoder_block(self, j):
        # minkunet.py:92,99,106,113: block(j+1) = _make_layer(BLOCK, PLANES[j], LAYERS[j])
        return self.PLANES[j], self.LAYERS[j]

    def network_initialization(self, in_channels, out_channels, D):
        n, P, L, exp = self.N_LEVELS, self.PLANES, self.LAYERS, self.BLOCK.expansion
        self.inplanes = self.INIT_DIM
        self.conv0p1s1 = ME.MinkowskiConvolution(in_channels, self.inplanes, kernel_size=3, dimension=D)
        self.bn0 = ME.MinkowskiBatchNorm(self.inplanes)
        for i in range(1, n + 1):  # encoder: stride-2 conv, BN, residual stack
            conv, bn, block = self._down_names(i)
            setattr(self, conv, ME.MinkowskiConvolution(self.inplanes, self.inplanes, kernel_size=2, stride=2,
                                                        dimension=D))
            setattr(self, bn, ME.MinkowskiBatchNorm(self.inplanes))
            setattr(self, block, self._make_layer(self.BLOCK, P[i - 1], L[i - 1]))
        skip_planes = [self.INIT_DIM] + [P[i] * exp for i in range(n - 1)]  # out_p1, block1 .. block(n-1)
        for j in range(n, 2 * n):  # decoder: transposed conv, BN, concat skip, residual stack
            conv, bn, block = self._up_names(j)
            setattr(self, conv, ME.MinkowskiConvolutionTranspose(self.inplanes, P[j], kernel_size=2, stride=2,
                                                                 dimension=D))
            setattr(self, bn, ME.MinkowskiBatchNorm(P[j]))
            self.inplanes = self._decoder_inplanes(j, skip_planes)
            setattr(self, block, self._make_layer(self.BLOCK, *self._decoder_block(j)))
        self.final = ME.MinkowskiConvolution(P[2 * n - 1] * exp, out_channels, kernel_size=1, bias=True, dimension=D)
        self.relu = ME.MinkowskiReLU(inplace=True)

    def encode(self, x):
        """conv0 + the n encoder stages; returns (deepest tensor, [out_p1, block1 .. block(n-1)] skips)."""
        out = self.conv0p1s1.forward_fused(x, bn=self.bn0, act=SV_ACT_RELU)
        skips = [out]
        for i in range(1, self.N_LEVELS + 1):
            conv, bn, block = self._down_names(i)
            out = getattr(self, conv).forward_fused(out, bn=getattr(self, bn), act=SV_ACT_RELU)
            out = getattr(self, block)(out)
            skips.append(out)
        return skips.pop(), skips

    def forward_except_final(self, x):
        out, skips = self.encode(x)
        n = self.N_LEVELS
        # optional per-frame callable the frame pipeline attaches to the frame's coordinate manager (app/pipeline.py): it is
        # told where the stride-1 decoder stage (the chip-filling 63 % of a frame) begins and ends
        phase_hook = getattr(x.coordinate_manager, "phase_hook", None)
        for j in range(n, 2 * n):
            conv, bn, block = self._up_names(j)
            hook = phase_hook if j == 2 * n - 1 else None
            if hook is not None:
                hook("level0_begin")
            # transposed conv + BN + ReLU written straight into the left columns of ME.cat(out, skip)
            out = getattr(self, conv).forward_fused(out, bn=getattr(self, bn), act=SV_ACT_RELU, cat_with=skips.pop())
            out = getattr(self, block)(out)
            if hook is not None:
                hook("level0_end")
        return out

    def forward(self, x, final_act=SV_ACT_NONE, final_slope=0.01):
        out = self.forward_except_final(x)
        return self.final.forward_fused(out, act=final_act, slope=final_slope)


class MinkUNet14(MinkUNetBase):
    BLOCK = BasicBlock
    LAYERS = (1, 1, 1, 1, 1, 1, 1, 1)


class MinkUNet18(MinkUNetBase):
    BLOCK = BasicBlock
    LAYERS = (2, 2, 2, 2, 2, 2, 2, 2)


class MinkUNet34(MinkUNetBase):
    BLOCK = BasicBlock
    LAYERS = (2, 3, 4, 6, 2, 2, 2, 2)


class MinkUNet50(MinkUNetBase):
    BLOCK = Bottleneck
    LAYERS = (2, 3, 4, 6, 2, 2, 2, 2)


class MinkUNet101(MinkUNetBase):
    BLOCK = Bottleneck
    LAYERS = (2, 3, 4, 23, 2, 2, 2, 2)


def _variant(base, name, planes):
    return type(name, (base,), {"PLANES": planes, "__module__": __name__})


MinkUNet14A = _variant(MinkUNet14, "MinkUNet14A", (32, 64, 128, 256, 128, 128, 96, 96))
MinkUNet14B = _variant(MinkUNet14, "MinkUNet14B", (32, 64, 128, 256, 128, 128, 128, 128))
MinkUNet14C = _variant(MinkUNet14, "MinkUNet14C", (32, 64, 128, 256, 192, 192, 128, 128))
MinkUNet14D = _variant(MinkUNet14, "MinkUNet14D", (32, 64, 128, 256, 384, 384, 384, 384))
MinkUNet18A = _variant(MinkUNet18, "MinkUNet18A", (32, 64, 128, 256, 128, 128, 96, 96))
MinkUNet18B = _variant(MinkUNet18, "MinkUNet18B", (32, 64, 128, 256, 128, 128, 128, 128))
MinkUNet18D = _variant(MinkUNet18, "MinkUNet18D", (32, 64, 128, 256, 384, 384, 384, 384))
MinkUNet34A = _variant(MinkUNet34, "MinkUNet34A", (32, 64, 128, 256, 256, 128, 64, 64))
MinkUNet34B = _variant(MinkUNet34, "MinkUNet34B", (32, 64, 128, 256, 256, 128, 64, 32))
MinkUNet34C = _variant(MinkUNet34, "MinkUNet34C", (32, 64, 128, 256, 256, 128, 96, 96))
