"""ResNetBase pieces the U-Nets inherit: `_make_layer` and `weight_initialization`.

Mirror of /root/reference/model/backbone/resnet.py:34-127 (the ResNet14..101 / ResFieldNet classes there are never
instantiated by a head, SURVEY.md §2.1, and are not rebuilt).  Same attribute names -> same state_dict keys.
"""
import torch.nn as nn

from ... import MinkowskiEngine as ME


class ResNetBase(nn.Module):
    BLOCK = None
    LAYERS = ()
    INIT_DIM = 64
    PLANES = (64, 128, 256, 512)

    def __init__(self, in_channels, out_channels, D=3):
        nn.Module.__init__(self)
        self.D = D
        assert self.BLOCK is not None
        self.network_initialization(in_channels, out_channels, D)
        self.weight_initialization()

    def network_initialization(self, in_channels, out_channels, D):
        raise NotImplementedError("only the U-Net subclasses are on the hot path")

    def weight_initialization(self):
        # resnet.py:86-93: kaiming-normal(fan_out, relu) conv kernels, BN gamma = 1, beta = 0
        for m in self.modules():
            if isinstance(m, ME.MinkowskiConvolution):
                ME.utils.kaiming_normal_(m.kernel, mode="fan_out", nonlinearity="relu")
            if isinstance(m, ME.MinkowskiBatchNorm):
                nn.init.constant_(m.bn.weight, 1)
                nn.init.constant_(m.bn.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1, bn_momentum=0.1):
        # resnet.py:95-127: first block may change width (1x1 conv + BN on the residual path), the rest keep it
        out_planes = planes * block.expansion
        downsample = None
        if stride != 1 or self.inplanes != out_planes:
            downsample = nn.Sequential(
                ME.MinkowskiConvolution(self.inplanes, out_planes, kernel_size=1, stride=stride, dimension=self.D),
                ME.MinkowskiBatchNorm(out_planes),
            )
        layers = [block(self.inplanes, planes, stride=stride, dilation=dilation, downsample=downsample,
                        dimension=self.D)]
        self.inplanes = out_planes
        layers += [block(self.inplanes, planes, stride=1, dilation=dilation, dimension=self.D)
                   for _ in range(1, blocks)]
        return nn.Sequential(*layers)
