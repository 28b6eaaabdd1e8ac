"""RobotNetSegmentation / RobotNetVote: per-voxel classification heads on a sparse U-Net.

Mirror of /root/reference/model/robotnet_segmentation.py:35-64 and model/robotnet_vote.py:36-71 (identical graphs:
U-Net(out=256) -> LeakyReLU -> Linear 256->1024 -> LeakyReLU -> Linear 1024->num_classes).  Attribute names
(`leaky_relu`, `regression.{0,1,2}`, `sigm`) and so state_dict keys (`regression.0.linear.weight` ...) are the
reference's.  Execution: three libsvhip calls after the U-Net body — `final` (bias + LeakyReLU in its epilogue),
regression.0 (bias + LeakyReLU fused), regression.2 (bias).
"""
import torch.nn as nn

from .. import MinkowskiEngine as ME
from .._lib import SV_ACT_LEAKY_RELU
from ..utils import config
from ._select import pose_backbone, segmentation_backbone

EPS = 1e-6


def _classification_head(UNet, default_classes, class_name):
    class _Head(UNet):
        name = "robotnet"

        def __init__(self, in_channels, out_channels=256, D=3, num_classes=None):
            UNet.__init__(self, in_channels, out_channels, D)
            if num_classes is None:
                num_classes = default_classes()
            self.leaky_relu = ME.MinkowskiLeakyReLU()
            self.regression = nn.Sequential(
                ME.MinkowskiOps.MinkowskiLinear(256, 1024),
                ME.MinkowskiLeakyReLU(),
                ME.MinkowskiOps.MinkowskiLinear(1024, num_classes),
            )
            self.sigm = ME.MinkowskiSigmoid()

        def forward(self, x):
            if isinstance(x, tuple):
                x, _joint_angles = x
            slope = self.leaky_relu.negative_slope
            out = UNet.forward(self, x, final_act=SV_ACT_LEAKY_RELU, final_slope=slope)
            out = self.regression[0].forward_fused(out, act=SV_ACT_LEAKY_RELU,
                                                   slope=self.regression[1].negative_slope)
            return self.regression[2].forward_fused(out)

    _Head.__name__ = _Head.__qualname__ = class_name
    return _Head


def make_robotnet_segmentation(backbone=None):
    return _classification_head(segmentation_backbone(backbone), lambda: config.Config().DATA.classes,
                                "RobotNetSegmentation")


def make_robotnet_vote(backbone=None):
    # robotnet_vote.py:39: 2 classes for ee_seg data, else 4
    return _classification_head(pose_backbone(backbone, "ROTATION"),
                                lambda: 2 if config.Config().DATA.data_type == "ee_seg" else 4, "RobotNetVote")


RobotNetSegmentation = make_robotnet_segmentation()
