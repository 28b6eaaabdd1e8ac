"""RobotNet / RobotNetEncode: 7-DoF pose regression heads (position xyz + quaternion wxyz [+ confidences]).

Mirror of /root/reference/model/robotnet.py:37-83 and model/robotnet_encode.py:36-119.
  RobotNet:       forward_except_final -> BN + ReLU -> global MAX pool -> [cat joint angles] -> Linear C->2048
                  -> LeakyReLU -> Linear 2048->out; sigmoid on [:, 7:]; L2-normalise [:, 3:7] in eval.
  RobotNetEncode: encoder half only (to tensor stride 16) -> BN + ReLU -> global AVG pool -> same MLP;
                  optional position * quantization_size.
Attribute names (global_pool, leaky_relu, final_bn, output_layer.{0,1}, pose_regression.{0,1,2}) are the reference's.
The MLP on the pooled [B, C] rows runs through the same dense MFMA kernel as MinkowskiLinear.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import MinkowskiEngine as ME
from .. import nn as svnn
from .._lib import SV_ACT_LEAKY_RELU, SV_ACT_RELU
from ..utils import config
from ._select import pose_backbone


def _linear_fused(linear, x, act=0, slope=0.01):
    w3 = linear.weight.detach().t().contiguous().unsqueeze(0)
    bias = linear.bias.detach() if linear.bias is not None else None
    return svnn.conv_forward(x.contiguous(), w3, None, x.shape[0], None, bias, None, act, slope)


class _PoseHeadMixin:
    def _init_head(self, feat_channels, out_channels, avg_pool):
        cfg = config.Config()
        self.global_pool = ME.MinkowskiGlobalAvgPooling() if avg_pool else ME.MinkowskiGlobalMaxPooling()
        self.final_bn = ME.MinkowskiBatchNorm(out_channels)
        self.output_layer = nn.Sequential(ME.MinkowskiBatchNorm(feat_channels), self.relu)
        self.use_joint_angles = bool(cfg.STRUCTURE.use_joint_angles)
        self.pose_regression_input_size = feat_channels + (9 if self.use_joint_angles else 0)
        self.pose_regression = nn.Sequential(
            nn.Linear(self.pose_regression_input_size, 2048),
            nn.LeakyReLU(),
            nn.Linear(2048, out_channels),
        )

    def _regress(self, feats, joint_angles):
        scale, shift = self.output_layer[0].folded()
        feats = feats.new(svnn.affine_act(feats.F, scale, shift, act=SV_ACT_RELU))
        pooled = self.global_pool(feats).features
        if self.use_joint_angles:
            pooled = torch.cat((pooled, joint_angles.to(pooled)), dim=1)
        h = _linear_fused(self.pose_regression[0], pooled, SV_ACT_LEAKY_RELU, self.pose_regression[1].negative_slope)
        out = _linear_fused(self.pose_regression[2], h)
        out[:, 7:] = torch.sigmoid(out[:, 7:])  # confidences
        if not self.training:
            out[:, 3:7] = F.normalize(out[:, 3:7], p=2, dim=1)
        return out


def make_robotnet(backbone=None):
    UNet = pose_backbone(backbone, "ROTATION")

    class RobotNet(UNet, _PoseHeadMixin):
        name = "robotnet"

        def __init__(self, in_channels, out_channels, D=3):
            UNet.__init__(self, in_channels, out_channels, D)
            self.leaky_relu = ME.MinkowskiLeakyReLU(inplace=False)
            self._init_head(self.PLANES[-1] * self.BLOCK.expansion, out_channels, avg_pool=False)

        def forward(self, x):  # WXYZ
            joint_angles = None
            if isinstance(x, tuple):
                x, joint_angles = x
            return self._regress(self.forward_except_final(x), joint_angles)

    return RobotNet


def make_robotnet_encode(backbone=None):
    UNet = pose_backbone(backbone, "TRANSLATION")

    class RobotNetEncode(UNet, _PoseHeadMixin):
        name = "robotnet"

        def __init__(self, in_channels, out_channels, D=3):
            UNet.__init__(self, in_channels, out_channels, D)
            cfg = config.Config()
            self.leaky_relu = nn.LeakyReLU()
            self._init_head(self.PLANES[3] * self.BLOCK.expansion, out_channels, avg_pool=True)
            self.quantization_size = cfg()["DATA"].get("quantization_size", 1 / cfg.DATA.scale)
            self.voxelize_position = cfg()["DATA"].get("voxelize_position", False)

        def forward(self, x):  # WXYZ
            joint_angles = None
            if isinstance(x, tuple):
                x, joint_angles = x
            # robotnet_encode.py:72-95: conv0 + the four encoder stages only (MinkUNet naming, N_LEVELS = 4 stages)
            out = self.conv0p1s1.forward_fused(x, bn=self.bn0, act=SV_ACT_RELU)
            for i in range(1, 5):
                conv, bn, block = self._down_names(i)
                out = getattr(self, conv).forward_fused(out, bn=getattr(self, bn), act=SV_ACT_RELU)
                out = getattr(self, block)(out)
            out = self._regress(out, joint_angles)
            if not self.training and self.voxelize_position:
                out[:, :3] *= self.quantization_size
            return out

    return RobotNetEncode


RobotNet = make_robotnet()
