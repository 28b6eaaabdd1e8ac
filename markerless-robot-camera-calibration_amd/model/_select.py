"""Backbone selection shared by the heads.

The reference picks the U-Net class at import time from the config singleton (model/robotnet_segmentation.py:17-28,
model/robotnet.py:17-30, model/robotnet_vote.py:18-31, model/robotnet_encode.py:18-31).  Here the same tables are
functions evaluated when a head class is requested, so a process can build several configurations.
"""
from ..utils import config
from .backbone import minkunet
from .backbone.aliveunet import make_alive_unet


def segmentation_backbone(name=None):
    # model/robotnet_segmentation.py:17-28 (INFERENCE.SEGMENTATION.backbone in inference mode)
    cfg = config.Config()
    if name is None:
        name = cfg()["STRUCTURE"].get("backbone")
        if cfg.MODE == "inference":
            name = cfg.INFERENCE.SEGMENTATION.backbone
    table = {"minkunet101": minkunet.MinkUNet101, "minkunet34C": minkunet.MinkUNet34C,
             "minkunet14A": minkunet.MinkUNet14A}
    return table.get(name, minkunet.MinkUNet18D)


def pose_backbone(name=None, section="ROTATION"):
    # model/robotnet.py:17-30, robotnet_vote.py:18-31 (ROTATION), robotnet_encode.py:18-31 (TRANSLATION)
    cfg = config.Config()
    if name is None:
        name = cfg()["STRUCTURE"].get("backbone")
        if cfg.MODE == "inference":
            name = getattr(cfg.INFERENCE, section).backbone
    table = {"minkunet": minkunet.MinkUNet18D, "minkunet101": minkunet.MinkUNet101,
             "minkunet34C": minkunet.MinkUNet34C, "minkunet14A": minkunet.MinkUNet14A}
    if name in table:
        return table[name]
    return make_alive_unet()
