"""RobotNetVote (model/robotnet_vote.py:36-71 in the reference): same head as RobotNetSegmentation, 2 or 4 classes."""
import torch

from ..utils import config
from .robotnet_segmentation import make_robotnet_vote

RobotNetVote = make_robotnet_vote()


def get_criterion():
    # robotnet_vote.py:74-79 — kept for import compatibility of train_vote.py-shaped callers (training is out of scope)
    cfg = config.Config()
    return torch.nn.CrossEntropyLoss(reduction=cfg().get("TRAIN", {}).get("loss_reduction", "mean"),
                                     ignore_index=cfg.DATA.ignore_label)
