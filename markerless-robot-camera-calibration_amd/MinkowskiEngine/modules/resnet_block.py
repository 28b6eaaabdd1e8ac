"""MinkowskiEngine.modules.resnet_block (model/backbone/minkunet.py:30, resnet.py:29)."""
from ...nn import BasicBlock, Bottleneck  # noqa: F401
