"""ME.MinkowskiOps.MinkowskiLinear is how the heads spell it (model/robotnet_segmentation.py:44-49)."""
from ..nn import MinkowskiLinear  # noqa: F401
from ..sparse import cat  # noqa: F401
