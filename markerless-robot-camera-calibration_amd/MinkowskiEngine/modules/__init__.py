from . import resnet_block  # noqa: F401
