"""RobotNetEncode (model/robotnet_encode.py:36-119 in the reference) — see model/robotnet.py."""
from .robotnet import make_robotnet_encode

RobotNetEncode = make_robotnet_encode()
