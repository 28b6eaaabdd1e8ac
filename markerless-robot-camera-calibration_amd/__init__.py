"""MI355X-native sparse-voxel inference for markerless robot-camera calibration.

Package layout (only what the hot path of SURVEY.md §8 needs):
  csrc/            hand-written HIP kernels + the C-ABI (include/sv_hip.h) -> libsvhip.so
  _lib.py          ctypes binding (no fallback: raises if the library is missing)
  sparse.py nn.py  sparse-tensor runtime and layers under MinkowskiEngine's names
  MinkowskiEngine/ the ME-shaped namespace (drop-in boundary)
  model/ utils/ app/  host-side mirror of the reference's model / dense-solve / InferenceEngine interface
  synth.py         deterministic synthetic inputs
"""
import sys

from . import _lib, synth  # noqa: F401

__all__ = ["install_as_minkowski_engine", "MinkowskiEngine"]


def __getattr__(name):
    if name == "MinkowskiEngine":
        import importlib

        return importlib.import_module(__name__ + ".MinkowskiEngine")
    raise AttributeError(name)


def install_as_minkowski_engine():
    """Register the ME-shaped namespace as `MinkowskiEngine` so `import MinkowskiEngine as ME` resolves to it."""
    import importlib

    me = importlib.import_module(__name__ + ".MinkowskiEngine")
    sys.modules["MinkowskiEngine"] = me
    sys.modules["MinkowskiEngine.modules"] = me.modules
    sys.modules["MinkowskiEngine.modules.resnet_block"] = me.modules.resnet_block
    sys.modules["MinkowskiEngine.utils"] = me.utils
    sys.modules["MinkowskiEngine.MinkowskiOps"] = me.MinkowskiOps
    return me
