"""`MinkowskiEngine`-shaped namespace over the MI355X-native runtime (the drop-in boundary).

The reference does `import MinkowskiEngine as ME` and `from MinkowskiEngine.modules.resnet_block import BasicBlock`
(model/backbone/minkunet.py:28-30).  `mrcc_amd.install_as_minkowski_engine()` registers this package under that
name so reference-shaped model files import unchanged; the build's own model mirror imports it directly.
"""
from ..nn import (BasicBlock, Bottleneck, MinkowskiBatchNorm, MinkowskiConvolution, MinkowskiConvolutionTranspose,
                  MinkowskiGlobalAvgPooling, MinkowskiGlobalMaxPooling, MinkowskiLeakyReLU, MinkowskiLinear,
                  MinkowskiReLU, MinkowskiSigmoid)
from ..sparse import (CoordinateManager, MinkowskiAlgorithm, SparseTensor, SparseTensorQuantizationMode, TensorField,
                      cat)
from . import MinkowskiOps, modules, utils  # noqa: F401

__version__ = "0.5.4+mi355x"
