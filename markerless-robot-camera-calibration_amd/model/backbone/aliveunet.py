"""AliveUNet: the 7-down / 7-up variant (fallback backbone when STRUCTURE.backbone is not a minkunet name).

Mirror of /root/reference/model/backbone/aliveunet.py:45-275: attribute names conv0p1s1, bn0, conv{i}p{2^(i-1)}s2,
bn{i}, block{i} (i = 1..7), convtr{j}, bntr{j}, block{j+1} (j = 7..13), final, relu.  `forward` returns block14's
output — the reference never applies `final` here (:177-265) and has no forward_except_final.
"""
from ...MinkowskiEngine.modules.resnet_block import BasicBlock, Bottleneck
from ...utils import config
from .minkunet import MinkUNetBase


class AliveUNetBase(MinkUNetBase):
    BLOCK = BasicBlock
    DILATIONS = (1,) * 14
    LAYERS = (1,) * 14
    PLANES = (32, 64, 96, 128, 160, 192, 224, 224, 192, 160, 128, 96, 64, 32)
    INIT_DIM = 32
    N_LEVELS = 7

    @classmethod
    def _up_names(cls, j):
        return f"convtr{j}", f"bntr{j}", f"block{j + 1}"

    def _decoder_inplanes(self, j, skip_planes):
        # aliveunet.py:122,129,...,164: the reference sizes block(j+1) as PLANES[j+1] + PLANES[13-j]*expansion
        # (equal to the real concat width for the symmetric plane tables it is used with); last stage uses INIT_DIM
        if j == 13:
            return self.PLANES[13] + self.INIT_DIM
        return self.PLANES[j + 1] + self.PLANES[13 - j] * self.BLOCK.expansion

    def _decoder_block(self, j):
        # aliveunet.py:123,130,137,144,151,158: block(j+1) = _make_layer(BLOCK, PLANES[j+1], LAYERS[j+1]) for
        # j = 7..12 — one table entry further than MinkUNet — and :165: block14 = (PLANES[13], LAYERS[13]).  The
        # transposed conv after block(j+1) therefore maps PLANES[j+1]*expansion -> PLANES[j+1] (:124-126).
        k = min(j + 1, 13)
        return self.PLANES[k], self.LAYERS[k]

    def forward(self, x):
        return self.forward_except_final(x)


def make_alive_unet(m=None, block_reps=None, bottleneck=None):
    """AliveUNet class for the given STRUCTURE.{m, block_reps, bottleneck} (aliveunet.py:268-275)."""
    cfg = config.Config()
    m = cfg.STRUCTURE.m if m is None else m
    block_reps = cfg.STRUCTURE.block_reps if block_reps is None else block_reps
    bottleneck = cfg()["STRUCTURE"].get("bottleneck") if bottleneck is None else bottleneck
    planes = tuple(i * m for i in (list(range(1, 8)) + list(range(7, 0, -1))))
    return type("AliveUNet", (AliveUNetBase,), {
        "BLOCK": Bottleneck if bottleneck else BasicBlock,
        "PLANES": planes,
        "LAYERS": tuple(block_reps for _ in planes),
        "__module__": __name__,
    })
