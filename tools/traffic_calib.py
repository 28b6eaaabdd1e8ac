"""Known-traffic launch for calibrating FETCH_SIZE / WRITE_SIZE on this kernel's access pattern (MI355X guide, HBM
section): a dense 1x1 conv (K = 1) over V rows x 384 channels -> 192 channels reads every input row exactly once
(TN = 192 covers all output channels, so there is one column slice) and writes every output row once.
    input  V * 384 * 4 B   (V = 1.3M -> 2.0 GB: far beyond the 256 MiB Infinity Cache and just below the 2 GB extent above
                            which sv_conv_fwd splits a dense layer into several launches)
    output V * 192 * 4 B
Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) and compare."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrcc_amd import nn as svnn
V = int(sys.argv[1]) if len(sys.argv) > 1 else 1_300_000
dev = torch.device("cuda:0")
x = torch.randn(V, 384, device=dev)
W = torch.randn(1, 384, 192, device=dev) * 0.05
for _ in range(3):
    out = svnn.conv_forward(x, W, None, V)
torch.cuda.synchronize()
print(f"known: read {V * 384 * 4 + 384 * 192 * 4} B, write {V * 192 * 4} B per launch")
