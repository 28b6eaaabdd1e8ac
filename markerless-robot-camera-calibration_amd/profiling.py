"""Live per-kernel timing with HIP events on the stream the kernels are launched on (torch's current stream).

bench.py installs a KernelTimer; nn.conv_forward reports each sv_conv_fwd launch to it with the quantities the
roofline needs (kernel offsets K, Cin, Cout, output rows, and the device-side pair count of the plan).  Nothing is
read back inside the timed region: events and pair counts are resolved after the final synchronize.
"""
import torch

TIMER = None  # set by bench.py


def conv_kernel_config(Cout):
    """Mirror of the dispatch in csrc/sv_conv.hip:sv_conv_fwd -> template instance name as rocprofv3 prints it."""
    if Cout > 128:
        return "conv_fwd_kernel<4, 3>" if (Cout % 192 == 0 or Cout > 2048) else "conv_fwd_kernel<4, 2>"
    if Cout > 64:
        return "conv_fwd_kernel<4, 2>"
    if Cout > 32:
        return "conv_fwd_kernel<4, 1>"
    if Cout > 16:
        return "conv_fwd_kernel<2, 1>"
    return "conv_fwd_kernel<1, 1>"


class KernelTimer:
    def __init__(self):
        self.records = []
        self.enabled = True

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, start_evt, kernel, K, Cin, Cout, V_out, pairs_dev):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.records.append((kernel, K, Cin, Cout, V_out, pairs_dev, start_evt, e))

    def clear(self):
        self.records = []

    def summarize(self):
        """After torch.cuda.synchronize(): per-kernel {launches, ms, flops, gather_bytes} (SURVEY.md §8d formulas:
        flops = 2 P Cin Cout; gather-bytes = P (4 Cin + 8) + 4 N_out Cout + 4 K Cin Cout)."""
        out = {}
        for kernel, K, Cin, Cout, V_out, pairs_dev, s, e in self.records:
            P = int(pairs_dev.item()) if pairs_dev is not None else V_out
            ms = s.elapsed_time(e)
            d = out.setdefault(kernel, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += ms
            d["flops"] += 2.0 * P * Cin * Cout
            d["bytes"] += P * (4.0 * Cin + 8) + 4.0 * V_out * Cout + 4.0 * K * Cin * Cout
        return out
