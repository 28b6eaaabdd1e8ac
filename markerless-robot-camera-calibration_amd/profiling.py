"""Live per-kernel timing with HIP events on the stream the kernels are launched on (torch's current stream).

bench.py installs a KernelTimer; nn.conv_forward reports each sv_conv_fwd launch to it with the quantities the
roofline needs (kernel offsets K, Cin, Cout, output rows, and the device-side pair count of the plan).  Nothing is
read back inside the timed region: events and pair counts are resolved after the final synchronize.
"""
import math

import torch

TIMER = None  # set by bench.py
# Optional list: nn.conv_forward appends (instance name, {"fast", "ring", "full"}, K, Cin, Cout, rows) per launch, as
# reported by the library itself (sv_conv_last_instance) - tests check which instances a configuration really ran on.
INSTANCE_LOG = None

_CANDIDATES = {
    "wide3": [(64, 4, 3, 2500), (32, 4, 3, 1500), (32, 2, 3, 0)],
    "wide3_128": [(64, 4, 3, 2500), (64, 4, 2, 1200), (32, 4, 3, 1200), (16, 4, 3, 0)],
    "wide3_128_full": [(64, 4, 3, 2500), (32, 4, 3, 1200), (16, 4, 3, 0)],  # Cin a multiple of the 16-row tile's 128-channel step
    "wide2": [(128, 4, 2, 1300), (64, 4, 2, 1500), (32, 4, 2, 1500), (32, 2, 2, 0)],
    "wide2_64": [(128, 4, 2, 1300), (64, 4, 2, 1500), (32, 4, 2, 1500), (16, 4, 1, 0)],
    "c64": [(64, 4, 1, 1500), (32, 4, 1, 1500), (16, 4, 1, 0)],
    "c32": [(128, 2, 1, 1500), (64, 2, 1, 1500), (32, 2, 1, 0)],
    "c16": [(128, 1, 1, 600), (64, 1, 1, 0)],
}


import os  # noqa: E402

# the same experiment switches select_and_launch() reads (tools/sweep_pipeline.sh, tools/ab_env*.sh vary them); what a launch
# really ran on is reported by the library itself: _lib.conv_last_instance()
CONV_TAIL_FRACTION = float(os.environ.get("SV_CONV_TAIL", "0.15"))  # SV_CONV_TAIL_DEFAULT of csrc/sv_conv.hip
CONV_WANT_SCALE = float(os.environ.get("SV_CONV_WANT_SCALE", "0.3"))  # SV_CONV_WANT_SCALE_DEFAULT of csrc/sv_conv.hip


_FUSED = {  # (Cin, Cout) -> candidates of the fused-offset form (thin layers, K > 1)
    (32, 32): [(64, 2, 1, 3000), (32, 2, 1, 0)],
    (32, 64): [(32, 4, 1, 1500), (16, 4, 1, 0)],
    (64, 64): [(32, 4, 1, 1500), (16, 4, 1, 0)],
    (64, 128): [(32, 4, 1, 1500), (16, 4, 1, 0)],
}


def conv_kernel_config(Cout, Vpad, Cin=None, K=1):
    """Mirror of select_and_launch() in csrc/sv_conv.hip -> kernel instance name: conv_fwd_kernel<TM, WAVES_N, NT>,
    with ", fused Cin" appended for the fused-offset form of the thin layers."""
    if K == 1 and Cout <= 4 and Cin is not None and Cin >= 64 and Cin % 4 == 0:
        return f"linear_narrow_kernel<{Cout}>"  # dense rows only; every K = 1 layer of the path is dense
    if K > 1 and Cin == 3 and Cout == 32:
        return "conv_first_mfma_kernel<3, 32>"  # (SV_CONV_FIRST_VALU: the thread-per-voxel VALU kernel it replaced)
    if K > 1 and Cin == 32 and Cout == 32:
        # conv_thin_lds_kernel (weights resident in LDS, one 16-wave workgroup per CU) under sv_conv_set_dispatch(>= 1): one
        # frame alone on the GPU; the recorded name is the library's (sv_conv_last_instance), this one only gates timing
        return "conv_thin_kernel<32, 32>"
    fused = None
    if K > 1 and Cin is not None:
        if Cin == 3 and 16 < Cout <= 32:
            fused = [(64, 2, 1, 3000), (32, 2, 1, 0)]
        else:
            fused = _FUSED.get((Cin, Cout))
    if fused is not None:
        cands = fused
    elif Cout > 128 and (Cout % 192 == 0 or Cout % 96 == 0 or Cout > 2048):
        if Cout % 128 != 0:
            cands = _CANDIDATES["wide3"]
        else:
            cands = _CANDIDATES["wide3_128_full" if (Cin is not None and Cin % 128 == 0) else "wide3_128"]
    elif Cout > 64:
        cands = _CANDIDATES["wide2_64" if Cout % 64 == 0 else "wide2"]
    elif Cout > 32:
        cands = _CANDIDATES["c64"]
    elif Cout > 16:
        cands = _CANDIDATES["c32"]
    else:
        cands = _CANDIDATES["c16"]
    suffix = f", fused {Cin}>" if fused is not None else ">"
    for tm, wn, nt, want in cands:
        tn = wn * nt * 16
        if (Vpad // tm) * ((Cout + tn - 1) // tn) >= want * CONV_WANT_SCALE:
            if ((tm, wn, nt) == (64, 4, 3) and fused is None and K > 1 and Cin is not None and Cin % 4 == 0
                    and Cout % tn == 0 and CONV_TAIL_FRACTION > 0 and int((Vpad // 128) * CONV_TAIL_FRACTION) >= 1):
                return "conv_fwd_dual_kernel<64, 32, 4, 3>"  # chip-filling layer: half-height tiles at the end of the grid
            return f"conv_fwd_kernel<{tm}, {wn}, {nt}{suffix}"
    tm, wn, nt, _ = cands[-1]
    return f"conv_fwd_kernel<{tm}, {wn}, {nt}{suffix}"


class KernelTimer:
    def __init__(self, capacity=0):
        self.records = []
        self.enabled = True
        self.only = None  # optional set of kernel names to time (None = all)
        # events are created up front: creating ~250 of them per frame inside the timed region costs milliseconds
        self._pool = [torch.cuda.Event(enable_timing=True) for _ in range(capacity)]
        self._next = 0
        self._first_of_frame = False

    def _event(self):
        if self._next < len(self._pool):
            e = self._pool[self._next]
            self._next += 1
            return e
        return torch.cuda.Event(enable_timing=True)

    def want(self, kernel):
        return self.enabled and (self.only is None or kernel in self.only)

    def start(self):
        e = self._event()
        e.record()
        return e

    def stop(self, start_evt, kernel, K, Cin, Cout, V_out, pairs_dev, level=None, passes=1):
        e = self._event()
        e.record()
        self.records.append((kernel, K, Cin, Cout, V_out, pairs_dev, start_evt, e, self._first_of_frame, level, passes))
        self._first_of_frame = False

    def frame_boundary(self):
        """The next timed launch is the first of a frame: its event interval can absorb the wait for the prepared
        frame (cross-stream) or for the host, so summarize() leaves it out."""
        self._first_of_frame = True

    def clear(self):
        self.records = []

    @staticmethod
    def layer_key(kernel, K, Cin, Cout, V_out, level=None):
        """(kernel instance, layer shape): launches that process the same kind of unit - the same kernel volume and
        channel counts on the same pyramid level, named by the output's tensor stride ("s1", "s2", ...) where the launch
        has a plan; dense layers (no plan, hence no level) by their rows to the nearest power of two ("~2^17": the frames
        of a pool differ by < 1 % in voxel count, pyramid levels by 3-4x)."""
        where = f"s{level}" if level is not None else f"~2^{int(round(math.log2(max(V_out, 1))))}"
        return (kernel, K, Cin, Cout, where)

    def summarize(self, by_layer=False):
        """After torch.cuda.synchronize(): per-kernel {launches, ms, flops, gather_bytes} (SURVEY.md §8d formulas:
        flops = 2 P Cin Cout; gather-bytes = P (4 Cin + 8) + 4 N_out Cout + 4 K Cin Cout).  by_layer: keyed by
        layer_key() instead of the kernel name alone (an instance that serves several layer shapes has no one
        "flops per launch")."""
        out = {}
        for kernel, K, Cin, Cout, V_out, pairs_dev, s, e, first, level, passes in self.records:
            if first:
                continue
            P = int(pairs_dev.item()) if pairs_dev is not None else V_out
            ms = s.elapsed_time(e)
            key = self.layer_key(kernel, K, Cin, Cout, V_out, level) if by_layer else kernel
            d = out.setdefault(key, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "rows": 0.0, "kernel_launches": 0})
            d["rows"] += V_out
            d["launches"] += 1  # layers; a layer run as offset-range passes is several kernel launches
            d["kernel_launches"] += passes
            d["ms"] += ms
            d["flops"] += 2.0 * P * Cin * Cout
            d["bytes"] += P * (4.0 * Cin + 8) + 4.0 * V_out * Cout + 4.0 * K * Cin * Cout
        return out


def pass_gflop(fn):
    """Algorithmic GFLOP (2 P Cin Cout per conv / linear layer, P = kernel-map pairs or dense rows) of the
    sv_conv_fwd launches fn() makes - one forward pass's worth, counted by timing it once under a private KernelTimer."""
    global TIMER
    saved, TIMER = TIMER, KernelTimer()
    try:
        fn()
        torch.cuda.synchronize()
        return sum(d["flops"] for d in TIMER.summarize().values()) / 1e9
    finally:
        TIMER = saved
