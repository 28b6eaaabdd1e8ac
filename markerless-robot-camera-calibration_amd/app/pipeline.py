"""Two-stream frame pipeline: coordinate work of frame i+1 overlaps the U-Net of frame i.

Per frame the host must read a handful of sizes back from the GPU (voxel count of every pyramid level) before it can
allocate and launch the next stage.  On a single stream each read-back waits for everything queued before it —
including the previous frame's ~30 ms of convolutions — so host and GPU take turns.  Here the voxelisation, coordinate
maps, kernel maps and conv plans of a frame are built on a PREP stream (its read-backs only wait for its own small
kernels) while the previous frame's convolutions run on the COMPUTE stream; an event hands the prepared frame over.
Prepared frames are kept alive until the compute stream has finished with them, so the caching allocator cannot
recycle their memory under a running kernel.
"""
import collections
import os

import torch

from .. import profiling

from .. import MinkowskiEngine as ME


class PreparedFrame:
    __slots__ = ("field", "x", "ready", "done", "tag")

    def __init__(self, field, x, ready, tag=None):
        self.field, self.x, self.ready, self.done, self.tag = field, x, ready, None, tag


def build_unet_plans(cm, levels=4):
    """Everything a MinkUNet-shaped graph will ask the coordinate manager for (model/backbone/minkunet.py:125-183):
    3x3x3 maps at strides 1..2^levels, stride-2 down maps and transposed up maps between neighbouring levels."""
    for l in range(levels):
        ts = 1 << l
        cm.plan_k3(ts)
        cm.plan_down(ts)
    cm.plan_k3(1 << levels)
    for l in range(levels, 0, -1):
        cm.plan_up(1 << l)


class FramePipeline:
    def __init__(self, device, levels=4, encoder_only=False, compute_streams=1):
        self.device = torch.device(device)
        self.levels = levels
        self.encoder_only = encoder_only
        # the prep stream's ~150 small kernels per frame must not queue behind thousands of conv workgroups: the host
        # blocks on their size read-backs, and a late prepare() starves a compute stream (high priority = -1)
        prio = int(os.environ.get("MRCC_PREP_PRIORITY", "-1"))
        self.prep_stream = torch.cuda.Stream(device=self.device, priority=prio)
        # compute_streams > 1: consecutive frames run on alternating streams, so the small-pyramid-level and thin-layer
        # kernels of one frame (which cannot fill 256 CUs) overlap the big convolutions of its neighbour
        self.compute_streams = [torch.cuda.Stream(device=self.device) for _ in range(compute_streams)] \
            if compute_streams > 1 else []
        self._next_stream = 0
        # The stride-1 decoder stages (level 0: 63 % of a frame's flops, chip-filling launches) of consecutive frames take
        # turns instead of overlapping each other; everything else of the other frames still runs underneath them.  Two
        # streams: 61.7 frames/s against 62.1 without, but the level-0 launches then run at a steady 0.61 of the matrix
        # peak each instead of 0.48-0.61 depending on how the two frames happen to be phased; three / four / five
        # streams: 64.2 / 63.4 / 63.8 frames/s (63.9-64.5 without at three) at 0.50 / 0.58 / 0.55 per launch.
        self.stagger_level0 = os.environ.get("MRCC_STAGGER_LEVEL0", "1") == "1"
        self._level0_done = None
        self.single = False  # True: run every frame on the first compute stream (isolated kernel timing)
        self._retired = collections.deque()

    def prepare(self, coords4, feats, tag=None):
        """Enqueue voxelisation + coordinate/kernel maps + plans of one frame on the prep stream.
        coords4: float32 [N,4] (batch, x*scale, y*scale, z*scale) on the GPU; feats: float32 [N,C] on the GPU."""
        with torch.cuda.stream(self.prep_stream):
            field = ME.TensorField(features=feats, coordinates=coords4,
                                   quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE,
                                   minkowski_algorithm=ME.MinkowskiAlgorithm.SPEED_OPTIMIZED, device=self.device)
            x = field.sparse()
            cm = x.coordinate_manager
            if self.encoder_only:
                for l in range(self.levels):
                    cm.plan_k3(1 << l)
                    cm.plan_down(1 << l)
                cm.plan_k3(1 << self.levels)
            else:
                build_unet_plans(cm, self.levels)
            ready = torch.cuda.Event()
            ready.record(self.prep_stream)
        return PreparedFrame(field, x, ready, tag)

    def run(self, prepared, fn):
        """Run fn(x, field) on a compute stream once the frame is ready; returns fn's result (valid on that stream;
        `drain()` or an event wait orders it against other streams)."""
        if self.compute_streams:
            compute = self.compute_streams[0 if self.single else self._next_stream]
            self._next_stream = (self._next_stream + 1) % len(self.compute_streams)
        else:
            compute = torch.cuda.current_stream(self.device)
        compute.wait_event(prepared.ready)
        with torch.cuda.stream(compute):
            if self.stagger_level0 and self.compute_streams and not self.single:
                profiling.PHASE_HOOK = self._phase_hook
                try:
                    out = fn(prepared.x, prepared.field)
                finally:
                    profiling.PHASE_HOOK = None
            else:
                out = fn(prepared.x, prepared.field)
        prepared.done = torch.cuda.Event()
        prepared.done.record(compute)
        self._retired.append(prepared)
        while self._retired and self._retired[0].done.query():
            self._retired.popleft()
        return out

    def _phase_hook(self, tag):
        """level-0 stages of consecutive frames take turns: a frame's first stride-1 decoder launch waits for the previous
        frame's last one (event wait on the device, nothing blocks on the host)"""
        st = torch.cuda.current_stream(self.device)
        if tag == "level0_begin":
            if self._level0_done is not None:
                st.wait_event(self._level0_done)
        elif tag == "level0_end":
            self._level0_done = torch.cuda.Event()
            self._level0_done.record(st)

    def drain(self):
        for st in self.compute_streams:
            st.synchronize()
        torch.cuda.current_stream(self.device).synchronize()
        self._retired.clear()
