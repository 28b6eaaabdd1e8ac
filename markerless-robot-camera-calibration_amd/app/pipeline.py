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
import time

import torch

from .. import MinkowskiEngine as ME


_STREAMS = {}


def shared_stream(device, role, index=0, priority=0):
    """One HIP stream per (device, role, index, priority) for the whole process.  HIP multiplexes streams onto a few hardware
    queues (GPU_MAX_HW_QUEUES: 8 here); every pipeline object used to create its own prep / compute / crop streams, and once
    a process had made more streams than queues, the compute streams of a NEW pipeline could share a queue - its frames then
    ran one after the other (engine stream 70.2 -> 62.6 frames/s for the third engine of a process, bench.py's engine block
    behind the other blocks).  Pipelines that are not used at the same time can share streams: stream order only adds
    dependencies.  Roles: "prep", "compute" (index = position in the rotation), "crop", "crop_side"."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device(), role, index, priority)
    st = _STREAMS.get(key)
    if st is None:
        st = _STREAMS[key] = torch.cuda.Stream(device=dev, priority=priority)
    return st


class PreparedFrame:
    __slots__ = ("field", "x", "ready", "done", "tag", "stream", "result", "sizes")

    def __init__(self, field, x, ready, tag=None):
        self.field, self.x, self.ready, self.done, self.tag = field, x, ready, None, tag
        self.sizes = None  # prepare_group: points per frame of the group
        self.stream = None  # compute stream run() put the frame on
        self.result = None


def build_unet_plans(cm, levels=4):
    """Everything a MinkUNet-shaped graph will ask the coordinate manager for (model/backbone/minkunet.py:125-183):
    3x3x3 maps at strides 1..2^levels, stride-2 down maps and transposed up maps between neighbouring levels, and the
    offset-range plans of the wide decoder layers on big levels - one sv_frame_plans call (the stride maps it needs come
    from TensorField.sparse(pyramid_levels=levels), or are built here one by one)."""
    cm.build_plans(levels)


class FramePipeline:
    def __init__(self, device, levels=4, encoder_only=False, compute_streams=1, stagger_level0=None, one_frame=False):
        self.device = torch.device(device)
        self.levels = levels
        self.encoder_only = encoder_only
        # one_frame: the caller runs ONE frame at a time and waits for it (InferenceEngine.predict per frame, the reference's
        # consumer loop app/main.py:432-456) - nothing else fills the GPU, so (i) the network starts as soon as the encoder's
        # plans exist and the offset-range plans of the decoder are built on the prep stream meanwhile, (ii) only level 0 runs
        # as offset-range passes (nn.SPLIT_RULES_ONE_FRAME) and (iii) the conv dispatch uses its one-launch-at-a-time
        # thresholds (sv_conv_set_dispatch(1.0): short tiles on the small levels, whose tails nobody would fill)
        self.one_frame = one_frame
        if one_frame:
            from .. import nn as svnn

            self.split_rules = svnn.SPLIT_RULES_ONE_FRAME
            self.want_scale = float(os.environ.get("MRCC_ONE_FRAME_WANT_SCALE", "1.0"))
        else:
            self.split_rules = None
            self.want_scale = None
        # the prep stream's ~150 small kernels per frame must not queue behind thousands of conv workgroups: the host
        # blocks on their size read-backs, and a late prepare() starves a compute stream (high priority = -1)
        prio = int(os.environ.get("MRCC_PREP_PRIORITY", "-1"))
        self.prep_stream = shared_stream(self.device, "prep", 0, prio)
        # compute_streams > 1: consecutive frames run on alternating streams, so the small-pyramid-level and thin-layer
        # kernels of one frame (which cannot fill 256 CUs) overlap the big convolutions of its neighbour
        self.compute_streams = [shared_stream(self.device, "compute", i) for i in range(compute_streams)] \
            if compute_streams > 1 else []
        self._next_stream = 0
        # The stride-1 decoder stages (level 0: 63 % of a frame's flops, chip-filling launches) of consecutive frames take
        # turns instead of overlapping each other; everything else of the other frames still runs underneath them.  Two
        # streams: 61.7 frames/s against 62.1 without, but the level-0 launches then run at a steady 0.61 of the matrix
        # peak each instead of 0.48-0.61 depending on how the two frames happen to be phased; three / four / five
        # streams: 64.2 / 63.4 / 63.8 frames/s (63.9-64.5 without at three) at 0.50 / 0.58 / 0.55 per launch.
        self.stagger_level0 = os.environ.get("MRCC_STAGGER_LEVEL0", "1") == "1" if stagger_level0 is None else bool(stagger_level0)
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
            x = field.sparse(pyramid_levels=self.levels)  # voxelise + the stride-2 maps: one host call, sizes read back once
            cm = x.coordinate_manager
            cm.split_rules = self.split_rules
            ready = torch.cuda.Event()
            if self.encoder_only:
                cm.build_plans(self.levels, up=False, split=False)
                ready.record(self.prep_stream)
            elif self.one_frame:
                cm.build_plans(self.levels, split=False)
                ready.record(self.prep_stream)
                # the decoder's offset-range plans: built while the compute stream already runs the encoder; the first layer
                # that uses one waits for this event (nn._ConvBase._plan)
                cm.build_plans(self.levels, k3=False, down=False, up=False, split=True)
                cm.split_ready = torch.cuda.Event()
                cm.split_ready.record(self.prep_stream)
            else:
                cm.build_plans(self.levels)
                ready.record(self.prep_stream)
        return PreparedFrame(field, x, ready, tag)

    def prepare_group(self, frames, tag=None):
        """Several resident frames as ONE sparse tensor - the reference's batched format (batch index in column 0,
        data/alivev2.py:358-383; its test loaders run TEST.batch_size frames per tensor, config/default.yaml:109).
        frames: [(coords4, feats), ...], each one frame (its own batch column is ignored); the group's batch column is the
        frame's position.  Frames of a batch never interact (batch index in the voxel key, BN in eval mode): every frame's
        rows carry the bits of its single-frame run (tests/test_gpu_engine.py), and the launches are len(frames) times
        longer - the ramp and decay of a chip-filling launch and the latency-bound small-level launches are paid once per
        group instead of once per frame.  The returned PreparedFrame's field / tensor cover the group, points and voxels
        in frame order; `.sizes` = points per frame."""
        if len(frames) == 1:
            prepared = self.prepare(frames[0][0], frames[0][1], tag)
            prepared.sizes = [int(frames[0][0].shape[0])]
            return prepared
        with torch.cuda.stream(self.prep_stream):
            coords = torch.cat([c for c, _ in frames])
            off = 0
            for b, (c, _) in enumerate(frames):
                coords[off:off + c.shape[0], 0] = float(b)
                off += c.shape[0]
            feats = torch.cat([f for _, f in frames])
        prepared = self.prepare(coords, feats, tag)
        prepared.sizes = [int(c.shape[0]) for c, _ in frames]
        return prepared

    def run(self, prepared, fn):
        """Run fn(x, field) on a compute stream once the frame is ready; returns fn's result (valid on that stream;
        `drain()` or an event wait orders it against other streams)."""
        if self.compute_streams:
            compute = self.compute_streams[0 if self.single else self._next_stream]
            self._next_stream = (self._next_stream + 1) % len(self.compute_streams)
        else:
            compute = torch.cuda.current_stream(self.device)
        compute.wait_event(prepared.ready)
        prepared.stream = compute
        # the hook travels with the frame (its coordinate manager), not in a module global: two pipelines or threads do not
        # see each other's hooks, and model code needs no knowledge of this class
        cm = prepared.x.coordinate_manager
        cm.phase_hook = self._phase_hook if (self.stagger_level0 and self.compute_streams and not self.single) else None
        with torch.cuda.stream(compute):
            try:
                if self.want_scale is not None:
                    from .. import _lib

                    with _lib.conv_dispatch(self.want_scale):
                        out = fn(prepared.x, prepared.field)
                else:
                    out = fn(prepared.x, prepared.field)
            finally:
                cm.phase_hook = None
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


class _HostSlot:
    """pinned staging buffers of one in-flight frame (host -> device inputs, device -> host labels)"""

    def __init__(self):
        self.cap = 0
        self.pts = self.rgb = self.labels = None
        self.uploaded = None  # event: the slot's H2D copies have finished (safe to refill)

    def reserve(self, n, channels):
        if self.rgb is None or n > self.cap or self.rgb.shape[1] != channels:
            # ONE pinned allocation per slot, carved into labels | points | colours: pinning is a driver call of milliseconds
            # (two orders of magnitude more when processes share a GPU), so there are few of them and none per frame
            self.cap = cap = max(n, int(self.cap * 1.25), 1)
            buf = torch.empty(cap * (8 + 12 + 4 * channels), dtype=torch.uint8).pin_memory()
            self.labels = buf[:8 * cap].view(torch.int64)
            self.pts = buf[8 * cap:20 * cap].view(torch.float32).view(cap, 3)
            self.rgb = buf[20 * cap:].view(torch.float32).view(cap, channels)


class HostFrameStream:
    """Host arrays in -> per-point labels out, frames overlapped (the reference's consumer is a per-frame loop over
    `InferenceEngine.predict(data)`, app/main.py:432-456; one synchronous frame at a time leaves the GPU idle during
    staging, H2D, the coordinate read-backs and D2H: 33 ms per 200k-point frame against 16 ms of kernels).

    Per frame: points / colours are copied into PINNED host buffers and uploaded asynchronously on the prep stream, the
    scaled (batch, x, y, z) rows are formed on the device (`points * scale` in float32 - the same IEEE product the
    reference computes on the host, app/inference_engine.py:405-409), FramePipeline.prepare() builds the coordinate
    maps / plans there, `stage(x, field)` (the network + slice/argmax) runs on the next compute stream, and `finish`
    (whatever needs host decisions: the largest-cluster rule with its count read-back) plus the D2H copy of the labels
    into pinned memory run on that frame's stream while LATER frames compute.  Results come back in input order and are
    bit-identical to the synchronous path: the same kernels run on the same data, only their interleaving changes."""

    def __init__(self, device, scale, stage, finish=None, levels=4, compute_streams=3, depth=None, stagger_level0=None,
                 one_frame=False, group=1):
        self.device = torch.device(device)
        self.scale = scale
        self.stage, self.finish = stage, finish
        self.pipe = FramePipeline(self.device, levels=levels, compute_streams=compute_streams, stagger_level0=stagger_level0,
                                  one_frame=one_frame)
        self.depth = depth or max(2, compute_streams)
        # group > 1 (run() only): that many consecutive frames go through `stage` as ONE sparse tensor (batch column =
        # position in the group, as FramePipeline.prepare_group; frames never interact, so every frame's labels are the bits
        # of its own pass) - launches `group` times longer, results delivered a group at a time; `finish` and the download
        # stay per frame
        self.group = max(1, int(group))
        self._slots = [_HostSlot() for _ in range((self.depth + 2) * self.group)]
        self._n = 0
        # host wall time per phase, summed over frames (perf_counter deltas; tools/engine_stream_phases.py prints them)
        self.host_s = {"stage": 0.0, "prepare": 0.0, "launch": 0.0, "finalize": 0.0, "frames": 0}

    def preallocate(self, n, channels=3):
        """Pin the staging buffers of every slot for frames of up to n points now (a warm-up step: otherwise each slot pins
        its buffer at its first use, inside whatever is being timed)."""
        for slot in self._slots:
            slot.reserve(int(n), channels)

    def _upload_and_prepare(self, points, rgb):
        import numpy as np

        t0 = time.perf_counter()
        slot = self._slots[self._n % len(self._slots)]
        self._n += 1
        if slot.uploaded is not None:
            slot.uploaded.synchronize()
        n = len(points)
        rgb_t = rgb if torch.is_tensor(rgb) else None
        channels = (rgb_t.shape[1] if rgb_t is not None else np.asarray(rgb).shape[1])
        slot.reserve(n, channels)
        prep = self.pipe.prep_stream
        np.copyto(slot.pts.numpy()[:n], np.asarray(points), casting="same_kind")  # float64 sources are rounded here
        with torch.cuda.stream(prep):  # the points are on their way while the colours are staged
            d_pts = slot.pts[:n].to(self.device, non_blocking=True)
            coords4 = torch.zeros((n, 4), dtype=torch.float32, device=self.device)
            torch.mul(d_pts, self.scale, out=coords4[:, 1:])
        if rgb_t is not None:
            slot.rgb[:n].copy_(rgb_t)
        else:
            np.copyto(slot.rgb.numpy()[:n], np.asarray(rgb), casting="same_kind")
        t1 = time.perf_counter()
        with torch.cuda.stream(prep):
            d_rgb = slot.rgb[:n].to(self.device, non_blocking=True)
            slot.uploaded = torch.cuda.Event()
            slot.uploaded.record(prep)
        out = self.pipe.prepare(coords4, d_rgb, tag=(slot, d_pts, n))
        t2 = time.perf_counter()
        self.host_s["stage"] += t1 - t0
        self.host_s["prepare"] += t2 - t1
        self.host_s["frames"] += 1
        return out

    def _upload_and_prepare_group(self, members):
        """members: [(points, rgb), ...] host arrays of consecutive frames -> one PreparedFrame over all of them; its tag is
        the list of (slot, device points, n) per frame."""
        import numpy as np

        if len(members) == 1:
            out = self._upload_and_prepare(*members[0])
            out.tag = [out.tag]
            return out
        t0 = time.perf_counter()
        prep = self.pipe.prep_stream
        sizes = [len(p) for p, _ in members]
        first_rgb = members[0][1]
        channels = first_rgb.shape[1] if torch.is_tensor(first_rgb) else np.asarray(first_rgb).shape[1]
        with torch.cuda.stream(prep):
            coords4 = torch.empty((sum(sizes), 4), dtype=torch.float32, device=self.device)
            feats = torch.empty((sum(sizes), channels), dtype=torch.float32, device=self.device)
        tags, off, t_stage = [], 0, 0.0
        for b, (points, rgb) in enumerate(members):
            slot = self._slots[self._n % len(self._slots)]
            self._n += 1
            if slot.uploaded is not None:
                slot.uploaded.synchronize()
            n = sizes[b]
            slot.reserve(n, channels)
            np.copyto(slot.pts.numpy()[:n], np.asarray(points), casting="same_kind")
            with torch.cuda.stream(prep):  # the points are on their way while the colours are staged
                d_pts = slot.pts[:n].to(self.device, non_blocking=True)
                rows = coords4[off:off + n]
                rows[:, 0] = float(b)
                torch.mul(d_pts, self.scale, out=rows[:, 1:])
            if torch.is_tensor(rgb):
                slot.rgb[:n].copy_(rgb)
            else:
                np.copyto(slot.rgb.numpy()[:n], np.asarray(rgb), casting="same_kind")
            with torch.cuda.stream(prep):
                feats[off:off + n].copy_(slot.rgb[:n], non_blocking=True)
                slot.uploaded = torch.cuda.Event()
                slot.uploaded.record(prep)
            tags.append((slot, d_pts, n))
            off += n
        t1 = time.perf_counter()
        out = self.pipe.prepare(coords4, feats, tag=tags)
        out.sizes = sizes
        t2 = time.perf_counter()
        self.host_s["stage"] += t1 - t0
        self.host_s["prepare"] += t2 - t1
        self.host_s["frames"] += len(members)
        return out

    def _finalize_group(self, prepared):
        """per frame of the group: `finish` on its slice of the labels, download; one wait for the whole group"""
        import numpy as np

        t0 = time.perf_counter()
        stream = prepared.stream if prepared.stream is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.stream(stream):
            off = 0
            for slot, d_pts, n in prepared.tag:
                label = prepared.result[off:off + n]
                if self.finish is not None:
                    label = self.finish(label, d_pts)
                slot.labels[:n].copy_(label, non_blocking=True)
                off += n
            done = torch.cuda.Event()
            done.record(stream)
        done.synchronize()
        outs = [np.array(slot.labels.numpy()[:n]) for slot, _, n in prepared.tag]
        prepared.result = prepared.tag = None
        self.host_s["finalize"] += time.perf_counter() - t0
        return outs

    def _launch(self, prepared):
        t0 = time.perf_counter()
        prepared.result = self.pipe.run(prepared, self.stage)
        self.host_s["launch"] += time.perf_counter() - t0

    def _finalize(self, prepared):
        import numpy as np

        t0 = time.perf_counter()
        slot, d_pts, n = prepared.tag
        stream = prepared.stream if prepared.stream is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.stream(stream):
            label = prepared.result
            if self.finish is not None:
                label = self.finish(label, d_pts)
            slot.labels[:n].copy_(label, non_blocking=True)
            done = torch.cuda.Event()
            done.record(stream)
        done.synchronize()
        out = np.array(slot.labels.numpy()[:n])  # leave the pinned buffer free for the next frame
        prepared.result = prepared.tag = None
        self.host_s["finalize"] += time.perf_counter() - t0
        return out

    def run_one(self, points, rgb):
        """One frame, start to finish (the per-frame call of the reference's loop): staged, uploaded and prepared on the prep
        stream, the network on the compute stream behind it, labels downloaded - nothing is enqueued between a size
        read-back and the kernels it waits for except this frame's own coordinate work."""
        with torch.no_grad():
            cur = self._upload_and_prepare(points, rgb)
            self._launch(cur)
            return self._finalize(cur)

    def run(self, frames):
        """frames: iterable of (points [N,3], rgb [N,C]) host arrays -> generator of int64 label arrays, in order."""
        pending = collections.deque()
        it = iter(frames)
        if self.group > 1:
            import itertools

            def take():
                members = list(itertools.islice(it, self.group))
                return self._upload_and_prepare_group(members) if members else None

            with torch.no_grad():
                nxt = take()
                while nxt is not None:
                    cur = nxt
                    self._launch(cur)
                    pending.append(cur)
                    nxt = take()
                    while len(pending) >= self.depth:
                        yield from self._finalize_group(pending.popleft())
                while pending:
                    yield from self._finalize_group(pending.popleft())
                self.pipe.drain()
            return
        try:
            first = next(it)
        except StopIteration:
            return
        with torch.no_grad():
            nxt = self._upload_and_prepare(*first)
            while nxt is not None:
                cur = nxt
                self._launch(cur)  # asynchronous: the GPU works on `cur` while the host stages the next frame
                pending.append(cur)
                try:
                    nxt = self._upload_and_prepare(*next(it))
                except StopIteration:
                    nxt = None
                while len(pending) >= self.depth:
                    yield self._finalize(pending.popleft())
            while pending:
                yield self._finalize(pending.popleft())
            self.pipe.drain()


class PinnedRing:
    """A few pinned host buffers handed out in turn; a slot is reused once the event recorded for its last transfer has
    completed (pinning memory costs hundreds of microseconds: the buffers are kept and grown, never freed per frame)."""

    def __init__(self, slots=6):
        self.bufs = [None] * slots
        self.events = [None] * slots
        self.i = 0

    def take(self, nbytes):
        i = self.i
        self.i = (i + 1) % len(self.bufs)
        if self.events[i] is not None:
            self.events[i].synchronize()
            self.events[i] = None
        buf = self.bufs[i]
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes * 1.25), 1 << 16), dtype=torch.uint8).pin_memory()
            self.bufs[i] = buf
        return i, buf[:nbytes]

    def busy_until(self, i, event):
        self.events[i] = event


class CropBatchRunner:
    """One sparse network over the end-effector crops of several frames as ONE sparse tensor - batch column = crop, the
    training-format batch of data/alivev2.py:358-383, whose frames never interact (every crop's rows have the bits of
    running it alone).  The reference runs its pose networks once per frame (app/inference_engine.py:304-319), each a
    U-Net of ~50-100 launches on a few thousand voxels: launch-bound.  Batched over the frames in flight, the launches, the
    H2D copy, the coordinate work (sv_frame_maps / sv_frame_plans) and the result download are paid once per group, on
    this runner's own stream, beside the segmentation networks of the following frames."""

    def __init__(self, device, levels=4):
        self.device = torch.device(device)
        self.levels = levels
        # high priority, as the prep stream: the crops' few hundred short workgroups must not queue behind the thousands of
        # convolution workgroups of the segmentation frames in flight (the host waits for this stream's size read-backs)
        self.stream = shared_stream(self.device, "crop", 0, int(os.environ.get("MRCC_PREP_PRIORITY", "-1")))
        # two networks that read the same voxelised batch are independent of each other: the second one runs here
        self.side = shared_stream(self.device, "crop_side", 0, int(os.environ.get("MRCC_PREP_PRIORITY", "-1")))
        self.up = PinnedRing()
        self.down = PinnedRing(slots=12)

    def run(self, pts_list, feat_list, scale, fn, encoder_only=False, one_frame=False):
        """pts_list[i] [n_i, 3] (already preprocessed; float64 sources are rounded to float32 first, then multiplied by
        `scale` in float32 - the product the reference forms on the host, app/inference_engine.py:446-450), feat_list[i]
        [n_i, C].  fn(x, field, seg_start) runs on this runner's stream with x = the voxelised batch; seg_start[i] = first
        point of crop i.  Returns fn's result (valid on self.stream).  Several networks that take the SAME voxelisation
        (same preprocessing, scale and features) are run by one fn on one x: the maps and plans are built once."""
        import contextlib

        import numpy as np

        from .. import _lib

        G = len(pts_list)
        sizes = [len(p) for p in pts_list]
        n = int(sum(sizes))
        C = int(np.asarray(feat_list[0]).shape[1]) if not torch.is_tensor(feat_list[0]) else int(feat_list[0].shape[1])
        slot, buf = self.up.take(n * (4 + C) * 4)
        host = buf.view(torch.float32).numpy()
        hc, hf = host[: 4 * n].reshape(n, 4), host[4 * n:].reshape(n, C)
        seg_start = [0]
        for b in range(G):
            o, m = seg_start[-1], sizes[b]
            hc[o:o + m, 0] = b
            np.multiply(np.asarray(pts_list[b], dtype=np.float32), np.float32(scale), out=hc[o:o + m, 1:])
            f = feat_list[b]
            hf[o:o + m] = f.detach().cpu().numpy() if torch.is_tensor(f) else np.asarray(f, dtype=np.float32)
            seg_start.append(o + m)
        with torch.cuda.stream(self.stream), torch.no_grad():
            dev = buf.to(self.device, non_blocking=True).view(torch.float32)
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self.up.busy_until(slot, ev)
            field = ME.TensorField(features=dev[4 * n:].view(n, C), coordinates=dev[: 4 * n].view(n, 4),
                                   quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE,
                                   minkowski_algorithm=ME.MinkowskiAlgorithm.SPEED_OPTIMIZED, device=self.device)
            x = field.sparse(pyramid_levels=self.levels)
            cm = x.coordinate_manager
            cm.num_batches = G  # known here: no read-back of the largest batch index
            cm.build_plans(self.levels, up=not encoder_only, split=not encoder_only)
            with (_lib.conv_dispatch(1.0) if one_frame else contextlib.nullcontext()):
                return fn(x, field, seg_start)

    def fork(self, fn):
        """fn() on the side stream, behind everything enqueued on the main stream so far; the main stream waits for it at
        join().  Launch-bound networks on a few thousand voxels leave most of the GPU idle: two of them side by side take
        about as long as the longer one.  Returns fn's result (tensors: valid on the main stream after join)."""
        ev = torch.cuda.Event()
        ev.record(self.stream)
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            out = fn()
            self._joined = torch.cuda.Event()
            self._joined.record(self.side)
        for t in (out if isinstance(out, (tuple, list)) else (out,)):
            if torch.is_tensor(t):
                t.record_stream(self.stream)
        return out

    def join(self):
        ev = getattr(self, "_joined", None)
        if ev is not None:
            self.stream.wait_event(ev)
            self._joined = None

    def download(self, tensors):
        """device tensors -> (list of pinned host views, event): one asynchronous D2H each on the runner's stream"""
        outs = []
        with torch.cuda.stream(self.stream):
            for t in tensors:
                t = t.contiguous()
                slot, buf = self.down.take(t.numel() * t.element_size())
                h = buf.view(t.dtype).view(t.shape)
                h.copy_(t, non_blocking=True)
                outs.append((slot, h, t))  # the device tensor stays referenced until the copy has been waited for
            ev = torch.cuda.Event()
            ev.record(self.stream)
        for slot, _, _ in outs:
            self.down.busy_until(slot, ev)
        return [h for _, h, _ in outs], ev, [t for _, _, t in outs]
