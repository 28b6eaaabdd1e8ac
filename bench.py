#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json: point-cloud frames/sec at 200k pts/frame on N MI355X.

A step = one pass of the hot path over one synthetic frame that is already resident in HBM:
    voxelise (floor, sort, unique, mean) -> coordinate maps + kernel maps + conv plans -> RobotNetSegmentation
    (MinkUNet18D, fp32 MFMA, fused BN/ReLU/residual) -> fused slice + argmax -> per-point labels.
Every step sees a different frame object; nothing (coordinate maps, plans, outputs) is cached across steps.

N > 1: frames shard across ranks (one process per GPU, launched by torch.distributed.run); there is no data-path
collective — one RCCL all_gather of a small metrics record at the end.  value = frames processed by
all ranks / max-over-ranks wall time.

Also in the JSON line: `roofline` for the dominant kernel (live HIP-event timing of every sv_conv_fwd launch in the
timed region) and, at N = 1, `cpu_baseline` = the C oracle (oracle/sv_oracle.c, OpenMP) timed on the host cores.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _spawn_ranks_if_needed():
    """`python bench.py --gpus N` with N > 1 and no torch.distributed environment: start N fresh ranks (one process per
    GPU) with torch.distributed.run and relay rank 0's JSON line.  This runs BEFORE torch / HIP are imported, so the
    parent never touches the GPU and nothing is re-exec'ed; the parent exits with the children's return code."""
    if __name__ != "__main__" or "WORLD_SIZE" in os.environ:
        return
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--gpus", type=int, default=1)
    known, _ = pre.parse_known_args()
    if known.gpus <= 1:
        return
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, len(os.sched_getaffinity(0)) // known.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(known.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for ln in res.stdout.splitlines():  # stdout carries the JSON line only; anything else a rank printed goes to stderr
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    if res.returncode == 0 and not any(l.startswith("{") for l in res.stdout.splitlines()):
        sys.stderr.write("bench.py launcher: the ranks exited 0 without printing the JSON line\n")
        sys.exit(1)
    sys.exit(res.returncode)


_spawn_ranks_if_needed()

import numpy as np  # noqa: E402
import torch  # noqa: E402

# HIP multiplexes streams onto a few hardware queues (4 by default); the pipeline uses a prep stream plus up to five compute
# streams besides torch's default one, and compute streams sharing a queue would serialise.  Must be set before HIP
# initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# idle OpenMP workers sleep instead of spinning: the oracle's team (cpu_baseline / accuracy legs) and torch's CPU pool
# are both as wide as the host
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import mrcc_amd  # noqa: E402
from mrcc_amd import MinkowskiEngine as ME  # noqa: E402
from mrcc_amd import profiling  # noqa: E402

_T0 = time.perf_counter()


def _log(msg):
    """progress line on stderr (stdout carries only the JSON line)"""
    sys.stderr.write(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}\n")
    sys.stderr.flush()


POINTS = 200_000
ROOM = 2.4
SCALE = 50  # 2 cm voxels
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32 matrix peak
PEAK_HBM_GBS = 8000.0


def build_model(device):
    from mrcc_amd.model.robotnet_segmentation import RobotNetSegmentation

    torch.manual_seed(1)
    model = RobotNetSegmentation(in_channels=3, num_classes=3)
    # random-init weights (no checkpoints ship); BN statistics are randomised too so that the eval-mode
    # network is not the identity-normalised special case and the label histogram is not degenerate
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
    return model.to(device).eval()


def make_frame(seed, device, n=POINTS, L=ROOM, batch=1):
    """One step's input: `batch` frames of n points in the reference's batched format (batch index in column 0,
    data/alivev2.py:358-383); batch = 1 is the headline workload."""
    parts = [mrcc_amd.synth.gen_room(n, L, seed * batch + b) for b in range(batch)]
    coords4 = np.concatenate([np.concatenate([np.full((n, 1), b, np.float32), p[0] * np.float32(SCALE)], axis=1)
                              for b, p in enumerate(parts)])
    rgb = np.concatenate([p[1] for p in parts])
    return (torch.from_numpy(coords4).to(device), torch.from_numpy(rgb).to(device), parts[0][0], parts[0][1],
            parts[0][2])


def run_frames(model, pipe, frames, steps, hist=None, group=1):
    """Process `steps` frames through the multi-stream pipeline (mrcc_amd/app/pipeline.py): while the U-Net of frame i
    runs on the compute stream, the host builds frame i+1's coordinate maps / plans on the prep stream.  Every frame
    is voxelised and mapped from scratch; all work of all `steps` frames is enqueued (and, by the caller's
    synchronize, finished) inside the caller's timed region.  group > 1: consecutive frames go through the network
    `group` at a time as one sparse tensor (FramePipeline.prepare_group: the concatenation is part of the timed work);
    the last group of a region holds the remainder."""
    voxels = 0
    cls = torch.arange(3, device=frames[0][1].device).unsqueeze(1)

    def unet(x, field):
        if profiling.TIMER is not None:
            profiling.TIMER.frame_boundary()
        out = model(x)
        label, conf = out.slice_argmax(field)
        if hist is not None:  # asynchronous label histogram on the frame's own stream
            return (label.unsqueeze(0) == cls).sum(dim=1)
        return label

    def members(i):  # the frames of steps [i, i + group)
        return [frames[(i + j) % len(frames)][:2] for j in range(min(group, steps - i))]

    partial = []
    starts = list(range(0, steps, group))
    nxt = pipe.prepare_group(members(0)) if starts else None
    for gi, i in enumerate(starts):
        cur = nxt
        res = pipe.run(cur, unet)
        if hist is not None:
            partial.append(res)
        voxels += cur.x.F.shape[0]
        if gi + 1 < len(starts):
            nxt = pipe.prepare_group(members(starts[gi + 1]))
    if hist is not None:
        pipe.drain()  # per-frame histograms live on their frames' streams
        hist += torch.stack(partial).sum(dim=0)
    return voxels


def oracle_pass(model, budget_s=25.0, world=1):
    """Run the oracle (C restatement, OpenMP over output rows) on the host cores, timed: a quarter-size frame first,
    the full 200k-point seed-0 frame only if the estimate says it fits the budget.  Returns (cpu_baseline record,
    checked frame = dict(pts, rgb, lab, ref) of the LARGEST frame the oracle labelled)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sv_oracle as O

    cores = len(os.sched_getaffinity(0))
    # explicit thread count: torch.distributed.run exports OMP_NUM_THREADS=1 to every rank, and the other ranks of this
    # host are parked in the final barrier while rank 0 checks the labels
    O.NUM_THREADS = max(1, min(cores, 128) // (1 if world == 1 else 2))
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    pts, rgb, lab = mrcc_amd.synth.gen_room(POINTS // 4, ROOM / 2, 0)
    _log(f"cpu baseline: oracle on a quarter-size frame, {O.NUM_THREADS} host threads")
    t0 = time.perf_counter()
    r = O.predict_segmentation(sd, pts, rgb, SCALE)
    t_small = time.perf_counter() - t0
    v_small = len(r["vox"]["keys"])
    _log(f"cpu baseline: quarter frame took {t_small:.2f} s; gather-GEMM (torch.mm) formulation next")
    checked = dict(pts=pts, rgb=rgb, lab=lab, ref=r, what=f"quarter-size frame ({POINTS // 4} pts, {v_small} voxels)")
    note = ("scalar order-preserving fmaf chain per output row (the bit-exact oracle, AVX2 across output channels, "
            "OpenMP over rows) - NOT a BLAS gather-GEMM; see cpu_baseline_gather_gemm for that")
    est_full = t_small * 4.0
    mm = gather_gemm_baseline(sd, r["label"]) if world == 1 else None
    if est_full <= budget_s:
        pts, rgb, lab = mrcc_amd.synth.gen_room(POINTS, ROOM, 0)
        _log("cpu baseline: oracle on the full 200k-point frame")
        t0 = time.perf_counter()
        r = O.predict_segmentation(sd, pts, rgb, SCALE)
        t_full = time.perf_counter() - t0
        checked = dict(pts=pts, rgb=rgb, lab=lab, ref=r, what=f"full frame ({POINTS} pts, seed 0)")
        base = {"value": 1.0 / t_full, "unit": "frames/s", "cores": O.NUM_THREADS, "kind": "port",
                "sample": f"1 full frame: {POINTS} pts, {len(r['vox']['keys'])} voxels, {t_full:.2f} s "
                          f"(C oracle, OpenMP, fp32 fmaf chain; quarter frame took {t_small:.2f} s)", "note": note}
    else:
        base = {"value": 1.0 / est_full, "unit": "frames/s", "cores": O.NUM_THREADS, "kind": "port",
                "sample": f"quarter-size frame ({POINTS // 4} pts, L={ROOM / 2} m, {v_small} voxels) took "
                          f"{t_small:.2f} s; value = 1 / (4 x that): work is linear in voxels", "note": note}
    return base, mm, checked


def gather_gemm_baseline(sd, chain_labels, timeout_s=150):
    """Second CPU baseline, as SURVEY.md 8(d) describes the reference's CPU path: per kernel offset gather rows, torch.mm
    (sgemm on the host cores), scatter-add - on the quarter frame.  Runs in a CHILD process with its own thread pools
    : inside this process the oracle's OpenMP team and torch's intra-op pool, both as wide as
    the host, spin against each other.  Bounded by timeout_s; a failure is reported, never fatal."""
    import subprocess
    import tempfile

    cores = len(os.sched_getaffinity(0))
    threads = min(cores, 64)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "sd.pt")
        torch.save(sd, path)
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env.pop("OMP_WAIT_POLICY", None)  # one pool at a time in the child: default (spin-then-sleep) waits
        env.update(OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
        try:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "--gather-gemm-child", path], env=env,
                                 capture_output=True, text=True, timeout=timeout_s)
            rec = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
        except (subprocess.TimeoutExpired, IndexError, ValueError) as e:
            _log(f"cpu baseline: gather-GEMM child failed: {type(e).__name__}")
            return {"value": None, "unit": "frames/s", "kind": "port", "note": f"not measured: {type(e).__name__}"}
    agree = rec.pop("label_histogram")
    _log(f"cpu baseline: gather-GEMM quarter frame took {rec['seconds']:.2f} s on {threads} threads")
    rec["note"] = ("same graph with every conv as gather -> torch.mm -> scatter-add per kernel offset (what "
                   "MinkowskiEngine's CPU path does); sgemm reassociates sums, so it is a timing baseline, not the "
                   f"parity oracle; label histogram {agree} vs the chain oracle's "
                   f"{np.bincount(chain_labels, minlength=3).tolist()}")
    return rec


def gather_gemm_child(sd_path):
    """--gather-gemm-child: time the gather-GEMM formulation on the quarter frame; never touches the GPU."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sv_oracle as O

    sd = torch.load(sd_path)
    pts, rgb, _ = mrcc_amd.synth.gen_room(POINTS // 4, ROOM / 2, 0)
    t0 = time.perf_counter()
    with _gather_gemm_conv(O):
        r = O.predict_segmentation(sd, pts, rgb, SCALE)
    t_mm = time.perf_counter() - t0
    print(json.dumps({"value": 1.0 / (4.0 * t_mm), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                      "seconds": t_mm, "label_histogram": np.bincount(r["label"], minlength=3).tolist(),
                      "sample": f"quarter-size frame ({POINTS // 4} pts, L={ROOM / 2} m, {len(r['vox']['keys'])} voxels) "
                                f"took {t_mm:.2f} s; value = 1 / (4 x that)"}), flush=True)


class _gather_gemm_conv:
    """Context manager: swap the oracle's conv for a gather -> torch.mm -> scatter-add formulation (timing only)."""

    def __init__(self, O):
        self.O = O

    def __enter__(self):
        O = self.O
        self.orig = O.conv

        def conv_mm(feats, W, nbr, V_out, scale=None, shift=None, residual=None, act=O.ACT_NONE, slope=0.01,
                    nthreads=None):
            x = torch.from_numpy(np.ascontiguousarray(feats, dtype=np.float32))
            Wt = torch.from_numpy(np.ascontiguousarray(W if W.ndim == 3 else W[None], dtype=np.float32))
            K, Cin, Cout = Wt.shape
            if nbr is None:
                acc = x @ Wt[0]
            else:
                acc = torch.zeros((V_out, Cout), dtype=torch.float32)
                nb = torch.from_numpy(np.ascontiguousarray(nbr)).long()
                for k in range(K):
                    rows = torch.nonzero(nb[k] >= 0).squeeze(1)
                    if rows.numel():
                        acc.index_add_(0, rows, x[nb[k][rows]] @ Wt[k])
            if scale is not None:
                acc = acc * torch.from_numpy(np.asarray(scale, np.float32)) + torch.from_numpy(
                    np.asarray(shift, np.float32).reshape(-1))
            elif shift is not None:
                acc = acc + torch.from_numpy(np.asarray(shift, np.float32).reshape(-1))
            if residual is not None:
                acc = acc + torch.from_numpy(np.ascontiguousarray(residual, dtype=np.float32))
            if act == O.ACT_RELU:
                acc = torch.relu(acc)
            elif act == O.ACT_LEAKY:
                acc = torch.nn.functional.leaky_relu(acc, slope)
            return acc.numpy()

        O.conv = conv_mm
        return self

    def __exit__(self, *exc):
        self.O.conv = self.orig


def accuracy_block(model, device, checked):
    """The "seg mIoU + pose ADD vs ref" half of the metric (SURVEY.md 8(d) parity gates): the GPU path and the oracle on
    the SAME frame and weights - labels equal, mIoU(GPU vs oracle labels), mIoU / reference accuracy against the
    synthetic ground truth for both (random-init weights: the absolute numbers only say both sides agree) - and 64
    Kabsch problems: ADD between the GPU pose and the oracle pose over the end-effector points (utils/metrics.py:139-150)."""
    import sv_oracle as O
    from mrcc_amd.utils import metrics as M
    from mrcc_amd.utils import transformation as T

    pts, rgb, lab, ref = checked["pts"], checked["rgb"], checked["lab"], checked["ref"]
    coords4 = np.concatenate([np.zeros((len(pts), 1), np.float32), pts * np.float32(SCALE)], axis=1)
    with torch.no_grad():
        field = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(coords4),
                               quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE, device=device)
        x = field.sparse()
        out = model(x)
        label, _ = out.slice_argmax(field)
    keys_equal = np.array_equal(x.coordinate_map.keys.cpu().numpy().view(np.uint64), ref["vox"]["keys"])
    g = label.cpu().numpy()
    o = ref["label"]
    logits = out.F.cpu().numpy()
    m_go = M.segmentation_metrics_from_confusion(M.confusion_matrix(g, o, 3))
    m_g = M.segmentation_metrics_from_confusion(M.confusion_matrix(g, lab, 3))
    m_o = M.segmentation_metrics_from_confusion(M.confusion_matrix(o, lab, 3))
    B = 64
    crops = [mrcc_amd.synth.gen_ee_crop(s, n=512) for s in range(B)]
    kp_ref = np.repeat(mrcc_amd.synth.REFERENCE_KEY_POINTS[None], B, axis=0)
    kp_tgt = np.stack([c[3] for c in crops])
    R, t, q = T.get_rigid_transform_3D_batched(kp_ref, kp_tgt, device=device)
    add_go, add_gt, dR = [], [], 0.0
    for b in range(B):
        Ro, to = O.get_rigid_transform_3D(kp_ref[b], kp_tgt[b])
        qo = O.get_q_from_matrix(Ro)
        qg = q[b] if np.dot(q[b], qo) >= 0 else -q[b]
        local = (crops[b][0].astype(np.float64) - crops[b][2][:3]) @ mrcc_amd.synth.quat_to_matrix(crops[b][2][3:])
        add_go.append(O.compute_ADD_np(local, np.concatenate([to, qo]), np.concatenate([t[b], qg])))
        add_gt.append(O.compute_ADD_np(local, crops[b][2], np.concatenate([t[b], qg])))
        dR = max(dR, float(np.abs(R[b] - Ro).max()))
    return {
        "frame": checked["what"], "voxels": int(logits.shape[0]),
        "voxel_keys_equal": bool(keys_equal),
        "logits_bit_exact": bool(np.array_equal(logits, ref["logits"])),
        "labels_equal": bool(np.array_equal(g, o)),
        "miou_gpu_vs_oracle": float(m_go["miou"]),
        "miou_vs_gt_gpu": float(m_g["miou"]), "miou_vs_gt_oracle": float(m_o["miou"]),
        "accuracy_vs_gt_gpu": float(m_g["accuracy"]), "accuracy_vs_gt_oracle": float(m_o["accuracy"]),
        "kabsch_problems": B, "add_gpu_vs_oracle_max_m": float(max(add_go)),
        "add_vs_gt_pose_mean_m": float(np.mean(add_gt)), "kabsch_max_abs_dR": dR,
        "note": "weights are random-init (no checkpoints ship), so mIoU/accuracy against the synthetic ground truth are "
                "chance-level by construction; the parity gates are labels_equal / miou_gpu_vs_oracle = 1.0 and "
                "add_gpu_vs_oracle <= 1e-4 m",
    }


def hbm_bound_layers(model, device, iters=60):
    """The gather-bound layers of the network, each timed as `iters` back-to-back launches (HIP events on the launch stream):
    achieved GB/s = SURVEY.md 8(d) algorithmic gather-bytes / time, against the 8 TB/s HBM peak.  north_star target:
    >= 40 % on the sparse-conv gather at 80k active voxels.  Two ways of issuing the series: `us_per_launch` = a hipGraph
    replay of the series (device-side back-to-back: what the kernels cost the GPU) and `us_per_launch_eager` = one host call
    per launch (a one-row launch issued that way already takes `eager_launch_floor_us`: the host's issue rate).  Each layer
    on the seed-0 frame (88k / 26k voxels) and on four frames in one tensor (352k / 106k voxels): at frame size the
    launches are one round of waves whose duration is a single wave's chain of dependent round trips, not bytes
    (profiles/r03_launch_floor.txt: the same kernels on a 16th of the frame take as long)."""
    from mrcc_amd import nn as svnn

    out = {}

    def series(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        eager = e0.elapsed_time(e1) / iters * 1e3
        graph_us = None
        try:
            st = torch.cuda.Stream(device=device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(st):
                with torch.cuda.graph(g, stream=st):
                    for _ in range(iters):
                        fn()
                g.replay()
                st.synchronize()
                e0.record(st)
                g.replay()
                e1.record(st)
                st.synchronize()
            graph_us = e0.elapsed_time(e1) / iters * 1e3
        except Exception as exc:  # capture is an optimisation of the measurement, never a requirement
            _log(f"hbm_bound_layers: graph capture unavailable ({type(exc).__name__}); eager series only")
        return eager, graph_us

    # every layer here runs ALONE on the GPU: the dispatch of a GPU that holds one frame (sv_conv_set_dispatch(1.0), what the
    # per-frame InferenceEngine.predict path uses) - for the thin 32 -> 32 layers that is the LDS-weights kernel
    with torch.no_grad(), mrcc_amd._lib.conv_dispatch(1.0):
        tiny = torch.zeros(1, 32, device=device)
        tiny_out = torch.empty_like(tiny)
        w1 = torch.zeros(1, 32, 32, device=device)
        floor, floor_graph = series(lambda: svnn.conv_forward(tiny, w1, None, 1, out=tiny_out))
        out["eager_launch_floor_us"] = round(floor, 2)
        out["graph_launch_floor_us"] = None if floor_graph is None else round(floor_graph, 2)
        for batch, tag in ((1, ""), (4, " x4 frames")):
            frame = make_frame(0, device, batch=batch)
            field = ME.TensorField(frame[1], frame[0], quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE,
                                   device=device)
            x = field.sparse()
            cm = x.coordinate_manager
            V0, V1 = cm.stride_map(1).V, cm.stride_map(2).V
            p0, p1 = cm.plan_k3(1), cm.plan_k3(2)
            blk = model.block1[0]
            s0, b0 = model.bn0.folded()
            s1, b1 = blk.norm1.folded()
            f1 = torch.randn(V1, 32, device=device)
            f0 = torch.randn(V0, 32, device=device)
            f1024 = torch.randn(V0, 1024, device=device)
            head = model.regression[2]
            cases = {
                "conv0 3->32 k27 (level 0)": (x.F, model.conv0p1s1.weight3().detach(), p0, V0, s0, b0),
                "block1 32->32 k27 (level 1)": (f1, blk.conv1.weight3().detach(), p1, V1, s1, b1),
                # the same layer shape on the level-0 map: the north_star's "gather at 80k active voxels"
                "32->32 k27 on the level-0 map": (f0, blk.conv1.weight3().detach(), p0, V0, s1, b1),
                "regression.2 1024->3 (level 0)": (f1024, head.weight3(), None, V0, None, head.linear.bias.detach()),
            }
            for name, (f, w, plan, V, sc, sh) in cases.items():
                K, Cin, Cout = w.shape
                dst = torch.empty(V, Cout, device=device)
                eager, graph_us = series(lambda: svnn.conv_forward(f, w, plan, V, sc, sh, None, 1, out=dst))
                us = graph_us if graph_us is not None else eager
                P = plan.num_pairs() if plan is not None else V
                gb = (P * (4.0 * Cin + 8) + 4.0 * V * Cout + 4.0 * K * Cin * Cout) / 1e9
                out[name + tag] = {"kernel": mrcc_amd._lib.conv_last_instance()[0], "rows": int(V),
                                   "us_per_launch": round(us, 2), "us_per_launch_eager": round(eager, 2),
                                   "timing": "hipGraph replay" if graph_us is not None else "eager series",
                                   "algorithmic_MB": round(gb * 1e3, 2), "GBps": round(gb / (us * 1e-6), 1),
                                   "frac_of_hbm_peak": round(gb / (us * 1e-6) / PEAK_HBM_GBS, 4)}
            del frame, field, x, f1, f0, f1024
    return out


def launcher_selftest(args, world, rank):
    """What the N-rank launch does around the GPU work, on the CPU: pin the rank, init (gloo), rank r takes frames
    r, r + world, ... (weak: --steps frames per rank; strong: --total-frames in all), R repeats of a fake timed region,
    ONE all_gather of the metrics record, max-over-ranks time per repeat, rank 0 prints the line.
    --selftest-fail-rank R: that rank exits non-zero before the rendezvous (the launcher must relay the failure)."""
    import torch.distributed as dist
    from mrcc_amd.app.sharding import frame_seeds_for_rank, gather_metrics, ordered_prefetch, pin_rank

    if args.selftest_fail_rank is not None and rank == args.selftest_fail_rank:
        sys.stderr.write(f"bench.py selftest: rank {rank} fails on purpose\n")
        sys.exit(3)
    pin = pin_rank(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    if world > 1:
        dist.init_process_group("gloo")
    total = args.total_frames if args.total_frames > 0 else args.steps * world
    seeds = frame_seeds_for_rank(total, rank, world)
    # the strong mode's frame source: seeds of this rank through the background host threads, in order
    produced = list(ordered_prefetch(lambda s_: s_ * 2, seeds, threads=2))
    assert produced == [2 * s_ for s_ in seeds]
    reps = [0.001 * (rank + 1) * (1 + 0.1 * r) for r in range(args.repeats)]
    agg = gather_metrics({"frames": len(seeds), "elapsed": reps[0], "confusion": np.eye(3, dtype=np.int64),
                          "seed_sum": int(sum(seeds)), "elapsed_repeats": reps,
                          "per_rank": {"host_prepare_ms": 1.5 + rank, "pinned_cores": pin.get("cores") or 0}}, device="cpu")
    if rank == 0:
        print(json.dumps({"metric": "launcher selftest (no GPU work)", "n_gpus": world, "steps": args.steps,
                          "scaling": "strong" if args.total_frames > 0 else "weak",
                          "frames": agg["frames"], "per_rank_frames": agg["per_rank_frames"],
                          "seed_sum": agg["seed_sum"], "elapsed_max": agg["elapsed_max"],
                          "elapsed_repeats_max": agg["elapsed_repeats_max"], "pinned_cores": pin.get("cores"),
                          "per_rank_elapsed": agg["per_rank_elapsed"], "per_rank": agg["per_rank"],
                          "selftest": True}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _median(xs):
    xs = sorted(xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


def strong_scaling_block(model, device, rank, world, total_frames, streams, threads, barrier, group=1):
    """Cfg-4 (BASELINE configs[3]): a FIXED job of `total_frames` frames sharded over the ranks (rank r takes seeds
    r, r + world, ...: app/sharding.py), host arrays in -> labels out.  Inside the timed region, per rank: background
    host threads produce the rank's frames in order (synthetic generation stands in for the reference's per-frame
    loader, test_segmentation.py:58-73), each frame is staged through pinned memory, uploaded, voxelised, run through
    the network, and its labels are copied back (app/pipeline.py HostFrameStream) - so host-side cost, the one thing that
    can break scaling on a shared host, is inside the measurement.  Returns this rank's record."""
    from mrcc_amd.app.pipeline import HostFrameStream
    from mrcc_amd.app.sharding import frame_seeds_for_rank, ordered_prefetch

    seeds = frame_seeds_for_rank(total_frames, rank, world)

    def stage(x, field):
        return model(x).slice_argmax(field, with_conf=False)[0]

    stream = HostFrameStream(device, SCALE, stage, None, compute_streams=streams, group=group)  # group: frames per sparse tensor

    def source(which):
        return ordered_prefetch(lambda sd: mrcc_amd.synth.gen_room(POINTS, ROOM, sd)[:2], which, threads=threads,
                                lookahead=2 * threads + 2)

    stream.preallocate(POINTS)  # every slot's pinned staging buffer: a driver call each, not part of a frame's work
    for _ in stream.run(source(seeds[:min(2 * group, len(seeds))])):  # warm the source's threads and the launch path
        pass
    torch.cuda.synchronize()
    barrier()
    hist = np.zeros(3, dtype=np.int64)
    t0 = time.perf_counter()
    n = 0
    for labels in stream.run(source(seeds)):
        hist += np.bincount(labels, minlength=3)[:3]
        n += 1
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    hs = stream.host_s
    return {"frames": n, "elapsed": elapsed, "hist": hist, "seed_sum": int(sum(seeds)),
            "host_ms_per_frame": {k: round(v / max(hs["frames"], 1) * 1e3, 3) for k, v in hs.items() if k != "frames"}}


def other_configs_block(model, device, streams, checked):
    """BASELINE configs[2] and [4] on the driver-run line (bounded: a few seconds each), each with a parity flag.
      cfg5: 500k points / 1 cm frames through the headline pipeline (voxelise + maps + U-Net + slice/argmax) - frames/s;
            parity = voxel keys equal the oracle's, the last dense layers of the head bit-exact against the oracle on
            the GPU's own input rows, labels = argmax of the logits.
      cfg3: 64 Cfg-2 frames in ONE sparse tensor (batch column, data/alivev2.py:358-383: 12.8 M points, 5.6 M voxels) -
            seg forward + 64 Kabsch problems, frames/s; parity = the batch's frame 0 equals the one-frame run bit for
            bit (and, through it, the oracle's labels when the accuracy block checked that frame); every launch of the
            wide layers on the buffer-addressed (FAST) instances although the tensors exceed 2 GB."""
    import sv_oracle as O
    from mrcc_amd.app.pipeline import FramePipeline
    from mrcc_amd.utils import transformation as T

    from mrcc_amd.model.robotnet_vote import RobotNetVote

    out = {}
    torch.manual_seed(4)
    vote = RobotNetVote(3).to(device).eval()
    crop1 = mrcc_amd.synth.gen_ee_crop(0, n=16)
    kp_ref1, kp_tgt1 = mrcc_amd.synth.REFERENCE_KEY_POINTS[None], crop1[3][None]
    # ---- Cfg-5
    _log("other configs: cfg5 (500k points, 1 cm)")
    frames5 = []
    for sd in range(2):
        pts, rgb, _ = mrcc_amd.synth.gen_room(500_000, ROOM, sd)
        c4 = np.concatenate([np.zeros((len(pts), 1), np.float32), pts * np.float32(100)], axis=1)
        frames5.append((torch.from_numpy(c4).to(device), torch.from_numpy(rgb).to(device), c4))
    pipe = FramePipeline(device, levels=4, compute_streams=streams)
    with torch.no_grad():
        g5 = 2  # frames per sparse tensor, as the headline's --group (two 500k-point frames: 611k voxels per launch)
        run_frames(model, pipe, frames5, 4, group=g5)
        pipe.drain()
        torch.cuda.synchronize()
        n5 = 10
        t0 = time.perf_counter()
        vox5 = run_frames(model, pipe, frames5, n5, group=g5)
        torch.cuda.synchronize()
        dt5 = (time.perf_counter() - t0) / n5

        # the configuration as BASELINE words it - seg -> vote -> pose: the vote head (model/robotnet_vote.py:62-71) on the
        # same sparse tensor and one Kabsch solve per frame (the pose legs on the end-effector crop are parity-tested at
        # this size in tests/test_gpu_cfg.py; their networks see a few thousand points)
        # one rigid-transform problem per frame of the group
        kp_ref1_d = torch.from_numpy(np.ascontiguousarray(kp_ref1, dtype=np.float64)).to(device).repeat(g5, 1, 1)
        kp_tgt1_d = torch.from_numpy(np.ascontiguousarray(kp_tgt1, dtype=np.float64)).to(device).repeat(g5, 1, 1)

        def seg_vote_pose(x_, f_):
            lab_ = model(x_).slice_argmax(f_)[0]
            v_ = vote(x_).slice_argmax(f_)[0]
            # the solve is enqueued behind the networks on the frame's stream; its result stays on the device (a download
            # here would block the host until the frame has finished, i.e. stop it from preparing the next one)
            pose_ = T.get_rigid_transform_3D_batched(kp_ref1_d, kp_tgt1_d, device=device, as_tensors=True)
            return lab_, v_, pose_

        def run5(n):  # n frames, g5 per sparse tensor (n a multiple of g5)
            members = lambda i: [frames5[(i + j) % len(frames5)][:2] for j in range(g5)]  # noqa: E731
            nxt = pipe.prepare_group(members(0))
            for i in range(0, n, g5):
                cur = nxt
                pipe.run(cur, seg_vote_pose)
                if i + g5 < n:
                    nxt = pipe.prepare_group(members(i + g5))
            pipe.drain()

        run5(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run5(6)
        torch.cuda.synchronize()
        dt5_full = (time.perf_counter() - t0) / 6
        # parity on frame 0
        field = ME.TensorField(frames5[0][1], frames5[0][0], device=device)
        x = field.sparse()
        feats = model.forward_except_final(x)
        h0 = model.final.forward_fused(feats, act=2, slope=model.leaky_relu.negative_slope)
        h1 = model.regression[0].forward_fused(h0, act=2, slope=model.regression[1].negative_slope)
        logits = model.regression[2].forward_fused(h1)
        label, _ = logits.slice_argmax(field)
        vox = O.voxelize(frames5[0][2])
        rows = np.sort(np.random.default_rng(0).choice(x.F.shape[0], size=2048, replace=False))
        ridx = torch.from_numpy(rows).to(device)
        sd_ = {k: v.cpu() for k, v in model.state_dict().items()}
        w0 = sd_["final.kernel"].numpy()[None]
        o0 = O.conv(feats.F[ridx].cpu().numpy(), w0, None, len(rows), None, sd_["final.bias"].numpy().reshape(-1), None,
                    O.ACT_LEAKY, 0.01)
        o1 = O.conv(o0, sd_["regression.0.linear.weight"].numpy().T[None], None, len(rows), None,
                    sd_["regression.0.linear.bias"].numpy(), None, O.ACT_LEAKY, 0.01)
        o2 = O.conv(o1, sd_["regression.2.linear.weight"].numpy().T[None], None, len(rows), None,
                    sd_["regression.2.linear.bias"].numpy())
        lg = logits.F.cpu().numpy()
        gf5_seg = profiling.pass_gflop(lambda: model(x))
        gf5_vote = profiling.pass_gflop(lambda: vote(x))
        e2e = lambda gf, sec: {"gflop_per_frame": round(gf, 1), "tflops": round(gf / sec / 1e3, 2),  # noqa: E731
                               "frac_of_f32_mfma_peak": round(gf / sec / 1e3 / PEAK_F32_MFMA_TFLOPS, 4)}
        out["cfg5"] = {
            "workload": "cfg5: 500k-pt cloud, 1 cm voxels, seg -> vote -> pose: voxelise + maps, RobotNetSegmentation and "
                        "RobotNetVote (both MinkUNet18D heads) on the frame, slice/argmax of both, one Kabsch solve; "
                        "two frames per sparse tensor, as the headline's --group",
            "frames_per_sparse_tensor": g5,
            "value": round(1.0 / dt5_full, 3), "unit": "frames/s", "ms_per_frame": round(dt5_full * 1e3, 3), "frames_timed": 6,
            "end_to_end": e2e(gf5_seg + gf5_vote, dt5_full),
            "seg_only": {"value": round(1.0 / dt5, 3), "unit": "frames/s", "ms_per_frame": round(dt5 * 1e3, 3), "frames_timed": n5,
                         "what": "voxelise + maps + seg U-Net + slice/argmax (the headline pipeline at 3.5x the voxels)",
                         "end_to_end": e2e(gf5_seg, dt5)},
            "active_voxels_per_frame": int(vox5 // n5),
            "parity": {"voxel_keys_equal": bool(np.array_equal(x.coordinate_map.keys.cpu().numpy().view(np.uint64), vox["keys"])),
                       "inverse_equal": bool(np.array_equal(field.inverse_mapping.cpu().numpy(), vox["inverse"])),
                       "head_layers_bit_exact_on_2048_rows": bool(np.array_equal(lg[rows], o2)),
                       "labels_are_argmax": bool(np.array_equal(label.cpu().numpy(), lg.argmax(1)[vox["inverse"]])),
                       "note": "vote and pose legs at this size: tests/test_gpu_cfg.py::test_cfg5_vote_and_pose_legs"}}
        del frames5, pipe, field, x, feats, h0, h1, logits
        torch.cuda.empty_cache()
        # ---- Cfg-3
        B = 64
        _log(f"other configs: cfg3 ({B} frames in one sparse tensor)")
        coords, feats_in, *_ = make_frame(0, device, batch=B)
        crops = [mrcc_amd.synth.gen_ee_crop(sd, n=16) for sd in range(B)]
        kp_ref = np.repeat(mrcc_amd.synth.REFERENCE_KEY_POINTS[None], B, axis=0)
        kp_tgt = np.stack([c[3] for c in crops])

        def step3(with_vote=False):
            f = ME.TensorField(feats_in, coords, device=device)
            xs = f.sparse()
            o = model(xs)
            lab = o.slice_argmax(f)[0]
            if with_vote:
                vote(xs).slice_argmax(f)
            R, t, q = T.get_rigid_transform_3D_batched(kp_ref, kp_tgt, device=device)
            return xs, o, lab, (R, t, q)

        profiling.INSTANCE_LOG = log = []
        xs, o, lab, pose = step3()
        profiling.INSTANCE_LOG = None
        torch.cuda.synchronize()
        wide = [e for e in log if e[3] >= 32 and e[3] % 4 == 0 and e[4] >= 32]
        bs = xs.coordinate_manager.batch_offsets(1, B).tolist()
        f1 = ME.TensorField(feats_in[:POINTS], coords[:POINTS], device=device)
        x1 = f1.sparse()
        o1_ = model(x1)
        l1 = o1_.slice_argmax(f1)[0]
        same = bool(torch.equal(o.F[bs[0]:bs[1]], o1_.F) and torch.equal(lab[:POINTS], l1))
        oracle_labels = None
        if checked is not None and checked["what"].startswith("full frame"):
            oracle_labels = bool(np.array_equal(lab[:POINTS].cpu().numpy(), checked["ref"]["label"]))
        Ro, to = O.get_rigid_transform_3D(kp_ref[0], kp_tgt[0])
        V3 = xs.F.shape[0]
        gf3_seg = profiling.pass_gflop(lambda: model(xs))
        gf3_vote = profiling.pass_gflop(lambda: vote(xs))
        del xs, o, lab, x1, o1_, f1
        t0 = time.perf_counter()
        n3 = 2
        for _ in range(n3):
            step3()
        torch.cuda.synchronize()
        dt3 = (time.perf_counter() - t0) / n3
        step3(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step3(True)
        torch.cuda.synchronize()
        dt3_full = time.perf_counter() - t0
        out["cfg3"] = {
            "workload": f"cfg3: {B} synthetic 200k-pt frames in ONE sparse tensor (batch column): seg + keypoint-vote (two "
                        f"MinkUNet18D heads on the batch) + slice/argmax of both + {B} Kabsch problems",
            "value": round(B / dt3_full, 3), "unit": "frames/s", "ms_per_batch": round(dt3_full * 1e3, 2), "batches_timed": 1,
            "end_to_end": dict(e2e(gf3_seg + gf3_vote, dt3_full), gflop_per_frame=round((gf3_seg + gf3_vote) / B, 1)),
            "seg_only": {"value": round(B / dt3, 3), "unit": "frames/s", "ms_per_batch": round(dt3 * 1e3, 2), "batches_timed": n3,
                         "what": f"seg U-Net forward + slice/argmax + {B} Kabsch problems",
                         "end_to_end": dict(e2e(gf3_seg, dt3), gflop_per_frame=round(gf3_seg / B, 1))},
            "active_voxels": int(V3),
            "parity": {"frame0_equals_single_frame_run_bit_exact": same, "frame0_labels_equal_oracle": oracle_labels,
                       "all_wide_layer_launches_on_fast_instances": bool(wide and all(e[1]["fast"] == 1 for e in wide)),
                       "conv_launches": len(log),
                       "kabsch_max_abs_dR_vs_oracle": float(np.abs(pose[0][0] - Ro).max())}}
        del coords, feats_in
        torch.cuda.empty_cache()
    return out


def engine_block(device):
    """Secondary configuration through the REFERENCE'S API (app/inference_engine.py), host numpy arrays in -> results out,
    everything (pinned staging, H2D, voxelisation, networks, largest-cluster rule, D2H) inside the timed regions:
      * predict_segmentation one frame at a time (the reference's consumer loop, app/main.py:432-456) and streamed
        (predict_segmentation_stream) on the headline's 200k-point room frames;
      * `predict_full`: whole predict() - segmentation -> EE crop -> rotation network -> translation -> key-point network ->
        selection -> Kabsch -> base poses (reference :281-382) - per frame and streamed (predict_stream, pose stages batched
        over groups of frames) on labelled 200k-point scenes whose 4 096-point end-effector crop exists by construction
        (synth.gen_scene(keyed_colors=True) + synth.wire_color_keyed_labels: two colour channels wired to the logits, every
        other weight random, so the kernels run full-size random operands)."""
    from mrcc_amd.app.dto import PointCloudDTO
    from mrcc_amd.app.inference_engine import InferenceEngine
    from mrcc_amd.utils.config import Config

    Config.reset()
    Config().update({"INFERENCE": {"SEGMENTATION": {"scale": SCALE}, "ROTATION": {"scale": 100},
                                   "KEY_POINTS": {"scale": 100, "conf_threshold": 0.0}}})
    try:
        eng = InferenceEngine(allow_random_init=True, seed=1)
        pool = [mrcc_amd.synth.gen_room(POINTS, ROOM, sd)[:2] for sd in range(4)]
        frames = [pool[i % 4] for i in range(32)]
        for i in range(4):  # first calls: code objects, pinned buffers, allocator pools
            ref = eng.predict_segmentation(*pool[(i + 2) % 4])  # the last one is pool[1], compared below
        lat = []
        for i in range(12):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.predict_segmentation(*frames[i])
            lat.append((time.perf_counter() - t0) * 1e3)
        sync_ms = _median(lat)
        for _ in range(2):  # steady state: pinned slots, the streams' allocator pools (emptied by the previous block)
            list(eng.predict_segmentation_stream(iter(frames[:16])))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = list(eng.predict_segmentation_stream(iter(frames)))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / len(frames) * 1e3
        for _ in range(2):
            list(eng.predict_segmentation_stream(iter(frames[:16]), group=4))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got4 = list(eng.predict_segmentation_stream(iter(frames), group=4))
        torch.cuda.synchronize()
        ms4 = (time.perf_counter() - t0) / len(frames) * 1e3
        out = {"workload": "engine: InferenceEngine.predict_segmentation_stream, host numpy in -> labels out "
                           "(H2D, voxelise, U-Net, slice/argmax, EE cluster rule, D2H inside the timed region), 200k-pt frames",
               "value": round(1e3 / ms, 3), "unit": "frames/s", "ms_per_frame": round(ms, 3), "frames_timed": len(frames),
               "per_frame_predict_segmentation_ms": round(sync_ms, 3),
               "per_frame_predict_segmentation_ms_min_max": [round(min(lat), 3), round(max(lat), 3)],
               "within_budget": {"per_frame_ms_le_21": bool(sync_ms <= 21.0), "stream_ms_le_20": bool(ms <= 20.0)},
               "labels_equal_predict_segmentation": bool(np.array_equal(got[1], ref) and np.array_equal(got[5], ref)),
               "stream_group4": {"what": "predict_segmentation_stream(group=4): four consecutive frames per sparse tensor, labels "
                                         "delivered a group at a time", "value": round(1e3 / ms4, 3), "unit": "frames/s",
                                 "ms_per_frame": round(ms4, 3),
                                 "labels_equal": bool(all(np.array_equal(a, b) for a, b in zip(got4, got)))}}
        # ---- whole predict() on labelled scenes
        mrcc_amd.synth.wire_color_keyed_labels(eng._segmentation_model)
        scenes = [mrcc_amd.synth.gen_scene(sd, n_bg=POINTS - 4000 - 4096, n_arm=4000, n_ee=4096, room=ROOM, keyed_colors=True)
                  for sd in range(4)]
        dtos = [PointCloudDTO(points=sc["points"], rgb=sc["rgb"], ee2base_pose=sc["ee2base_pose"]) for sc in scenes]
        seq = [dtos[i % 4] for i in range(24)]
        for i in range(3):
            refs = [eng.predict(d) for d in dtos]
        lat = []
        for i in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.predict(seq[i])
            lat.append((time.perf_counter() - t0) * 1e3)
        for _ in range(2):  # pinned rings, allocator pools of the crop stream, the pose thread's first launches
            list(eng.predict_stream(iter(seq[:12])))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = list(eng.predict_stream(iter(seq)))
        torch.cuda.synchronize()
        full_ms = (time.perf_counter() - t0) / len(seq) * 1e3
        for _ in range(2):
            list(eng.predict_stream(iter(seq[:12]), seg_group=4))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res4 = list(eng.predict_stream(iter(seq), seg_group=4))
        torch.cuda.synchronize()
        full4_ms = (time.perf_counter() - t0) / len(seq) * 1e3
        same4 = all(np.array_equal(a.segmentation, b.segmentation) and np.array_equal(a.ee_pose, b.ee_pose) and
                    np.array_equal(a.key_points_pose, b.key_points_pose) and np.array_equal(a.base_pose, b.base_pose)
                    for a, b in zip(res4, res))
        same = all(np.array_equal(r.segmentation, refs[i % 4].segmentation) and r.ee_pose is not None and
                   np.array_equal(r.ee_pose, refs[i % 4].ee_pose) and np.array_equal(r.key_points_pose, refs[i % 4].key_points_pose)
                   and np.array_equal(r.base_pose, refs[i % 4].base_pose) for i, r in enumerate(res))
        ee_counts = [int((r.segmentation == 2).sum()) for r in refs]
        out["predict_full"] = {
            "workload": "InferenceEngine.predict / predict_stream on labelled 200k-pt scenes (4096-pt EE crop by construction): "
                        "segmentation -> crop -> rotation net -> translation -> key-point net -> selection -> Kabsch -> base poses",
            "stream": {"value": round(1e3 / full_ms, 3), "unit": "frames/s", "ms_per_frame": round(full_ms, 3),
                       "frames_timed": len(seq), "pose_group": 4},
            "stream_seg_group4": {"what": "predict_stream(seg_group=4): the segmentation stage on groups of four frames too",
                                  "value": round(1e3 / full4_ms, 3), "unit": "frames/s", "ms_per_frame": round(full4_ms, 3),
                                  "results_equal": bool(same4)},
            "per_frame": {"value": round(1e3 / _median(lat), 3), "unit": "frames/s", "ms_per_frame": round(_median(lat), 3),
                          "ms_min_max": [round(min(lat), 3), round(max(lat), 3)]},
            "stream_over_segmentation_stream": round(ms / full_ms, 3),
            "ee_points_labelled": ee_counts,
            "stream_results_equal_per_frame_predict": bool(same)}
        return out
    finally:
        Config.reset()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region (exactly --steps steps between barrier + synchronize) is run this many times; "
                         "ms_per_step is the median, min / max are on the line.  Small --steps get more repeats "
                         "(ceil(64 / steps)) so that the timed regions total about a second or more")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pool", type=int, default=4, help="distinct frames resident per rank")
    ap.add_argument("--no-kernel-timer", action="store_true", help="diagnostic: drop the per-launch HIP events")
    ap.add_argument("--streams", type=int, default=3,
                    help="compute streams alternating between frames (app/pipeline.py); default = the configuration with "
                         "the best measured frames/s (DESIGN.md section 8)")
    ap.add_argument("--stagger-level0", type=int, default=None, choices=(0, 1),
                    help="level-0 stages of consecutive frames take turns (default: MRCC_STAGGER_LEVEL0 / the pipeline's default)")
    ap.add_argument("--group", type=int, default=4,
                    help="frames the pipeline puts through the network at a time as ONE sparse tensor (batch column = position "
                         "in the group; FramePipeline.prepare_group). A step stays one frame: --steps frames are timed, the "
                         "last group of a region holds the remainder. 1 = every frame its own launches")
    ap.add_argument("--frames-per-step", type=int, default=1,
                    help="frames fused into one sparse tensor per step (batch column); 1 = the headline workload")
    ap.add_argument("--batched-frames", type=int, default=4,
                    help="secondary measurement: this many frames per step in one sparse tensor (0/1 = skip)")
    ap.add_argument("--total-frames", type=int, default=0,
                    help="STRONG-scaling mode (Cfg-4): a fixed job of this many frames sharded over the ranks, host arrays "
                         "in -> labels out, frame source and H2D inside the timed region; becomes the headline of the line")
    ap.add_argument("--strong-frames", type=int, default=512,
                    help="frames of the secondary strong-scaling block of a default (weak) run; 0 = skip")
    ap.add_argument("--source-threads", type=int, default=0, help="host threads producing frames in the strong mode (0 = auto)")
    ap.add_argument("--no-extras", action="store_true",
                    help="headline measurement only: no batched / other-config / engine / strong blocks, no HBM-bound-layer "
                         "block (PMC passes)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="CPU rehearsal of the N-rank launch: pinning, rendezvous (gloo), frame sharding (weak or strong), "
                         "the one all_gather and the JSON line, without touching a GPU (tests/test_dist_cpu.py)")
    ap.add_argument("--selftest-fail-rank", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--gather-gemm-child", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gather_gemm_child:
        return gather_gemm_child(args.gather_gemm_child)
    import faulthandler

    faulthandler.dump_traceback_later(300, repeat=True, file=sys.stderr)  # a stuck run says where it is stuck

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the torch.distributed environment has WORLD_SIZE={world}; "
                         "launch with --nproc-per-node equal to --gpus (or run `python bench.py --gpus N`, which starts "
                         "the ranks itself)")
    backend = os.environ.get("MRCC_DIST_BACKEND", "nccl")  # "gloo" = rehearsal with several ranks on one GPU
    if args.launcher_selftest:
        return launcher_selftest(args, world, rank)
    # ---- host placement, BEFORE the first GPU call (no re-exec): this rank's threads stay on its GPU's NUMA node
    from mrcc_amd.app.sharding import gather_metrics, pin_rank

    full_affinity = os.sched_getaffinity(0)
    pin = pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))) if world > 1 else \
        {"pinned": False, "reason": "single rank: left on the whole host", "cores": len(full_affinity)}
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)
    mrcc_amd._lib.load()

    def barrier():
        if world > 1:
            dist.barrier()

    _log(f"rank {rank}/{world} on {device} ({pin}): building the model and {args.pool} resident frames")
    model = build_model(device)
    # ---- the oracle pass comes FIRST, outside every timed region: the reference labels of the accuracy half of the metric
    #      (every N) and, at N = 1, the timed CPU baseline.  With N > 1 it is capped to the quarter frame (~2.5 s) so that
    #      the other ranks wait at the first barrier for seconds, not for the 17 s of the full N = 1 pass.
    base = base_mm = checked = None
    if rank == 0 and not args.no_cpu_baseline:
        base, base_mm, checked = oracle_pass(model, budget_s=25.0 if world == 1 else 0.0, world=world)
    # rank r owns frames r, r + world, ... of the global seed sequence (Cfg-4 sharding rule)
    frames = [make_frame(rank + world * i, device, batch=args.frames_per_step) for i in range(args.pool)]

    from mrcc_amd.app.pipeline import FramePipeline

    stagger = None if args.stagger_level0 is None else bool(args.stagger_level0)
    group = max(1, args.group) if args.frames_per_step == 1 else 1  # an explicitly batched workload is not grouped again
    pipe = FramePipeline(device, levels=4, compute_streams=args.streams, stagger_level0=stagger)
    src_threads = args.source_threads or max(2, min(6, pin.get("cores", 8) - 2))
    if args.total_frames > 0:
        # ------------------------------------------------------------------------------------------------ strong mode
        with torch.no_grad():
            rec = strong_scaling_block(model, device, rank, world, args.total_frames, args.streams, src_threads, barrier,
                                       group=group)
        agg = gather_metrics({"frames": rec["frames"], "elapsed": rec["elapsed"], "confusion": np.diag(rec["hist"]),
                              "seed_sum": rec["seed_sum"],
                              "per_rank": dict({f"host_{k}_ms": v for k, v in rec["host_ms_per_frame"].items()},
                                               pinned_cores=pin.get("cores") or 0)},
                             device=device if backend == "nccl" else "cpu")
        if rank == 0:
            t_max = agg["elapsed_max"]
            line = {"metric": "point-cloud frames/sec at 200k pts/frame", "value": round(agg["frames"] / t_max, 3),
                    "unit": "frames/s", "n_gpus": world, "steps": max(agg["per_rank_frames"]), "warmup": 2 * group,
                    "ms_per_step": round(t_max / max(agg["per_rank_frames"]) * 1e3, 3), "higher_is_better": True,
                    "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": f"cfg4: a fixed job of {args.total_frames} synthetic 200k-pt frames (2 cm voxels) "
                                           f"sharded over the ranks, host arrays in -> labels out: frame source ({src_threads} "
                                           "host threads per rank), pinned staging, H2D, voxelise, RobotNetSegmentation"
                                           "(MinkUNet18D), slice/argmax, D2H all inside the timed region",
                               "total_frames": args.total_frames, "per_rank_frames": agg["per_rank_frames"],
                               "frames_per_sparse_tensor": group,
                               "points_per_frame": POINTS, "label_histogram": [int(v) for v in np.diag(agg["confusion"])],
                               "host_ms_per_frame_rank0": rec["host_ms_per_frame"], "host_placement": pin,
                               "per_rank_elapsed_s": [round(x, 4) for x in agg["per_rank_elapsed"]],
                               "per_rank": {k: [round(x, 3) for x in v] for k, v in agg["per_rank"].items()},
                               "parallelism": f"frame-sharded x{world} (rank r: seeds r, r + {world}, ...), one all_gather "
                                              "of metrics"},
                    "roofline": None, "cpu_baseline": base}
            print(json.dumps(line), flush=True)
        faulthandler.cancel_dump_traceback_later()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    with torch.no_grad():
        # warm-up (untimed), part 1 on ONE compute stream: every conv launch is event-timed -> per-kernel table and the
        # dominant kernel instance measured in isolation; part 2 warms the multi-stream pipeline used in the timed region
        nwarm = -(-max(args.warmup, 2) // group) * group  # whole groups: every launch of the pass covers `group` frames
        warm_timer = profiling.KernelTimer(capacity=2 * 64 * nwarm + 64)
        profiling.TIMER = warm_timer
        pipe.single = True
        warm_hist = torch.zeros(3, dtype=torch.int64, device=device)  # same code path as the timed region: the
        run_frames(model, pipe, frames, nwarm, warm_hist, group=group)  # first use of a torch kernel loads its code object
        pipe.drain()
        torch.cuda.synchronize()
        pipe.single = False
        profiling.TIMER = None
        warm = warm_timer.summarize()
        warm_by_layer = warm_timer.summarize(by_layer=True)
        # dominant = the (kernel instance, layer shape) with the most algorithmic flops - launches that all process the
        # same kind of unit, so that "flops per launch / average launch duration" means something (one instance serves
        # several layer shapes: since the dispatch thresholds favour tall tiles, the dual-body kernel runs levels 0 AND
        # 1).  Time-ranking would be fooled by the first launch after an idle gap, whose event interval absorbs the gap.
        dominant_layer = max(warm_by_layer.items(), key=lambda kv: kv[1]["flops"])[0]
        dominant = dominant_layer[0]
        run_frames(model, pipe, frames, max(nwarm, 2 * group), warm_hist, group=group)
        pipe.drain()
        torch.cuda.synchronize()
        # ---- the timed region, R times: each is EXACTLY --steps steps bracketed by barrier + synchronize on both sides
        elapsed_list, voxels, hist, timer = [], 0, None, None

        def timed_region():
            nonlocal voxels, hist, timer
            timer_r = profiling.KernelTimer(capacity=2 * 32 * args.steps + 64)
            timer_r.only = {dominant}
            profiling.TIMER = None if args.no_kernel_timer else timer_r
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hist_r = torch.zeros(3, dtype=torch.int64, device=device)
            voxels = run_frames(model, pipe, frames, args.steps, hist_r, group=group)
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            elapsed_list.append(time.perf_counter() - t0)
            profiling.TIMER = None
            if timer is None:
                timer, hist = timer_r, hist_r
            else:
                timer.records += timer_r.records

        # R repeats; short regions (small --steps) get more of them so that the timed regions total about a second or more.
        # The count is a pure function of the arguments - identical on every rank without a collective (the run's only
        # collective stays the one all_gather of the metrics record; the barriers are the contract's).
        repeats = min(max(1, args.repeats, -(-64 // max(args.steps * args.frames_per_step, 1))), 64)
        _log(f"warm-up done; timing {args.steps} steps x {repeats} repeats")
        for _ in range(repeats):
            timed_region()
    elapsed = _median(elapsed_list)
    _log(f"timed regions: median {elapsed * 1e3 / args.steps:.2f} ms/step over {len(elapsed_list)} repeats "
         f"(min {min(elapsed_list) * 1e3 / args.steps:.2f}, max {max(elapsed_list) * 1e3 / args.steps:.2f})")
    # ---- secondary, separately named configurations (N = 1 only: with more ranks they would only make the others wait)
    extras = world == 1 and not args.no_extras and args.frames_per_step == 1
    batched = None
    if extras and args.batched_frames > 1:
        bsteps = max(4, args.steps // args.batched_frames)
        with torch.no_grad():
            bframes = [make_frame(rank + world * i, device, batch=args.batched_frames) for i in range(2)]
            run_frames(model, pipe, bframes, 3)
            pipe.drain()
            torch.cuda.synchronize()
            tb = time.perf_counter()
            run_frames(model, pipe, bframes, bsteps)
            torch.cuda.synchronize()
            tb = time.perf_counter() - tb
            del bframes
        batched = {"config": f"cfg2 frames, {args.batched_frames} per step in one sparse tensor (batch column)",
                   "frames_per_step": args.batched_frames, "steps": bsteps, "ms_per_step": round(tb / bsteps * 1e3, 3),
                   "value_this_rank": round(bsteps * args.batched_frames / tb, 3), "unit": "frames/s"}
        _log(f"batched x{args.batched_frames}: {batched['value_this_rank']} frames/s on this rank")
    ungrouped = None
    if extras and group > 1:  # the same frames one per launch (rounds 1-3's headline configuration), for continuity
        with torch.no_grad():
            run_frames(model, pipe, frames, 6)
            pipe.drain()
            torch.cuda.synchronize()
            tu = time.perf_counter()
            run_frames(model, pipe, frames, args.steps)
            torch.cuda.synchronize()
            tu = time.perf_counter() - tu
        ungrouped = {"config": "the headline's frames, one frame per sparse tensor (--group 1)", "steps": args.steps,
                     "ms_per_step": round(tu / args.steps * 1e3, 3), "value_this_rank": round(args.steps / tu, 3), "unit": "frames/s"}
        _log(f"one frame per launch: {ungrouped['value_this_rank']} frames/s on this rank")
    hbm_layers = hbm_bound_layers(model, device) if extras else None
    other = engine = None
    if extras:
        _log("engine block: InferenceEngine per frame and streamed")
        engine = engine_block(device)
    # the strong-scaling block runs at EVERY N (same total job), so the driver's N = 1, 2, 4, 8 lines carry the curve
    strong = None
    if not args.no_extras and args.strong_frames > 0 and args.frames_per_step == 1:
        _log(f"strong-scaling block: {args.strong_frames} frames over {world} rank(s), {src_threads} source threads per rank")
        with torch.no_grad():
            strong = strong_scaling_block(model, device, rank, world, args.strong_frames, args.streams, src_threads, barrier,
                                          group=group)
    if extras:
        # LAST: Cfg-3 holds 60-77 GiB for a moment and the block returns its memory to the driver (empty_cache); whatever runs
        # after such an episode is ~10 % slower for the rest of the process (engine stream 70.2 -> 62.6 frames/s after a 20 GB
        # allocate / free / empty_cache, measured in isolation: the regrown pools are placed worse), so the frame-rate blocks
        # above come first
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        other = other_configs_block(model, device, args.streams, checked)

    # the run's ONE collective: all_gather of a small per-rank record (RCCL over xGMI when world > 1)
    h = hist.cpu().numpy()
    per_rank = {"median_repeat_s": _median(elapsed_list), "pinned_cores": pin.get("cores") or 0}
    if strong:
        per_rank.update({"strong_elapsed_s": strong["elapsed"], "strong_frames": strong["frames"]})
        per_rank.update({f"strong_host_{k}_ms": v for k, v in strong["host_ms_per_frame"].items()})
    rec = {"frames": args.steps * args.frames_per_step, "elapsed": elapsed, "confusion": np.diag(h), "seed_sum": voxels,
           "elapsed_repeats": elapsed_list + ([strong["elapsed"], float(strong["frames"])] if strong else []),
           "per_rank": per_rank}
    agg = gather_metrics(rec, device=device if backend == "nccl" else "cpu")
    reps_max = agg["elapsed_repeats_max"][:len(elapsed_list)]
    t_med = _median(reps_max)
    total_frames = float(agg["frames"])
    voxels_per_frame = voxels // max(args.steps * args.frames_per_step, 1)

    if rank == 0:
        ksum = timer.summarize()
        ksum_by_layer = timer.summarize(by_layer=True)
        roofline = None
        nwarm = -(-max(args.warmup, 2) // group) * group  # frames of the warm-up pass (see above)
        gflop_step = sum(v["flops"] for v in warm.values()) / nwarm / 1e9  # every conv launch of one step
        if dominant_layer in ksum_by_layer:
            name, d = dominant, ksum_by_layer[dominant_layer]
            _, lK, lCin, lCout, lwhere = dominant_layer
            tflops = d["flops"] / (d["ms"] * 1e-3) / 1e12
            roofline = {
                "kernel": name, "layer": f"kernel volume {lK}, {lCin} -> {lCout} channels, output map {lwhere} "
                                         "(s<tensor stride>: s1 = level 0)",
                "bound": "mfma", "achieved": round(tflops, 3), "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(tflops / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                # a layer run as offset-range passes is several kernel launches behind ONE event interval: "launch" below =
                # layer; the per-kernel-launch figures (what a rocprofv3 kernel trace lists) are beside it
                "launches": d["launches"], "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                "algorithmic_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 3),
                "kernel_launches": d["kernel_launches"],
                "avg_kernel_launch_ms": round(d["ms"] / max(d["kernel_launches"], 1), 4),
                "algorithmic_gflop_per_kernel_launch": round(d["flops"] / max(d["kernel_launches"], 1) / 1e9, 3),
                "compute_streams": args.streams,
                "note": "achieved / frac = per-launch HIP-event intervals INSIDE the timed regions, where a launch shares "
                        "the chip with the other frames' kernels; `isolated` = the same launches alone on the GPU; "
                        "`end_to_end` = all algorithmic flops of a step / ms_per_step",
            }
            # HBM traffic of this kernel from PMC counters (separate rocprofv3 --pmc passes over this command, calibrated
            # on a known-traffic launch as the MI355X guide prescribes): newest profiles/rNN_traffic.json, GB per launch
            try:
                import glob

                tfile = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))[-1]
                tj = json.load(open(tfile))
                tr = tj["kernels"].get(name)
                groups = tj.get("kernels_by_grid", {}).get(name)
                if groups:  # the launches of this layer shape: the group whose grid is closest to this layer's
                    rows = warm_by_layer[dominant_layer]["rows"] / warm_by_layer[dominant_layer]["launches"]
                    want_grid = rows / 64.0 * 2.3  # 64-row tiles x 2 column halves + 15 % of the tiles at half height (per pass)
                    tr = min(groups, key=lambda g: abs(math.log(g["grid_workgroups"] / want_grid)))
                if tr:
                    wl = warm_by_layer[dominant_layer]
                    alg_gb = wl["bytes"] / wl["launches"] / 1e9
                    passes = wl["kernel_launches"] / wl["launches"]  # offset-range passes: kernel launches per layer
                    tr = dict(tr, traffic_GB_per_launch=round(tr["traffic_GB_per_launch"] * passes, 4))
                    roofline["kernel_launches_per_layer"] = passes
                    roofline["traffic"] = tr["traffic_GB_per_launch"]
                    roofline["traffic_unit"] = "GB per layer = per pass (PMC, calibrated) x passes"
                    roofline["traffic_source"] = {"file": "profiles/" + os.path.basename(tfile),
                                                  "collected_at_commit": tj.get("commit", "unknown")}
                    roofline["algorithmic_GB_per_launch"] = round(alg_gb, 4)
                    roofline["traffic_over_algorithmic"] = round(tr["traffic_GB_per_launch"] / alg_gb, 2)
            except (OSError, KeyError, ValueError, IndexError):
                pass
            # every launch of that kernel instance in the timed region, whatever the layer (levels 0 and 1, K = 27 and 8,
            # Cin 384 and 416): the aggregate the roofline object carried before it was split by layer shape
            allk = ksum[name]
            all_tf = allk["flops"] / (allk["ms"] * 1e-3) / 1e12
            roofline["all_launches_of_kernel"] = {"achieved": round(all_tf, 3), "frac": round(all_tf / PEAK_F32_MFMA_TFLOPS, 4),
                                                  "launches": allk["launches"]}
            iso = warm_by_layer[dominant_layer]
            iso_tf = iso["flops"] / (iso["ms"] * 1e-3) / 1e12
            # the same instance with nothing else on the GPU (warm-up pass on one compute stream): with several compute
            # streams the timed-region duration of a launch includes the time it shares the CUs with the other frame
            roofline["isolated"] = {"achieved": round(iso_tf, 3), "frac": round(iso_tf / PEAK_F32_MFMA_TFLOPS, 4),
                                    "avg_launch_ms": round(iso["ms"] / iso["launches"], 4),
                                    "avg_kernel_launch_ms": round(iso["ms"] / max(iso["kernel_launches"], 1), 4)}
            # the whole step against the matrix peak: every conv launch's algorithmic flops (2 P Cin Cout, warm-up pass,
            # where all of them are counted) over the median ms_per_step of this rank's timed regions
            e2e = gflop_step / (elapsed / args.steps * 1e3) if elapsed > 0 else 0.0  # GFLOP / ms = TFLOP/s
            roofline["end_to_end"] = {"gflop_per_step": round(gflop_step, 1), "tflops": round(e2e, 2),
                                      "frac": round(e2e / PEAK_F32_MFMA_TFLOPS, 4)}
        kernels = {k: {"launches_per_step": round(v["launches"] / nwarm, 2), "ms_per_step": round(v["ms"] / nwarm, 3),
                       "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                       "gather_GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)} for k, v in warm.items()}
        per_step = [t / args.steps * 1e3 for t in reps_max]
        line = {
            "metric": "point-cloud frames/sec at 200k pts/frame", "value": round(total_frames / t_med, 3),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(t_med / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "repeats": {"n": len(per_step), "ms_per_step_median": round(_median(per_step), 3),
                        "ms_per_step_min": round(min(per_step), 3), "ms_per_step_max": round(max(per_step), 3),
                        "timed_seconds_total": round(sum(reps_max), 3),
                        "note": "each repeat = exactly `steps` steps between barrier + synchronize, max over ranks; value and "
                                "ms_per_step are the MEDIAN repeat"},
            "config": {"workload": "cfg2: synthetic 200k-pt RGB-D cloud, 2 cm voxels, RobotNetSegmentation(MinkUNet18D) "
                                   "forward = voxelise + sparse U-Net + slice/argmax, random-init weights",
                       "points_per_frame": POINTS, "frames_per_step": args.frames_per_step,
                       "frames_per_sparse_tensor": group,
                       "grouping": (f"a step is one frame; the pipeline voxelises {group} consecutive resident frames into one "
                                    "sparse tensor (batch column) and runs the network once per group, the reference's own batched "
                                    "format (data/alivev2.py:358-383, TEST.batch_size in config/default.yaml:109); per-frame results "
                                    "are bit-identical to single-frame launches (tests/test_gpu_engine.py); --group 1 = round 1-3's "
                                    "one frame per launch") if group > 1 else "one frame per sparse tensor",
                       "voxel_size_m": 1.0 / SCALE,
                       "active_voxels_per_frame": int(voxels_per_frame),
                       "label_histogram": [int(x) for x in np.diag(agg["confusion"])],
                       "host_placement": pin,
                       "parallelism": f"frame-sharded x{world}, one RCCL all_gather of metrics; per rank: prep stream + "
                                      f"{args.streams} compute stream(s) alternating between {'groups of frames' if group > 1 else 'frames'}"},
            "roofline": roofline,
            "kernels_warmup": kernels,
            "per_rank": {k: [round(x, 4) for x in v] for k, v in agg["per_rank"].items() if not k.startswith("strong_")},
        }
        if strong is not None:
            extra = agg["elapsed_repeats_max"][len(elapsed_list):]
            t_strong = extra[0]
            line["strong_scaling"] = {
                "workload": f"cfg4: a FIXED job of {args.strong_frames} synthetic 200k-pt frames sharded over {world} rank(s) "
                            "(rank r: seeds r, r + N, ...), host arrays in -> labels out; frame source, pinned staging, H2D "
                            "and D2H inside the timed region",
                "scaling": "strong", "total_frames": args.strong_frames, "value": round(args.strong_frames / t_strong, 3),
                "unit": "frames/s", "seconds": round(t_strong, 3), "frames_rank0": strong["frames"],
                "source_threads_per_rank": src_threads, "frames_per_sparse_tensor": group,
                "host_ms_per_frame_rank0": strong["host_ms_per_frame"],
                "per_rank": {k[len("strong_"):]: [round(x, 3) for x in v] for k, v in agg["per_rank"].items()
                             if k.startswith("strong_")},
                "note": "same total job at every N: value(N) / value(1) is the strong-scaling speed-up; per_rank: every "
                        "rank's own elapsed time, frames and host milliseconds per frame and phase"}
        if hbm_layers is not None:
            line["hbm_bound_layers"] = hbm_layers
        if batched is not None:
            line["batched"] = batched
        if ungrouped is not None:
            line["one_frame_per_launch"] = ungrouped
        if other is not None:
            line["other_configs"] = other
        if engine is not None:
            line["engine"] = engine
        if checked is not None:
            _log("accuracy: GPU path vs oracle labels on " + checked["what"])
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            line["accuracy"] = accuracy_block(model, device, checked)
            if world == 1:
                line["cpu_baseline"] = base
                line["cpu_baseline_gather_gemm"] = base_mm
        print(json.dumps(line), flush=True)
    faulthandler.cancel_dump_traceback_later()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
