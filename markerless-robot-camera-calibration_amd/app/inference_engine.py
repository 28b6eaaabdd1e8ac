"""InferenceEngine on the MI355X-native path — mirror of the reference's app/inference_engine.py:46-559.

Same public surface (predict, predict_segmentation, predict_rotation, predict_translation, predict_key_points,
predict_pose_from_kp, check_sanity, calibrate) and the same DTOs; frames go H2D once per stage, every network and
the Kabsch / averaging solves run through libsvhip.  Differences that are deliberate and documented:
  * checkpoints: the reference's paths are not shipped.  `allow_random_init=True` keeps
    `pred_enabled` when a checkpoint is missing so the pipeline can be exercised with seeded random weights;
    with the reference's behaviour (False) a missing checkpoint disables prediction (all-zero segmentation, :281-283).
  * ICP refinement (utils/icp.py) runs on libsvhip (sv_icp_point2point); the CAD model points are passed in
    (`cad_points=`) because the reference's mesh asset does not ship with this build.
  * check_sanity: the reference derives ground-truth key points from the EE crop with utils/data.py:141-335
    get_6_key_points (label synthesis, out of scope); here the expected key points are the six constant
    reference_key_points moved by the predicted EE pose — the same quantity the reference compares against.
  * predict_translation with q=None: the reference reads an unbound `rot_mat` (F8d); here it raises ValueError.
"""
import collections
import os

import numpy as np
import torch

from .. import MinkowskiEngine as ME
from ..model.backbone import minkunet
from ..model.pointnet2 import PointNet2SSG
from ..model.robotnet import make_robotnet, make_robotnet_encode
from ..model.robotnet_segmentation import _classification_head, make_robotnet_segmentation
from ..utils import calibration as calib_util
from ..utils import config, preprocess
from ..utils import output as out_utils
from ..utils.data import get_farthest_point_sample_idx
from ..utils.transformation import (get_base2cam_matrix, get_q_from_matrix,
                                    get_quaternion_rotation_matrix, get_rigid_transform_3D,
                                    get_rigid_transform_3D_batched, transform_pose2pose)
from .dto import CalibrationResultDTO, PointCloudDTO, ResultDTO, TestResultDTO

# app/inference_engine.py:128-137
REFERENCE_KEY_POINTS = np.array([
    [0.01982731, 0.08085986, 0.00321919],
    [0.02171595, -0.08986182, 0.00388430],
    [0.01288678, 0.09103118, 0.06127814],
    [0.02079032, -0.09790908, 0.05609143],
    [-0.00185802, 0.04654205, 0.11564558],
    [0.00241113, -0.04262756, 0.11564558],
])


def checkpoint_restore(model, f=None, device="cuda"):
    """utils/utils.py:87-126 reduced to the explicit-file case: load {"epoch", "model_state_dict", ...};
    returns epoch + 1, or -99 when there is no file."""
    if not f or not os.path.isfile(f):
        return -99
    ckpt = torch.load(f, map_location=device)
    model.load_state_dict(ckpt["model_state_dict"])
    return ckpt.get("epoch", 0) + 1


class InferenceEngine:
    def __init__(self, calibration_only=False, device="cuda", allow_random_init=False, seed=1, cad_points=None):
        self._config = config.Config()
        cfg = self._config
        self.device = torch.device(device)
        # CAD-to-crop ICP (utils/icp.py): the reference samples its CAD points from app/hand_files/hand_notblender.obj,
        # which does not ship with this build -> the caller supplies the model points
        self.match_icp = None
        if cfg.INFERENCE.icp_enabled:
            if cad_points is None:
                raise ValueError("INFERENCE.icp_enabled needs cad_points (the CAD model of the end effector, [P,3])")
            from ..utils.icp import get_point2point_matcher

            self.match_icp = get_point2point_matcher(cad_points, device=self.device)
        self.reference_key_points = REFERENCE_KEY_POINTS.copy()
        self.ee_min_width = abs(self.reference_key_points[0][1] - self.reference_key_points[1][1]) - 0.02
        self.ee_min_height = abs(self.reference_key_points[0][2] - self.reference_key_points[2][2]) - 0.01
        self.camera_link_transformation_pose = cfg.INFERENCE.camera_link_transformation_pose
        if self.camera_link_transformation_pose is not None:
            self.camera_link_transformation_pose = np.array(self.camera_link_transformation_pose, dtype=np.float32)
        if calibration_only:
            return
        self.cluster_util = out_utils.ClusterUtil()
        torch.manual_seed(seed)
        self.pred_enabled = True

        def restore(model, section):
            rc = checkpoint_restore(model, getattr(cfg.INFERENCE, section).checkpoint, self.device)
            ok = rc > -1 or allow_random_init
            self.pred_enabled = self.pred_enabled and ok
            return model.to(self.device).eval()

        seg_cls = make_robotnet_segmentation(cfg.INFERENCE.SEGMENTATION.backbone)
        self._segmentation_model = restore(
            seg_cls(in_channels=cfg.DATA.input_channel, num_classes=cfg.DATA.classes), "SEGMENTATION")
        compute_confidence = cfg()["STRUCTURE"].get("compute_confidence", False)
        rot_cls = (make_robotnet_encode if cfg.INFERENCE.ROTATION.encode_only else make_robotnet)(
            cfg.INFERENCE.ROTATION.backbone)
        self._rotation_model = restore(
            rot_cls(in_channels=cfg.DATA.input_channel, out_channels=(10 if compute_confidence else 7)), "ROTATION")
        kp = cfg.INFERENCE.KEY_POINTS
        if kp.backbone == "pointnet2":
            in_ch = 6 if kp.use_coordinates_as_features else 9
            # the reference feeds cat(points, rgb) = 6 channels either way (app/inference_engine.py:523-528)
            self._key_points_model = restore(PointNet2SSG(num_classes=kp.num_of_keypoints, in_channels=6), "KEY_POINTS")
            del in_ch
        else:
            head = _classification_head(minkunet.MinkUNet18D, lambda: kp.num_of_keypoints, "RobotNetKeyPoints")
            self._key_points_model = restore(
                head(in_channels=cfg.DATA.input_channel, num_classes=kp.num_of_keypoints), "KEY_POINTS")

    # ---- helpers --------------------------------------------------------------------------------------------
    def _field(self, points, feats, scale):
        pts = torch.as_tensor(np.asarray(points), dtype=torch.float32)
        f = feats if torch.is_tensor(feats) else torch.as_tensor(np.asarray(feats), dtype=torch.float32)
        coords = ME.utils.batched_coordinates([pts * scale], dtype=torch.float32)
        return ME.TensorField(features=f.to(torch.float32), coordinates=coords,
                              quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE,
                              minkowski_algorithm=ME.MinkowskiAlgorithm.SPEED_OPTIMIZED, device=self.device)

    # ---- stages (reference :384-559) -------------------------------------------------------------------------
    def predict_pose_from_kp(self, kp_coords, kp_classes):
        if len(kp_classes) < 4:
            return None
        R, t = get_rigid_transform_3D(self.reference_key_points[np.asarray(kp_classes)], np.asarray(kp_coords))
        return np.concatenate((t, get_q_from_matrix(R)))

    def _largest_ee_cluster_rule(self, label, xyz):
        """app/inference_engine.py:419-433 on the device: every EE prediction becomes arm, except the members of the
        largest single-linkage cluster among them.  label: int64 CUDA [N]; xyz: CUDA [N, 3] (float32, as the reference
        clusters `seg_points`, the float32 copy of the raw points, :403)."""
        ee_idx = out_utils.select_equal(label, 2)
        if ee_idx.numel() > 1:
            inside = self.cluster_util.get_largest_cluster(xyz, idx=ee_idx)
            label[ee_idx] = 1
            label[ee_idx[inside]] = 2
        elif ee_idx.numel() == 1:
            label[ee_idx] = 1
        return label

    def _segment(self, x, field):
        return self._segmentation_model(x).slice_argmax(field, with_conf=False)[0]

    def predict_segmentation(self, points, rgb):
        """app/inference_engine.py:395-435.  One frame start to finish, as the reference's consumer calls it (one
        predict() per frame, app/main.py:432-456): pinned staging and asynchronous copies, the frame's coordinate work as
        two libsvhip calls on a prep stream (its size read-backs wait for nothing but that work), the network enqueued
        behind it without a host synchronisation, the decoder's offset-range plans built while the encoder runs
        (app/pipeline.py FramePipeline one_frame).  F8a: the reference centres the points and then voxelises the RAW
        ones (:396-408)."""
        from .pipeline import HostFrameStream

        scale = self._config.INFERENCE.SEGMENTATION.scale
        st = self.__dict__.get("_one_frame_stream")
        if st is None or st.scale != scale:
            st = HostFrameStream(self.device, scale, self._segment, self._largest_ee_cluster_rule, compute_streams=1,
                                 one_frame=True)
            self._one_frame_stream = st
        return st.run_one(points, rgb)

    def predict_segmentation_stream(self, frames, compute_streams=3, group=1):
        """Streaming form of predict_segmentation for a sequence of frames (the reference's consumer is the per-frame
        loop of app/main.py:432-456): `frames` yields (points, rgb) host arrays, the generator yields the label arrays in
        order, each IDENTICAL to predict_segmentation(points, rgb) - while frame i's network runs, frame i+1 is staged
        through pinned memory, uploaded and voxelised, and frame i-1's cluster rule and label download complete
        (app/pipeline.py HostFrameStream).  group > 1: that many consecutive frames share one sparse tensor (the
        reference's batched format, data/alivev2.py:358-383) - the same labels, launches `group` times longer (≈ 4 % more
        frames/s at 4), results delivered a group at a time.  Engine-path throughput: see bench.py's `engine` block."""
        from .pipeline import HostFrameStream

        # one stream object per configuration, kept: its pinned staging buffers and HIP streams are expensive to create
        key = (compute_streams, self._config.INFERENCE.SEGMENTATION.scale, max(1, int(group)))
        streams = self.__dict__.setdefault("_seg_streams", {})
        if key not in streams:
            streams[key] = HostFrameStream(self.device, key[1], self._segment, self._largest_ee_cluster_rule,
                                           compute_streams=compute_streams, group=key[2])
        return streams[key].run(frames)

    # ---- pose stages on end-effector crops (reference :437-559), one or several crops per network run ---------------
    def _crop_runner(self):
        from .pipeline import CropBatchRunner

        r = self.__dict__.get("_crops")
        if r is None:
            # pyramid depth of the deeper of the two pose networks (MinkUNet: 4 stride-2 steps, AliveUNet: 7); a network with
            # fewer levels simply leaves the deeper maps unused
            levels = max(getattr(self._rotation_model, "N_LEVELS", 4), getattr(self._key_points_model, "N_LEVELS", 4))
            r = self._crops = CropBatchRunner(self.device, levels=levels)
        return r

    def _pose_nets_share_voxels(self):
        """the rotation and the key-point network read the same voxelised crop (same centring, scale and colour features:
        the default configuration, config/default.yaml:129-160) - one sparse tensor, one set of maps and plans for both"""
        cfg = self._config
        r, k = cfg.INFERENCE.ROTATION, cfg.INFERENCE.KEY_POINTS
        return (k.backbone != "pointnet2" and r.scale == k.scale and bool(r.center_at_origin) == bool(k.center_at_origin)
                and not k.use_coordinates_as_features)

    def _pose_nets_enqueue(self, crops_pts, crops_rgb, conf_th, one_frame):
        """rotation network AND key-point network (+ its batched selection) on one voxelisation of G crops; returns
        ((rot host view, event, keep), (kp host views, event, keep))"""
        cfg = self._config
        pts = []
        for p in crops_pts:
            p = np.asarray(p)
            if cfg.INFERENCE.ROTATION.center_at_origin:
                p, _ = preprocess.center_at_origin(p)
            pts.append(p)
        G = len(pts)
        runner = self._crop_runner()

        def nets(x, field, seg_start):
            # the two networks only share their input: the rotation network runs on the runner's side stream while the
            # key-point network is enqueued and runs on the main one
            rot = runner.fork(lambda: self._rotation_model(x))
            kp = self._kp_select(x, field, seg_start, G, conf_th)
            runner.join()
            return rot, kp

        rot, kp = runner.run(pts, crops_rgb, cfg.INFERENCE.ROTATION.scale, nets, one_frame=one_frame)
        (host,), ev, keep = runner.download([rot])
        return (host, ev, keep), runner.download(list(kp))

    def _kp_select(self, x, field, seg_start, G, conf_th):
        """key-point network on the voxelised batch x, slice to the points, per-crop selection of utils/output.py:81-87 for
        all crops (sv_key_point_predictions_batched) -> (prob, idx, selected) [G, C] on the device"""
        from ctypes import c_float, c_int, c_int64, c_size_t

        from .._lib import call, ptr, stream_ptr

        logits = self._key_points_model(x).slice(field).features
        C = logits.shape[1]
        dev = logits.device
        buf = torch.empty(G * C, dtype=torch.int64, device=dev)
        idx = torch.empty((G, C), dtype=torch.int64, device=dev)
        prob = torch.empty((G, C), dtype=torch.float32, device=dev)
        sel = torch.empty((G, C), dtype=torch.int32, device=dev)
        segs = (c_int64 * (G + 1))(*seg_start)
        call("sv_key_point_predictions_batched", ptr(logits), c_int64(logits.stride(0)), c_int(C), segs, c_int(G),
             c_float(conf_th), ptr(buf), c_size_t(8 * G * C), ptr(prob), ptr(idx), ptr(sel), stream_ptr())
        return prob, idx, sel

    def _rotation_enqueue(self, crops_pts, crops_rgb, one_frame):
        """the rotation network on G crops as one sparse tensor (batch column = crop): enqueued on the crop stream, its
        [G, 7 or 10] output on its way to pinned memory; returns (host view, event, keep-alive)"""
        cfg = self._config
        pts = []
        for p in crops_pts:
            p = np.asarray(p)
            if cfg.INFERENCE.ROTATION.center_at_origin:
                p, _ = preprocess.center_at_origin(p)
            pts.append(p)
        runner = self._crop_runner()
        out = runner.run(pts, crops_rgb, cfg.INFERENCE.ROTATION.scale, lambda x, field, seg: self._rotation_model(x),
                         encoder_only=bool(cfg.INFERENCE.ROTATION.encode_only), one_frame=one_frame)
        (host,), ev, keep = runner.download([out])
        return host, ev, keep

    def predict_rotation(self, ee_raw_points, ee_rgb):
        host, ev, _ = self._rotation_enqueue([ee_raw_points], [ee_rgb], one_frame=True)
        ev.synchronize()
        return np.array(host[0][3:].numpy())  # quaternion (+ confidences when STRUCTURE.compute_confidence), :446-454

    def predict_translation(self, ee_raw_points, ee_rgb, q=None):
        cfg = self._config
        if q is None:
            raise ValueError("predict_translation needs the predicted quaternion (the reference reads an unbound "
                             "rot_mat when q is None)")
        ee_raw_points = np.asarray(ee_raw_points)
        rot_mat = get_quaternion_rotation_matrix(q, switch_w=False)
        ee_points = ee_raw_points
        if cfg.INFERENCE.TRANSLATION.move_ee_to_origin or cfg.INFERENCE.TRANSLATION.magic_enabled:
            ee_points = (rot_mat.T @ ee_raw_points.reshape((-1, 3, 1))).reshape((-1, 3))
        if cfg.INFERENCE.TRANSLATION.center_at_origin or cfg.INFERENCE.TRANSLATION.magic_enabled:
            ee_pos_points, offset = preprocess.center_at_origin(ee_points)
        else:
            ee_pos_points, offset = ee_points, np.array([0.0, 0.0, 0.0])
        min_z = ee_pos_points.min(axis=0)[2]
        magic = np.array([-0.015, 0.0, min_z]) + offset
        return rot_mat @ magic, offset

    def _key_points_enqueue(self, crops_pts, crops_rgb, conf_th, one_frame):
        """the key-point network (sparse U-Net backbone) on G crops as one sparse tensor, then the per-crop selection of
        utils/output.py:81-87 for all of them (sv_key_point_predictions_batched); returns (host views [prob, idx, selected],
        event, keep-alive)"""
        cfg = self._config
        kp = cfg.INFERENCE.KEY_POINTS
        pts, feats = [], []
        for p, f in zip(crops_pts, crops_rgb):
            p = np.array(np.asarray(p), copy=True)
            if kp.center_at_origin:
                p, _ = preprocess.center_at_origin(p)
            if kp.use_coordinates_as_features:
                f = preprocess.normalize_points(p)
            pts.append(p)
            feats.append(f)
        G = len(pts)

        runner = self._crop_runner()
        outs = runner.run(pts, feats, kp.scale, lambda x, field, seg: self._kp_select(x, field, seg, G, conf_th),
                          one_frame=one_frame)
        return runner.download(list(outs))

    def predict_key_points(self, raw_points, rgb, conf_th=None):
        cfg = self._config
        kp = cfg.INFERENCE.KEY_POINTS
        raw_points = np.asarray(raw_points)
        th = conf_th or kp.conf_threshold
        if kp.backbone != "pointnet2":
            (prob, idx, sel), ev, _ = self._key_points_enqueue([raw_points], [rgb], th, one_frame=True)
            ev.synchronize()
            classes = np.where(sel[0].numpy() != 0)[0]
            return raw_points[idx[0].numpy()[classes]], classes, torch.from_numpy(np.array(prob[0].numpy()[classes]))
        points = np.array(raw_points, copy=True)
        if kp.center_at_origin:
            points, _ = preprocess.center_at_origin(points)
        if kp.use_coordinates_as_features:
            rgb = preprocess.normalize_points(points)
        rgb_t = rgb if torch.is_tensor(rgb) else torch.from_numpy(np.asarray(rgb)).to(torch.float32)
        with torch.no_grad():
            n_dense = cfg.INFERENCE.num_of_dense_input_points
            if len(points) < n_dense:
                return [], [], []
            if kp.pointcloud_sampling_method == "uniform":
                sample_idx = np.random.choice(len(points), n_dense, replace=False)
            else:
                sample_idx = get_farthest_point_sample_idx(points, n_dense)
            pts_t = torch.from_numpy(points).to(torch.float32)
            inp = torch.cat((pts_t[sample_idx], rgb_t.cpu()[sample_idx]), dim=-1).view(1, n_dense, -1)
            out = self._key_points_model(inp.transpose(2, 1).to(self.device))[0].view(n_dense, -1)
            kp_idx, kp_classes, probs = out_utils.get_key_point_predictions(out, conf_th=th)
            kp_idx = sample_idx[kp_idx]
        return raw_points[kp_idx], kp_classes, probs

    def check_sanity(self, data: PointCloudDTO, result: ResultDTO, kp_error_margin=None):
        cfg = self._config
        if kp_error_margin is None:
            kp_error_margin = cfg.INFERENCE.KEY_POINTS.error_margin
        if (result.segmentation == 2).sum() < cfg.INFERENCE.SANITY.min_num_of_ee_points:
            return False
        if result.ee_pose is None:
            return False
        if result.key_points is not None and len(result.key_points) > 3:
            classes, coords = zip(*result.key_points)
            classes = np.array(classes, dtype=np.int64)
            coords = np.array(coords, dtype=np.float32)
            R = get_quaternion_rotation_matrix(np.asarray(result.ee_pose[3:], dtype=np.float64), switch_w=False)
            expected = self.reference_key_points @ R.T + np.asarray(result.ee_pose[:3], dtype=np.float64)
            if np.linalg.norm(expected[classes] - coords, axis=1).mean() > kp_error_margin:
                return False
        return True

    def predict(self, data: PointCloudDTO):
        if not self.pred_enabled:
            return ResultDTO(segmentation=np.zeros(len(data.points), dtype=np.int64))
        rgb = preprocess.normalize_colors(data.rgb)
        seg = self.predict_segmentation(data.points, rgb)
        return self._pose_collect(self._pose_enqueue([(data, rgb, seg)], one_frame=True))[0]

    def predict_stream(self, frames, compute_streams=3, group=4, pose_thread=True, seg_group=1):
        """predict() over a sequence of PointCloudDTOs, everything pipelined: the segmentation stage across frames
        (predict_segmentation_stream), and the pose stages (reference :304-319: rotation network, translation, key-point
        network, key-point selection, Kabsch, base poses) for GROUPS of `group` consecutive frames - their end-effector crops
        form one sparse tensor per network (batch column = frame), the selection and the rigid-transform solves run once per
        group, and a group's results are collected only after the next group's work has been enqueued, so the host never
        waits for a pose kernel while segmentation networks could be launched.  pose_thread: the groups' pose work (its
        launches, and above all its blocking size read-backs and result waits, which sit behind the crop stream's kernels)
        runs on a second host thread, so that the thread feeding the segmentation pipeline never waits for it.  Yields the
        same ResultDTOs, in order, as calling predict() frame by frame (results arrive up to 2 * group - 1 frames after
        their input)."""
        if not self.pred_enabled:
            for data in frames:
                yield ResultDTO(segmentation=np.zeros(len(data.points), dtype=np.int64))
            return
        window = collections.deque()

        def inputs():
            for data in frames:
                rgb = preprocess.normalize_colors(data.rgb)
                window.append((data, rgb))
                yield data.points, rgb

        cur, pending = [], collections.deque()
        # seg_group > 1: the segmentation stage also runs groups of frames per sparse tensor (predict_segmentation_stream)
        seg_stream = self.predict_segmentation_stream(inputs(), compute_streams=compute_streams, group=seg_group)
        if not pose_thread:
            for seg in seg_stream:
                data, rgb = window.popleft()
                cur.append((data, rgb, seg))
                if len(cur) >= group:
                    pending.append(self._pose_enqueue(cur, one_frame=False))
                    cur = []
                    while len(pending) > 1:
                        yield from self._pose_collect(pending.popleft())
            if cur:
                pending.append(self._pose_enqueue(cur, one_frame=False))
            while pending:
                yield from self._pose_collect(pending.popleft())
            return
        import queue
        import threading

        jobs, done = queue.Queue(), queue.Queue()

        def worker():
            try:
                if self.device.index is not None:
                    torch.cuda.set_device(self.device)
                waiting = None  # one group enqueued ahead of the one being collected
                while True:
                    items = jobs.get()
                    nxt = self._pose_enqueue(items, one_frame=False) if items is not None else None
                    if waiting is not None:
                        done.put(self._pose_collect(waiting))
                    waiting = nxt
                    if items is None:
                        return
            except BaseException as e:  # whatever happens on this thread is handed to the consumer, which re-raises it
                done.put(e)

        th = threading.Thread(target=worker, name="mrcc-pose", daemon=True)
        th.start()
        outstanding = 0

        def ready(block):
            nonlocal outstanding
            while outstanding:
                try:
                    r = done.get(block=block, timeout=1.0 if block else None)
                except queue.Empty:
                    if not block:
                        return
                    if not th.is_alive() and done.empty():  # never wait for a thread that is gone
                        raise RuntimeError("the pose thread ended without delivering its results")
                    continue
                if isinstance(r, BaseException):
                    raise r
                outstanding -= 1
                yield from r

        try:
            for seg in seg_stream:
                data, rgb = window.popleft()
                cur.append((data, rgb, seg))
                if len(cur) >= group:
                    jobs.put(cur)
                    outstanding += 1
                    cur = []
                yield from ready(block=False)
            if cur:
                jobs.put(cur)
                outstanding += 1
            jobs.put(None)
            yield from ready(block=True)
        finally:
            if th.is_alive():
                jobs.put(None)
            th.join(timeout=60)

    def _predict_after_segmentation(self, data, rgb, seg):
        return self._pose_collect(self._pose_enqueue([(data, rgb, seg)], one_frame=True))[0]

    def _pose_enqueue(self, items, one_frame):
        """items: [(PointCloudDTO, normalised colours, labels)].  Crops the end effector of every frame (host: the labels
        are host arrays, as in the reference :297-303) and enqueues the rotation and key-point networks for all crops that
        pass the point-count threshold; nothing here waits for the GPU."""
        cfg = self._config
        crops = []
        for data, rgb, seg in items:
            ee_idx = np.where(seg == 2)[0]
            if len(ee_idx) < cfg.INFERENCE.ee_point_counts_threshold:
                crops.append(None)
            else:
                crops.append((data.points[ee_idx], rgb[ee_idx]))
        live = [c for c in crops if c is not None]
        handle = {"items": items, "crops": crops, "rot": None, "kp": None}
        if live:
            pts, cols = [c[0] for c in live], [c[1] for c in live]
            if self._pose_nets_share_voxels():
                handle["rot"], handle["kp"] = self._pose_nets_enqueue(pts, cols, cfg.INFERENCE.KEY_POINTS.conf_threshold,
                                                                      one_frame)
            else:
                handle["rot"] = self._rotation_enqueue(pts, cols, one_frame)
                if cfg.INFERENCE.KEY_POINTS.backbone != "pointnet2":
                    handle["kp"] = self._key_points_enqueue(pts, cols, cfg.INFERENCE.KEY_POINTS.conf_threshold, one_frame)
        return handle

    def _pose_collect(self, handle):
        """wait for a group's networks (one event each), then the host side of the pose stages per frame and the group's
        rigid-transform problems - key-point Kabsch and the quaternions of the base poses - as two batched solves"""
        cfg = self._config
        results = []
        rot = kp = None
        if handle["rot"] is not None:
            host, ev, _ = handle["rot"]
            ev.synchronize()
            rot = np.array(host.numpy())
        if handle["kp"] is not None:
            (prob, idx, sel), ev, _ = handle["kp"]
            ev.synchronize()
            kp = (np.array(prob.numpy()), np.array(idx.numpy()), np.array(sel.numpy()))
        j = 0
        work = []  # (result, data, ee_pts) of the frames that have a crop
        for (data, rgb, seg), crop in zip(handle["items"], handle["crops"]):
            result = ResultDTO(segmentation=seg)
            results.append(result)
            if crop is None:
                continue
            ee_pts, ee_rgb = crop
            q = rot[j][3:]
            pos, _ = self.predict_translation(ee_pts, None, q=q)
            result.ee_pose = np.concatenate((pos, q))
            if kp is not None:
                classes = np.where(kp[2][j] != 0)[0]
                kp_coords = ee_pts[kp[1][j][classes]]
            else:
                kp_coords, classes, _ = self.predict_key_points(ee_pts, torch.from_numpy(ee_rgb).to(torch.float32))
            result.key_points = list(zip(classes, kp_coords))
            work.append((result, data, ee_pts, np.asarray(classes), np.asarray(kp_coords)))
            j += 1
        if not work:
            return results
        # ---- solve 1: Kabsch of every frame with >= 4 key points (:384-393)
        probs = [(self.reference_key_points[c], k) for _, _, _, c, k in work if len(c) >= 4]
        sol = iter(self._solve_rigid(probs))
        for result, data, ee_pts, classes, kp_coords in work:
            if len(classes) >= 4:
                _, t, q = next(sol)
                result.key_points_pose = np.concatenate((t, q))
            result.is_confident = self.check_sanity(data, result)
            if self.match_icp is not None:  # app/inference_engine.py:358-362
                result.ee_pose = self.match_icp(ee_pts, result.ee_pose)
                result.key_points_pose = self.match_icp(ee_pts, result.key_points_pose)
        # ---- solve 2: the base poses' quaternions (get_base2cam_pose -> get_q_from_matrix), all frames at once
        mats = []
        for result, data, *_ in work:
            if data.ee2base_pose is None:
                continue
            for pose in (result.ee_pose, result.key_points_pose):  # the ICP step may have rejected a pose (:364-369)
                if pose is not None:
                    mats.append(get_base2cam_matrix(pose, data.ee2base_pose))
        eye = np.concatenate([np.eye(3), np.zeros((1, 3))])  # get_q_from_matrix: identity points -> rows of R
        sol = iter(self._solve_rigid([(eye, eye @ m[:3, :3].T) for m in mats]))
        mi = iter(mats)
        for result, data, *_ in work:
            if data.ee2base_pose is None:
                continue
            if result.ee_pose is not None:
                result.base_pose = np.concatenate((next(mi)[:3, 3], next(sol)[2]))
            if result.key_points_pose is not None:
                result.key_points_base_pose = np.concatenate((next(mi)[:3, 3], next(sol)[2]))
        return results

    def _solve_rigid(self, problems):
        """[(reference [K, 3], target [K, 3])] -> [(R, t, q)]: ONE sv_kabsch_batched launch (K <= 6 padded to 6 rows; a
        problem's result does not depend on what else is in the batch)"""
        if not problems:
            return []
        B = len(problems)
        ref = np.zeros((B, 6, 3))
        tgt = np.zeros((B, 6, 3))
        K = np.zeros(B, dtype=np.int32)
        for b, (a, t) in enumerate(problems):
            K[b] = len(a)
            ref[b, : len(a)] = a
            tgt[b, : len(a)] = t
        R, t, q = get_rigid_transform_3D_batched(ref, tgt, K=K, device=self.device)
        return [(R[b], t[b], q[b]) for b in range(B)]

    # ---- calibration (reference :152-244) ---------------------------------------------------------------------
    def calibrate(self, data) -> CalibrationResultDTO:
        individual = [self._calibrate_individual(v) for v in data.values()]
        individual = [v for v in individual if v is not None]
        if len(data) == 1 and len(individual) > 0:
            raw = individual[0]
        else:
            raw = self._calibrate_individual(individual)
            if raw is None:
                return CalibrationResultDTO(pose_camera_link=None)
        stack = np.stack((raw.base_pose, raw.key_points_base_pose), axis=0)
        calibration = CalibrationResultDTO(pose_camera_link=calib_util.compute_poses_average(stack))
        calibration.load_from_test_result(raw)
        return calibration

    def _calibrate_individual(self, data, weights=None, confident_count=2):
        result = TestResultDTO(segmentation=None, is_confident=True)
        try:
            confident = [d for d in data if d.is_confident]
            if len(confident) < confident_count:
                return None
            if weights is not None:
                weights = weights[np.array([d.is_confident for d in data], dtype=bool)]

            def avg(values):
                arr = np.array([v for v in values if v is not None], dtype=np.float32)
                return calib_util.compute_poses_average(calib_util.remove_pose_outliers(arr), weights=weights)

            result.ee_pose = avg(d.ee_pose for d in confident)
            result.base_pose = avg(d.base_pose for d in confident)
            result.key_points_pose = avg(d.key_points_pose for d in confident)
            result.key_points_base_pose = avg(d.key_points_base_pose for d in confident)
            cl = self.camera_link_transformation_pose
            if isinstance(confident[0], TestResultDTO):
                result.base_pose_camera_link = avg(d.base_pose_camera_link for d in confident)
                result.key_points_base_pose_camera_link = avg(d.key_points_base_pose_camera_link for d in confident)
            elif cl is not None:
                result.base_pose_camera_link = avg(transform_pose2pose(d.base_pose, cl) for d in confident
                                                   if d.base_pose is not None)
                result.key_points_base_pose_camera_link = avg(
                    transform_pose2pose(d.key_points_base_pose, cl) for d in confident
                    if d.key_points_base_pose is not None)
        except Exception:
            result.is_confident = False
        return result
