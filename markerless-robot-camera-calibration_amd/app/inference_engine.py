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
from ..utils import config, metrics, preprocess
from ..utils import output as out_utils
from ..utils.data import get_farthest_point_sample_idx
from ..utils.transformation import (get_base2cam_pose, get_q_from_matrix, get_quaternion_rotation_matrix,
                                    get_rigid_transform_3D, transform_pose2pose)
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

    def predict_segmentation_stream(self, frames, compute_streams=3):
        """Streaming form of predict_segmentation for a sequence of frames (the reference's consumer is the per-frame
        loop of app/main.py:432-456): `frames` yields (points, rgb) host arrays, the generator yields the label arrays in
        order, each IDENTICAL to predict_segmentation(points, rgb) - while frame i's network runs, frame i+1 is staged
        through pinned memory, uploaded and voxelised, and frame i-1's cluster rule and label download complete
        (app/pipeline.py HostFrameStream).  Engine-path throughput: see bench.py's `engine` block."""
        from .pipeline import HostFrameStream

        # one stream object per configuration, kept: its pinned staging buffers and HIP streams are expensive to create
        key = (compute_streams, self._config.INFERENCE.SEGMENTATION.scale)
        streams = self.__dict__.setdefault("_seg_streams", {})
        if key not in streams:
            streams[key] = HostFrameStream(self.device, key[1], self._segment, self._largest_ee_cluster_rule,
                                           compute_streams=compute_streams)
        return streams[key].run(frames)

    def predict_rotation(self, ee_raw_points, ee_rgb):
        cfg = self._config
        pts = np.asarray(ee_raw_points)
        if cfg.INFERENCE.ROTATION.center_at_origin:
            pts, _ = preprocess.center_at_origin(pts)
        with torch.no_grad():
            x = self._field(pts, ee_rgb, cfg.INFERENCE.ROTATION.scale).sparse()
            out = self._rotation_model(x)
        return out[0][3:].cpu().numpy()  # quaternion (+ confidences when STRUCTURE.compute_confidence), :446-454

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

    def predict_key_points(self, raw_points, rgb, conf_th=None):
        cfg = self._config
        kp = cfg.INFERENCE.KEY_POINTS
        raw_points = np.asarray(raw_points)
        points = np.array(raw_points, copy=True)
        if kp.center_at_origin:
            points, _ = preprocess.center_at_origin(points)
        if kp.use_coordinates_as_features:
            rgb = preprocess.normalize_points(points)
        rgb_t = rgb if torch.is_tensor(rgb) else torch.from_numpy(np.asarray(rgb)).to(torch.float32)
        th = conf_th or kp.conf_threshold
        with torch.no_grad():
            if kp.backbone == "pointnet2":
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
            else:
                field = self._field(points, rgb_t, kp.scale)
                out = self._key_points_model(field.sparse())
                logits = out.slice(field).features
                kp_idx, kp_classes, probs = out_utils.get_key_point_predictions(logits, conf_th=th)
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
        return self._predict_after_segmentation(data, rgb, self.predict_segmentation(data.points, rgb))

    def predict_stream(self, frames, compute_streams=3):
        """predict() over a sequence of PointCloudDTOs with the segmentation stage pipelined across frames
        (predict_segmentation_stream); the pose stages of a frame (end-effector crop: a few thousand points) run when its
        labels arrive, while the following frames' segmentation networks are already on the GPU.  Yields the same
        ResultDTOs, in order, as calling predict() frame by frame."""
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

        for seg in self.predict_segmentation_stream(inputs(), compute_streams=compute_streams):
            data, rgb = window.popleft()
            yield self._predict_after_segmentation(data, rgb, seg)

    def _predict_after_segmentation(self, data, rgb, seg):
        cfg = self._config
        result = ResultDTO(segmentation=seg)
        ee_idx = np.where(seg == 2)[0]
        if len(ee_idx) < cfg.INFERENCE.ee_point_counts_threshold:
            return result
        ee_pts = data.points[ee_idx]
        ee_rgb = torch.from_numpy(rgb[ee_idx]).to(dtype=torch.float32)
        q = self.predict_rotation(ee_pts, ee_rgb)
        pos, _ = self.predict_translation(ee_pts, ee_rgb, q=q)
        result.ee_pose = np.concatenate((pos, q))
        kp_coords, kp_classes, _ = self.predict_key_points(ee_pts, ee_rgb)
        result.key_points = list(zip(kp_classes, kp_coords))
        result.key_points_pose = self.predict_pose_from_kp(kp_coords, kp_classes)
        result.is_confident = self.check_sanity(data, result)
        if self.match_icp is not None:  # app/inference_engine.py:358-362
            result.ee_pose = self.match_icp(ee_pts, result.ee_pose)
            result.key_points_pose = self.match_icp(ee_pts, result.key_points_pose)
        if data.ee2base_pose is not None:
            if result.ee_pose is not None:  # the ICP step may have rejected the pose (:364-369)
                result.base_pose = get_base2cam_pose(result.ee_pose, data.ee2base_pose)
            if result.key_points_pose is not None:
                result.key_points_base_pose = get_base2cam_pose(result.key_points_pose, data.ee2base_pose)
        return result

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
