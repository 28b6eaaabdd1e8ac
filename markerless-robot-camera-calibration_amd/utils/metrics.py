"""Metric definitions of the reference (utils/metrics.py), ADD on libsvhip.

compute_ADD_np            utils/metrics.py:139-150   (the reported pose accuracy)
compute_rotational_diff   :153-165, compute_translational_diff :168-176
compute_segmentation_metrics  :51-107 (accuracy / precision / recall per class, balanced accuracy) + mIoU, which the
                          reference never computes on this path (BASELINE.json asks for it).
"""
from ctypes import c_int

import numpy as np
import torch

from .._lib import call, ptr, stream_ptr


def compute_ADD_batched(points, P, gt_pose, pred_pose, device=None):
    """points [B,Pmax,3], P int[B], poses [B,7] (x,y,z,qw,qx,qy,qz) -> ADD [B] float64."""
    dev = torch.device("cuda" if device is None else device)
    pts = torch.as_tensor(np.ascontiguousarray(points, dtype=np.float64)).to(dev)
    B, Pmax, _ = pts.shape
    Pt = None if P is None else torch.as_tensor(np.asarray(P, dtype=np.int32)).to(dev)
    gt = torch.as_tensor(np.ascontiguousarray(gt_pose, dtype=np.float64)).to(dev)
    pr = torch.as_tensor(np.ascontiguousarray(pred_pose, dtype=np.float64)).to(dev)
    out = torch.empty(B, dtype=torch.float64, device=dev)
    call("sv_add_metric_batched", ptr(pts), ptr(Pt), c_int(Pmax), ptr(gt), ptr(pr), c_int(B), ptr(out), stream_ptr())
    return out.cpu().numpy()


def compute_ADD_np(points, gt_pose, pred_pose):
    return float(compute_ADD_batched(np.asarray(points)[None], None, np.asarray(gt_pose)[None],
                                     np.asarray(pred_pose)[None])[0])


def compute_rotational_diff(q1, q2, degree=True):
    diff = 2 * np.arccos(min(1.0, abs(np.sum(np.asarray(q1) * np.asarray(q2)))))
    return diff * 57.2958 if degree else diff


def compute_translational_diff(t1, t2, cm=True, method="euclidean"):
    dist = np.linalg.norm(np.asarray(t1) - np.asarray(t2)) if method == "euclidean" else -1
    return dist * 100 if cm else dist


def compute_pose_metrics(gt_pose, pred_pose):
    """utils/metrics.py:110-127 reduced to the two numbers app/test.py reports: position (m) and angle (rad) error."""
    gt_pose, pred_pose = np.asarray(gt_pose, dtype=np.float64), np.asarray(pred_pose, dtype=np.float64)
    return {"dist_position": float(np.linalg.norm(gt_pose[:3] - pred_pose[:3])),
            "angle_diff": float(compute_rotational_diff(gt_pose[3:7], pred_pose[3:7], degree=False))}


def confusion_matrix(pred, gt, num_classes):
    pred = np.asarray(pred).astype(np.int64)
    gt = np.asarray(gt).astype(np.int64)
    valid = (gt >= 0) & (gt < num_classes)
    return np.bincount(gt[valid] * num_classes + pred[valid], minlength=num_classes ** 2).reshape(num_classes,
                                                                                                  num_classes)


def segmentation_metrics_from_confusion(cm):
    """accuracy, per-class precision / recall / IoU, mIoU, balanced accuracy ((sens + spec) / 2, metrics.py:98-102)."""
    cm = np.asarray(cm, dtype=np.float64)
    tp = np.diag(cm)
    fp = cm.sum(axis=0) - tp
    fn = cm.sum(axis=1) - tp
    tn = cm.sum() - tp - fp - fn
    with np.errstate(divide="ignore", invalid="ignore"):
        prec = np.where(tp + fp > 0, tp / (tp + fp), 0.0)
        rec = np.where(tp + fn > 0, tp / (tp + fn), 0.0)
        iou = np.where(tp + fp + fn > 0, tp / (tp + fp + fn), np.nan)
        spec = np.where(tn + fp > 0, tn / (tn + fp), 0.0)
    return {"accuracy": float(tp.sum() / max(cm.sum(), 1)), "precision": prec, "recall": rec, "iou": iou,
            "miou": float(np.nanmean(iou)), "balanced_accuracy": (rec + spec) / 2}


def compute_segmentation_metrics(gt, pred, classes=("background", "arm", "ee")):
    return segmentation_metrics_from_confusion(confusion_matrix(pred, gt, len(classes)))
