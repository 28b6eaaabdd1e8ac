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


def _qmul(q, r):
    """Hamilton product, (w, x, y, z) (utils/quaternion.py qmul_np)."""
    w0, x0, y0, z0 = q
    w1, x1, y1, z1 = r
    return np.array([w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1, w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1,
                     w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1, w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1])


def compute_pose_metrics(gt, pred):
    """utils/metrics.py:110-127: position error (m) and the rotation angle of gt * conj(pred) (rad, in [0, pi])."""
    gt, pred = np.asarray(gt, dtype=np.float64), np.asarray(pred, dtype=np.float64)
    gt_rot = gt[3:7] / np.linalg.norm(gt[3:7])
    pred_rot = pred[3:7] / np.linalg.norm(pred[3:7])
    q = _qmul(gt_rot, pred_rot * np.array([1.0, -1.0, -1.0, -1.0]))
    angle = np.abs(2 * np.arctan2(np.linalg.norm(q[1:]), q[0]))
    return {"dist_position": float(np.linalg.norm(gt[:3] - pred[:3])),
            "angle_diff": float(min(angle, 2 * np.pi - angle))}


def compute_kp_error(gt_coords, kp_coords, kp_classes):
    """utils/metrics.py:130-136: mean distance between predicted key points and the ground-truth ones of their class;
    100 when fewer than two are available."""
    if len(gt_coords) < 2 or len(kp_coords) < 2 or len(kp_classes) < 2:
        return 100
    return float(np.linalg.norm(np.asarray(gt_coords)[np.asarray(kp_classes)] - np.asarray(kp_coords), axis=1).mean())


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
    """utils/metrics.py:51-107, quirks included: per class accuracy/precision/recall with precision = 1 when there is no
    false positive (recall = 1 when no false negative); overall "accuracy" = (sensitivity + specificity) / 2 over the
    summed one-vs-rest counts; overall precision / recall = plain means over classes.  `miou` is this build's addition
    (BASELINE.json asks for it; the reference never computes IoU on this path)."""
    gt = np.asarray(gt)
    pred = np.asarray(pred)
    n = len(gt)
    results = {"class_results": {}}
    precisions, recalls, ious = [], [], []
    tp_s = tn_s = fp_s = fn_s = 0
    for ci, cn in enumerate(classes):
        g = gt == ci
        p = pred == ci
        tp = int((g & p).sum())
        tn = int(n - (g | p).sum())
        fp = int(p.sum()) - tp
        fn = int(g.sum()) - tp
        tp_s, tn_s, fp_s, fn_s = tp_s + tp, tn_s + tn, fp_s + fp, fn_s + fn
        precision = 1 if fp == 0 else tp / (tp + fp)
        recall = 1 if fn == 0 else tp / (tp + fn)
        results["class_results"][cn] = {"accuracy": (tp + tn) / (tp + tn + fp + fn), "precision": precision,
                                        "recall": recall}
        precisions.append(precision)
        recalls.append(recall)
        if tp + fp + fn > 0:
            ious.append(tp / (tp + fp + fn))
    sensitivity = tp_s / (tp_s + fn_s)
    specificity = tn_s / (tn_s + fp_s)
    results["accuracy"] = (sensitivity + specificity) / 2
    results["precision"] = float(np.mean(precisions))
    results["recall"] = float(np.mean(recalls))
    results["miou"] = float(np.mean(ious)) if ious else float("nan")
    return results
