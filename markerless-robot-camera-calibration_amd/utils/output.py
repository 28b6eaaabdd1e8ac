"""Post-ops of the reference (utils/output.py:45-87) on device tensors."""
import numpy as np
import torch


def get_segmentations_from_tensor_field(field):
    """row max -> label, sigmoid(max) -> confidence (utils/output.py:67-73).  Callers that still hold the sparse
    output should prefer SparseTensor.slice_argmax(field), which fuses slice + this into one kernel."""
    logits = field.features
    conf, preds = logits.max(1)
    return preds.cpu().numpy(), torch.sigmoid(conf).cpu().numpy()


def get_key_point_predictions(logits, conf_th=0.999):
    """softmax over classes, max over points per class, threshold (utils/output.py:81-87)."""
    softmax = logits.softmax(1).max(0)
    classes = np.where(softmax[0].cpu() > conf_th)[0]
    idx = softmax[1][classes].cpu().numpy()
    return idx, classes, softmax[0].cpu()[classes]


def get_pred_center(out, coords, ee_r=0.03, q=None):
    """mean of the 8 highest-vote points (utils/output.py:45-64, without the optional quaternion offset)."""
    sel = out[:, 1].sort(descending=True)[1][:8]
    return np.asarray(coords)[sel.cpu().numpy()].mean(axis=0)
