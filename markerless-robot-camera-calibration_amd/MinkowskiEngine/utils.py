"""ME.utils: batched_coordinates, sparse_quantize, kaiming_normal_ (data/alivev2.py:290-296,363; resnet.py:89)."""
from ctypes import c_int, c_int64, c_size_t

import numpy as np
import torch

from .. import _lib
from ..nn import kaiming_normal_  # noqa: F401


def batched_coordinates(coords, dtype=torch.int32, device=None):
    """Prepend the batch index column: list of [N_i, 3] -> [sum N_i, 4] (batch, x, y, z).
    float dtypes keep the continuous coordinates (TensorField input, app/inference_engine.py:407-410)."""
    out = []
    for b, c in enumerate(coords):
        c = torch.as_tensor(c)
        if not dtype.is_floating_point:
            c = torch.floor(c) if c.dtype.is_floating_point else c
        bc = torch.full((c.shape[0], 1), b, dtype=dtype, device=c.device)
        out.append(torch.cat([bc, c.to(dtype)], dim=1))
    res = torch.cat(out, dim=0) if out else torch.zeros((0, 4), dtype=dtype)
    return res.to(device) if device is not None else res


def sparse_quantize(coordinates, features=None, labels=None, ignore_label=-100, return_index=False,
                    return_inverse=False, return_maps_only=False, quantization_size=None, device="cuda"):
    """floor(coordinates / quantization_size) -> unique voxels in canonical order; the lowest original point index
    represents a voxel; a voxel whose points carry different labels gets ignore_label (data/alivev2.py:290-296)."""
    from ..sparse import _voxelize

    is_np = isinstance(coordinates, np.ndarray)
    c = torch.as_tensor(coordinates)
    if quantization_size is not None:
        c = torch.floor(c.to(torch.float64) / quantization_size)
    elif c.dtype.is_floating_point:
        c = torch.floor(c)
    c = c.to(torch.int32)
    if c.shape[1] == 3:
        c = torch.cat([torch.zeros((c.shape[0], 1), dtype=torch.int32), c], dim=1)
    dev = torch.device(device)
    cd = c.to(dev).contiguous()
    cmap, inverse, order, seg_start = _voxelize(cd, dev, coords_are_int=True)
    first = order[seg_start[:-1].long()].long()  # representative point of each voxel
    discrete = cmap.coords[:, 1:] if coordinates.shape[1] == 3 else cmap.coords

    def back(t):
        t = t.cpu()
        return t.numpy() if is_np else t

    if return_maps_only:
        return (back(first), back(inverse)) if return_inverse else back(first)
    res = [back(discrete)]
    if features is not None:
        f = torch.as_tensor(features)
        res.append(back(f.to(dev)[first]) if not isinstance(features, np.ndarray) else features[first.cpu().numpy()])
    if labels is not None:
        lab = torch.as_tensor(labels).to(dev).long()
        rep = lab[first]
        differs = torch.zeros(cmap.V, dtype=torch.bool, device=dev)
        differs.index_put_((inverse,), lab != rep[inverse], accumulate=True)
        out_lab = torch.where(differs, torch.full_like(rep, ignore_label), rep)
        res.append(back(out_lab.to(torch.as_tensor(labels).dtype)))
    if return_index:
        res.append(back(first))
    if return_inverse:
        res.append(back(inverse))
    return res[0] if len(res) == 1 else tuple(res)
