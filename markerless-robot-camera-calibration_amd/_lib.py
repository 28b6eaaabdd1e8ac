"""ctypes binding of libsvhip.so (the C-ABI declared in include/sv_hip.h).

There is NO fallback: if the HIP library is missing or a call fails, the product path raises.  PyTorch-ROCm is used
only to own device memory and streams; every computation on the hot path is a call into the library.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 4  # SV_ABI_VERSION of include/sv_hip.h
LIB_PATH = os.environ.get("SVHIP_LIB") or os.path.join(_HERE, "libsvhip.so")  # SVHIP_LIB: kernel A/B experiments

SV_ACT_NONE, SV_ACT_RELU, SV_ACT_LEAKY_RELU = 0, 1, 2
SV_POOL_MAX, SV_POOL_AVG = 0, 1
SV_REDUCE_MEAN, SV_REDUCE_FIRST = 0, 1
SV_TILE_ROWS = 128
SV_COORD_BIAS = 1 << 17
SV_COORD_BITS = 18
SV_MAX_BATCH = 1024
SV_FRAME_MAX_LEVELS, SV_FRAME_MAX_CUTS, SV_FRAME_RECORD = 8, 4, 16
SV_FRAME_K3, SV_FRAME_DOWN, SV_FRAME_UP, SV_FRAME_SPLIT = 1, 2, 4, 8
SV_FRAME_REC_HASH, SV_FRAME_REC_K3, SV_FRAME_REC_DOWN, SV_FRAME_REC_UP, SV_FRAME_REC_SPLIT = 1, 2, 3, 4, 5


class SvHipError(RuntimeError):
    pass


# name -> (restype, argtypes); kept in one table so tests can check that every symbol of include/sv_hip.h is exported
_P = c_void_p
SIGNATURES = {
    "sv_last_error": (c_char_p, []),
    "sv_abi_version": (c_int, []),
    "sv_voxelize_workspace_bytes": (c_size_t, [c_int64]),
    "sv_voxelize": (c_int, [_P, c_int, c_int64, _P, c_size_t, _P, _P, _P, _P, _P, _P, _P]),
    "sv_voxel_reduce": (c_int, [_P, c_int, _P, _P, c_int64, c_int, _P, _P]),
    "sv_hash_build": (c_int, [_P, c_int64, _P, _P, c_int64, _P]),
    "sv_stride_map_workspace_bytes": (c_size_t, [c_int64]),
    "sv_stride_map": (c_int, [_P, c_int64, c_int, _P, c_size_t, _P, _P, _P, _P, _P, _P]),
    "sv_kernel_map_k3": (c_int, [_P, c_int64, c_int, c_int, _P, _P, c_int64, _P, c_int64, _P, _P]),
    "sv_kernel_map_down": (c_int, [_P, _P, c_int64, c_int, c_int64, _P, c_int64, _P, _P]),
    "sv_kernel_map_up": (c_int, [_P, _P, c_int64, c_int, _P, c_int64, _P, _P]),
    "sv_plan_workspace_bytes": (c_size_t, [c_int64]),
    "sv_plan_build": (c_int, [_P, c_int64, _P, c_int, c_int64, c_int64, _P, c_size_t, _P, _P, _P, _P, c_int64, _P]),
    "sv_conv_fwd": (
        c_int,
        [_P, c_int64, c_int64, c_int, _P, c_int, c_int, _P, _P, _P, _P, c_int64, c_int64, _P, _P, _P, c_int64, c_int,
         c_float, _P, c_int64, _P],
    ),
    "sv_conv_fwd_acc": (
        c_int,
        [_P, c_int64, c_int64, c_int, _P, c_int, c_int, _P, _P, _P, _P, c_int64, c_int64, _P, c_int64, _P, _P, _P, c_int64, c_int,
         c_float, _P, c_int64, _P],
    ),
    "sv_conv_last_instance": (c_char_p, []),
    "sv_conv_set_dispatch": (c_int, [c_double, c_double]),
    "sv_frame_maps_arena_bytes": (c_size_t, [c_int64, c_int]),
    "sv_frame_maps_scratch_bytes": (c_size_t, [c_int64]),
    "sv_frame_maps": (c_int, [_P, c_int, c_int64, c_int, _P, c_size_t, _P, c_size_t, _P, _P, _P]),
    "sv_frame_plans_arena_bytes": (c_size_t, [_P, c_int, c_int, _P]),
    "sv_frame_plans_scratch_bytes": (c_size_t, [_P, c_int]),
    "sv_frame_plans": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, c_size_t, _P, c_size_t, _P, c_int, _P]),
    "sv_affine_act": (c_int, [_P, c_int64, c_int, c_int64, _P, _P, _P, c_int64, c_int, c_float, _P, c_int64, _P]),
    "sv_col_stats_workspace_bytes": (c_size_t, [c_int64]),
    "sv_col_stats": (c_int, [_P, c_int64, c_int64, c_int, _P, _P, c_size_t, _P, _P, _P, _P, _P]),
    "sv_center_scale": (c_int, [_P, c_int64, c_int64, c_int, _P, _P, _P, _P, c_int64, _P]),
    "sv_batch_offsets": (c_int, [_P, c_int64, c_int, _P, _P]),
    "sv_global_pool": (c_int, [_P, c_int64, c_int, _P, c_int, c_int, _P, _P]),
    "sv_slice_rows": (c_int, [_P, c_int64, c_int, _P, c_int64, _P, _P]),
    "sv_slice_argmax": (c_int, [_P, c_int64, c_int, _P, c_int64, _P, _P, _P]),
    "sv_key_point_predictions": (c_int, [_P, c_int64, c_int, c_int64, c_float, _P, c_size_t, _P, _P, _P, _P]),
    "sv_key_point_predictions_batched": (c_int, [_P, c_int64, c_int, _P, c_int, c_float, _P, c_size_t, _P, _P, _P, _P]),
    "sv_topk_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "sv_topk_indices": (c_int, [_P, c_int64, c_int64, c_int, _P, c_size_t, _P, _P]),
    "sv_kabsch_batched": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P, _P]),
    "sv_quat_avg_batched": (c_int, [_P, _P, _P, c_int, c_int, _P, _P]),
    "sv_add_metric_batched": (c_int, [_P, _P, c_int, _P, _P, c_int, _P, _P]),
    "sv_icp_workspace_bytes": (c_size_t, [c_int64]),
    "sv_icp_point2point": (c_int, [_P, c_int64, _P, c_int64, _P, c_double, c_int, c_double, c_double, _P, c_size_t, _P, _P,
                                   _P]),
    "sv_fps": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "sv_three_nn_interpolate": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "sv_cluster_workspace_bytes": (c_size_t, [c_int64]),
    "sv_single_linkage_roots": (c_int, [_P, c_int, c_int64, _P, c_int64, c_double, _P, c_size_t, _P, _P, _P]),
    "sv_select_equal": (c_int, [_P, c_int, c_int64, c_int64, _P, _P, _P, _P]),
    "sv_ball_query": (c_int, [_P, _P, c_int, c_int, c_int, c_double, c_int, _P, _P]),
}

_lib = None


def load():
    """Load libsvhip.so (once).  Raises SvHipError when it has not been built: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SvHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C markerless-robot-camera-calibration_amd/csrc`). There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.sv_abi_version.restype = c_int
    if lib.sv_abi_version() != ABI_VERSION:  # a stale build would misread the arguments of a call whose signature moved
        raise SvHipError(f"{LIB_PATH} has C-ABI version {lib.sv_abi_version()}, this package binds version {ABI_VERSION}: "
                         "rebuild it (`python -c 'import __graft_entry__ as g; g.build()'`)")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc, name):
    if rc != 0:
        msg = _lib.sv_last_error().decode("utf-8", "replace")
        raise SvHipError(f"{name} failed with status {rc}: {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def stream_ptr():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    _check(rc, name)


def conv_last_instance():
    """(kernel instance name, {"fast": 0/1, "ring": 0/1, "full": 0/1}) of this thread's last sv_conv_fwd launch."""
    raw = load().sv_conv_last_instance().decode()
    name, _, flags = raw.partition("|")
    return name, {k: int(v) for k, v in (kv.split("=") for kv in flags.split(",") if kv)}


class conv_dispatch:
    """`with conv_dispatch(want_scale):` - dispatch thresholds of this thread's sv_conv_fwd calls inside the block
    (sv_conv_set_dispatch; 1.0 = one frame alone on the GPU), library defaults restored afterwards."""

    def __init__(self, want_scale, tail_fraction=-1.0):
        self.args = (c_double(want_scale), c_double(tail_fraction))

    def __enter__(self):
        call("sv_conv_set_dispatch", *self.args)

    def __exit__(self, *exc):
        call("sv_conv_set_dispatch", c_double(-1.0), c_double(-1.0))


def require_cuda(t, what="tensor"):
    if not t.is_cuda:
        raise SvHipError(f"{what} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    return t
