"""The C-ABI library loads and exports every symbol include/sv_hip.h declares (no compute calls: runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sv_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sv_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    import mrcc_amd

    lib = mrcc_amd._lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 24
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/sv_hip.h but not exported by libsvhip.so"
    # the ctypes table covers exactly the header
    assert sorted(mrcc_amd._lib.SIGNATURES) == declared
    assert lib.sv_abi_version() == 4 == mrcc_amd._lib.ABI_VERSION


def test_argument_validation_without_gpu():
    """Shape / pointer validation happens on the host before any HIP call, so it can be exercised on CPU."""
    import mrcc_amd

    lib = mrcc_amd._lib.load()
    rc = lib.sv_conv_fwd(None, 10, 4, 4, None, 27, 4, None, None, None, None, 10, 128, None, None, None, 0, 0,
                         ctypes.c_float(0.0), None, 4, None)
    assert rc == -1 and b"null pointer" in lib.sv_last_error()
    rc = lib.sv_conv_fwd(None, 10, 4, 4, None, 1, 4, None, None, None, None, 10, 100, None, None, None, 0, 0,
                         ctypes.c_float(0.0), None, 4, None)
    assert rc == -1 and b"multiple of 128" in lib.sv_last_error()
    rc = lib.sv_hash_build(None, 10, None, None, 24, None)
    assert rc == -1 and b"power of two" in lib.sv_last_error()
    rc = lib.sv_fps(None, 1, 100000, 16, None, None, None)
    assert rc == -1 and b"too large" in lib.sv_last_error()
    assert lib.sv_voxelize_workspace_bytes(200000) > 200000 * 8 * 2
    assert lib.sv_plan_workspace_bytes(88000) > 88000 * 4 * 3
    # frame composites (ABI v4): argument checks and arena sizes are host code
    rc = lib.sv_frame_maps(None, 0, 0, 4, None, 0, None, 0, None, None, None)
    assert rc == -1 and b"at least one point" in lib.sv_last_error()
    rc = lib.sv_frame_maps(None, 0, 100, 99, None, 0, None, 0, None, None, None)
    assert rc == -1 and b"levels out of range" in lib.sv_last_error()
    n = 200_000
    assert lib.sv_frame_maps_arena_bytes(n, 4) >= n * (8 + 16 + 8 + 4 + 4) + 4 * n * (8 + 16 + 4 + 4)  # worst case V_l <= N
    assert lib.sv_frame_maps_scratch_bytes(n) >= lib.sv_voxelize_workspace_bytes(n)
    V = (ctypes.c_int64 * 5)(88113, 26552, 6849, 1732, 418)
    cuts = (ctypes.c_int32 * 20)(9, 18, 0, 0, 9, 18, 0, 0)
    k3 = lib.sv_frame_plans_arena_bytes(V, 4, 1, None)
    every = lib.sv_frame_plans_arena_bytes(V, 4, 1 | 2 | 4 | 8, cuts)
    assert k3 >= 88113 * 27 * 4 * 2 and every > k3 + 88113 * 27 * 4  # 27-offset map + plan; + down / up / three range plans
    assert lib.sv_frame_plans_scratch_bytes(V, 4) == lib.sv_plan_workspace_bytes(88113)
    rc = lib.sv_conv_set_dispatch(ctypes.c_double(1.0), ctypes.c_double(1.5))
    assert rc == -1 and b"tail_fraction" in lib.sv_last_error()
    assert lib.sv_conv_set_dispatch(ctypes.c_double(-1.0), ctypes.c_double(-1.0)) == 0


def test_product_path_refuses_cpu_tensors():
    """No CPU fallback: asking the runtime for a CPU device raises instead of silently computing elsewhere."""
    import pytest
    import torch

    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME

    with pytest.raises(mrcc_amd._lib.SvHipError):
        ME.TensorField(torch.zeros(4, 3), torch.zeros(4, 4), device="cpu")
    with pytest.raises(mrcc_amd._lib.SvHipError):
        ME.SparseTensor(torch.zeros(4, 3), coordinates=torch.zeros(4, 4, dtype=torch.int32), device="cpu")


def test_oracle_is_not_imported_by_the_package():
    pkg = os.path.join(ROOT, "markerless-robot-camera-calibration_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "sv_oracle" not in src.replace("oracle/sv_oracle", ""), f"{f} references the oracle module"
                assert "libsvoracle" not in src
