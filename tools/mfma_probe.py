"""Sustained v_mfma_f32_16x16x4_f32 rate of the chip (tools/mfma_probe.hip) beside rocBLAS/hipBLASLt SGEMM.

Context for roofline.frac: the 157.3 TFLOP/s peak is 1024 SIMDs x 64 flop/cycle x 2.4 GHz; this prints what a loop of
nothing but matrix ops reaches once clocks and the power limit have settled, with sclk / power sampled while it runs.
    python tools/mfma_probe.py [seconds per case]
"""
import ctypes, os, re, subprocess, sys, time
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_build", "libmfma_probe.so")


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    src = os.path.join(ROOT, "tools", "mfma_probe.hip")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", SO, src])
    lib = ctypes.CDLL(SO)
    lib.mfma_probe_launch.restype = ctypes.c_longlong
    lib.mfma_probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:  # noqa: BLE001
        return f"rocm-smi unavailable ({e})"
    sclk = re.findall(r"sclk clock level.*?\((\d+)Mhz\)", out)
    power = re.findall(r"Power \(W\):\s*([\d.]+)", out)
    return f"sclk {sclk[:1] or '?'} MHz, power {power[:1] or '?'} W"


def timed(seconds, launch, flop_per_launch):
    """Back-to-back launches for `seconds`; TFLOP/s over the last quarter (settled), clocks sampled in the middle."""
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(); e1.record(); torch.cuda.synchronize()
    one = e0.elapsed_time(e1) * 1e-3
    n = max(8, int(seconds / one))
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    marks[0].record()
    for i in range(n):
        launch(); marks[i + 1].record()
    time.sleep(min(seconds * 0.5, 2.0))
    state = smi()
    torch.cuda.synchronize()
    q = max(1, n // 4)
    tail = marks[n - q].elapsed_time(marks[n]) * 1e-3 / q
    head = marks[0].elapsed_time(marks[q]) * 1e-3 / q
    return flop_per_launch / tail / 1e12, flop_per_launch / head / 1e12, state, n


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 3.0
    dev = torch.device("cuda:0")
    lib = build()
    out = torch.zeros(16, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    print(f"idle: {smi()}", flush=True)
    for variant, name in ((0, "registers only, 12 accumulators"), (1, "A via LDS (4x3 accumulators, 1 ds_read_b32 per 3 ops)")):
        for waves in (1, 2, 4):
            blocks, iters = 256 * waves, 4000
            ops = lib.mfma_probe_launch(variant, blocks, 1, out.data_ptr(), stream)
            flop = ops * iters * blocks * 4 * 2048.0

            def launch():
                lib.mfma_probe_launch(variant, blocks, iters, out.data_ptr(), stream)
            tf_tail, tf_head, state, n = timed(seconds, launch, flop)
            print(f"mfma 16x16x4 f32, {name}, {waves} wave(s)/SIMD: settled {tf_tail:6.1f} TFLOP/s "
                  f"(first quarter {tf_head:6.1f}; {n} launches) = {tf_tail / 157.3:.3f} of 157.3 | {state}", flush=True)
    # v_mfma_f32_4x4x1_16B_f32 (4-row skip granularity at the same nominal peak): registers only, A from LDS, A + B fed per k
    lib.mfma_4x4_launch.restype = ctypes.c_longlong
    lib.mfma_4x4_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    wb = torch.randn(1 << 16, device=dev)
    for feed, name in ((0, "registers only"), (1, "4 ds_read_b32 per k (A, reused by 3 column tiles)"),
                       (2, "4 ds_read_b32 + 1 buffer_load_dwordx3 per k (A and B: a conv step's operand traffic)")):
        for waves in (2, 4):
            blocks, iters = 256 * waves, 1000
            ops = lib.mfma_4x4_launch(feed, blocks, 1, out.data_ptr(), wb.data_ptr(), stream)
            flop = ops * iters * blocks * 4 * 512.0

            def launch():
                lib.mfma_4x4_launch(feed, blocks, iters, out.data_ptr(), wb.data_ptr(), stream)
            tf_tail, tf_head, state, n = timed(min(seconds, 2.0), launch, flop)
            print(f"mfma 4x4x1 16B f32, {name}, {waves} waves/SIMD: {tf_tail:6.1f} TFLOP/s = {tf_tail / 157.3:.3f} of 157.3", flush=True)
    if "--4x4-only" in sys.argv:
        return
    # the same LDS-fed loop with one more instruction class at a time (tools/mfma_probe.hip, mfma_mix_kernel)
    lib.mfma_mix_launch.restype = ctypes.c_longlong
    lib.mfma_mix_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_longlong, ctypes.c_void_p]
    wbuf = torch.randn(1 << 16, device=dev)
    rows = 1 << 18
    big = torch.randn(rows, 384, device=dev)
    names = {0: "nothing added", 1: "+ weight reload (1 dwordx3 / k-step, cache-resident)", 2: "+ gathers (2 dwordx4 / step, 8 lanes per random row)",
             3: "+ weight reloads + gathers", 4: "+ 2 64-bit VALU adds / k-step",
             16: "+ 4 ds_write_b64 + barrier / step", 19: "+ reloads + gathers + LDS stores + barrier", 23: "+ all of them",
             33: "+ weight reload as buffer_load (SGPR row offset, constant VGPR offset)",
             34: "+ gathers as buffer_load (offsets from an LDS table + 1 v_add)", 35: "+ both in buffer form",
             51: "+ both in buffer form + LDS stores + barrier (the kernel's step)"}
    for mode, name in names.items():
        blocks, iters = 1024, 1000
        ops = lib.mfma_mix_launch(mode, blocks, 1, out.data_ptr(), wbuf.data_ptr(), big.data_ptr(), rows, stream)
        assert ops > 0
        flop = ops * iters * blocks * 4 * 2048.0

        def launch():
            lib.mfma_mix_launch(mode, blocks, iters, out.data_ptr(), wbuf.data_ptr(), big.data_ptr(), rows, stream)
        tf_tail, tf_head, state, n = timed(min(seconds, 2.0), launch, flop)
        print(f"LDS-fed loop, 4 waves/SIMD, {name}: {tf_tail:6.1f} TFLOP/s = {tf_tail / 157.3:.3f}", flush=True)
    lib.mfma_ld_launch.restype = ctypes.c_longlong
    lib.mfma_ld_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    forms = {0: "global_load_dwordx3 v, v[64-bit address] + s_waitcnt vmcnt(7) per k-step", 1: "same load, drained once per step",
             2: "the s_waitcnt alone (no loads)", 3: "global_load_dwordx3 v, v_offset, s[base]", 4: "global_load_dword (64-bit address)",
             5: "buffer_load_dwordx3 offen", 6: "three ds_read_b32 from LDS instead"}
    for ld, name in forms.items():
        blocks, iters = 1024, 1000
        ops = lib.mfma_ld_launch(ld, blocks, 1, out.data_ptr(), wbuf.data_ptr(), stream)
        assert ops > 0
        flop = ops * iters * blocks * 4 * 2048.0

        def launch():
            lib.mfma_ld_launch(ld, blocks, iters, out.data_ptr(), wbuf.data_ptr(), stream)
        tf_tail, tf_head, state, n = timed(min(seconds, 2.0), launch, flop)
        print(f"LDS-fed loop + one load per k-step as {name}: {tf_tail:6.1f} TFLOP/s = {tf_tail / 157.3:.3f}", flush=True)
    if "--mix-only" in sys.argv:
        return
    torch.backends.cuda.matmul.allow_tf32 = False
    for m in (4096, 8192):
        a = torch.randn(m, m, device=dev); b = torch.randn(m, m, device=dev); c = torch.empty(m, m, device=dev)

        def launch():
            torch.mm(a, b, out=c)
        tf_tail, tf_head, state, n = timed(seconds, launch, 2.0 * m ** 3)
        print(f"torch.mm fp32 {m}^3 (library SGEMM): settled {tf_tail:6.1f} TFLOP/s (first quarter {tf_head:6.1f}) "
              f"= {tf_tail / 157.3:.3f} of 157.3 | {state}", flush=True)


if __name__ == "__main__":
    main()
