#!/usr/bin/env python3
"""Micro-benchmark of the MFMA conv kernels on the dominant C2 shape (B=8, 32^3, 64->64, 3x3x3) — for
rocprofv3 --pmc passes and A/B timing.  Usage: python tools/conv_bench.py [fwd|bwdw|both] [iters] [B S Cin Cout]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusioniqt_amd import ops, _lib

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B, S, Cin, Cout = (int(v) for v in sys.argv[3:7]) if len(sys.argv) > 6 else (8, 32, 64, 64)
_lib.load()
dev = "cuda"
SP = tuple(int(v) for v in os.environ["SHAPE"].split(",")) if "SHAPE" in os.environ else (S, S, S)    # non-cubic volumes (Family B: frames x H x W)
x = torch.randn(B, *SP, Cin, device=dev)
KS = tuple(int(v) for v in os.environ.get("KSHAPE", "3,3,3").split(","))      # filter extents (padding = k // 2)
PADS = tuple(k // 2 for k in KS)
w = (torch.randn(Cout, Cin, *KS, device=dev) * 0.02).requires_grad_()
bias = torch.zeros(Cout, device=dev, requires_grad=True)
dy = torch.randn(B, *SP, Cout, device=dev)
flops = 2.0 * B * SP[0] * SP[1] * SP[2] * Cin * Cout * KS[0] * KS[1] * KS[2]


def timeit(fn, n):
    for _ in range(max(3, int(os.environ.get("WARM", "300")))):     # DVFS: clocks need ~100 ms of load to settle
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


if mode in ("fwd", "both"):
    with torch.no_grad():
        ms = timeit(lambda: ops.conv3d(x, w, bias, PADS), iters)
    print(f"conv fwd  B={B} {SP} k={KS} {Cin}->{Cout}: {ms*1e3:.1f} us  {flops/ms/1e9:.1f} TFLOP/s")
if mode == "gn":                # Block on the sampling path: GroupNorm-apply inside the conv staging vs groupnorm_act + conv3d
    gamma, beta = torch.randn(Cin, device=dev), torch.randn(Cin, device=dev)
    ss = torch.randn(B, 2 * Cin, device=dev) * 0.3
    act = ops.ACT_SILU if KS[0] == 1 else ops.ACT_MISH
    with torch.no_grad():
        assert ops.gn_conv3d(x, gamma, beta, ss, 8, act, 1e-5, w, bias, PADS) is not None, "shape not taken"
        ms_f = timeit(lambda: ops.gn_conv3d(x, gamma, beta, ss, 8, act, 1e-5, w, bias, PADS), iters)
        ms_c = timeit(lambda: ops.conv3d(x, w, bias, PADS), iters)
        ms_u = timeit(lambda: ops.conv3d(ops.groupnorm_act(x, gamma, beta, ss, 8, act), w, bias, PADS), iters)
    print(f"GN+act+conv B={B} {SP} k={KS} {Cin}->{Cout}: fused {ms_f*1e3:.1f} us (incl. statistics pass + coefficient launch), "
          f"two-kernel path {ms_u*1e3:.1f} us, conv alone {ms_c*1e3:.1f} us = {flops/ms_c/1e9:.1f} TFLOP/s")
if mode in ("fwdh",):           # fp16 / bf16 operand kernel (LP=fp16|bf16)
    with torch.no_grad(), ops.low_precision(os.environ.get("LP", "fp16")):
        ms = timeit(lambda: ops.conv3d(x, w, bias, PADS), iters)
    print(f"conv fwd {os.environ.get('LP', 'fp16')}  B={B} {S}^3 {Cin}->{Cout}: {ms*1e3:.1f} us  {flops/ms/1e9:.1f} TFLOP/s")
    if os.environ.get("DIQT_CONVH_DBG") == "1":
        import ctypes
        import numpy as np
        lib = _lib.load()
        buf = np.zeros((65536, 8), dtype=np.uint64)
        n = lib.diqt_debug_convh_stamps(buf.ctypes.data_as(ctypes.c_void_p), 65536)
        st = buf[:n].astype(np.int64)
        names = ["prologue", "tap loops", "store+barrier", "epilogue", "lifetime", "steps"]
        print(f"{n} workgroups; median cycles:", {k: int(np.median(st[:, i])) for i, k in enumerate(names)})
        print("p10/p90 lifetime:", int(np.percentile(st[:, 4], 10)), int(np.percentile(st[:, 4], 90)))
if mode in ("bwdw", "both"):
    xr = x.clone()
    y = ops.conv3d(xr, w, bias, PADS)

    def bw():
        w.grad = None
        bias.grad = None
        y.backward(dy, retain_graph=True, inputs=[w, bias])
    ms = timeit(bw, iters)
    print(f"conv bwd-weight(+bias): {ms*1e3:.1f} us  {flops/ms/1e9:.1f} TFLOP/s")

if os.environ.get("DIQT_CONV_DBG") == "1" and mode == "bwdw":
    import ctypes
    import numpy as np
    lib = _lib.load()
    for _ in range(20):
        bw()
    torch.cuda.synchronize()
    lib.diqt_debug_wgrad3_stamps.restype = ctypes.c_int
    buf = np.zeros((65536, 8), dtype=np.uint64)
    n3 = lib.diqt_debug_wgrad3_stamps(buf.ctypes.data_as(ctypes.c_void_p), 65536)
    if n3 > 0:                      # version-3 kernel: one record per wave
        st = buf[:n3].astype(np.int64).reshape(-1, 4, 8)
        print(f"conv_wgrad3: {st.shape[0]} workgroups, {int(np.median(st[:, :, 7]))} tiles each; medians per wave (cycles):")
        print("   wave taps   lifetime     k-loops  wait+barrier  prologue  epilogue   clock MHz   k-loop per tile / MFMA floor")
        for wv in range(4):
            r = st[:, wv]
            clk = np.median(r[:, 0] / np.maximum(r[:, 5], 1) * 100.0)
            floor = r[:, 6] * 16 * 64
            print(f"   {wv}    {np.median(r[:, 6]) / 2:4.1f}  {np.median(r[:, 0]):10.0f}  {np.median(r[:, 1]):10.0f}  {np.median(r[:, 2]):10.0f}  "
                  f"{np.median(r[:, 3]):8.0f}  {np.median(r[:, 4]):8.0f}  {clk:9.0f}   {np.median(r[:, 1] / r[:, 7]):8.0f} / {np.median(floor):6.0f}")
        sys.exit(0)
    buf = np.zeros((65536, 8), dtype=np.uint64)
    n = lib.diqt_debug_conv_stamps(buf.ctypes.data_as(ctypes.c_void_p), 65536)
    st = buf[:n, :6].astype(np.int64).reshape(-1, 8, 6)
    names = ["barrier wait", "store regs->LDS", "tables+barrier", "issue loads", "MFMA k-loop"]
    print("per-wave medians (cycles per workgroup, 32 tiles):  " + "  ".join(names) + "  | SIMD")
    for wv in range(8):
        print(f"  wave {wv}: " + "  ".join(f"{np.median(st[:, wv, i]):12.0f}" for i in range(5)) + f"  | {np.bincount(st[:, wv, 5], minlength=4)}")
    sys.exit(0)
if os.environ.get("DIQT_CONV_DBG") == "1":
    import ctypes
    import numpy as np
    lib = _lib.load()
    with torch.no_grad():
        for _ in range(50):
            ops.conv3d(x, w, bias, PADS)
    torch.cuda.synchronize()
    buf = np.zeros((65536, 8), dtype=np.uint64)
    n = lib.diqt_debug_conv_stamps(buf.ctypes.data_as(ctypes.c_void_p), 65536)
    st = buf[:n].astype(np.int64)
    if os.environ.get("DIQT_CONV_STAGGER") == "-1":
        cyc = (st[:, 7] - st[:, 0]).astype(np.float64)
        rt = (st[:, 6] - st[:, 1]).astype(np.float64)          # 100 MHz ticks
        f = cyc / rt * 100.0
        print(f"effective shader clock inside the kernel (s_memtime / s_memrealtime): median {np.median(f):.0f} MHz, "
              f"p10 {np.percentile(f, 10):.0f}, p90 {np.percentile(f, 90):.0f}; workgroup lifetime median {np.median(rt) / 100:.1f} us")
        sys.exit(0)
    d = np.diff(st[:, :8], axis=1)
    names = ["tables+stage chunk0", "taps chunk0", "chunk1: entry barrier", "chunk1: halo load+store", "chunk1: W0 + barrier", "taps chunk1", "epilogue"]
    print(f"stamps from {n} workgroups (cycles of the constant-rate counter, median / p90):")
    for i, nm in enumerate(names):
        print(f"  {nm:22s} {np.median(d[:, i]):10.0f} {np.percentile(d[:, i], 90):10.0f}")
    tot = st[:, 7] - st[:, 0]
    print(f"  {'workgroup lifetime':22s} {np.median(tot):10.0f} {np.percentile(tot, 90):10.0f}")
    # only the LAST launch's stamps are in the buffer; the cycle counter is per-XCD, so compare within an XCD
    for xcd in range(8):
        m = np.arange(n) % 8 == xcd
        t0, t1 = st[m, 0], st[m, 7]
        print(f"  XCD {xcd}: lifetime median {np.median(t1 - t0):8.0f}  first start -> last end {t1.max() - t0.min():9.0f} cycles;"
              f"  end-time spread of the last 64 finishers {np.sort(t1)[-1] - np.sort(t1)[-64]:8.0f}")
    span = st[:, 7].max() - st[:, 0].min()
    print(f"  kernel span {span} ticks; sum of lifetimes / (512 slots) = {tot.sum() / 512:.0f}")
