#!/usr/bin/env python3
"""Per-shape timing of the MFMA contraction kernel on the C3 @ batch-64 shapes (HIP events, random data)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L

rt = Runtime("bf16", "cuda:0")
dev = rt.device


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


rows = []
B = 64
# ---- linear layers of the fusion transformer (M = B*306)
M = B * 306
for name, N, K, act, res, pre in [("qkv", 1536, 512, 0, 0, 0), ("proj+res", 512, 512, 0, 1, 0), ("fc1+gelu", 2048, 512, 2, 0, 1), ("fc2+res", 512, 2048, 0, 1, 0)]:
    x, w, b = rnd(M, K), rnd(N, K), torch.randn(N, device=dev)
    y, r, p = rnd(M, N), rnd(M, N), rnd(M, N)
    fl = 2.0 * M * N * K
    t = timeit(lambda: rt.call("msc_linear_fwd", x, w, b, r if res else None, y, p if pre else None, M, N, K, act, rt.dt))
    rows.append((f"linear_fwd {name} M{M} N{N} K{K}", t, fl))
    dy, wt, dx = rnd(M, N), rnd(K, N), rnd(M, K)
    t = timeit(lambda: rt.call("msc_linear_dgrad", dy, wt, None, None, 0, None, dx, M, N, K, rt.dt))
    rows.append((f"linear_dgrad {name}", t, fl))
    dw = torch.empty(N, K, device=dev)
    wsb = L.query("msc_wgrad_workspace_bytes", N, K, M)
    ws = torch.empty(wsb // 4, device=dev)
    t = timeit(lambda: rt.call("msc_linear_wgrad", x, dy, dw, None, ws, wsb, M, N, K, 0, rt.dt))
    rows.append((f"linear_wgrad {name}", t, fl))

# ---- convolutions (ResNet-18 @224 + BEV)
convs = [("conv1 7x7/2", 224, 8, 64, 7, 2, 3), ("l1 3x3", 56, 64, 64, 3, 1, 1), ("l2 3x3/2", 56, 64, 128, 3, 2, 1), ("l2 3x3", 28, 128, 128, 3, 1, 1),
         ("l2 1x1/2", 56, 64, 128, 1, 2, 0), ("l3 3x3", 14, 256, 256, 3, 1, 1), ("l4 3x3", 7, 512, 512, 3, 1, 1),
         ("bev1 3x3/2", 128, 64, 128, 3, 2, 1), ("bev2 3x3/2", 64, 128, 256, 3, 2, 1), ("bev3 3x3/2", 32, 256, 512, 3, 2, 1)]
for name, H, Cin, Cout, K, s, p in convs:
    Ho = (H + 2 * p - K) // s + 1
    x, w = rnd(B, H, H, Cin), rnd(Cout, K, K, Cin)
    y = rnd(B, Ho, Ho, Cout)
    part = torch.empty((B * Ho * Ho + 63) // 64 * 2 * Cout, device=dev)
    cin_real = 3 if Cin == 8 else Cin
    fl = 2.0 * B * Ho * Ho * Cout * K * K * cin_real
    wsb = L.query("msc_conv2d_splitk_workspace_bytes", B * Ho * Ho, Cout, K * K * Cin, rt.dt)
    wsf = torch.empty(max(wsb // 4, 1), device=dev) if wsb else None
    t = timeit(lambda: rt.call("msc_conv2d_fwd", x, w, y, part, B, H, H, Cin, Cout, K, K, s, p, wsf, wsb, rt.dt))
    rows.append((f"conv_fwd {name} H{H} {Cin}->{Cout}", t, fl))
    if Cin != 8:
        wt, dx = rnd(Cin, K, K, Cout), rnd(B, H, H, Cin)
        wsd = L.query("msc_conv2d_splitk_workspace_bytes", B * H * H, Cin, K * K * Cout, rt.dt) if s == 1 else 0
        wsg = torch.empty(max(wsd // 4, 1), device=dev) if wsd else None
        t = timeit(lambda: rt.call("msc_conv2d_dgrad", y, wt, None, None, dx, B, H, H, Cin, Cout, K, K, s, p, wsg, wsd, rt.dt))
        rows.append((f"conv_dgrad {name}", t, fl))
    dw = torch.empty(Cout, K, K, Cin, device=dev)
    wsb = L.query("msc_wgrad_workspace_bytes", Cout, K * K * Cin, B * Ho * Ho)
    ws = torch.empty(wsb // 4, device=dev)
    t = timeit(lambda: rt.call("msc_conv2d_wgrad", x, y, dw, ws, wsb, B, H, H, Cin, Cout, K, K, s, p, 0, rt.dt))
    rows.append((f"conv_wgrad {name}", t, fl))

# ---- attention batched GEMMs
T, h, hd, D = 306, 8, 64, 512
Tp = 320
qkv = rnd(B * T, 3 * D)
S = torch.empty(B * h, T, Tp, device=dev)
ld = 3 * D
t = timeit(lambda: rt.call("msc_gemm_batched", qkv, qkv[:, D:], S, T, T, hd, ld, ld, Tp, 1, 1, B * h, h, T * ld, hd, T * ld, hd, h * T * Tp, T * Tp, 0.125, 1, rt.dt))
rows.append(("attn QK^T (f32 out)", t, 2.0 * B * h * T * T * hd))
P = rnd(B * h, T, Tp)
o = rnd(B * T, D)
t = timeit(lambda: rt.call("msc_gemm_batched", P, qkv[:, 2 * D:], o, T, hd, T, Tp, ld, D, 1, 0, B * h, h, h * T * Tp, T * Tp, T * ld, hd, T * D, hd, 1.0, 0, rt.dt))
rows.append(("attn PV", t, 2.0 * B * h * T * T * hd))

tot_t = tot_f = 0
for name, t, fl in rows:
    print(f"{name:44s} {t*1e3:9.1f} us  {fl/t/1e9:8.1f} TFLOP/s")
