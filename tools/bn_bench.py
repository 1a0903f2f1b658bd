#!/usr/bin/env python3
"""BatchNorm streaming kernels at the ResNet-18 / BEV shapes of C3 @ 64 frames: achieved HBM GB/s (algorithmic bytes / time), knob bn_un 1 vs 4"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L
rt = Runtime("bf16", "cuda:0")
dev = rt.device
lib = L.lib.load()
B = 64


def timeit(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {"apply": 0.0, "bwd_un1": 0.0, "bwd_un4": 0.0}
for name, H, C in [("l1", 56, 64), ("l2", 28, 128), ("l3", 14, 256), ("l4", 7, 512), ("bev1", 64, 128), ("bev2", 32, 256), ("bev3", 16, 512)]:
    rows = B * H * H
    x = torch.randn(rows, C, device=dev).to(torch.bfloat16)
    dy = torch.randn(rows, C, device=dev).to(torch.bfloat16)
    res = torch.randn(rows, C, device=dev).to(torch.bfloat16)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    mean, rstd, gamma = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    ws = torch.empty(L.query("msc_bn_bwd_workspace_bytes", C) // 4, device=dev)
    nb = rows * C * 2
    t_apply = timeit(lambda: rt.call("msc_bn_apply", x, sc, sh, None, y, rows, C, 1, rt.dt))
    t_apply_res = timeit(lambda: rt.call("msc_bn_apply", x, sc, sh, res, y, rows, C, 1, rt.dt))
    out = {}
    for un in (1, 4):
        lib.msc_set_knob(b"bn_un", un)
        out[un] = timeit(lambda: rt.call("msc_bn_bwd", dy, x, None, mean, rstd, gamma, sc, sh, dx, None, dg, db, ws, rows, C, 1, 0, rt.dt))
    lib.msc_set_knob(b"bn_un", 4)
    tot["apply"] += t_apply; tot["bwd_un1"] += out[1]; tot["bwd_un4"] += out[4]
    print(f"{name:5s} rows {rows:7d} C {C:3d}: apply {t_apply:6.1f} us {2 * nb / t_apply / 1e6:5.2f} TB/s | apply+res {t_apply_res:6.1f} us {3 * nb / t_apply_res / 1e6:5.2f} TB/s | "
          f"bwd (reduce+finalize+apply, 5 passes) un1 {out[1]:6.1f} us {5 * nb / out[1] / 1e6:5.2f} TB/s, un4 {out[4]:6.1f} us {5 * nb / out[4] / 1e6:5.2f} TB/s", flush=True)
print("sum (one kernel per shape):", {k: round(v, 1) for k, v in tot.items()})
