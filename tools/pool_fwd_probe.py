#!/usr/bin/env python3
"""stem BatchNorm + ReLU + max-pool forward alone (64 frames): unrolled 3x3 / stride-2 kernel vs generic window loops; outputs compared bit for bit"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L
rt = Runtime("bf16", "cuda:0")
B, H, C, Ho = 64, 112, 64, 56
x = (torch.randn(B, H, H, C, device=rt.device) * 0.5).to(torch.bfloat16)
scale, shift = torch.randn(C, device=rt.device), torch.randn(C, device=rt.device) * 0.1
res = {}
for on in (1, 0):
    L.lib.msc_set_knob(b"pool_k3s2", on)
    y = torch.empty(B, Ho, Ho, C, device=rt.device, dtype=torch.bfloat16); xsel = torch.empty_like(y)
    idx = torch.empty(B, Ho, Ho, C, device=rt.device, dtype=torch.uint8)
    f = lambda: rt.call("msc_bn_relu_pool_fwd", x, scale, shift, y, idx, xsel, B, H, H, C, 3, 2, 1, rt.dt)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"bn_relu_pool_fwd {'unrolled k3s2' if on else 'generic'}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
    res[on] = (y.clone(), idx.clone(), xsel.clone())
print("bit-identical:", all(torch.equal(a, b) for a, b in zip(res[0], res[1])))
L.lib.msc_set_knob(b"pool_k3s2", 1)
