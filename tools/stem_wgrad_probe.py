#!/usr/bin/env python3
"""stem weight gradient alone (64 frames): halo-resident kernel vs gathered split-K GEMM (HIP events, kernel + its reduce)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L
rt = Runtime("bf16", "cuda:0")
rnd = lambda *s: (torch.randn(*s, device=rt.device) * 0.5).to(torch.bfloat16)
B, H, C = 64, 112, 64
xs, dy, dw = rnd(B, H, H, 16), rnd(B, H, H, C), torch.empty(C, 256, device=rt.device)
wsb = L.query("msc_wgrad_workspace_bytes", C, 256, B * H * H)
ws = torch.empty(wsb // 4, device=rt.device)
for on in (1, 0):
    L.lib.msc_set_knob(b"stem_wgrad_halo", on)
    f = lambda: rt.call("msc_stem_s2d_conv_wgrad", xs, dy, dw, ws, wsb, B, H, H, C, 0, rt.dt)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"stem wgrad {'halo kernel' if on else 'gathered GEMM'}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us (kernel + reduce)")
L.lib.msc_set_knob(b"stem_wgrad_halo", 1)
