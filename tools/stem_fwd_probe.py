#!/usr/bin/env python3
"""stem forward convolution alone (64 frames): halo-resident kernel vs implicit GEMM (HIP events)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L
rt = Runtime("bf16", "cuda:0")
rnd = lambda *s: (torch.randn(*s, device=rt.device) * 0.5).to(torch.bfloat16)
B, H, C = 64, 112, 64
xs, w, y = rnd(B, H, H, 16), rnd(C, 4, 4, 16), rnd(B, H, H, C)
nblk = L.query("msc_conv2d_fwd_stat_rows", B * H * H, C, rt.dt)
part = torch.empty(nblk * 2 * C, device=rt.device)
for on in (1, 0):
    L.lib.msc_set_knob(b"stem_fwd_halo", on)
    f = lambda: rt.call("msc_stem_s2d_conv_fwd", xs, w, y, part, B, H, H, C, rt.dt)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"stem forward {'halo kernel' if on else 'implicit GEMM'}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
L.lib.msc_set_knob(b"stem_fwd_halo", 1)
