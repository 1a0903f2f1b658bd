#!/usr/bin/env python3
"""1x1 / stride-2 data gradient (ResNet downsample convolutions) alone: time and the kernel variant"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L
rt = Runtime("bf16", "cuda:0")
rnd = lambda *s: (torch.randn(*s, device=rt.device) * 0.5).to(torch.bfloat16)
for H, Cin, Cout in [(56, 64, 128), (28, 128, 256), (14, 256, 512)]:
    B = 64
    y, wt, dx = rnd(B, H // 2, H // 2, Cout), rnd(Cin, 1, 1, Cout), rnd(B, H, H, Cin)
    f = lambda: rt.call("msc_conv2d_dgrad", y, wt, None, None, dx, B, H, H, Cin, Cout, 1, 1, 2, 0, None, 0, rt.dt)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    info = (ctypes.c_int * 8)(); L.lib.msc_debug_last_gemm(ctypes.cast(info, ctypes.c_void_p))
    print(f"H{H} {Cin}<-{Cout}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us, variant {list(info)}, zeros ok: {bool((dx[:, 1::2].float().abs().max() == 0).item())}")
