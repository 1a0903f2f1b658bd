#!/usr/bin/env python3
"""Where a workgroup of conv_c64.hip spends its cycles (diagnostic build: make -C multimodal_sc_amd/csrc alt ALT=c6stamp ALT_SRC=conv_c64
ALT_FLAGS=-DC6_STAMP; run with MSC_LIB_PATH=multimodal_sc_amd/csrc/libmsc_hip_c6stamp.so)."""
import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L

rt = Runtime("bf16", "cuda:0")
dll = ctypes.CDLL(L.LIB_PATH)
dll.msc_debug_c64_stamps.restype = ctypes.c_int
dll.msc_debug_c64_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
rnd = lambda *s: (torch.randn(*s, device=rt.device) * 0.5).to(torch.bfloat16)
B, H, C = 64, 56, 64
x, w, y = rnd(B, H, H, C), rnd(C, 3, 3, C), rnd(B, H, H, C)
part = torch.empty(B * H * H // 64 * 2 * C, device=rt.device)
for mode in ("fwd + statistics", "dgrad plain", "dgrad + res + relu"):
    for _ in range(3):
        if mode.startswith("fwd"):
            rt.call("msc_conv2d_fwd", x, w, y, part, B, H, H, C, C, 3, 3, 1, 1, None, 0, rt.dt)
        else:
            r = x if "res" in mode else None
            rt.call("msc_conv2d_dgrad", y, w, r, r, x if r is None else y, B, H, H, C, C, 3, 3, 1, 1, None, 0, rt.dt)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (256 * 8))()
    assert dll.msc_debug_c64_stamps(buf, 256) == 0
    rows = [buf[i * 8:(i + 1) * 8] for i in range(256)]
    med = lambda i: statistics.median(r[i] for r in rows)
    nt = med(4)
    print(f"{mode}: per workgroup (median of 256, s_memtime ticks): life {med(5):.0f} | weights {med(0):.0f} | tiles {nt:.0f} | per tile: halo wait + barrier {med(1) / nt:.0f}, "
          f"K loop {med(2) / nt:.0f}, epilogue {med(3) / nt:.0f} | prologue: setup + DMA issue {med(6):.0f}, DMA wait + barrier {med(7):.0f}, fragment reads {med(0) - med(6) - med(7):.0f}")
