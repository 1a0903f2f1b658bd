#!/usr/bin/env python3
"""One ResNet layer-1 weight gradient (64->64, 3x3, 56x56, 64 frames) launched N times: run under
`rocprofv3 --pmc FETCH_SIZE` to read the L2-miss bytes per launch (A/B knobs: MSC_WGRAD_XCD_SPLIT, MSC_WGRAD_BLOCKS)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L
rt = Runtime("bf16", "cuda:0")
B, H, C, K = 64, int(os.environ.get("WG_H", "56")), int(os.environ.get("WG_C", "64")), 3
x = (torch.randn(B, H, H, C, device="cuda") * 0.5).to(torch.bfloat16)
dy = (torch.randn(B, H, H, C, device="cuda") * 0.5).to(torch.bfloat16)
dw = torch.empty(C, K, K, C, device="cuda")
wsb = L.query("msc_wgrad_workspace_bytes", C, K * K * C, B * H * H)
ws = torch.empty(wsb // 4, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    rt.call("msc_conv2d_wgrad", x, dy, dw, ws, wsb, B, H, H, C, C, K, K, 1, 1, 0, rt.dt)
torch.cuda.synchronize()
print("slabs", wsb // (C * (K * K * C + 1) * 4))
