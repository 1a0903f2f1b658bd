#!/usr/bin/env python3
"""debug aid: two half-batch ping-pong launches beside tiled weight gradients; where are the wrong elements?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L
rt = Runtime("bf16", "cuda:0")
dev = rt.device
torch.manual_seed(0)
s2, s3 = torch.cuda.Stream(), torch.cuda.Stream()
M, N, K = 9792, 512, 512
BM = int(sys.argv[1]) if len(sys.argv) > 1 else 160
if len(sys.argv) > 1:
    L.lib.msc_set_knob(b"gemm_pp", BM)
dy = (torch.randn(2 * M, N, device=dev) * 0.5).to(torch.bfloat16)
wt = (torch.randn(K, N, device=dev) * N ** -0.5).to(torch.bfloat16)
x = (torch.randn(2 * M, K, device=dev) * 0.5).to(torch.bfloat16)
ref = dy.float() @ wt.float().t()
dw = torch.empty(N, K, device=dev)
wsb = L.query("msc_wgrad_workspace_bytes", N, K, 2 * M)
ws = torch.empty(wsb // 4, device=dev)
junk = torch.empty(1 << 26, device=dev)
nbad = 0
for rep in range(40):
    dx = torch.full((2 * M, K), float("nan"), device=dev, dtype=torch.bfloat16)
    junk.normal_()                       # other work in between so the start conditions vary
    torch.cuda.synchronize()
    with torch.cuda.stream(s3):
        for _ in range(1 + rep % 3):
            L.call("msc_linear_wgrad", x, dy, dw, None, ws, wsb, 2 * M, N, K, 0, rt.dt, s3.cuda_stream)
    L.call("msc_linear_dgrad", dy, wt, None, None, 0, None, dx, M, N, K, rt.dt, torch.cuda.current_stream().cuda_stream)
    with torch.cuda.stream(s2):
        L.call("msc_linear_dgrad", dy[M:], wt, None, None, 0, None, dx[M:], M, N, K, rt.dt, s2.cuda_stream)
    torch.cuda.synchronize()
    d = (dx.float() - ref).abs()
    bad = ~(d < 0.05)
    if bad.any():
        nbad += 1
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        half = "A" if rows[0] < M else "B"
        rr = (rows % M)
        tiles = sorted(set((int(r) // BM) for r in rr.tolist()))
        print(f"rep {rep}: {int(bad.sum())} bad elements, nan {int(torch.isnan(dx.float()).sum())}; chain {half}; row tiles {tiles[:12]} (of {len(tiles)}); "
              f"rows in tile {sorted(set(int(r) % BM for r in rr.tolist()))[:40]}; cols {cols[:48].tolist()} ({len(cols)})", flush=True)
print("bad reps:", nbad, "of 40")
