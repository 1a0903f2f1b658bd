#!/usr/bin/env python3
"""Microbenchmark of the contraction kernel on one large dense shape (default 8192 x 8192 x 4096, bf16, random data), per
tile variant (MSC_GEMM_TILE is read per call, so the variants are interleaved in ONE process: rounds x variants, median)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime

M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (8192, 8192, 4096)))
rt = Runtime("bf16", "cuda:0")
x = (torch.randn(M, K, device=rt.device) * 0.5).to(torch.bfloat16)
w = (torch.randn(N, K, device=rt.device) * 0.5).to(torch.bfloat16)
y = torch.empty(M, N, device=rt.device, dtype=torch.bfloat16)
tiles = ["128x128", "256x128", "256x256"]
res = {t: [] for t in tiles}


def run(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        rt.call("msc_linear_fwd", x, w, None, None, y, None, M, N, K, 0, rt.dt)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for rnd in range(6):
    for t in tiles:
        os.environ["MSC_GEMM_TILE"] = t
        run(2)
        res[t].append(run(10))
ref = (x[:64].float() @ w[:64].float().t())
os.environ["MSC_GEMM_TILE"] = "256x256"
run(1)
err = (y[:64, :64].float() - ref).abs().max().item() / ref.abs().max().item()
for t in tiles:
    ms = statistics.median(res[t])
    print(f"{M}x{N}x{K} tile {t}: median {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s   (min {min(res[t])*1e3:.1f} us)")
print(f"spot check rel err {err:.2e}")
