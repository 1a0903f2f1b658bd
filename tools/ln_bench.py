#!/usr/bin/env python3
"""LayerNorm forward / backward alone at the path's sizes (whole batch and one half-batch chain): us per call and achieved GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L
rt = Runtime("bf16", "cuda:0")
D = 512
def timeit(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = []
for rows in (19584, 9792):
    x = (torch.randn(rows, D, device=rt.device)).to(torch.bfloat16); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
    g, b = torch.ones(D, device=rt.device), torch.zeros(D, device=rt.device)
    mu, rs = torch.empty(rows, device=rt.device), torch.empty(rows, device=rt.device)
    ws = torch.empty(L.query("msc_layernorm_bwd_workspace_bytes", rows, D) // 4, device=rt.device)
    tf = timeit(lambda: rt.call("msc_layernorm_fwd", x, g, b, y, mu, rs, rows, D, 1e-5, rt.dt))
    tb = timeit(lambda: rt.call("msc_layernorm_bwd", dy, x, mu, rs, g, None, dx, None, None, ws, rows, D, 0, rt.dt))
    out.append(f"rows {rows}: fwd {tf:5.1f} us ({2 * rows * D * 2 / tf / 1e3:5.0f} GB/s)  bwd {tb:5.1f} us ({3 * rows * D * 2 / tb / 1e3:5.0f} GB/s)")
print(" | ".join(out))
