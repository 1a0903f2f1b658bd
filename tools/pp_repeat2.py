#!/usr/bin/env python3
"""debug aid: the ping-pong kernel beside OTHER contraction kernels on other streams (a tiled weight gradient; a second ping-pong launch)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L
rt = Runtime("bf16", "cuda:0")
dev = rt.device
lib = L.lib.load()
torch.manual_seed(0)
s2, s3 = torch.cuda.Stream(), torch.cuda.Stream()
M, N, K = 9792, 512, 512
dy = (torch.randn(2 * M, N, device=dev) * 0.5).to(torch.bfloat16)
wt = (torch.randn(K, N, device=dev) * N ** -0.5).to(torch.bfloat16)
x = (torch.randn(2 * M, K, device=dev) * 0.5).to(torch.bfloat16)
ref = dy.float() @ wt.float().t()
dw = torch.empty(N, K, device=dev)
wsb = L.query("msc_wgrad_workspace_bytes", N, K, 2 * M)
ws = torch.empty(wsb // 4, device=dev)
for mode in ("quiet", "two_pp", "pp_and_wgrad", "two_pp_and_wgrad"):
    res = []
    for rep in range(6):
        dx = torch.full((2 * M, K), float("nan"), device=dev, dtype=torch.bfloat16)
        torch.cuda.synchronize()
        if "wgrad" in mode:
            with torch.cuda.stream(s3):
                for _ in range(3):
                    L.call("msc_linear_wgrad", x, dy, dw, None, ws, wsb, 2 * M, N, K, 0, rt.dt, s3.cuda_stream)
        if mode == "quiet" or mode == "pp_and_wgrad":
            rt.call("msc_linear_dgrad", dy, wt, None, None, 0, None, dx, 2 * M, N, K, rt.dt)
        else:       # two half-batch launches on two streams, like the fusion's two chains
            L.call("msc_linear_dgrad", dy, wt, None, None, 0, None, dx, M, N, K, rt.dt, torch.cuda.current_stream().cuda_stream)
            with torch.cuda.stream(s2):
                L.call("msc_linear_dgrad", dy[M:], wt, None, None, 0, None, dx[M:], M, N, K, rt.dt, s2.cuda_stream)
        torch.cuda.synchronize()
        res.append(dx)
    err = [((o.float() - ref).abs().max() / ref.abs().max()).item() for o in res]
    print(f"{mode:18s} err {['%.1e' % e for e in err]} same {[bool(torch.equal(res[0], o)) for o in res]}", flush=True)
