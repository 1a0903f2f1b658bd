#!/usr/bin/env python3
"""debug aid: plain dgrad shapes of the fusion block on the ping-pong kernel, repeated, quiet and beside other streams' kernels:
bitwise repeatability + error against fp32"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L
rt = Runtime("bf16", "cuda:0")
dev = rt.device
lib = L.lib.load()
torch.manual_seed(0)
side, side2 = torch.cuda.Stream(), torch.cuda.Stream()
big, big2 = torch.empty(1 << 27, device=dev), torch.empty(1 << 27, device=dev)
small = torch.randn(4096, 512, device=dev)
for M in (9792, 19584):
    for N, K in ((512, 512), (1536, 512), (2048, 512)):       # reduction N, output K (msc_linear_dgrad's naming)
        dy = (torch.randn(M, N, device=dev) * 0.5).to(torch.bfloat16)
        wt = (torch.randn(K, N, device=dev) * N ** -0.5).to(torch.bfloat16)
        ref = dy.float() @ wt.float().t()
        outs = []
        for rep in range(6):
            if rep >= 2:
                with torch.cuda.stream(side):
                    big2.copy_(big)
                with torch.cuda.stream(side2):
                    for _ in range(20):
                        small.add_(1.0)
            dx = torch.full((M, K), float("nan"), device=dev, dtype=torch.bfloat16)
            rt.call("msc_linear_dgrad", dy, wt, None, None, 0, None, dx, M, N, K, rt.dt)
            torch.cuda.synchronize()
            outs.append(dx)
        err = [((o.float() - ref).abs().max() / ref.abs().max()).item() for o in outs]
        same = [bool(torch.equal(outs[0], o)) for o in outs]
        print(f"M={M} red={N} out={K}: err {['%.1e' % e for e in err]} same-as-first {same}", flush=True)
