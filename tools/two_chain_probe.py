#!/usr/bin/env python3
"""Does splitting a transformer linear into two half-batch launches on two streams beat one full-batch launch?  (The workgroups
of one launch run load / MFMA / store in lock-step; two concurrent launches might interleave those phases.)  Chains of the four
forward linears of a layer, full batch on one stream vs two half-batch chains on two streams."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L
rt = Runtime("bf16", "cuda:0")
dev = "cuda"
M = 19584
shapes = [("qkv", 1536, 512, 0, 0, 0), ("proj", 512, 512, 0, 1, 0), ("fc1", 2048, 512, 2, 0, 1), ("fc2", 512, 2048, 0, 1, 0)]
def mk(m):
    t = []
    for name, N, K, act, res, pre in shapes:
        x = (torch.randn(m, K, device=dev) * 0.5).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        b = torch.randn(N, device=dev); y = torch.empty(m, N, device=dev, dtype=torch.bfloat16)
        r = torch.empty_like(y) if res else None; p = torch.empty_like(y) if pre else None
        t.append((x, w, b, r, y, p, m, N, K, act))
    return t
full, ha, hb = mk(M), mk(M // 2), mk(M // 2)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def chain(ts, stream):
    for a in ts:
        L.call("msc_linear_fwd", *a, rt.dt, stream.cuda_stream)
def run_full():
    chain(full, s1)
def run_split():
    chain(ha, s1); chain(hb, s2)
def run_split_interleaved():
    for a, b in zip(ha, hb):
        L.call("msc_linear_fwd", *a, rt.dt, s1.cuda_stream); L.call("msc_linear_fwd", *b, rt.dt, s2.cuda_stream)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e0.record()
    s1.wait_event(e0); s2.wait_event(e0)
    for _ in range(n): fn()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, fn in (("one chain, full batch", run_full), ("two half-batch chains, two streams", run_split), ("... launches interleaved", run_split_interleaved)):
    print(f"{name:40s} {timeit(fn):7.1f} us per layer-forward of linears")

# ---- same question for the camera trunk's convolutions (layer 1: 64->64 3x3 at 56x56; layer 3: 256->256 at 14x14), forward GEMMs only
def conv_chain(B, H, C, st, bufs):
    x, w, y, part = bufs
    Ho = H
    L.call("msc_conv2d_fwd", x, w, y, part, B, H, H, C, C, 3, 3, 1, 1, None, 0, rt.dt, st.cuda_stream)


def mkconv(B, H, C):
    x = (torch.randn(B, H, H, C, device=dev) * 0.5).to(torch.bfloat16); w = (torch.randn(C, 3, 3, C, device=dev) * 0.05).to(torch.bfloat16)
    y = torch.empty(B, H, H, C, device=dev, dtype=torch.bfloat16)
    part = torch.empty(L.query("msc_conv2d_fwd_stat_rows", B * H * H, C, rt.dt) * 2 * C, device=dev)
    return (x, w, y, part)


for H, C in ((56, 64), (28, 128), (14, 256)):
    full, ha, hb = mkconv(64, H, C), mkconv(32, H, C), mkconv(32, H, C)
    one = lambda: [conv_chain(64, H, C, s1, full) for _ in range(4)]
    two = lambda: [(conv_chain(32, H, C, s1, ha), conv_chain(32, H, C, s2, hb)) for _ in range(4)]
    print(f"conv {C}->{C} 3x3 @{H}: 4 launches full batch {timeit(one):7.1f} us   two half-batch chains {timeit(two):7.1f} us")
