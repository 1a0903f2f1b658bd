#!/usr/bin/env python3
"""Achievable HBM rates of plain streaming kernels on this box: write-only (fill), read-only (sum), copy (1 read + 1 write), at several sizes.
The contraction kernel's store phase is priced against the write-only figure."""
import torch, time
dev = "cuda:0"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (16, 32, 64, 128, 512, 2048):
    n = mb * (1 << 20) // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev); b = torch.empty_like(a)
    a.fill_(1.0)
    tf = t(lambda: a.fill_(2.0)); tc = t(lambda: b.copy_(a)); ts = t(lambda: a.view(torch.int16).max())
    print(f"{mb:5d} MB: fill {mb / 1024 / tf / 1e3 * 1.073741824:6.2f} TB/s ({tf * 1e6:7.1f} us) | copy {2 * mb / 1024 / tc / 1e3 * 1.073741824:6.2f} TB/s ({tc * 1e6:7.1f} us) | "
          f"read (max) {mb / 1024 / ts / 1e3 * 1.073741824:6.2f} TB/s ({ts * 1e6:7.1f} us)")
