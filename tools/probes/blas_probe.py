#!/usr/bin/env python3
"""What the vendor library reaches on the path's dense shapes (torch.matmul -> hipBLASLt / rocBLAS, bf16, plain C = A B^T, no epilogue):
a yardstick for csrc/gemm.hip on THIS box, not part of the product."""
import torch
dev = "cuda:0"
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 64 * 306
for name, m, n, k in [("qkv fwd", M, 1536, 512), ("proj fwd", M, 512, 512), ("fc1 fwd", M, 2048, 512), ("fc2 fwd", M, 512, 2048),
                      ("qkv wgrad", 1536, 512, M), ("fc1 wgrad", 2048, 512, M), ("half-batch qkv", M // 2, 1536, 512), ("big", 8192, 8192, 4096)]:
    a = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
    b = (torch.randn(n, k, device=dev) * 0.5).to(torch.bfloat16)
    c = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    us = t(lambda: torch.matmul(a, b.t(), out=c))
    print(f"{name:16s} {m:6d} x {n:5d} x {k:6d}: {us:7.1f} us  {2.0 * m * n * k / us / 1e6:6.0f} TFLOP/s")
    if "wgrad" in name:      # the layout the weight gradient really has: both operands reduction-major
        at = a.t().contiguous(); bt = b.t().contiguous()
        us = t(lambda: torch.matmul(at.t(), bt, out=c))
        print(f"{'  (k-major in)':16s} {m:6d} x {n:5d} x {k:6d}: {us:7.1f} us  {2.0 * m * n * k / us / 1e6:6.0f} TFLOP/s")
