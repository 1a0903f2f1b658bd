#!/usr/bin/env python3
"""Is the contraction kernel bound by its memory side or by its own structure?  Same launch, operands aliased so that
every tile reads the same cache-resident rows (lda = ldb = 0) vs the real strided operands."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
rt = Runtime("bf16", "cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (M, N, K) in [(19584, 2048, 512), (19584, 512, 2048), (19584, 1536, 512), (65536, 256, 1152)]:
    A = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); B = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    for tag, lda, ldb in (("real", K, K), ("A aliased", 0, K), ("A,B aliased", 0, 0)):
        t = timeit(lambda: rt.call("msc_gemm_batched", A, B, C, M, N, K, lda, ldb, N, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, 0, rt.dt))
        print(f"M{M} N{N} K{K} {tag:12s} {t*1e3:8.1f} us {fl/t/1e9:8.1f} TFLOP/s")
