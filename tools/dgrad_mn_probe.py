#!/usr/bin/env python3
"""Linear dgrad dx = dy W with W read in its own [N][K] layout (reduction-major B operand, ds_read_b64_tr_b16 fragments) against the k-major
transposed copy the step keeps (msc_permute3_cast_batched after every optimizer step): is the copy worth its 55 us per step?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
rt = Runtime("bf16", "cuda:0")
def timeit(fn, n=20):
    for _ in range(4): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
rnd = lambda *s: (torch.randn(*s, device=rt.device) * 0.5).to(torch.bfloat16)
for M in (19584, 9792):
    for name, N, K in (("qkv", 1536, 512), ("proj", 512, 512), ("fc1", 2048, 512), ("fc2", 512, 2048)):
        dy, w = rnd(M, N), rnd(N, K)
        wt = w.t().contiguous()
        dx0, dx1 = torch.empty(M, K, device=rt.device, dtype=torch.bfloat16), torch.empty(M, K, device=rt.device, dtype=torch.bfloat16)
        f0 = lambda: rt.call("msc_linear_dgrad", dy, wt, None, None, 0, None, dx0, M, N, K, rt.dt)
        # C[m][k] = sum_n dy[m][n] * w[n][k]: A = dy k-major (ld N), B = w stored [reduction n][col k] = mn-major (ld K)
        f1 = lambda: rt.call("msc_gemm_batched", dy, w, dx1, M, K, N, N, K, K, 1, 0, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, 0, rt.dt)
        t0, t1 = timeit(f0), timeit(f1)
        torch.cuda.synchronize()
        err = (dx0.float() - dx1.float()).abs().max().item()
        print(f"M {M:6d} {name:5s} dgrad: transposed copy (k-major B) {t0:6.1f} us | W as stored (reduction-major B) {t1:6.1f} us | max diff {err:.3g}")
