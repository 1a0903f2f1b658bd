#!/usr/bin/env python3
"""GEMM_DEPHASE experiment (make alt ALT=dephase ALT_FLAGS=-DGEMM_DEPHASE; MSC_LIB_PATH=.../libmsc_hip_dephase.so): the odd
residency slot of every CU sleeps N x 1024 clocks once, in the first round of a launch, so the two workgroups of a CU run their
K loops (LDS-bound) and epilogues (store burst) out of phase.  Interleaved rounds, median us per shape and N."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime

rt = Runtime("bf16", "cuda:0")
dev = rt.device
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M = 64 * 306
cases = []
for name, N, K, act, res, pre in [("qkv", 1536, 512, 0, 0, 0), ("proj", 512, 512, 0, 1, 0), ("fc1", 2048, 512, 2, 0, 1), ("fc2", 512, 2048, 0, 1, 0)]:
    x, w, b, y, r, p = rnd(M, K), rnd(N, K), torch.randn(N, device=dev), rnd(M, N), rnd(M, N), rnd(M, N)
    cases.append((f"lin_fwd {name}", 2.0 * M * N * K, lambda x=x, w=w, b=b, y=y, r=r, p=p, N=N, K=K, act=act, res=res, pre=pre:
                  rt.call("msc_linear_fwd", x, w, b, r if res else None, y, p if pre else None, M, N, K, act, rt.dt)))
    dy, wt, dx = rnd(M, N), rnd(K, N), rnd(M, K)
    cases.append((f"lin_dgrad {name}", 2.0 * M * N * K, lambda dy=dy, wt=wt, dx=dx, N=N, K=K:
                  rt.call("msc_linear_dgrad", dy, wt, None, None, 0, None, dx, M, N, K, rt.dt)))
B = 64
for name, H, Cin, Cout in [("l1 3x3", 56, 64, 64), ("l2 3x3", 28, 128, 128), ("bev2 3x3/2", 64, 128, 256)]:
    s = 2 if "/2" in name else 1
    Ho = (H + 2 - 3) // s + 1
    x, w, y = rnd(B, H, H, Cin), rnd(Cout, 3, 3, Cin), rnd(B, Ho, Ho, Cout)
    part = torch.empty((B * Ho * Ho + 63) // 64 * 2 * Cout, device=dev)
    cases.append((f"conv_fwd {name}", 2.0 * B * Ho * Ho * Cout * 9 * Cin, lambda x=x, w=w, y=y, part=part, H=H, Cin=Cin, Cout=Cout, s=s:
                  rt.call("msc_conv2d_fwd", x, w, y, part, B, H, H, Cin, Cout, 3, 3, s, 1, None, 0, rt.dt)))
vals = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,3,6,9,12".split(","))]
for name, fl, fn in cases:
    res = {v: [] for v in vals}
    for _ in range(3):
        for v in vals:
            os.environ["MSC_GEMM_DEPHASE"] = str(v)
            res[v].append(timeit(fn))
    med = {v: statistics.median(t) for v, t in res.items()}
    print(f"{name:22s} " + "  ".join(f"N={v}: {med[v]:6.1f} us ({fl / med[v] / 1e6:4.0f} TF)" for v in vals), flush=True)
