#!/usr/bin/env python3
"""Per-launch time of the <= 64-row dense layers (codec, heads, state encoder) on gemm_skinny.hip vs the tiled family (HIP events)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L

rt = Runtime("bf16", "cuda:0")
dev = rt.device
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rows in (64, 32):
    tot = {1: 0.0, 0: 0.0}
    for name, N, K, act in [("enc.fc1", 256, 512, 1), ("enc.fc2", 128, 256, 0), ("dec.fc1", 256, 128, 1), ("dec.fc2", 512, 256, 0), ("head.fc", 256, 512, 3),
                            ("head.out", 16, 256, 0), ("recon.fc1", 3136, 512, 1), ("state.fc1", 512, 32, 1), ("state.fc2", 512, 512, 0)]:
        x, w, b, y = rnd(rows, K), rnd(N, K), torch.randn(N, device=dev), rnd(rows, N)
        dy, wt, dx = rnd(rows, N), rnd(K, N), rnd(rows, K)
        line = f"rows {rows} {name:10s} N{N} K{K}:"
        for sk in (1, 0):
            L.lib.msc_set_knob(b"gemm_skinny", sk)
            tf = timeit(lambda: rt.call("msc_linear_fwd", x, w, b, None, y, None, rows, N, K, act, rt.dt))
            tb = timeit(lambda: rt.call("msc_linear_dgrad", dy, wt, None, None, 0, None, dx, rows, N, K, rt.dt))
            tot[sk] += tf + tb
            line += f"  {'skinny' if sk else 'tiled '} fwd {tf:5.1f} dgrad {tb:5.1f} us |"
        print(line)
    print(f"rows {rows} TOTAL fwd+dgrad: skinny {tot[1]:.0f} us, tiled {tot[0]:.0f} us (back-to-back launches: launch-rate bound below ~4 us each)")
L.lib.msc_set_knob(b"gemm_skinny", 1)
