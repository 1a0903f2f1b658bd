#!/usr/bin/env python3
"""Tile sweep of the contraction kernel over the C3 @ 64-frame shapes: every legal MSC_GEMM_TILE per shape, interleaved
rounds in one process (the env knob is read per call), median us.  Marks the tile the built-in heuristic picks."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L

rt = Runtime("bf16", "cuda:0")
dev = rt.device
B = int(os.environ.get("SWEEP_B", "64"))


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


cases = []   # (name, flops, fn, legal tiles)
ALL = ["64x64", "64x128", "128x64", "128x128", "128x128p", "256x128", "256x256"]     # 128x128p: persistent, two workgroups per CU
if os.environ.get("SWEEP_TILES"):
    ALL = os.environ["SWEEP_TILES"].split(",")
MN = ["64x64", "64x128", "128x64", "128x128"]
M = B * 306
for name, N, K in [("qkv", 1536, 512), ("proj", 512, 512), ("fc1", 2048, 512), ("fc2", 512, 2048)]:
    x, w, b = rnd(M, K), rnd(N, K), torch.randn(N, device=dev)
    y, r, p = rnd(M, N), rnd(M, N), rnd(M, N)
    fl = 2.0 * M * N * K
    act, res, pre = (2, 0, 1) if name == "fc1" else (0, name in ("proj", "fc2"), 0)
    cases.append((f"lin_fwd {name} N{N} K{K}", fl, (lambda x=x, w=w, b=b, r=r, y=y, p=p, N=N, K=K, act=act, res=res, pre=pre:
                  rt.call("msc_linear_fwd", x, w, b, r if res else None, y, p if pre else None, M, N, K, act, rt.dt)), ALL))
    dy, wt, dx = rnd(M, N), rnd(K, N), rnd(M, K)
    cases.append((f"lin_dgrad {name} out{K} red{N}", fl, (lambda dy=dy, wt=wt, dx=dx, N=N, K=K:
                  rt.call("msc_linear_dgrad", dy, wt, None, None, 0, None, dx, M, N, K, rt.dt)), ALL))
    dw = torch.empty(N, K, device=dev)
    wsb = L.query("msc_wgrad_workspace_bytes", N, K, M) * 4
    ws = torch.empty(wsb // 4, device=dev)
    cases.append((f"lin_wgrad {name} {N}x{K}", fl, (lambda x=x, dy=dy, dw=dw, ws=ws, wsb=wsb, N=N, K=K:
                  rt.call("msc_linear_wgrad", x, dy, dw, None, ws, wsb, M, N, K, 0, rt.dt)), MN))
convs = [("conv1 7x7/2", 224, 8, 64, 7, 2, 3), ("l1 3x3", 56, 64, 64, 3, 1, 1), ("l2 3x3/2", 56, 64, 128, 3, 2, 1), ("l2 3x3", 28, 128, 128, 3, 1, 1),
         ("l3 3x3/2", 28, 128, 256, 3, 2, 1), ("l3 3x3", 14, 256, 256, 3, 1, 1), ("l4 3x3/2", 14, 256, 512, 3, 2, 1), ("l4 3x3", 7, 512, 512, 3, 1, 1),
         ("bev1 3x3/2", 128, 64, 128, 3, 2, 1), ("bev2 3x3/2", 64, 128, 256, 3, 2, 1), ("bev3 3x3/2", 32, 256, 512, 3, 2, 1)]
for name, H, Cin, Cout, K, s, p in convs:
    Ho = (H + 2 * p - K) // s + 1
    x, w = rnd(B, H, H, Cin), rnd(Cout, K, K, Cin)
    y = rnd(B, Ho, Ho, Cout)
    part = torch.empty((B * Ho * Ho + 63) // 64 * 2 * Cout, device=dev)
    fl = 2.0 * B * Ho * Ho * Cout * K * K * (3 if Cin == 8 else Cin)
    wsf = torch.empty(8 * B * Ho * Ho * Cout, device=dev)
    cases.append((f"conv_fwd {name} {Cin}->{Cout}", fl, (lambda x=x, w=w, y=y, part=part, wsf=wsf, H=H, Cin=Cin, Cout=Cout, K=K, s=s, p=p:
                  rt.call("msc_conv2d_fwd", x, w, y, part, B, H, H, Cin, Cout, K, K, s, p, wsf, wsf.numel() * 4, rt.dt)), ALL))
    if Cin != 8:
        wt, dx = rnd(Cin, K, K, Cout), rnd(B, H, H, Cin)
        wsg = torch.empty(8 * B * H * H * Cin, device=dev)
        cases.append((f"conv_dgrad {name}", fl, (lambda y=y, wt=wt, dx=dx, wsg=wsg, H=H, Cin=Cin, Cout=Cout, K=K, s=s, p=p:
                      rt.call("msc_conv2d_dgrad", y, wt, None, None, dx, B, H, H, Cin, Cout, K, K, s, p, wsg if s == 1 else None, wsg.numel() * 4 if s == 1 else 0, rt.dt)), ALL))
    dw = torch.empty(Cout, K, K, Cin, device=dev)
    wsb = L.query("msc_wgrad_workspace_bytes", Cout, K * K * Cin, B * Ho * Ho) * 4
    ws = torch.empty(wsb // 4, device=dev)
    cases.append((f"conv_wgrad {name}", fl, (lambda x=x, y=y, dw=dw, ws=ws, wsb=wsb, H=H, Cin=Cin, Cout=Cout, K=K, s=s, p=p:
                  rt.call("msc_conv2d_wgrad", x, y, dw, ws, wsb, B, H, H, Cin, Cout, K, K, s, p, 0, rt.dt)), MN))

sel = sys.argv[1] if len(sys.argv) > 1 else ""
tot_def = tot_best = 0.0
for name, fl, fn, tiles in cases:
    if sel and sel not in name:
        continue
    res = {t: [] for t in ["auto"] + tiles}
    for rnd_ in range(3):
        for t in res:
            if t == "auto":
                os.environ.pop("MSC_GEMM_TILE", None)
                os.environ.pop("MSC_GEMM_P128", None)
                os.environ.pop("MSC_GEMM_NP256", None)
            else:
                os.environ["MSC_GEMM_TILE"] = t.rstrip("pn")
                os.environ["MSC_GEMM_P128"] = "1" if t.endswith("p") else "0"
                os.environ["MSC_GEMM_NP256"] = "1" if t.endswith("n") else "0"      # 256x128n: experiment build (make alt)
            try:
                res[t].append(timeit(fn))
            except RuntimeError:
                res[t].append(float("inf"))
    os.environ.pop("MSC_GEMM_TILE", None)
    os.environ.pop("MSC_GEMM_P128", None)
    os.environ.pop("MSC_GEMM_NP256", None)
    med = {t: statistics.median(v) for t, v in res.items()}
    best = min((t for t in med if t != "auto"), key=lambda t: med[t])
    tot_def += med["auto"]; tot_best += med[best]
    print(f"{name:34s} auto {med['auto']:7.1f} us {fl/med['auto']/1e6:6.0f} TF | best {best:8s} {med[best]:7.1f} us {fl/med[best]/1e6:6.0f} TF | " +
          " ".join(f"{t}:{med[t]:.0f}" for t in tiles), flush=True)
print(f"TOTAL auto {tot_def:.0f} us, best-per-shape {tot_best:.0f} us")
