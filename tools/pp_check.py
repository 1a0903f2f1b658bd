#!/usr/bin/env python3
"""Dense linears of a fusion block at the C3 @ 64-frame shape: the persistent ping-pong kernel (csrc/gemm_pp.hip, knob gemm_pp = 1)
against the tiled 128x128 kernel (gemm_pp = 0) and the vendor library's plain matmul (yardstick only, never on the product path):
correctness against an fp32 matmul of the same bf16 inputs first, then interleaved rounds in one process, median us.
    usage: pp_check.py [rows=19584] [rounds=5]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L

rt = Runtime("bf16", "cuda:0")
dev = rt.device
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 306
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lib = L.lib.load()


def rnd(*s, sc=0.5):
    return (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)


def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def setpp(v):
    assert lib.msc_set_knob(b"gemm_pp", v) == 0


cases = []
for name, N, K in [("qkv", 1536, 512), ("proj", 512, 512), ("fc1", 2048, 512), ("fc2", 512, 2048)]:
    x, w, b = rnd(M, K), rnd(N, K, sc=K ** -0.5), torch.randn(N, device=dev) * 0.1
    y, r, p = torch.empty(M, N, device=dev, dtype=torch.bfloat16), rnd(M, N), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    act, res, pre = (L.ACT_GELU_DSAVE, 0, 1) if name == "fc1" else (0, name in ("proj", "fc2"), 0)
    fl = 2.0 * M * N * K

    def fwd(x=x, w=w, b=b, r=r, y=y, p=p, N=N, K=K, act=act, res=res, pre=pre):
        rt.call("msc_linear_fwd", x, w, b, r if res else None, y, p if pre else None, M, N, K, act, rt.dt)

    def fwd_ref(x=x, w=w, b=b, r=r, act=act, res=res):
        u = x.float() @ w.float().t() + b
        if act:
            u = torch.nn.functional.gelu(u)
        return u + (r.float() if res else 0)
    cases.append((f"fwd  {name:5s} N{N} K{K}", fl, fwd, fwd_ref, y, (lambda x=x, w=w: torch.matmul(x, w.t()))))
    dy, wt, dx = rnd(M, N), rnd(K, N, sc=N ** -0.5), torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    dref = rnd(M, K) if name == "fc2" else None

    def dgrad(dy=dy, wt=wt, dx=dx, dref=dref, N=N, K=K):
        rt.call("msc_linear_dgrad", dy, wt, None, dref, L.ACT_MUL_REF if dref is not None else 0, None, dx, M, N, K, rt.dt)

    def dgrad_ref(dy=dy, wt=wt, dref=dref):
        g = dy.float() @ wt.float().t()
        return g * dref.float() if dref is not None else g
    cases.append((f"dgrd {name:5s} out{K} red{N}", fl, dgrad, dgrad_ref, dx, (lambda dy=dy, wt=wt: torch.matmul(dy, wt.t()))))

MODES = [("pp", 1), ("pp128", 128), ("pp160", 160), ("pp192", 192), ("pp256", 256), ("tiled", 0)]      # knob gemm_pp: 1 = planner's tile height
tot = {k: 0.0 for k, _ in MODES}
tot["vendor"] = 0.0
for name, fl, fn, ref_fn, out, vend in cases:
    ref = ref_fn()
    errs = {}
    for mode, v in MODES:
        setpp(v)
        out.fill_(float("nan"))
        fn()
        torch.cuda.synchronize()
        errs[mode] = ((out.float() - ref).abs().max() / ref.abs().max().clamp_min(1.0)).item()
    res = {k: [] for k in tot}
    for _ in range(ROUNDS):
        for mode, v in MODES:
            setpp(v); res[mode].append(timeit(fn))
        res["vendor"].append(timeit(vend))
    setpp(1)
    med = {k: statistics.median(v) for k, v in res.items()}
    for k in tot:
        tot[k] += med[k]
    print(f"{name:26s} " + " | ".join(f"{k} {med[k]:6.1f} us {fl / med[k] / 1e6:5.0f} TF (err {errs[k]:.0e})" for k, _ in MODES) + f" | vendor plain matmul {med['vendor']:6.1f} us", flush=True)
print("TOTAL " + " | ".join(f"{k} {tot[k]:.0f} us" for k in tot))
# large square contraction (tools/gemm_big.py's shape)
for (m, n, k) in [(8192, 8192, 4096), (4096, 4096, 4096)]:
    x, w = rnd(m, k), rnd(n, k, sc=k ** -0.5)
    y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    fn = lambda: rt.call("msc_linear_fwd", x, w, None, None, y, None, m, n, k, 0, rt.dt)
    out = {}
    for mode, v in (("pp", 1), ("pp192", 192), ("pp256", 256), ("tiled", 0)):
        setpp(v)
        out[mode] = statistics.median(timeit(fn, 5) for _ in range(3))
    setpp(1)
    fn(); torch.cuda.synchronize()
    ref = x.float() @ w.float().t()
    err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
    vt = statistics.median(timeit(lambda: torch.matmul(x, w.t()), 5) for _ in range(3))
    fl = 2.0 * m * n * k
    print(f"{m}x{n}x{k}: " + " | ".join(f"{kk} {vv:.0f} us {fl / vv / 1e6:.0f} TF" for kk, vv in out.items()) + f" (pp err {err:.1e}) | vendor {vt:.0f} us {fl / vt / 1e6:.0f} TF")
