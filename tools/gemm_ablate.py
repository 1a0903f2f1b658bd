#!/usr/bin/env python3
"""Ablation timing of the contraction kernel: which of {LDS fill (DMA), fragment reads, MFMA, epilogue} bounds it.

    python tools/gemm_ablate.py build     # (CPU box) builds tools/abl/libmsc_abl{0..4}.so from csrc/gemm.hip -DGEMM_ABLATE=n
    python tools/gemm_ablate.py           # (GPU box) one child process per variant, same shapes, HIP-event timing
Variant results are wrong by construction (see GEMM_ABLATE in gemm.hip); only the times mean something.
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABL = os.path.join(ROOT, "tools", "abl")
CSRC = os.path.join(ROOT, "multimodal_sc_amd", "csrc")
NAMES = {0: "baseline", 1: "no loop DMA", 2: "no MFMA", 3: "no frag reads/MFMA", 4: "no epilogue"}


def build():
    os.makedirs(ABL, exist_ok=True)
    others = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".o") and f != "gemm.o"]
    procs = []
    for n in NAMES:
        obj = os.path.join(ABL, f"gemm_abl{n}.o")
        extra = os.environ.get("ABL_FLAGS", "").split()
        procs.append((n, obj, subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result",
                                                 f"-I{ROOT}/include", f"-I{CSRC}", f"-DGEMM_ABLATE={n}", *extra, "-c", os.path.join(CSRC, "gemm.hip"), "-o", obj])))
    for n, obj, pr in procs:
        assert pr.wait() == 0
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj, *others, "-o", os.path.join(ABL, f"libmsc_abl{n}.so")])
    print("built", sorted(os.listdir(ABL)))


def child(n):
    sys.path.insert(0, ROOT)
    import multimodal_sc_amd.lib as L
    L.LIB_PATH = os.path.join(ABL, f"libmsc_abl{n}.so")
    import torch
    from multimodal_sc_amd.runtime import Runtime
    rt = Runtime("bf16", "cuda:0")
    dev = rt.device

    def rnd(*s):
        return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)

    def timeit(fn, it=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it * 1e3

    out = []
    for name, M, N, K in [("lin qkv 19584x1536x512", 19584, 1536, 512), ("lin fc2 19584x512x2048", 19584, 512, 2048), ("lin 8192x8192x4096", 8192, 8192, 4096)]:
        x, w, y = rnd(M, K), rnd(N, K), rnd(M, N)
        t = timeit(lambda: rt.call("msc_linear_fwd", x, w, None, None, y, None, M, N, K, 0, rt.dt))
        out.append((name, t, 2.0 * M * N * K))
        if M == 19584:
            dy, dw = rnd(M, N), torch.empty(N, K, device=dev)
            wsb = L.query("msc_wgrad_workspace_bytes", N, K, M)
            ws = torch.empty(wsb // 4, device=dev)
            t = timeit(lambda: rt.call("msc_linear_wgrad", x, dy, dw, ws, wsb, M, N, K, 0, rt.dt))
            out.append(("wgrad " + name, t, 2.0 * M * N * K))
    B = 64
    for name, H, Cin, Cout, Kk, s, p in [("conv1 7x7/2 H224 8->64", 224, 8, 64, 7, 2, 3), ("l1 3x3 H56 64->64", 56, 64, 64, 3, 1, 1), ("l3 3x3 H14 256->256", 14, 256, 256, 3, 1, 1), ("bev2 3x3/2 H64 128->256", 64, 128, 256, 3, 2, 1)]:
        Ho = (H + 2 * p - Kk) // s + 1
        x, w, y = rnd(B, H, H, Cin), rnd(Cout, Kk, Kk, Cin), rnd(B, Ho, Ho, Cout)
        part = torch.empty((B * Ho * Ho + 63) // 64 * 2 * Cout, device=dev)
        t = timeit(lambda: rt.call("msc_conv2d_fwd", x, w, y, part, B, H, H, Cin, Cout, Kk, Kk, s, p, None, 0, rt.dt))
        out.append(("conv_fwd " + name, t, 2.0 * B * Ho * Ho * Cout * Kk * Kk * Cin))
        dw = torch.empty(Cout, Kk, Kk, Cin, device=dev)
        wsb = L.query("msc_wgrad_workspace_bytes", Cout, Kk * Kk * Cin, B * Ho * Ho)
        ws = torch.empty(max(wsb // 4, 1), device=dev)
        t = timeit(lambda: rt.call("msc_conv2d_wgrad", x, y, dw, ws, wsb, B, H, H, Cin, Cout, Kk, Kk, s, p, 0, rt.dt))
        out.append(("conv_wgrad " + name, t, 2.0 * B * Ho * Ho * Cout * Kk * Kk * Cin))
        wt, dx = rnd(Kk, Kk, Cin, Cout), rnd(B, H, H, Cin)
        t = timeit(lambda: rt.call("msc_conv2d_dgrad", y, wt, None, dx, B, H, H, Cin, Cout, Kk, Kk, s, p, None, 0, rt.dt))
        out.append(("conv_dgrad " + name, t, 2.0 * B * Ho * Ho * Cout * Kk * Kk * Cin))
    for name, t, fl in out:
        print(f"RES|{n}|{name}|{t:.1f}|{fl / t / 1e6:.0f}", flush=True)


def main():
    res = {}
    for n in NAMES:
        if not os.path.exists(os.path.join(ABL, f"libmsc_abl{n}.so")):
            continue
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n)], capture_output=True, text=True, timeout=300)
        for ln in o.stdout.splitlines():
            if ln.startswith("RES|"):
                _, v, name, t, tf = ln.split("|")
                res.setdefault(name, {})[int(v)] = (float(t), float(tf))
        if o.returncode:
            print("variant", n, "failed:", o.stderr[-400:])
    print(f"{'shape':44s}" + "".join(f"{NAMES[n]:>20s}" for n in NAMES))
    for name, d in res.items():
        print(f"{name:44s}" + "".join(f"{d.get(n, (0, 0))[0]:12.1f} us     " for n in NAMES) + f"  base {d.get(0, (0, 0))[1]:.0f} TF/s")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    elif len(sys.argv) > 2 and sys.argv[1] == "child":
        child(int(sys.argv[2]))
    else:
        main()
