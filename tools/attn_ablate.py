#!/usr/bin/env python3
"""Ablation timing of the fused attention forward kernel (see ATT_ABLATE in attention.hip).
    python tools/attn_ablate.py build   (CPU box)      python tools/attn_ablate.py   (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABL = os.path.join(ROOT, "tools", "abl")
CSRC = os.path.join(ROOT, "multimodal_sc_amd", "csrc")
NAMES = {0: "baseline", 1: "fwd: no K/V staging", 2: "fwd: no softmax", 3: "fwd: no PV", 4: "fwd: no QK", 5: "fwd: no Q loads",
         11: "bwd: no staging", 12: "bwd: no exp/VALU", 13: "bwd: no 2nd-stage MFMA", 14: "bwd: no 1st-stage MFMA"}


def build():
    os.makedirs(ABL, exist_ok=True)
    # the product's objects only (one per .hip source): alternative builds (make alt / stamp) leave extra objects beside them
    others = [os.path.join(CSRC, f[:-4] + ".o") for f in os.listdir(CSRC) if f.endswith(".hip") and f != "attention.hip"]
    procs = []
    for n in NAMES:
        obj = os.path.join(ABL, f"attn_abl{n}.o")
        procs.append((n, obj, subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result",
                                                 f"-I{ROOT}/include", f"-I{CSRC}", f"-DATT_ABLATE={n}", "-c", os.path.join(CSRC, "attention.hip"), "-o", obj])))
    for n, obj, pr in procs:
        assert pr.wait() == 0
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj, *others, "-o", os.path.join(ABL, f"libmsc_attn{n}.so")])
    print("built")


def child(n):
    sys.path.insert(0, ROOT)
    import multimodal_sc_amd.lib as L
    L.LIB_PATH = os.path.join(ABL, f"libmsc_attn{n}.so")
    import torch
    from multimodal_sc_amd.runtime import Runtime
    rt = Runtime("bf16", "cuda:0")
    B, T, H = 64, 306, 8
    D = H * 64
    qkv = (torch.randn(B * T, 3 * D, device=rt.device) * 0.5).to(torch.bfloat16)
    o = torch.empty(B * T, D, device=rt.device, dtype=torch.bfloat16)
    Tp = L.query("msc_attention_tpad", T)
    lse = torch.empty(B * H * Tp, device=rt.device)
    do = (torch.randn(B * T, D, device=rt.device) * 0.5).to(torch.bfloat16)
    dqkv, delta = torch.empty_like(qkv), torch.empty(2 * lse.numel(), device=rt.device)
    rt.call("msc_attention_fwd", qkv, o, lse, B, T, H, 64, 0.125, rt.dt)
    if n >= 10:
        fn = lambda: rt.call("msc_attention_bwd", qkv, o, do, lse, delta, dqkv, B, T, H, 64, 0.125, rt.dt)
    else:
        fn = lambda: rt.call("msc_attention_fwd", qkv, o, lse, B, T, H, 64, 0.125, rt.dt)
    if n == 0:
        fb = lambda: rt.call("msc_attention_bwd", qkv, o, do, lse, delta, dqkv, B, T, H, 64, 0.125, rt.dt)
        for _ in range(3):
            fb()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fb()
        e1.record()
        torch.cuda.synchronize()
        print(f"bwd baseline {e0.elapsed_time(e1) / 50 * 1e3:.1f} us", file=sys.stderr)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"RES|{n}|{e0.elapsed_time(e1) / 50 * 1e3:.1f}", flush=True)


def main():
    for n in NAMES:
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n)], capture_output=True, text=True, timeout=300)
        t = [ln.split("|")[2] for ln in o.stdout.splitlines() if ln.startswith("RES|")]
        print(f"{NAMES[n]:26s} {t[0] if t else 'FAILED ' + o.stderr[-300:]} us   {o.stderr.strip().splitlines()[-1] if n == 0 and o.stderr.strip() else ''}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    elif len(sys.argv) > 2 and sys.argv[1] == "child":
        child(int(sys.argv[2]))
    else:
        main()
