#!/usr/bin/env python3
"""Where a workgroup of the tiled contraction kernel spends its cycles on the CONVOLUTION launches of the trunk (diagnostic build:
make -C multimodal_sc_amd/csrc stamp; run with MSC_LIB_PATH=multimodal_sc_amd/csrc/libmsc_hip_stamp.so)."""
import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L

rt = Runtime("bf16", "cuda:0")
dll = L.lib.load()
dll.msc_debug_gemm_stamps.restype = ctypes.c_int
dll.msc_debug_gemm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dll.msc_debug_gemm_stamps_ep.restype = ctypes.c_int
dll.msc_debug_gemm_stamps_ep.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = rt.device
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
B = 64
convs = [c for c in [("l1 3x3", 56, 64, 64, 3, 1, 1), ("l2 3x3/2", 56, 64, 128, 3, 2, 1), ("l2 3x3", 28, 128, 128, 3, 1, 1), ("l2 1x1/2", 56, 64, 128, 1, 2, 0),
         ("l3 3x3", 14, 256, 256, 3, 1, 1), ("l4 3x3", 7, 512, 512, 3, 1, 1)] if len(sys.argv) < 2 or sys.argv[1] in c[0]]


def report(tag, nmax=1 << 16):
    torch.cuda.synchronize()
    info = (ctypes.c_int * 8)()
    dll.msc_debug_last_gemm(info)
    buf = (ctypes.c_ulonglong * (nmax * 8))()
    assert dll.msc_debug_gemm_stamps(buf, nmax) == 0
    rows = [buf[i * 8:(i + 1) * 8] for i in range(nmax)]
    rows = [r for r in rows if r[5] > r[0] > 0]
    # records of the LAST launch: the newest cluster of entry stamps (earlier launches with more workgroups leave stale records behind)
    rows.sort(key=lambda r: r[0])
    gaps = sorted(rows[i + 1][0] - rows[i][0] for i in range(len(rows) - 1))
    thr = max(50 * gaps[len(gaps) // 2], 200)
    k = len(rows) - 1
    while k > 0 and rows[k][0] - rows[k - 1][0] < thr:
        k -= 1
    rows = rows[k:]
    t0 = min(r[0] for r in rows)
    tot = max(r[5] for r in rows) - t0
    med = lambda f: statistics.median(f(r) for r in rows)
    print(f"{tag}: variant {list(info)[:6]} {len(rows)} wgs, span {tot} ticks; per wg median: life {med(lambda r: r[5]-r[0]):.0f} prologue {med(lambda r: r[1]-r[0]):.0f} "
          f"kloop {med(lambda r: r[3]):.0f} (wait {med(lambda r: r[2]):.0f}) epilogue {med(lambda r: r[4]):.0f} steps {med(lambda r: r[7]):.0f} "
          f"-> per step {med(lambda r: r[3] / max(r[7], 1)):.0f} (wait {med(lambda r: r[2] / max(r[7], 1)):.0f}); start spread {max(r[0] for r in rows) - t0}", flush=True)


for name, H, Cin, Cout, K, s, p in convs:
    Ho = (H + 2 * p - K) // s + 1
    x, w, y = rnd(B, H, H, Cin), rnd(Cout, K, K, Cin), rnd(B, Ho, Ho, Cout)
    part = torch.empty((B * Ho * Ho + 63) // 64 * 2 * Cout, device=dev)
    wsb = L.query("msc_conv2d_splitk_workspace_bytes", B * Ho * Ho, Cout, K * K * Cin, rt.dt)
    wsf = torch.empty(max(wsb // 4, 1), device=dev) if wsb else None
    for _ in range(2):
        rt.call("msc_conv2d_fwd", x, w, y, part, B, H, H, Cin, Cout, K, K, s, p, wsf, wsb, rt.dt)
    report(f"fwd   {name}")
    wt, dx = rnd(Cin, K, K, Cout), rnd(B, H, H, Cin)
    wsd = L.query("msc_conv2d_splitk_workspace_bytes", B * H * H, Cin, K * K * Cout, rt.dt) if s == 1 else 0
    wsg = torch.empty(max(wsd // 4, 1), device=dev) if wsd else None
    for _ in range(2):
        rt.call("msc_conv2d_dgrad", y, wt, None, None, dx, B, H, H, Cin, Cout, K, K, s, p, wsg, wsd, rt.dt)
    report(f"dgrad {name}")
