#!/usr/bin/env python3
"""Where a workgroup of the contraction kernel spends its cycles (diagnostic build: make -C multimodal_sc_amd/csrc stamp;
run with MSC_LIB_PATH=multimodal_sc_amd/csrc/libmsc_hip_stamp.so).  Shares of the shader clock per workgroup (median over
workgroups): prologue (entry -> first data), K loop, of which waiting for DMA + barrier, epilogue."""
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
M = B * 306
shapes = {"qkv": (M, 1536, 512), "fc2": (M, 512, 2048), "big": (8192, 8192, 4096), "fc1": (M, 2048, 512), "fc1plain": (M, 2048, 512)}
for name in (sys.argv[1:] or ["qkv", "fc2", "big"]):
    Mm, N, K = shapes[name]
    x, w, b, y = rnd(Mm, K), rnd(N, K), torch.randn(N, device=dev), rnd(Mm, N)
    pre = rnd(Mm, N) if name == "fc1" else None          # fc1: GELU epilogue + second (pre-activation) output
    for tile in ("128x128", "128x64", "256x256"):
        os.environ["MSC_GEMM_TILE"] = tile
        for _ in range(3):
            rt.call("msc_linear_fwd", x, w, b, None, y, pre, Mm, N, K, 2 if name == "fc1" else 0, rt.dt)
        torch.cuda.synchronize()
        bm, bn = (int(v) for v in tile.split("x"))
        ntiles = -(-Mm // bm) * -(-N // bn)
        nwg = min(ntiles, 256) if bm == 256 else ntiles
        buf = (ctypes.c_ulonglong * (nwg * 8))()
        assert dll.msc_debug_gemm_stamps(buf, nwg) == 0
        rows = [buf[i * 8:(i + 1) * 8] for i in range(nwg)]
        t0 = min(r[0] for r in rows)
        tot = max(r[5] for r in rows) - t0
        med = lambda f: statistics.median(f(r) for r in rows)
        print(f"{name} {Mm}x{N}x{K} tile {tile}: {nwg} wgs, kernel span {tot} cyc; per wg median: life {med(lambda r: r[5]-r[0]):.0f} "
              f"prologue {med(lambda r: r[1]-r[0]):.0f} kloop {med(lambda r: r[3]):.0f} (wait {med(lambda r: r[2]):.0f}) "
              f"epilogue {med(lambda r: r[4]):.0f} tiles {med(lambda r: r[6]):.0f} steps {med(lambda r: r[7]):.0f} "
              f"-> per step {med(lambda r: r[3] / max(r[7], 1)):.0f} cyc (wait {med(lambda r: r[2] / max(r[7], 1)):.0f}), "
              f"start spread {max(r[0] for r in rows) - t0}")
        if bm != 256:
            assert dll.msc_debug_gemm_stamps_ep(buf, nwg) == 0
            ep = [buf[i * 8:(i + 1) * 8] for i in range(nwg)]
            m2 = lambda i: statistics.median(r[i] for r in ep)
            print(f"    epilogue of wave 0: barrier after the K loop {m2(0):.0f} | statistics {m2(1):.0f} | accumulators -> slab + barrier {m2(2):.0f} | "
                  f"slab -> stores issued {m2(3):.0f} | barrier between passes (waits for the stores) {m2(4):.0f} | tail {m2(5):.0f}")
