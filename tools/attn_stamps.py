#!/usr/bin/env python3
"""Where the dK/dV pass of the fused attention backward spends a query-tile iteration (diagnostic build:
make -C multimodal_sc_amd/csrc alt ALT=attstamp ALT_SRC=attention ALT_FLAGS=-DATT_STAMP; MSC_LIB_PATH=.../libmsc_hip_attstamp.so).
Cycles of wave 0 per iteration, median over workgroups: S'/dP' stage, exp / dS / packing, dV / dK stage, whole iteration."""
import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L

rt = Runtime("bf16", "cuda:0")
dll = L.lib.load()
dll.msc_debug_attn_stamps.restype = ctypes.c_int
dll.msc_debug_attn_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
B, T, H = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 306, 8)))
D = H * 64
qkv = (torch.randn(B * T, 3 * D, device=rt.device) * 0.5).to(torch.bfloat16)
o = torch.empty(B * T, D, device=rt.device, dtype=torch.bfloat16)
do = (torch.randn(B * T, D, device=rt.device) * 0.5).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
Tp = L.query("msc_attention_tpad", T)
lse = torch.empty(B * H * Tp, device=rt.device)
delta = torch.empty(2 * lse.numel(), device=rt.device)
rt.call("msc_attention_fwd", qkv, o, lse, B, T, H, 64, 0.125, rt.dt)
for _ in range(3):
    rt.call("msc_attention_bwd", qkv, o, do, lse, delta, dqkv, B, T, H, 64, 0.125, rt.dt)
torch.cuda.synchronize()
n = min(B * H, 4096)
buf = (ctypes.c_ulonglong * (n * 8))()
assert dll.msc_debug_attn_stamps(buf, n) == 0
rows = [buf[i * 8:(i + 1) * 8] for i in range(n)]
rows = [r for r in rows if r[5]]
med = lambda f: statistics.median(f(r) for r in rows)
print(f"dK/dV pass, B*H = {B * H}, T = {T}: {len(rows)} workgroups stamped, {med(lambda r: r[5]):.0f} iterations per key tile; cycles per iteration (wave 0, median): "
      f"S'/dP' stage {med(lambda r: r[0] / r[5]):.0f} | exp + dS + packing {med(lambda r: r[1] / r[5]):.0f} | dV/dK stage {med(lambda r: r[2] / r[5]):.0f} | "
      f"whole loop {med(lambda r: r[3] / r[5]):.0f} | prologue (staging, once per workgroup) {med(lambda r: r[4]):.0f} | wave 0 entry -> last store {med(lambda r: r[7] - r[6]):.0f}")
t0 = min(r[6] for r in rows); t1 = max(r[7] for r in rows)
print(f"first entry -> last exit over the grid: {t1 - t0} ticks; workgroups entering in the first 10 % of that span: {sum(1 for r in rows if r[6] - t0 < 0.1 * (t1 - t0))}")
