import sys, os, torch
sys.path.insert(0, os.getcwd())
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L
rt = Runtime("bf16", "cuda:0")
B, T, H = 64, 306, 8
D = H * 64
qkv = (torch.randn(B * T, 3 * D, device=rt.device) * 0.5).to(torch.bfloat16)
o = torch.empty(B * T, D, device=rt.device, dtype=torch.bfloat16)
do = (torch.randn(B * T, D, device=rt.device) * 0.5).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
Tp = L.query("msc_attention_tpad", T)
lse = torch.empty(B * H * Tp, device=rt.device); delta = torch.empty(2 * lse.numel(), device=rt.device)
def t(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("fwd us", round(t(lambda: rt.call("msc_attention_fwd", qkv, o, lse, B, T, H, 64, 0.125, rt.dt)), 1),
      "bwd us", round(t(lambda: rt.call("msc_attention_bwd", qkv, o, do, lse, delta, dqkv, B, T, H, 64, 0.125, rt.dt)), 1))
