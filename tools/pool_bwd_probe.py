import sys, torch
sys.path.insert(0, '/root/repo')
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L
rt = Runtime("bf16", "cuda:0")
B, H, C = 64, 112, 64
Ho = 56
rnd = lambda *s: (torch.randn(*s, device=rt.device) * 0.5).to(torch.bfloat16)
x = rnd(B, H, H, C); dpool = rnd(B, Ho, Ho, C); xsel = rnd(B, Ho, Ho, C)
idx = torch.randint(0, 9, (B, Ho, Ho, C), dtype=torch.uint8, device=rt.device)
stats = torch.randn(4, C, device=rt.device).abs() + 0.5
gamma = torch.randn(C, device=rt.device); dg = torch.zeros(C, device=rt.device); db = torch.zeros(C, device=rt.device)
dx = torch.empty_like(x)
ws = torch.empty(L.query("msc_bn_bwd_workspace_bytes", C) // 4, device=rt.device)
f = lambda: rt.call("msc_bn_relu_pool_bwd", dpool, idx, xsel, x, stats[0], stats[1], gamma, stats[2], stats[3], dx, dg, db, ws, B, Ho, Ho, C, 3, 2, 1, 0, rt.dt)
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print("msc_bn_relu_pool_bwd (3 launches): %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
