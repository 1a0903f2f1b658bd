import os, sys, torch
sys.path.insert(0, '/root/repo')
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L
rt = Runtime("bf16", "cuda:0")
rnd = lambda *s: (torch.randn(*s, device=rt.device) * 0.5).to(torch.bfloat16)
M, N, K = 19584, 2048, 512
x, w, b, y, d = rnd(M, K), rnd(N, K), torch.randn(N, device=rt.device), rnd(M, N), rnd(M, N)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(os.environ.get("MSC_LIB_PATH", "default").split("_")[-1], "fc1 gelu+C2: %.1f us | plain N2048 K512: %.1f us" % (
    t(lambda: rt.call("msc_linear_fwd", x, w, b, None, y, d, M, N, K, L.ACT_GELU_DSAVE, rt.dt)),
    t(lambda: rt.call("msc_linear_fwd", x, w, b, None, y, None, M, N, K, L.ACT_NONE, rt.dt))))
