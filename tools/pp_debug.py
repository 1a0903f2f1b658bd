#!/usr/bin/env python3
"""debug aid: where does the ping-pong kernel's output differ from an fp32 matmul?  per 256x256 tile / per 64x32 quadrant block"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L
rt = Runtime("bf16", "cuda:0")
dev = rt.device
lib = L.lib.load()
M, N, K = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (19584, 1536, 512)
if len(sys.argv) > 4:
    assert lib.msc_set_knob(b"gemm_pp", int(sys.argv[4])) == 0
torch.manual_seed(0)
x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
ref = x.float() @ w.float().t()
for trial in range(3):
    y = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    rt.call("msc_linear_fwd", x, w, None, None, y, None, M, N, K, 0, rt.dt)
    torch.cuda.synchronize()
    d = (y.float() - ref).abs()
    bad = ~(d < 0.05)
    print(f"trial {trial}: bad elements {int(bad.sum())} of {bad.numel()} (nan {int(torch.isnan(y.float()).sum())})")
    if bad.any():
        Mp, Np = (M + 255) // 256 * 256, (N + 255) // 256 * 256
        bp = torch.zeros(Mp, Np, dtype=torch.bool, device=dev)
        bp[:M, :N] = bad
        tiles = bp.view(Mp // 256, 256, Np // 256, 256).permute(0, 2, 1, 3).reshape(Mp // 256, Np // 256, -1).sum(-1)
        nz = tiles.nonzero()
        print("  bad tiles (tile_m, tile_n, count):", [(int(a), int(b), int(tiles[a, b])) for a, b in nz[:40]], "total tiles", len(nz))
        a, b = [int(v) for v in nz[0]]
        t = bp[a * 256:(a + 1) * 256, b * 256:(b + 1) * 256]
        blk = t.view(16, 16, 8, 32).permute(0, 2, 1, 3).reshape(16, 8, -1).sum(-1)      # 16-row x 32-col blocks
        print("  first bad tile, bad count per (16-row frag, 32-col half):")
        print(blk.cpu().numpy())
        rows = t.any(1).nonzero().flatten()[:20].tolist(); cols = t.any(0).nonzero().flatten()[:40].tolist()
        print("  rows", rows, "cols", cols)
        r0, c0 = a * 256 + rows[0], b * 256 + cols[0]
        print("  sample", y[r0, c0:c0 + 8].float().cpu().numpy(), ref[r0, c0:c0 + 8].cpu().numpy())
