import sys, os, torch
sys.path.insert(0, os.getcwd())
from multimodal_sc_amd.runtime import Runtime
from multimodal_sc_amd import lib as L
for dt in ("f32", "bf16"):
    rt = Runtime(dt, "cuda:0")
    for (B, H, Cin, Cout, s) in [(2, 7, 512, 512, 1), (2, 14, 256, 256, 1), (64, 7, 512, 512, 1), (3, 7, 256, 512, 1)]:
        g = torch.Generator().manual_seed(1)
        x = (torch.randn(B, H, H, Cin, generator=g)).to(rt.tdtype).cuda()
        w = (torch.randn(Cout, 3, 3, Cin, generator=g) / 50).to(rt.tdtype).cuda()
        Ho = H
        M = B * Ho * Ho
        nrows = L.query("msc_conv2d_fwd_stat_rows", M, Cout, rt.dt)
        outs = []
        for use in (0, 1):
            y = torch.zeros(B, Ho, Ho, Cout, dtype=rt.tdtype, device="cuda")
            part = torch.zeros(nrows * 2 * Cout, device="cuda")
            wsb = L.query("msc_conv2d_splitk_workspace_bytes", M, Cout, 9 * Cin, rt.dt) if use else 0
            ws = torch.empty(max(wsb // 4, 1), device="cuda") if wsb else None
            rt.call("msc_conv2d_fwd", x, w, y, part, B, H, H, Cin, Cout, 3, 3, s, 1, ws, wsb, rt.dt)
            torch.cuda.synchronize()
            st = part.view(nrows, 2, Cout).sum(0)
            outs.append((y.float(), st, wsb))
        dy = (y.float() - outs[0][0])
        print(dt, (B, H, Cin, Cout), "fwd wsb", outs[1][2], "max|dy|", (outs[1][0] - outs[0][0]).abs().max().item(), "ref max", outs[0][0].abs().max().item(),
              "stat diff", (outs[1][1] - outs[0][1]).abs().max().item(), "stat max", outs[0][1].abs().max().item())
        # dgrad
        wt = (torch.randn(3 * 3 * Cout * Cin, generator=g) / 50).to(rt.tdtype).cuda()
        dyt = torch.randn(B, Ho, Ho, Cout, generator=g).to(rt.tdtype).cuda()
        res = torch.randn(B, H, H, Cin, generator=g).to(rt.tdtype).cuda()
        douts = []
        for use in (0, 1):
            dx = torch.zeros(B, H, H, Cin, dtype=rt.tdtype, device="cuda")
            wsd = L.query("msc_conv2d_splitk_workspace_bytes", B * H * H, Cin, 9 * Cout, rt.dt) if use else 0
            ws = torch.empty(max(wsd // 4, 1), device="cuda") if wsd else None
            rt.call("msc_conv2d_dgrad", dyt, wt, res, None, dx, B, H, H, Cin, Cout, 3, 3, s, 1, ws, wsd, rt.dt)
            torch.cuda.synchronize()
            douts.append(dx.float())
        print("   dgrad wsd", wsd, "max|d|", (douts[1] - douts[0]).abs().max().item(), "ref max", douts[0].abs().max().item())
