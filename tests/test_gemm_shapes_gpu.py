"""The PRODUCTION variants of the contraction kernels at the shapes that select them (round-3 review, weak item 1).

test_kernels_gpu.py holds every kernel to the tight per-kernel tolerances, but its shapes are small: the planner shrinks them
to 64-wide tiles, so the instantiations that carry the benchmark (128x128 / 512 threads, the XCD-pinned split-K weight
gradients, the class-interleaved strided dgrad, the persistent ping-pong kernel of the dense linears) were only ever compared
end to end.  Here every contraction of the C3 @ 64-frame step (tools/gemm_sweep.py's list) runs through the C-ABI with the
DEFAULT planner and is compared with fp32 PyTorch-CPU computed from the bf16-rounded inputs, at test_kernels_gpu.py's own
tolerances (1.2e-2 * max|ref| on bf16 outputs, 2e-3 on fp32 outputs); msc_debug_last_gemm says which variant ran, and the
test asserts it is the one the benchmark uses.  A 19584 x 1536 x 512 fp32 matmul is well under a second on the box's host cores.
"""
import math

import pytest
import torch
import torch.nn.functional as TF

from conftest import last_gemm

pytestmark = pytest.mark.gpu

B = 64
T = 306
M = B * T           # 19584 token rows of the fusion transformer at 64 frames


@pytest.fixture(scope="module")
def rt(cuda):
    from multimodal_sc_amd.runtime import Runtime
    return Runtime("bf16", "cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def qb(t):
    return t.to(torch.bfloat16).float()


def dev(t, rt):
    return t.to(rt.device).to(rt.tdtype).contiguous()


def check(name, got, ref, t):
    torch.cuda.synchronize()
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    err = (got - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    assert err <= t * scale, f"{name}: max err {err:.3e} > {t:.1e} * {scale:.3e}"


LINEARS = [("qkv", 1536, 512), ("proj", 512, 512), ("fc1", 2048, 512), ("fc2", 512, 2048)]


@pytest.mark.parametrize("pp", [1, 256, 0])        # planner's tile height / 256 rows forced / tiled family
@pytest.mark.parametrize("rows", [M, M // 2])          # whole batch / one of the two half-batch chains
@pytest.mark.parametrize("name,N,K", LINEARS)
def test_fusion_linear_forward_at_benchmark_shape(rt, knob, name, N, K, rows, pp):
    """forward of the four dense layers of a fusion block with exactly the epilogue the step uses: qkv bias; proj / fc2 bias +
    residual; fc1 bias + GELU with the derivative as second output"""
    from multimodal_sc_amd import lib as L
    knob("gemm_pp", pp)
    x, w, b = rnd(rows, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3, scale=0.1)
    res = name in ("proj", "fc2")
    r = rnd(rows, N, seed=4) if res else None
    pre = qb(x) @ qb(w).t() + b
    y = torch.full((rows, N), float("nan"), device=rt.device, dtype=rt.tdtype)
    if name == "fc1":
        u = pre.clone().requires_grad_()
        ref = TF.gelu(u)
        ref.sum().backward()
        d = torch.full((rows, N), float("nan"), device=rt.device, dtype=rt.tdtype)
        rt.call("msc_linear_fwd", dev(x, rt), dev(w, rt), b.to(rt.device), None, y, d, rows, N, K, L.ACT_GELU_DSAVE, rt.dt)
        check("gelu'", d, u.grad, 1.2e-2)
    else:
        ref = pre + (qb(r) if res else 0)
        rt.call("msc_linear_fwd", dev(x, rt), dev(w, rt), b.to(rt.device), dev(r, rt) if res else None, y, None, rows, N, K, L.ACT_NONE, rt.dt)
    info = last_gemm()
    if pp:
        assert info[0] == 3 and info[1] in ((128, 160, 192, 256) if pp == 1 else (pp,)) and info[2:4] == (256, 512), info      # persistent ping-pong, 8 waves
    else:
        assert info[0] == 1 and info[1:4] == (128, 128, 512), info          # one tile per workgroup, 128 x 128, 8 waves
    check("y", y, ref, 1.2e-2)
    y2 = torch.empty_like(y)          # bit-identical when repeated (fixed tile -> workgroup map, no atomics)
    if name == "fc1":
        rt.call("msc_linear_fwd", dev(x, rt), dev(w, rt), b.to(rt.device), None, y2, torch.empty_like(y), rows, N, K, L.ACT_GELU_DSAVE, rt.dt)
    else:
        rt.call("msc_linear_fwd", dev(x, rt), dev(w, rt), b.to(rt.device), dev(r, rt) if res else None, y2, None, rows, N, K, L.ACT_NONE, rt.dt)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)


@pytest.mark.parametrize("pp", [1, 0])
@pytest.mark.parametrize("name,N,K", LINEARS)
def test_fusion_linear_dgrad_at_benchmark_shape(rt, knob, name, N, K, pp):
    """data gradients: dx[M,K] = dy[M,N] wt[K,N]^T; fc2's dgrad multiplies by the stored GELU derivative (MSC_ACT_MUL_REF)"""
    from multimodal_sc_amd import lib as L
    knob("gemm_pp", pp)
    dy, w = rnd(M, N, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(N))
    wt = qb(w).t().contiguous()                     # [K][N]
    ref = qb(dy) @ qb(w)
    dref = None
    if name == "fc2":
        dref = rnd(M, K, seed=3, scale=0.5)
        ref = ref * qb(dref)
    dx = torch.full((M, K), float("nan"), device=rt.device, dtype=rt.tdtype)
    rt.call("msc_linear_dgrad", dev(dy, rt), dev(wt, rt), None, dev(dref, rt) if dref is not None else None,
            L.ACT_MUL_REF if dref is not None else L.ACT_NONE, None, dx, M, N, K, rt.dt)
    info = last_gemm()
    if pp:
        assert info[0] == 3 and info[1] in (128, 160, 192, 256) and info[2:4] == (256, 512), info
    else:
        assert info[0] == 1 and info[1:4] == (128, 128, 512), info
    check("dx", dx, ref, 2.4e-2)


@pytest.mark.parametrize("bm", [128, 160, 192, 256])
@pytest.mark.parametrize("name,N,K", LINEARS)
def test_ping_pong_kernel_bitwise_under_concurrent_load(rt, knob, name, N, K, bm):
    """The persistent ping-pong kernel keeps four LDS-DMA pieces and its epilogue stores in flight behind COUNTED waits: a count that
    is one instruction too generous reads a piece before it has landed — only when memory is slow.  Every tile height, every epilogue
    of the path (bias / bias + residual / GELU + derivative), repeated while another stream streams 1 GiB through HBM, must give the
    bits of the quiet run."""
    from multimodal_sc_amd import lib as L
    knob("gemm_pp", bm)
    x, w, b = dev(rnd(M, K, seed=1), rt), dev(rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rt), rnd(N, seed=3, scale=0.1).to(rt.device)
    res = name in ("proj", "fc2")
    r = dev(rnd(M, N, seed=4), rt) if res else None
    act = L.ACT_GELU_DSAVE if name == "fc1" else L.ACT_NONE

    def run():
        y = torch.full((M, N), float("nan"), device=rt.device, dtype=rt.tdtype)
        d = torch.full((M, N), float("nan"), device=rt.device, dtype=rt.tdtype) if name == "fc1" else None
        rt.call("msc_linear_fwd", x, w, b, r, y, d, M, N, K, act, rt.dt)
        return y, d
    quiet, quiet_d = run()
    torch.cuda.synchronize()
    assert last_gemm()[0] == 3 and last_gemm()[1] == bm
    side = torch.cuda.Stream()
    big = torch.empty(1 << 28, device=rt.device)          # 1 GiB
    big2 = torch.empty_like(big)
    for rep in range(4):
        with torch.cuda.stream(side):
            for _ in range(3):
                big2.copy_(big)
        y, d = run()
        torch.cuda.synchronize()
        assert torch.equal(y, quiet), f"repetition {rep}: output differs from the quiet run"
        if d is not None:
            assert torch.equal(d, quiet_d)


@pytest.mark.parametrize("name,N,K", LINEARS)
def test_fusion_linear_wgrad_at_benchmark_shape(rt, name, N, K):
    """weight + bias gradient over the whole batch: 128 x 128 tiles, 8 waves, K slices pinned to XCDs, fixed-order reduce"""
    from multimodal_sc_amd import lib as L
    x, dy = rnd(M, K, seed=1), rnd(M, N, seed=2, scale=0.1)
    ref_dw = qb(dy).t() @ qb(x)
    ref_db = qb(dy).sum(0)
    dw = torch.full((N, K), float("nan"), device=rt.device)
    db = torch.full((N,), float("nan"), device=rt.device)
    wsb = L.query("msc_wgrad_workspace_bytes", N, K, M)
    ws = torch.empty(wsb // 4, device=rt.device)
    rt.call("msc_linear_wgrad", dev(x, rt), dev(dy, rt), dw, db, ws, wsb, M, N, K, 0, rt.dt)
    info = last_gemm()
    assert info[0] == 1 and info[1:4] == (128, 128, 512) and info[4] >= 8 and info[4] % 8 == 0 and info[5] == 1, info
    check("dw", dw, ref_dw, 2e-3 * 4)
    check("db", db, ref_db, 2e-3 * 4)


# ResNet-18 trunk + BEV convolutions of C3 (tools/gemm_sweep.py): name, H, Cin, Cout, K, stride, pad
CONVS = [("l1 3x3", 56, 64, 64, 3, 1, 1), ("l2 3x3/2", 56, 64, 128, 3, 2, 1), ("l2 1x1/2", 56, 64, 128, 1, 2, 0), ("l2 3x3", 28, 128, 128, 3, 1, 1),
         ("l3 3x3/2", 28, 128, 256, 3, 2, 1), ("l3 3x3", 14, 256, 256, 3, 1, 1), ("l4 3x3/2", 14, 256, 512, 3, 2, 1), ("l4 1x1/2", 14, 256, 512, 1, 2, 0),
         ("l4 3x3", 7, 512, 512, 3, 1, 1), ("bev1 3x3/2", 128, 64, 128, 3, 2, 1), ("bev2 3x3/2", 64, 128, 256, 3, 2, 1), ("bev3 3x3/2", 32, 256, 512, 3, 2, 1)]


@pytest.mark.parametrize("name,H,Cin,Cout,K,s,p", CONVS)
def test_conv_at_benchmark_shape(rt, name, H, Cin, Cout, K, s, p):
    """implicit-GEMM convolution forward (with the BatchNorm partial sums and the split-K workspace, as the model calls it), data
    gradient (with the residual input) and weight gradient at 64 frames, default planner"""
    from multimodal_sc_amd import lib as L
    Ho = (H + 2 * p - K) // s + 1
    x = rnd(B, H, H, Cin, seed=1)
    w = rnd(Cout, K, K, Cin, seed=2, scale=1 / math.sqrt(K * K * Cin))
    dy = rnd(B, Ho, Ho, Cout, seed=3, scale=0.1)
    xq, wq = qb(x).requires_grad_(), qb(w).requires_grad_()
    ref = TF.conv2d(xq.permute(0, 3, 1, 2), wq.permute(0, 3, 1, 2), None, s, p)
    ref.backward(qb(dy).permute(0, 3, 1, 2))
    rows = B * Ho * Ho
    xd, wd = dev(x, rt), dev(w, rt)
    y = torch.full((B, Ho, Ho, Cout), float("nan"), device=rt.device, dtype=rt.tdtype)
    nrows = L.query("msc_conv2d_fwd_stat_rows", rows, Cout, rt.dt)
    part = torch.full((nrows, 2, Cout), float("nan"), device=rt.device)
    wsb = L.query("msc_conv2d_splitk_workspace_bytes", rows, Cout, K * K * Cin, rt.dt)
    ws = torch.empty(max(wsb // 4, 1), device=rt.device)
    rt.call("msc_conv2d_fwd", xd, wd, y, part, B, H, H, Cin, Cout, K, K, s, p, ws if wsb else None, wsb, rt.dt)
    fwd = last_gemm()
    yr = ref.permute(0, 2, 3, 1)
    check("y", y, yr, 1.2e-2)
    check("sum", part[:, 0].sum(0), yr.sum((0, 1, 2)), 2e-3 * math.sqrt(rows))
    check("sumsq", part[:, 1].sum(0), yr.pow(2).sum((0, 1, 2)), 2e-3 * math.sqrt(rows))
    layer1 = (Cin, Cout, K, s, p, H) == (64, 64, 3, 1, 1, 56)          # the halo-resident kernel of conv_c64.hip: 224-pixel tiles, 4 waves
    if layer1:
        assert fwd[0] == 5 and fwd[1:4] == (224, 64, 256), fwd
    elif rows * Cout >= 200 * 128 * 128:
        assert fwd[0] == 1 and fwd[1:4] == (128, 128 if Cout >= 128 else 64, 512 if Cout >= 128 else 256), fwd
    # data gradient with a residual input
    wt = torch.empty(w.numel(), dtype=rt.tdtype, device=rt.device)
    rt.call("msc_permute3_cast", w.to(rt.device).float().contiguous(), wt, Cout, K * K, Cin, rt.dt)
    rs = rnd(B, H, H, Cin, seed=5, scale=0.1)
    dx = torch.full((B, H, H, Cin), float("nan"), device=rt.device, dtype=rt.tdtype)
    wsb2 = L.query("msc_conv2d_splitk_workspace_bytes", B * H * H, Cin, K * K * Cout, rt.dt) if s == 1 else 0
    ws2 = torch.empty(max(wsb2 // 4, 1), device=rt.device)
    rt.call("msc_conv2d_dgrad", dev(dy, rt), wt, dev(rs, rt), None, dx, B, H, H, Cin, Cout, K, K, s, p, ws2 if wsb2 else None, wsb2, rt.dt)
    dg = last_gemm()
    check("dx", dx, xq.grad + qb(rs), 2.4e-2)
    if layer1:
        assert dg[0] == 5, dg
    if s == 2 and K > 1:
        assert dg[6] == 1, ("strided dgrad without the parity-class interleave", dg)
    if s == 2 and K == 1:
        assert dg[6] == 0, ("1x1 strided dgrad: only the tap-carrying class is launched (class-major order), its workgroups fill the siblings", dg)
    # weight gradient
    dw = torch.full((Cout, K, K, Cin), float("nan"), device=rt.device)
    wsw = L.query("msc_wgrad_workspace_bytes", Cout, K * K * Cin, rows)
    wsb3 = torch.empty(wsw // 4, device=rt.device)
    rt.call("msc_conv2d_wgrad", xd, dev(dy, rt), dw, wsb3, wsw, B, H, H, Cin, Cout, K, K, s, p, 0, rt.dt)
    wg = last_gemm()
    if layer1:
        assert wg[0] == 7 and wg[1:4] == (224, 576, 256), wg                  # conv_c64w.hip: both operands staged once per 224-pixel tile
    else:
        assert wg[0] == 1 and wg[3] == 256 and wg[4] >= 1, wg                 # gathered weight gradient: 4 waves, 64-byte K steps
    check("dw", dw, wq.grad.permute(0, 1, 2, 3), 2e-3 * 4)


def test_stem_at_benchmark_shape(rt):
    """ImageEncoder stem in space-to-depth form at 64 frames: forward + weight gradient (the last kernel of the backward)"""
    from multimodal_sc_amd import lib as L
    H = 224
    img = torch.rand(B, 3, H, H, generator=torch.Generator().manual_seed(1))
    w7 = rnd(64, 7, 7, 8, seed=2, scale=1 / math.sqrt(147))
    w7[..., 3:] = 0
    xs = torch.empty(B, H // 2, H // 2, 16, device=rt.device, dtype=rt.tdtype)
    rt.call("msc_image_to_s2d", img.to(rt.device), xs, B, H, H, rt.dt)
    ws2d = torch.empty(64, 4, 4, 16, device=rt.device, dtype=rt.tdtype)
    rt.call("msc_stem_weight_s2d", w7.to(rt.device).contiguous(), ws2d, 64, rt.dt)
    Hs = H // 2
    y = torch.full((B, Hs, Hs, 64), float("nan"), device=rt.device, dtype=rt.tdtype)
    nrows = L.query("msc_conv2d_fwd_stat_rows", B * Hs * Hs, 64, rt.dt)
    part = torch.empty(nrows, 2, 64, device=rt.device)
    rt.call("msc_stem_s2d_conv_fwd", xs, ws2d, y, part, B, Hs, Hs, 64, rt.dt)
    xq = qb(img).requires_grad_()
    wq = qb(w7[..., :3]).requires_grad_()
    ref = TF.conv2d(xq, wq.permute(0, 3, 1, 2), None, 2, 3)
    check("y", y, ref.permute(0, 2, 3, 1), 1.2e-2)
    dy = rnd(B, Hs, Hs, 64, seed=3, scale=0.1)
    ref.backward(qb(dy).permute(0, 3, 1, 2))
    dws = torch.full((64, 4, 4, 16), float("nan"), device=rt.device)
    wsw = L.query("msc_wgrad_workspace_bytes", 64, 256, B * Hs * Hs)
    ws = torch.empty(wsw // 4, device=rt.device)
    rt.call("msc_stem_s2d_conv_wgrad", xs, dev(dy, rt), dws, ws, wsw, B, Hs, Hs, 64, 0, rt.dt)
    dw7 = torch.zeros(64, 7, 7, 8, device=rt.device)
    rt.call("msc_stem_wgrad_unpack", dws, dw7, 64, 0)
    check("dw", dw7[..., :3], wq.grad, 2e-3 * 4)


# ---- the same small ragged shapes as test_kernels_gpu.py::test_linear_fwd_bwd / CONV_CASES, with the 128 x 128 tile FORCED: the
# ---- production instantiation on edge cases (rows / columns / K not multiples of the tile) the planner would give to 64-wide tiles
@pytest.mark.parametrize("Mr,N,K,act", [(128, 128, 64, 0), (300, 200, 96, 1), (1000, 512, 512, 2), (37, 3072, 64, 4), (130, 136, 8, 0)])
def test_linear_forced_128x128(rt, knob, Mr, N, K, act):
    knob("gemm_tile", "128x128")
    x, w, b = rnd(Mr, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3, scale=0.1)
    dy = rnd(Mr, N, seed=4)
    pre = qb(x) @ qb(w).t() + b
    ref = [pre, torch.relu(pre), TF.gelu(pre), torch.tanh(pre), torch.sigmoid(pre)][act]
    y = torch.full((Mr, N), float("nan"), device=rt.device, dtype=rt.tdtype)
    p2 = torch.full((Mr, N), float("nan"), device=rt.device, dtype=rt.tdtype)
    rt.call("msc_linear_fwd", dev(x, rt), dev(w, rt), b.to(rt.device), None, y, p2, Mr, N, K, act, rt.dt)
    assert last_gemm()[:4] == (1, 128, 128, 512), last_gemm()
    check("y", y, ref, 1.2e-2)
    check("pre", p2, pre, 1.2e-2)
    wt = qb(w).t().contiguous()
    dx = torch.full((Mr, K), float("nan"), device=rt.device, dtype=rt.tdtype)
    rt.call("msc_linear_dgrad", dev(dy, rt), dev(wt, rt), None, None, 0, None, dx, Mr, N, K, rt.dt)
    assert last_gemm()[:4] == (1, 128, 128, 512), last_gemm()
    check("dx", dx, qb(dy) @ qb(w), 2.4e-2)
    from multimodal_sc_amd import lib as L
    dw = torch.full((N, K), float("nan"), device=rt.device)
    db = torch.full((N,), float("nan"), device=rt.device)
    wsb = L.query("msc_wgrad_workspace_bytes", N, K, Mr)
    ws = torch.empty(max(wsb // 4, 1), device=rt.device)
    rt.call("msc_linear_wgrad", dev(x, rt), dev(dy, rt), dw, db, ws, wsb, Mr, N, K, 0, rt.dt)
    assert last_gemm()[:4] == (1, 128, 128, 512), last_gemm()
    check("dw", dw, qb(dy).t() @ qb(x), 2e-3 * 4)
    check("db", db, qb(dy).sum(0), 2e-3 * 4)


@pytest.mark.parametrize("case", [(2, 9, 9, 16, 64, 3, 1, 1), (4, 32, 32, 64, 64, 3, 1, 1), (2, 14, 14, 64, 128, 3, 2, 1), (2, 14, 14, 64, 128, 1, 2, 0),
                                  (3, 7, 7, 256, 512, 3, 1, 1)])
def test_conv_forced_128x128(rt, knob, case):
    knob("gemm_tile", "128x128")
    Bc, H, W, Cin, Cout, K, s, p = case
    Ho, Wo = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
    x = rnd(Bc, H, W, Cin, seed=1)
    w = rnd(Cout, K, K, Cin, seed=2, scale=1 / math.sqrt(K * K * Cin))
    dy = rnd(Bc, Ho, Wo, Cout, seed=3)
    xq, wq = qb(x).requires_grad_(), qb(w).requires_grad_()
    ref = TF.conv2d(xq.permute(0, 3, 1, 2), wq.permute(0, 3, 1, 2), None, s, p)
    ref.backward(qb(dy).permute(0, 3, 1, 2))
    y = torch.full((Bc, Ho, Wo, Cout), float("nan"), device=rt.device, dtype=rt.tdtype)
    rt.call("msc_conv2d_fwd", dev(x, rt), dev(w, rt), y, None, Bc, H, W, Cin, Cout, K, K, s, p, None, 0, rt.dt)
    assert last_gemm()[:4] == (1, 128, 128, 512), last_gemm()
    check("y", y, ref.permute(0, 2, 3, 1), 1.2e-2)
    wt = torch.empty(w.numel(), dtype=rt.tdtype, device=rt.device)
    rt.call("msc_permute3_cast", w.to(rt.device).float().contiguous(), wt, Cout, K * K, Cin, rt.dt)
    dx = torch.full((Bc, H, W, Cin), float("nan"), device=rt.device, dtype=rt.tdtype)
    rt.call("msc_conv2d_dgrad", dev(dy, rt), wt, None, None, dx, Bc, H, W, Cin, Cout, K, K, s, p, None, 0, rt.dt)
    info = last_gemm()
    assert info[0] == 1 and info[2] == 128, info                     # (the row tile is capped at 64 where a parity class asks for it)
    check("dx", dx, xq.grad, 2.4e-2)


# ------------------------------------------------------------------------------------------------ one row per frame: the skinny kernel
# (name, out features, in features, activation): the codec / head / state-encoder layers of C3 (model.py param_specs)
SKINNY_FWD = [("enc.fc1 relu", 256, 512, "relu"), ("head.fc tanh", 256, 512, "tanh"), ("head.out", 16, 256, "none"),
              ("recon.fc1 relu", 3136, 512, "relu"), ("state.fc1 relu", 512, 32, "relu"), ("dec.fc2", 512, 256, "none"), ("sigmoid", 48, 64, "sigmoid")]
_ACT = {"none": 0, "relu": 1, "tanh": 3, "sigmoid": 4}
_ACT_REF = {"none": lambda t: t, "relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid}


@pytest.mark.parametrize("rows", [64, 32, 33, 7, 1])
@pytest.mark.parametrize("name,N,K,act", SKINNY_FWD)
def test_skinny_linear_forward(rt, knob, name, N, K, act, rows):
    """forward of the <= 64-row dense layers on gemm_skinny.hip (no staging; the K split over the waves summed in wave order) against fp32
    PyTorch from the bf16-rounded inputs, against the tiled family on the same inputs, and bit-identical when repeated"""
    x, w, b = rnd(rows, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3, scale=0.1)
    r = rnd(rows, N, seed=4) if "dec" in name else None
    ref = _ACT_REF[act](qb(x) @ qb(w).t() + b) + (qb(r) if r is not None else 0)
    xd, wd, bd, rd = dev(x, rt), dev(w, rt), b.to(rt.device), (dev(r, rt) if r is not None else None)
    outs = {}
    for sk in (1, 0):
        knob("gemm_skinny", sk)
        y = torch.full((rows, N), float("nan"), device=rt.device, dtype=rt.tdtype)
        rt.call("msc_linear_fwd", xd, wd, bd, rd, y, None, rows, N, K, _ACT[act], rt.dt)
        info = last_gemm()
        assert (info[0] == 4) == bool(sk), info
        check(f"y skinny={sk}", y, ref, 1.2e-2)
        outs[sk] = y
    check("skinny vs tiled", outs[1], outs[0].float(), 1.2e-2)
    knob("gemm_skinny", 1)
    y2 = torch.empty_like(outs[1])
    rt.call("msc_linear_fwd", xd, wd, bd, rd, y2, None, rows, N, K, _ACT[act], rt.dt)
    torch.cuda.synchronize()
    assert torch.equal(y2, outs[1])


@pytest.mark.parametrize("rows", [64, 33, 2])
@pytest.mark.parametrize("name,N,K,dact,res", [("recon.fc1 dgrad + head branch", 3136, 512, "none", True), ("head.out dgrad x tanh'", 16, 256, "tanh", False),
                                               ("dec.fc2 dgrad x relu'", 512, 256, "relu", False), ("enc.fc1 dgrad", 256, 512, "none", False)])
def test_skinny_linear_dgrad(rt, knob, name, N, K, dact, res, rows):
    """data gradients of the same layers: dx[M,K] = dy[M,N] wt[K,N]^T (+ residual) x act'(ref) — the reduction runs over N (3136 for the
    reconstruction head: eight waves share it)"""
    dy, w = rnd(rows, N, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(N))
    wt = qb(w).t().contiguous()                     # [K][N]
    r = rnd(rows, K, seed=3) if res else None
    yref = torch.tanh(rnd(rows, K, seed=5)) if dact == "tanh" else torch.relu(rnd(rows, K, seed=5)) if dact == "relu" else None
    ref = qb(dy) @ qb(w) + (qb(r) if res else 0)
    if dact == "tanh":
        ref = ref * (1 - qb(yref) ** 2)
    elif dact == "relu":
        ref = ref * (qb(yref) > 0)
    outs = {}
    for sk in (1, 0):
        knob("gemm_skinny", sk)
        dx = torch.full((rows, K), float("nan"), device=rt.device, dtype=rt.tdtype)
        rt.call("msc_linear_dgrad", dev(dy, rt), dev(wt, rt), dev(r, rt) if res else None, dev(yref, rt) if yref is not None else None,
                _ACT[dact], None, dx, rows, N, K, rt.dt)
        info = last_gemm()
        assert (info[0] == 4) == bool(sk), info
        if sk:
            assert info[3] == (512 if N >= 2048 else 256), info
        check(f"dx skinny={sk}", dx, ref, 1.2e-2)
        outs[sk] = dx
    check("skinny vs tiled", outs[1], outs[0].float(), 1.2e-2)


# ------------------------------------------------------------------------------------------------ ResNet layer 1: the halo-resident 64 -> 64 kernel
@pytest.mark.parametrize("Bc,Hc", [(2, 56), (1, 4), (3, 12), (5, 8)])
def test_conv_c64_forward_and_statistics(rt, knob, Bc, Hc):
    """3x3 / stride 1 / pad 1, 64 -> 64 channels on 56-wide maps through conv_c64.hip (halo in LDS, weights in registers): output and the
    BatchNorm partial sums against fp32 PyTorch from bf16-rounded inputs, against the implicit-GEMM kernel, bit-identical when repeated.
    Heights 4 / 8 / 12: every tile touches the top or bottom zero padding, and with 5 images the tile count is odd."""
    from multimodal_sc_amd import lib as L
    Wc, C = 56, 64
    x = rnd(Bc, Hc, Wc, C, seed=1)
    w = rnd(C, 3, 3, C, seed=2, scale=1 / math.sqrt(9 * C))
    ref = TF.conv2d(qb(x).permute(0, 3, 1, 2), qb(w).permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
    Mrows = Bc * Hc * Wc
    nblk = L.query("msc_conv2d_fwd_stat_rows", Mrows, C, rt.dt)
    xd, wd = dev(x, rt), dev(w, rt)
    outs = {}
    for on in (1, 0):
        knob("conv_c64", on)
        y = torch.full((Bc, Hc, Wc, C), float("nan"), device=rt.device, dtype=rt.tdtype)
        part = torch.full((nblk, 2, C), float("nan"), device=rt.device)
        rt.call("msc_conv2d_fwd", xd, wd, y, part, Bc, Hc, Wc, C, C, 3, 3, 1, 1, None, 0, rt.dt)
        info = last_gemm()
        assert (info[0] == 5) == bool(on), info
        check(f"y c64={on}", y, ref, 1.2e-2)
        torch.cuda.synchronize()
        assert torch.isfinite(part).all()
        s = part.double().sum(0).cpu()
        flat = ref.reshape(-1, C).double()
        assert (s[0] - flat.sum(0)).abs().max() <= 2e-3 * max(1.0, flat.abs().sum(0).max().item())
        assert (s[1] - (flat * flat).sum(0)).abs().max() <= 2e-3 * max(1.0, (flat * flat).sum(0).max().item())
        outs[on] = (y, part.clone())
    check("c64 vs implicit GEMM", outs[1][0], outs[0][0].float(), 1.2e-2)
    knob("conv_c64", 1)
    y2, part2 = torch.empty_like(outs[1][0]), torch.empty_like(outs[1][1])
    rt.call("msc_conv2d_fwd", xd, wd, y2, part2, Bc, Hc, Wc, C, C, 3, 3, 1, 1, None, 0, rt.dt)
    torch.cuda.synchronize()
    assert torch.equal(y2, outs[1][0]) and torch.equal(part2, outs[1][1])


@pytest.mark.parametrize("mode", ["plain", "res", "res+relu"])
@pytest.mark.parametrize("Bc,Hc", [(2, 56), (3, 8)])
def test_conv_c64_dgrad(rt, knob, Bc, Hc, mode):
    """data gradient of the same convolution (transposed weight layout, taps negated), with the residual add and the ReLU mask of the
    epilogue, against the autograd gradient of fp32 PyTorch"""
    Wc, C = 56, 64
    dy = rnd(Bc, Hc, Wc, C, seed=1)
    w = rnd(C, 3, 3, C, seed=2, scale=1 / math.sqrt(9 * C))                 # [Cout][KH][KW][Cin]
    xin = torch.zeros(Bc, C, Hc, Wc, requires_grad=True)
    TF.conv2d(xin, qb(w).permute(0, 3, 1, 2), None, 1, 1).backward(qb(dy).permute(0, 3, 1, 2))
    ref = xin.grad.permute(0, 2, 3, 1)
    r = rnd(Bc, Hc, Wc, C, seed=3) if mode != "plain" else None
    mref = rnd(Bc, Hc, Wc, C, seed=4) if mode == "res+relu" else None
    if r is not None:
        ref = ref + qb(r)
    if mref is not None:
        ref = ref * (qb(mref) > 0)
    wt = qb(w).permute(3, 1, 2, 0).contiguous()                              # [Cin][KH][KW][Cout]
    outs = {}
    for on in (1, 0):
        knob("conv_c64", on)
        dx = torch.full((Bc, Hc, Wc, C), float("nan"), device=rt.device, dtype=rt.tdtype)
        rt.call("msc_conv2d_dgrad", dev(dy, rt), dev(wt, rt), dev(r, rt) if r is not None else None, dev(mref, rt) if mref is not None else None, dx,
                Bc, Hc, Wc, C, C, 3, 3, 1, 1, None, 0, rt.dt)
        info = last_gemm()
        assert (info[0] == 5) == bool(on), info
        check(f"dx c64={on}", dx, ref, 1.2e-2)
        outs[on] = dx
    check("c64 vs implicit GEMM", outs[1], outs[0].float(), 1.2e-2)


def test_conv_c64_at_benchmark_shape(rt, knob):
    """layer 1 at 64 frames (200704 pixels, 896 tiles over the persistent workgroups): against the implicit-GEMM kernel and an fp32 reference
    on a slice of the batch"""
    Bc, Hc, Wc, C = 64, 56, 56, 64
    x = rnd(Bc, Hc, Wc, C, seed=1)
    w = rnd(C, 3, 3, C, seed=2, scale=1 / math.sqrt(9 * C))
    xd, wd = dev(x, rt), dev(w, rt)
    ys = {}
    for on in (1, 0):
        knob("conv_c64", on)
        y = torch.full((Bc, Hc, Wc, C), float("nan"), device=rt.device, dtype=rt.tdtype)
        rt.call("msc_conv2d_fwd", xd, wd, y, None, Bc, Hc, Wc, C, C, 3, 3, 1, 1, None, 0, rt.dt)
        ys[on] = y
    check("c64 vs implicit GEMM", ys[1], ys[0].float(), 1.2e-2)
    for b in (0, 31, 63):
        ref = TF.conv2d(qb(x[b:b + 1]).permute(0, 3, 1, 2), qb(w).permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
        check(f"image {b}", ys[1][b:b + 1], ref, 1.2e-2)


# ------------------------------------------------------------------------------------------------ stem weight gradient: the halo-resident kernel
@pytest.mark.parametrize("Bs,Hs", [(2, 112), (1, 2), (3, 6), (64, 112)])
def test_stem_wgrad_halo(rt, knob, Bs, Hs):
    """weight gradient of the space-to-depth stem (4x4 / stride 1, pad 2 before / 1 after, 16 -> 64 channels on 112-wide maps) through conv_stem.hip
    (both operands staged once per tile, transposed LDS reads, accumulators resident across the workgroup's tiles, fixed-order reduce of the
    workgroups' partials) against the autograd weight gradient of fp32 PyTorch from bf16-rounded inputs, against the gathered split-K GEMM,
    and bit-identical when repeated.  Heights 2 / 6: every tile touches the zero padding above or below; 3 images x 3 tiles: odd tile count."""
    from multimodal_sc_amd import lib as L
    Ws, Cout = 112, 64
    xs = rnd(Bs, Hs, Ws, 16, seed=1)
    dy = rnd(Bs, Hs, Ws, Cout, seed=2, scale=0.1)
    w = torch.zeros(Cout, 16, 4, 4, requires_grad=True)
    xin = TF.pad(qb(xs).permute(0, 3, 1, 2), (2, 1, 2, 1))
    TF.conv2d(xin, w).backward(qb(dy).permute(0, 3, 1, 2))
    ref = w.grad.permute(0, 2, 3, 1).reshape(Cout, 256)                   # [co][kh][kw][c]
    wsb = L.query("msc_wgrad_workspace_bytes", Cout, 256, Bs * Hs * Ws)
    ws = torch.empty(wsb // 4, device=rt.device)
    xd, dyd = dev(xs, rt), dev(dy, rt)
    outs = {}
    for on in (1, 0):
        knob("stem_wgrad_halo", on)
        dw = torch.full((Cout, 256), float("nan"), device=rt.device)
        rt.call("msc_stem_s2d_conv_wgrad", xd, dyd, dw, ws, wsb, Bs, Hs, Ws, Cout, 0, rt.dt)
        info = last_gemm()
        assert (info[0] == 6) == bool(on), info
        check(f"dw halo={on}", dw, ref, 2e-3)
        outs[on] = dw
    check("halo vs gathered GEMM", outs[1], outs[0], 2e-3)
    knob("stem_wgrad_halo", 1)
    dw2 = torch.full((Cout, 256), float("nan"), device=rt.device)
    rt.call("msc_stem_s2d_conv_wgrad", xd, dyd, dw2, ws, wsb, Bs, Hs, Ws, Cout, 0, rt.dt)
    acc = outs[1].clone()
    rt.call("msc_stem_s2d_conv_wgrad", xd, dyd, acc, ws, wsb, Bs, Hs, Ws, Cout, 1, rt.dt)        # accumulate = 1: adds to what is there
    torch.cuda.synchronize()
    assert torch.equal(dw2, outs[1])
    check("accumulate", acc, 2 * ref, 2e-3)


@pytest.mark.parametrize("Bc,Hc", [(2, 56), (1, 4), (3, 12), (64, 56)])
def test_conv_c64_wgrad(rt, knob, Bc, Hc):
    """weight gradient of the layer-1 convolutions through conv_c64w.hip (dy tile and x halo staged once, transposed LDS reads, 36 accumulators per
    wave resident across the workgroup's tiles, fixed-order reduce of the partials) against the autograd weight gradient of fp32 PyTorch from
    bf16-rounded inputs, against the gathered split-K GEMM, bit-identical when repeated, accumulate mode"""
    from multimodal_sc_amd import lib as L
    Wc, C = 56, 64
    x = rnd(Bc, Hc, Wc, C, seed=1)
    dy = rnd(Bc, Hc, Wc, C, seed=2, scale=0.1)
    w = torch.zeros(C, C, 3, 3, requires_grad=True)
    TF.conv2d(qb(x).permute(0, 3, 1, 2), w, None, 1, 1).backward(qb(dy).permute(0, 3, 1, 2))
    ref = w.grad.permute(0, 2, 3, 1).contiguous()                           # [Cout][KH][KW][Cin]
    wsb = L.query("msc_wgrad_workspace_bytes", C, 9 * C, Bc * Hc * Wc)
    assert wsb >= 256 * 64 * 576 * 4
    ws = torch.empty(wsb // 4, device=rt.device)
    xd, dyd = dev(x, rt), dev(dy, rt)
    outs = {}
    for on in (1, 0):
        knob("conv_c64", on)
        dw = torch.full((C, 3, 3, C), float("nan"), device=rt.device)
        rt.call("msc_conv2d_wgrad", xd, dyd, dw, ws, wsb, Bc, Hc, Wc, C, C, 3, 3, 1, 1, 0, rt.dt)
        info = last_gemm()
        assert (info[0] == 7) == bool(on), info
        check(f"dw c64={on}", dw, ref, 2e-3)
        outs[on] = dw
    check("c64 vs gathered GEMM", outs[1], outs[0], 2e-3)
    knob("conv_c64", 1)
    dw2 = torch.full((C, 3, 3, C), float("nan"), device=rt.device)
    rt.call("msc_conv2d_wgrad", xd, dyd, dw2, ws, wsb, Bc, Hc, Wc, C, C, 3, 3, 1, 1, 0, rt.dt)
    acc = outs[1].clone()
    rt.call("msc_conv2d_wgrad", xd, dyd, acc, ws, wsb, Bc, Hc, Wc, C, C, 3, 3, 1, 1, 1, rt.dt)
    torch.cuda.synchronize()
    assert torch.equal(dw2, outs[1])
    check("accumulate", acc, 2 * ref, 2e-3)


@pytest.mark.parametrize("Bs,Hs", [(2, 112), (1, 2), (3, 6), (64, 112)])
def test_stem_fwd_halo(rt, knob, Bs, Hs):
    """forward of the space-to-depth stem (4x4 / stride 1, pad 2 before / 1 after, 16 -> 64 channels on 112-wide maps) through conv_stem.hip (halo staged
    once, weights in registers): output and BatchNorm partial sums against fp32 PyTorch from bf16-rounded inputs, against the implicit-GEMM
    kernel, bit-identical when repeated"""
    from multimodal_sc_amd import lib as L
    Ws, Cout = 112, 64
    xs = rnd(Bs, Hs, Ws, 16, seed=1)
    w = rnd(Cout, 4, 4, 16, seed=2, scale=1 / 16.0)
    xin = TF.pad(qb(xs).permute(0, 3, 1, 2), (2, 1, 2, 1))
    ref = TF.conv2d(xin, qb(w).permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    rows = Bs * Hs * Ws
    nblk = L.query("msc_conv2d_fwd_stat_rows", rows, Cout, rt.dt)
    xd, wd = dev(xs, rt), dev(w, rt)
    outs = {}
    for on in (1, 0):
        knob("stem_fwd_halo", on)
        y = torch.full((Bs, Hs, Ws, Cout), float("nan"), device=rt.device, dtype=rt.tdtype)
        part = torch.full((nblk, 2, Cout), float("nan"), device=rt.device)
        rt.call("msc_stem_s2d_conv_fwd", xd, wd, y, part, Bs, Hs, Ws, Cout, rt.dt)
        info = last_gemm()
        assert (info[0] == 8) == bool(on), info
        check(f"y halo={on}", y, ref, 1.2e-2)
        torch.cuda.synchronize()
        assert torch.isfinite(part).all()
        s = part.double().sum(0).cpu()
        flat = ref.reshape(-1, Cout).double()
        assert (s[0] - flat.sum(0)).abs().max() <= 2e-3 * max(1.0, flat.abs().sum(0).max().item())
        assert (s[1] - (flat * flat).sum(0)).abs().max() <= 2e-3 * max(1.0, (flat * flat).sum(0).max().item())
        outs[on] = (y, part.clone())
    check("halo vs implicit GEMM", outs[1][0], outs[0][0].float(), 1.2e-2)
    knob("stem_fwd_halo", 1)
    y2, part2 = torch.empty_like(outs[1][0]), torch.empty_like(outs[1][1])
    rt.call("msc_stem_s2d_conv_fwd", xd, wd, y2, part2, Bs, Hs, Ws, Cout, rt.dt)
    torch.cuda.synchronize()
    assert torch.equal(y2, outs[1][0]) and torch.equal(part2, outs[1][1])
