"""Per-kernel parity: every HIP op (through the C-ABI) against plain PyTorch-CPU fp32 on the same seeded inputs.

Tolerances (stated here, used below):
  f32 mode : |gpu - ref| <= 2e-4 * max(1, max|ref|)   (exact-fp32 MFMA; only the summation order differs)
  bf16 mode: the reference is computed in fp32 FROM THE bf16-ROUNDED INPUTS, so the only differences are the
             bf16 rounding of the stored output (2^-8 relative) and summation order: 1.2e-2 * max(1, max|ref|);
             fp32 outputs of bf16 kernels (weight gradients, statistics) use 2e-3 (inputs identical, fp32 accumulate;
             the slack covers bf16-rounded intermediate activations inside fused chains).
Integer outputs (arg-max, pool indices) are compared bit-exactly.
"""
import math

import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

DTYPES = ["f32", "bf16"]


def tol(dtype, fp32_out=False):
    if dtype == "f32":
        return 2e-4
    return 2e-3 if fp32_out else 1.2e-2


def check(name, got, ref, t):
    torch.cuda.synchronize()          # weight-gradient GEMMs run on the runtime's side stream
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    err = (got - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    assert err <= t * scale, f"{name}: max err {err:.3e} > {t:.1e} * {scale:.3e}"


def check_grad(name, got, ref, t, dtype, relu):
    """Gradients behind a ReLU in bf16 mode: an activation within bf16 rounding of 0 can land on the other side of the
    mask than in the fp32 reference, which moves a few isolated entries by O(|dy|) — a max-abs bound is the wrong
    yardstick for that.  Those cases use the relative Frobenius error (<= 4e-2); everything else uses max-abs."""
    if dtype == "bf16" and relu:
        torch.cuda.synchronize()
        g, r = got.detach().double().cpu(), ref.detach().double().cpu()
        assert g.shape == r.shape and torch.isfinite(g).all(), name
        rel = ((g - r).norm() / r.norm().clamp_min(1e-30)).item()
        assert rel <= 4e-2, f"{name}: relative Frobenius error {rel:.3e} > 4e-2"
    else:
        check(name, got, ref, t)


@pytest.fixture(scope="module")
def rts(cuda):
    from multimodal_sc_amd.runtime import Runtime
    return {d: Runtime(d, "cuda:0") for d in DTYPES}


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(t, dtype):
    """value as the kernel sees it (bf16-rounded in bf16 mode), as fp32 CPU"""
    return t.to(torch.bfloat16).float() if dtype == "bf16" else t.clone()


def ste(t, dtype):
    """bf16 mode: round to bf16 at the points where the kernels store bf16 activations (straight-through for the
    gradient), so that ReLU masks / BN inputs of the reference see exactly the stored values."""
    return t if dtype == "f32" else t + (t.detach().to(torch.bfloat16).float() - t.detach())


def dev(t, rt):
    return t.to(rt.device).to(rt.tdtype).contiguous()


class FakeParam:
    """ParamRef stand-in built from CPU tensors for single-op tests."""

    def __init__(self, rt, w, kind):
        from multimodal_sc_amd import lib as L
        self.shape = tuple(w.shape)
        self.kind = kind
        self.p = w.to(rt.device).float().contiguous()
        self.c = self.p if rt.dt == L.MSC_F32 else self.p.to(rt.tdtype)
        self.g = torch.full_like(self.p, float("nan"))
        self.t = None
        if kind == "conv":
            co, kh, kw, ci = w.shape
            self.t = torch.empty(w.numel(), dtype=rt.tdtype, device=rt.device)
            rt.call("msc_permute3_cast", self.p, self.t, co, kh * kw, ci, rt.dt)
        elif kind == "linear":
            n, k = w.shape
            self.t = torch.empty(w.numel(), dtype=rt.tdtype, device=rt.device)
            rt.call("msc_permute3_cast", self.p, self.t, n, 1, k, rt.dt)


# ------------------------------------------------------------------------------------------------ GEMM / linear
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,act", [(128, 128, 64, 0), (300, 200, 96, 1), (64, 16, 256, 3), (1000, 512, 512, 2), (37, 3072, 64, 4), (50, 64, 8, 0)])
def test_linear_fwd_bwd(rts, dtype, M, N, K, act):
    from multimodal_sc_amd import functional as F
    rt = rts[dtype]
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3, scale=0.1)
    dy = rnd(M, N, seed=4)
    xq, wq = q(x, dtype).requires_grad_(), q(w, dtype).requires_grad_()
    bq = b.clone().requires_grad_()
    pre = xq @ wq.t() + bq
    ref = [pre, torch.relu(pre), TF.gelu(pre), torch.tanh(pre), torch.sigmoid(pre)][act]
    W, Bp = FakeParam(rt, w, "linear"), FakeParam(rt, b, "vec")
    xd = dev(x, rt).requires_grad_()
    y = F.Linear.apply(xd, None, rt, W, Bp, act, True)
    check("y", y, ref, tol(dtype))
    # backward: feed the same upstream gradient; the reference activation-gradient uses its own (fp32) output
    ref.backward(q(dy, dtype))
    y.backward(dev(dy, rt))
    check("dx", xd.grad, xq.grad, tol(dtype) * 2)
    check("dw", W.g, wq.grad, tol(dtype, True) * 4)
    check("db", Bp.g, bq.grad, tol(dtype, True) * 4)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,act", [(300, 128, 256, 2), (1000, 512, 2048, 2), (70, 64, 64, 1)])
def test_linear_dgrad_fused_act_and_bias(rts, dtype, M, N, K, act):
    """dgrad epilogue: dx = (dy W) * act'(ref) and the per-row-block column sums that become the previous bias gradient"""
    from multimodal_sc_amd import lib as L
    rt = rts[dtype]
    dy, w, pre = rnd(M, N, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(N)), rnd(M, K, seed=3)
    pq = q(pre, dtype).requires_grad_()
    y = TF.gelu(pq) if act == 2 else torch.relu(pq)
    ref_in = pq if act == 2 else y                      # gelu: pre-activation, relu: output
    (y @ q(w, dtype).t()).backward(q(dy, dtype))
    Wp = FakeParam(rt, w, "linear")
    dx = rt.empty(M, K)
    nblk = L.query("msc_linear_dgrad_colsum_rows", M, K, rt.dt)
    part = torch.full((nblk, K), float("nan"), device=rt.device)
    rt.call("msc_linear_dgrad", dev(dy, rt), Wp.t, None, dev(ref_in.detach(), rt), act, part, dx, M, N, K, rt.dt)
    check("dx", dx, pq.grad, tol(dtype) * 2)
    db = torch.empty(K, device=rt.device)
    rt.call("msc_colsum_finalize", part, nblk, K, db, 0)
    check("db", db, pq.grad.sum(0), tol(dtype, True) * 8)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (1000, 2048, 512), (37, 72, 64)])
def test_linear_gelu_derivative_save(rts, dtype, M, N, K):
    """MSC_ACT_GELU_DSAVE: y is bit-identical to MSC_ACT_GELU's y, the side output is gelu'(pre-activation); MSC_ACT_MUL_REF in the
    dgrad epilogue multiplies by it: same gradient as the evaluate-in-the-backward path within one rounding of the stored derivative"""
    from multimodal_sc_amd import functional as F, lib as L
    rt = rts[dtype]
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3, scale=0.1)
    W, Bp = FakeParam(rt, w, "linear"), FakeParam(rt, b, "vec")
    y0, pre = F.linear_fwd(rt, dev(x, rt), W, Bp, L.ACT_GELU, want_pre=True)
    y1, dsv = F.linear_fwd(rt, dev(x, rt), W, Bp, L.ACT_GELU_DSAVE, want_pre=True)
    assert torch.equal(y0, y1)
    u = (q(x, dtype) @ q(w, dtype).t() + b).requires_grad_()
    TF.gelu(u).sum().backward()
    check("gelu'", dsv, u.grad, tol(dtype))
    # backward through a following layer [N -> K2]: dx = (dy W2) * gelu'
    K2 = 64
    dy, w2 = rnd(M, K2, seed=4), rnd(K2, N, seed=5, scale=1 / math.sqrt(N))
    W2 = FakeParam(rt, w2, "linear")
    dx0, dx1 = rt.empty(M, N), rt.empty(M, N)
    rt.call("msc_linear_dgrad", dev(dy, rt), W2.t, None, pre, L.ACT_GELU, None, dx0, M, K2, N, rt.dt)
    rt.call("msc_linear_dgrad", dev(dy, rt), W2.t, None, dsv, L.ACT_MUL_REF, None, dx1, M, K2, N, rt.dt)
    ref = (q(dy, dtype) @ q(w2, dtype)) * u.grad
    check("dx (derivative evaluated in the backward)", dx0, ref, tol(dtype) * 2)
    check("dx (stored derivative)", dx1, ref, tol(dtype) * 2)
    if dtype == "f32":
        check("two paths, fp32", dx1, dx0.float().cpu(), 1e-6)
    # the element-wise activation backward accepts the stored derivative too
    d = rt.empty(M, N)
    rt.call("msc_act_bwd", dev(dy @ w2, rt), dsv, d, M * N, L.ACT_MUL_REF, rt.dt)
    check("act_bwd", d, q(dy @ w2, dtype) * q(dsv.float().cpu(), dtype), tol(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_residual(rts, dtype):
    from multimodal_sc_amd import functional as F
    rt = rts[dtype]
    M, N, K = 200, 128, 64
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3), rnd(M, N, seed=5)
    W, Bp = FakeParam(rt, w, "linear"), FakeParam(rt, b, "vec")
    y, _ = F.linear_fwd(rt, dev(x, rt), W, Bp, 0, res=dev(r, rt))
    check("y", y, q(x, dtype) @ q(w, dtype).t() + b + q(r, dtype), tol(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_batched_layouts(rts, dtype):
    """all four operand-layout combinations of the general kernel, with strides and a ragged K (306)"""
    rt = rts[dtype]
    Bz, M, N, K = 6, 70, 64, 306
    Kp = 320
    A = rnd(Bz, M, Kp, seed=1)
    A[..., K:] = 0
    Bm = rnd(Bz, N, Kp, seed=2)
    Bm[..., K:] = 0
    ref = torch.einsum("zmk,znk->zmn", q(A, dtype), q(Bm, dtype)) * 0.5
    for a_k in (1, 0):
        for b_k in (1, 0):
            Ad = dev(A if a_k else A[..., :K].transpose(1, 2).contiguous(), rt)     # [z][M][Kp] or [z][K][M]
            Bd = dev(Bm if b_k else Bm[..., :K].transpose(1, 2).contiguous(), rt)
            if not a_k:  # pad M to a multiple of 8 for the mn-major leading dimension
                Ad = dev(torch.nn.functional.pad(A[..., :K].transpose(1, 2), (0, 2)), rt)  # [z][K][72]
            lda = Kp if a_k else Ad.shape[-1]
            ldb = Kp if b_k else N
            C = torch.full((Bz, M, N), float("nan"), dtype=torch.float32, device=rt.device)
            rt.call("msc_gemm_batched", Ad, Bd, C, M, N, K, lda, ldb, N, a_k, b_k, Bz, 2, 2 * Ad.shape[1] * Ad.shape[2],
                    Ad.shape[1] * Ad.shape[2], 2 * Bd.shape[1] * Bd.shape[2], Bd.shape[1] * Bd.shape[2], 2 * M * N, M * N, 0.5, 1, rt.dt)
            check(f"C a_k={a_k} b_k={b_k}", C, ref, tol(dtype, True))


# ------------------------------------------------------------------------------------------------ persistent 256-row tiles
PERSIST_TILES = ["256x128", "256x256"]


@pytest.mark.parametrize("tile", PERSIST_TILES)
@pytest.mark.parametrize("M,N,K,act,res", [(70000, 384, 192, 0, True),    # > 256 tiles: every workgroup walks several; ragged last row tile
                                           (2100, 512, 64, 2, False),     # K = one step: the issue side rolls over tiles faster than the ring depth
                                           (1024, 256, 2048, 3, True),    # long K, few tiles (fewer tiles than CUs)
                                           (33000, 200, 328, 1, False)])  # N not a tile multiple, K not a step multiple
def test_persistent_linear_fwd(rts, knob, tile, M, N, K, act, res):
    """the persistent kernels (one workgroup per CU walking the tiles, next tile's DMA in flight under the epilogue) forced
    on shapes that stress the tile roll-over, against fp32 from the bf16-rounded inputs"""
    rt = rts["bf16"]
    knob("gemm_tile", tile)
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3, scale=0.1)
    r = rnd(M, N, seed=4) if res else None
    pre = q(x, "bf16") @ q(w, "bf16").t() + b
    ref = [pre, torch.relu(pre), TF.gelu(pre), torch.tanh(pre), torch.sigmoid(pre)][act]
    if res:
        ref = ref + q(r, "bf16")
    y = torch.full((M, N), float("nan"), device=rt.device, dtype=rt.tdtype)
    p2 = torch.full((M, N), float("nan"), device=rt.device, dtype=rt.tdtype)
    rt.call("msc_linear_fwd", dev(x, rt), dev(w, rt), b.to(rt.device), dev(r, rt) if res else None, y, p2, M, N, K, act, rt.dt)
    check("y", y, ref, tol("bf16"))
    check("pre", p2, pre, tol("bf16"))
    # the same launch twice is bit-identical (fixed tile -> workgroup assignment, no atomics)
    y2 = torch.empty_like(y)
    rt.call("msc_linear_fwd", dev(x, rt), dev(w, rt), b.to(rt.device), dev(r, rt) if res else None, y2, None, M, N, K, act, rt.dt)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)


@pytest.mark.parametrize("tile", PERSIST_TILES)
def test_persistent_linear_dgrad_fused(rts, knob, tile):
    """extended epilogue in the persistent kernels: act' of the saved pre-activation.  (Launches that also emit per-row-block
    column sums or BatchNorm partials never take the persistent kernels — their side buffers are sized by queries that know
    only (rows, cols) — so the same call WITH the column sums must fall back to the 128-row tiles and still be right.)"""
    from multimodal_sc_amd import lib as L
    rt = rts["bf16"]
    knob("gemm_tile", tile)
    M, N, K = 40000, 192, 512
    dy, w, pre = rnd(M, N, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(N)), rnd(M, K, seed=3)
    pq = q(pre, "bf16").requires_grad_()
    (TF.gelu(pq) @ q(w, "bf16").t()).backward(q(dy, "bf16"))
    Wp = FakeParam(rt, w, "linear")
    dx = torch.full((M, K), float("nan"), device=rt.device, dtype=rt.tdtype)
    rt.call("msc_linear_dgrad", dev(dy, rt), Wp.t, None, dev(pre, rt), 2, None, dx, M, N, K, rt.dt)
    check("dx", dx, pq.grad, tol("bf16") * 2)
    nblk = L.query("msc_linear_dgrad_colsum_rows", M, K, rt.dt)
    assert nblk == (M + 127) // 128
    part = torch.full((nblk, K), float("nan"), device=rt.device)
    dx2 = torch.empty_like(dx)
    rt.call("msc_linear_dgrad", dev(dy, rt), Wp.t, None, dev(pre, rt), 2, part, dx2, M, N, K, rt.dt)
    check("dx2", dx2, pq.grad, tol("bf16") * 2)
    db = torch.empty(K, device=rt.device)
    rt.call("msc_colsum_finalize", part, nblk, K, db, 0)
    check("db", db, pq.grad.sum(0), tol("bf16", True) * 16)


@pytest.mark.parametrize("tile", PERSIST_TILES)
@pytest.mark.parametrize("case", [(6, 48, 48, 64, 128, 3, 1, 1), (5, 40, 40, 64, 256, 3, 2, 1), (3, 64, 64, 128, 256, 1, 2, 0)])
def test_persistent_conv_fwd_dgrad(rts, knob, tile, case):
    """implicit-GEMM convolution through the persistent kernels: forward with the fused BatchNorm partial sums, data
    gradient with the residual add — stride 1, stride 2 (parity-class-major rows, per-tile tap schedules) and 1x1/2."""
    from multimodal_sc_amd import lib as L
    rt = rts["bf16"]
    knob("gemm_tile", tile)
    B, H, W, Cin, Cout, K, s, p = case
    Ho, Wo = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
    x = rnd(B, H, W, Cin, seed=1)
    w = rnd(Cout, K, K, Cin, seed=2, scale=1 / math.sqrt(K * K * Cin))
    xq, wq = q(x, "bf16").requires_grad_(), q(w, "bf16")
    ref = TF.conv2d(xq.permute(0, 3, 1, 2), wq.permute(0, 3, 1, 2), None, s, p)
    M = B * Ho * Wo
    y = torch.full((B, Ho, Wo, Cout), float("nan"), device=rt.device, dtype=rt.tdtype)
    Wp = FakeParam(rt, w, "conv")
    rt.call("msc_conv2d_fwd", dev(x, rt), Wp.c, y, None, B, H, W, Cin, Cout, K, K, s, p, None, 0, rt.dt)     # persistent (no BN partials)
    check("y", y, ref.permute(0, 2, 3, 1), tol("bf16"))
    nrows = L.query("msc_conv2d_fwd_stat_rows", M, Cout, rt.dt)
    part = torch.full((nrows, 2, Cout), float("nan"), device=rt.device)
    y.fill_(float("nan"))
    rt.call("msc_conv2d_fwd", dev(x, rt), Wp.c, y, part, B, H, W, Cin, Cout, K, K, s, p, None, 0, rt.dt)      # with partials: 128-row tiles
    check("y2", y, ref.permute(0, 2, 3, 1), tol("bf16"))
    check("sum", part[:, 0].sum(0), ref.sum((0, 2, 3)), 2e-3 * math.sqrt(M))
    check("sumsq", part[:, 1].sum(0), ref.pow(2).sum((0, 2, 3)), 2e-3 * math.sqrt(M))
    dy, rs = rnd(B, Ho, Wo, Cout, seed=5), rnd(B, H, W, Cin, seed=6)
    ref.backward(q(dy, "bf16").permute(0, 3, 1, 2))
    dx = torch.full((B, H, W, Cin), float("nan"), device=rt.device, dtype=rt.tdtype)
    # dgrad output has Cin columns: the 256-column tile needs Cin >= 256, else the knob falls back inside choose_tile's rules
    rt.call("msc_conv2d_dgrad", dev(dy, rt), Wp.t, dev(rs, rt), None, dx, B, H, W, Cin, Cout, K, K, s, p, None, 0, rt.dt)
    check("dx", dx, xq.grad + q(rs, "bf16"), tol("bf16") * 2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H", [(2, 32), (3, 20), (1, 224)])
def test_stem_space_to_depth(rts, dtype, B, H):
    """The stem convolution (7x7 / stride 2 / pad 3 over RGB) in space-to-depth form against torch.nn.functional.conv2d and its
    autograd weight gradient: re-blocking kernels exact (values are only moved), forward and weight gradient within the
    per-kernel tolerances, BatchNorm partial sums consistent with the output."""
    from multimodal_sc_amd import lib as L
    rt = rts[dtype]
    Cout = 64
    img = torch.rand(B, 3, H, H, generator=torch.Generator().manual_seed(1))
    w = rnd(Cout, 7, 7, 3, seed=2, scale=0.1)
    w8 = torch.zeros(Cout, 7, 7, 8)
    w8[..., :3] = w
    Hs = H // 2
    xs = torch.full((B, Hs, Hs, 16), float("nan"), device=rt.device, dtype=rt.tdtype)
    rt.call("msc_image_to_s2d", img.to(rt.device), xs, B, H, H, rt.dt)
    want = img.view(B, 3, Hs, 2, Hs, 2).permute(0, 2, 4, 3, 5, 1).reshape(B, Hs, Hs, 12)       # [b][i][j][(dy,dx,c)]
    torch.cuda.synchronize()
    assert torch.equal(xs[..., :12].float().cpu(), q(want, dtype)) and (xs[..., 12:] == 0).all()
    ws2d = torch.full((Cout, 4, 4, 16), float("nan"), device=rt.device, dtype=rt.tdtype)
    rt.call("msc_stem_weight_s2d", w8.to(rt.device), ws2d, Cout, rt.dt)
    torch.cuda.synchronize()
    wr = ws2d.float().cpu().view(Cout, 4, 4, 16)
    for kh in range(7):
        for kw in range(7):
            i, dy, j, dx = (kh + 1) >> 1, (kh + 1) & 1, (kw + 1) >> 1, (kw + 1) & 1
            assert torch.equal(wr[:, i, j, (dy * 2 + dx) * 3:(dy * 2 + dx) * 3 + 3], q(w[:, kh, kw], dtype))
    assert wr[..., 12:].abs().sum() == 0 and wr[:, 0, :, :6].abs().sum() == 0 and wr[:, :, 0, [0, 1, 2, 6, 7, 8]].abs().sum() == 0
    # forward
    iq, wq = q(img, dtype), q(w, dtype).requires_grad_()
    ref = TF.conv2d(iq, wq.permute(0, 3, 1, 2), None, 2, 3)
    M = B * Hs * Hs
    nrows = L.query("msc_conv2d_fwd_stat_rows", M, Cout, rt.dt)
    part = torch.full((nrows, 2, Cout), float("nan"), device=rt.device)
    y = torch.full((B, Hs, Hs, Cout), float("nan"), device=rt.device, dtype=rt.tdtype)
    rt.call("msc_stem_s2d_conv_fwd", xs, ws2d, y, part, B, Hs, Hs, Cout, rt.dt)
    check("y", y, ref.permute(0, 2, 3, 1), tol(dtype))
    check("sum", part[:, 0].sum(0), ref.sum((0, 2, 3)), 2e-3 * math.sqrt(M))
    check("sumsq", part[:, 1].sum(0), ref.pow(2).sum((0, 2, 3)), 2e-3 * math.sqrt(M))
    # weight gradient, back in the OHWI layout (padded channels exactly zero)
    dy = rnd(B, Hs, Hs, Cout, seed=5)
    ref.backward(q(dy, dtype).permute(0, 3, 1, 2))
    wsb = L.query("msc_wgrad_workspace_bytes", Cout, 256, M)
    wsw = torch.empty(max(wsb // 4, 1), device=rt.device)
    dws = torch.full((Cout * 256,), float("nan"), device=rt.device)
    rt.call("msc_stem_s2d_conv_wgrad", xs, dev(dy, rt), dws, wsw, wsb, B, Hs, Hs, Cout, 0, rt.dt)
    g = torch.full((Cout, 7, 7, 8), 5.0, device=rt.device)
    rt.call("msc_stem_wgrad_unpack", dws, g, Cout, 0)
    check("dw", g[..., :3], wq.grad, tol(dtype, True) * 8)
    assert (g[..., 3:] == 0).all()
    rt.call("msc_stem_wgrad_unpack", dws, g, Cout, 1)                 # accumulate: doubles
    check("dw x2", g[..., :3], 2 * wq.grad, tol(dtype, True) * 16)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 14, 14, 64, 64, 3, 1, 1), (3, 7, 7, 256, 512, 3, 1, 1), (2, 14, 14, 64, 128, 3, 2, 1), (2, 14, 14, 64, 128, 1, 2, 0)])
def test_conv_dgrad_relu_ref(rts, dtype, case):
    """msc_conv2d_dgrad with relu_ref: dx = (conv-transpose(dy) + res) * [relu_ref > 0], bit-identical to masking the plain result
    (the mask is a 0/1 multiply of the value that is rounded either way) — un-split epilogue, split-K finish (7x7x256) and the
    strided / class-major row order."""
    from multimodal_sc_amd import lib as L
    rt = rts[dtype]
    B, H, W, Cin, Cout, K, s, p = case
    Ho, Wo = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
    w = rnd(Cout, K, K, Cin, seed=1, scale=0.1)
    Wp = FakeParam(rt, w, "conv")
    dy, rs, ref = dev(rnd(B, Ho, Wo, Cout, seed=2), rt), dev(rnd(B, H, W, Cin, seed=3), rt), dev(rnd(B, H, W, Cin, seed=4), rt)
    ref[0, 0, 0, :4] = 0                                     # exact zeros: relu'(0) = 0
    wsb = L.query("msc_conv2d_splitk_workspace_bytes", B * H * W, Cin, K * K * Cout, rt.dt) if s == 1 else 0
    ws = torch.empty(max(wsb // 4, 1), device=rt.device) if wsb else None
    plain = torch.empty(B, H, W, Cin, device=rt.device, dtype=rt.tdtype)
    masked = torch.full_like(plain, float("nan"))
    rt.call("msc_conv2d_dgrad", dy, Wp.t, rs, None, plain, B, H, W, Cin, Cout, K, K, s, p, ws, wsb, rt.dt)
    rt.call("msc_conv2d_dgrad", dy, Wp.t, rs, ref, masked, B, H, W, Cin, Cout, K, K, s, p, ws, wsb, rt.dt)
    torch.cuda.synchronize()
    assert torch.equal(masked, torch.where(ref > 0, plain, torch.zeros_like(plain)))
    assert (masked[0, 0, 0, :4] == 0).all() and masked.abs().max() > 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(4, 28, 28, 64, 64, 3, 1, 1), (3, 7, 7, 256, 512, 3, 1, 1), (2, 14, 14, 128, 128, 3, 1, 1), (2, 9, 9, 24, 64, 3, 1, 1)])
def test_conv_dgrad_bn_statistics(rts, dtype, case):
    """msc_conv2d_dgrad_bnstat: the same dx as msc_conv2d_dgrad, bit for bit, plus per-row-block sums of dz and dz*xhat for the BatchNorm + ReLU
    that produced the layer's input — msc_bn_bwd_from_partials then gives the dx / dgamma / dbeta of msc_bn_bwd on that dx (same terms, another
    summation order).  Un-split epilogue (128x64 and 128x128 tiles, a ragged 24-channel case) and the split-K finish (7x7x256)."""
    from multimodal_sc_amd import lib as L
    rt = rts[dtype]
    B, H, W, Cin, Cout, K, s, p = case
    w = rnd(Cout, K, K, Cin, seed=1, scale=0.1)
    Wp = FakeParam(rt, w, "conv")
    dy, rs = dev(rnd(B, H, W, Cout, seed=2), rt), dev(rnd(B, H, W, Cin, seed=3), rt)
    x_raw = dev(rnd(B, H, W, Cin, seed=4) * 1.5 + 0.3, rt)                    # the input of the BatchNorm in front of this conv
    M = B * H * W
    xf = x_raw.float().reshape(M, Cin)
    mean, var = xf.mean(0), xf.var(0, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    gamma = torch.linspace(0.5, 1.5, Cin, device=rt.device)
    beta = torch.linspace(-0.3, 0.3, Cin, device=rt.device)
    scale = gamma * rstd
    stats = torch.stack([mean, rstd, scale, beta - mean * scale]).contiguous()
    wsb = L.query("msc_conv2d_splitk_workspace_bytes", M, Cin, K * K * Cout, rt.dt)
    ws = torch.empty(max(wsb // 4, 1), device=rt.device) if wsb else None
    plain = torch.empty(B, H, W, Cin, device=rt.device, dtype=rt.tdtype)
    fused = torch.full_like(plain, float("nan"))
    nblk = L.query("msc_conv2d_dgrad_stat_rows", M, Cin, rt.dt)
    part = torch.full((nblk, 2, Cin), float("nan"), device=rt.device)
    rt.call("msc_conv2d_dgrad", dy, Wp.t, rs, None, plain, B, H, W, Cin, Cout, K, K, s, p, ws, wsb, rt.dt)
    rt.call("msc_conv2d_dgrad_bnstat", dy, Wp.t, rs, fused, x_raw, stats, part, B, H, W, Cin, Cout, K, K, s, p, ws, wsb, rt.dt)
    torch.cuda.synchronize()
    assert torch.equal(plain, fused)
    assert torch.isfinite(part).all()
    # the sums themselves against fp64 of the stored values
    mask = (xf * stats[2] + stats[3]) > 0
    dz = torch.where(mask, plain.float().reshape(M, Cin), torch.zeros((), device=rt.device)).double()
    xhat = ((xf - mean) * rstd).double()
    check("sum dz", part[:, 0].double().sum(0), dz.sum(0), 2e-5 if dtype == "f32" else 2e-4)
    check("sum dz xhat", part[:, 1].double().sum(0), (dz * xhat).sum(0), 2e-5 if dtype == "f32" else 2e-4)
    # and through the BatchNorm backward
    outs = []
    for mode in (0, 1):
        dx = torch.full_like(plain, float("nan"))
        dg, db = torch.zeros(Cin, device=rt.device), torch.zeros(Cin, device=rt.device)
        wsn = torch.empty(L.query("msc_bn_bwd_workspace_bytes", Cin) // 4, device=rt.device)
        if mode == 0:
            rt.call("msc_bn_bwd", plain, x_raw, None, stats[0], stats[1], gamma, stats[2], stats[3], dx, None, dg, db, wsn, M, Cin, 1, 0, rt.dt)
        else:
            rt.call("msc_bn_bwd_from_partials", plain, x_raw, stats[0], stats[1], gamma, stats[2], stats[3], part, nblk, dx, dg, db, wsn, M, Cin, 0, rt.dt)
        outs.append((dx.float().clone(), dg.clone(), db.clone()))
    t = 1e-5 if dtype == "f32" else 1e-2
    check("dx", outs[1][0], outs[0][0], t)
    check("dgamma", outs[1][1], outs[0][1], 1e-4)
    check("dbeta", outs[1][2], outs[0][2], 1e-4)


# ------------------------------------------------------------------------------------------------ conv + BN
CONV_CASES = [  # B,H,W,Cin,Cout,K,stride,pad
    (2, 16, 16, 8, 64, 7, 2, 3),
    (2, 14, 14, 64, 64, 3, 1, 1),
    (2, 14, 14, 64, 128, 3, 2, 1),
    (2, 14, 14, 64, 128, 1, 2, 0),
    (3, 7, 7, 256, 512, 3, 1, 1),
    (4, 32, 32, 64, 64, 3, 1, 1),
    (8, 1, 1, 512, 512, 3, 1, 1),
    (8, 2, 2, 256, 512, 3, 2, 1),
]


# cases that MUST take the split-K path (forward and, being stride 1, dgrad) / the case that must NOT, in both dtypes:
# the strict per-kernel tolerances below therefore cover plan_conv_split + splitk_finish, not only the un-split kernel
CONV_SPLIT_CASES = {(3, 7, 7, 256, 512, 3, 1, 1), (8, 1, 1, 512, 512, 3, 1, 1)}
CONV_UNSPLIT_CASES = {(2, 16, 16, 8, 64, 7, 2, 3), (2, 14, 14, 64, 128, 1, 2, 0)}


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("relu", [True, False])
def test_conv_bn_fwd_bwd(rts, dtype, case, relu):
    from multimodal_sc_amd import functional as F, lib as L
    rt = rts[dtype]
    B, H, W, Cin, Cout, K, s, p = case
    Ho, Wo = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
    wsb_f = L.query("msc_conv2d_splitk_workspace_bytes", B * Ho * Wo, Cout, K * K * Cin, rt.dt)
    wsb_d = L.query("msc_conv2d_splitk_workspace_bytes", B * H * W, Cin, K * K * Cout, rt.dt)
    if case in CONV_SPLIT_CASES:
        assert wsb_f > 0 and wsb_d > 0, "this case is meant to exercise the split-K convolution path"
        assert wsb_f % (B * Ho * Wo * Cout * 4) == 0 and wsb_f // (B * Ho * Wo * Cout * 4) >= 2
    if case in CONV_UNSPLIT_CASES:
        assert wsb_f == 0, "this case is meant to exercise the un-split kernel"
    x = rnd(B, H, W, Cin, seed=1)
    w = rnd(Cout, K, K, Cin, seed=2, scale=1 / math.sqrt(K * K * Cin))
    gamma, beta = 1 + 0.1 * rnd(Cout, seed=3), 0.1 * rnd(Cout, seed=4)
    xq, wq = q(x, dtype).requires_grad_(), q(w, dtype).requires_grad_()
    gq, bq = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    conv = ste(TF.conv2d(xq.permute(0, 3, 1, 2), wq.permute(0, 3, 1, 2), None, s, p), dtype)
    bn = TF.batch_norm(conv, None, None, gq, bq, True, 0.0, 1e-5)
    ref = (torch.relu(bn) if relu else bn).permute(0, 2, 3, 1)
    Wp, G, Bt = FakeParam(rt, w, "conv"), FakeParam(rt, gamma, "vec"), FakeParam(rt, beta, "vec")
    xd = dev(x, rt).requires_grad_()
    y = F.ConvBNAct.apply(xd, None, rt, Wp, G, Bt, s, p, relu, True)
    # BN divides by the batch std: a 1x1 output map with batch 8 amplifies rounding; scale tolerance accordingly
    amp = 4.0 if B * y.shape[1] * y.shape[2] < 64 else 1.0
    check("y", y, ref, tol(dtype) * amp)
    dy = rnd(*ref.shape, seed=5)
    ref.backward(q(dy, dtype))
    y.backward(dev(dy, rt))
    check_grad("dx", xd.grad, xq.grad, tol(dtype) * 4 * amp, dtype, relu)
    check_grad("dw", Wp.g, wq.grad, tol(dtype, True) * 8 * amp, dtype, relu)
    check_grad("dgamma", G.g, gq.grad, tol(dtype, True) * 8 * amp, dtype, relu)
    check_grad("dbeta", Bt.g, bq.grad, tol(dtype, True) * 8 * amp, dtype, relu)


@pytest.mark.parametrize("dtype", DTYPES)
def test_stem_fused_bn_relu_pool(rts, dtype):
    """ImageEncoder stem as one node (conv -> BN -> ReLU -> max-pool, the last three fused; the backward gathers the pooled
    gradient through the arg-max ids and rebuilds the ReLU mask from the conv output) against torch, and bit for bit against the
    unfused kernels in the forward."""
    from multimodal_sc_amd import functional as F
    rt = rts[dtype]
    B, H, Cin, Cout = 3, 38, 8, 64          # odd-sized maps: ragged pooling windows at the border
    x = rnd(B, H, H, Cin, seed=1)
    w = rnd(Cout, 7, 7, Cin, seed=2, scale=1 / math.sqrt(49 * Cin))
    gamma, beta = 1 + 0.1 * rnd(Cout, seed=3), 0.1 * rnd(Cout, seed=4)
    xq, wq = q(x, dtype), q(w, dtype).requires_grad_()
    gq, bq = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    conv = ste(TF.conv2d(xq.permute(0, 3, 1, 2), wq.permute(0, 3, 1, 2), None, 2, 3), dtype)
    act = ste(torch.relu(TF.batch_norm(conv, None, None, gq, bq, True, 0.0, 1e-5)), dtype)
    ref = TF.max_pool2d(act, 3, 2, 1).permute(0, 2, 3, 1)
    Wp, G, Bt = FakeParam(rt, w, "conv"), FakeParam(rt, gamma, "vec"), FakeParam(rt, beta, "vec")
    anchor = torch.zeros(1, device=rt.device, requires_grad=True)
    y = F.StemFn.apply(dev(x, rt), anchor, rt, Wp, G, Bt, 2, 3, 3, 2, 1)
    check("y", y, ref, tol(dtype))
    # forward values equal the unfused path exactly (same rounding point, same tie-break)
    W2, G2, B2 = FakeParam(rt, w, "conv"), FakeParam(rt, gamma, "vec"), FakeParam(rt, beta, "vec")
    y2 = F.MaxPool.apply(F.ConvBNAct.apply(dev(x, rt), anchor, rt, W2, G2, B2, 2, 3, True, False), rt, 3, 2, 1)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    dy = rnd(*ref.shape, seed=5)
    ref.backward(q(dy, dtype))
    y.backward(dev(dy, rt))
    y2.backward(dev(dy, rt))
    check_grad("dw", Wp.g, wq.grad, tol(dtype, True) * 8, dtype, True)
    check_grad("dgamma", G.g, gq.grad, tol(dtype, True) * 8, dtype, True)
    check_grad("dbeta", Bt.g, bq.grad, tol(dtype, True) * 8, dtype, True)
    # ... and the fused backward agrees with the unfused kernels (which round the un-pooled gradient to the storage type)
    check("dw vs unfused", Wp.g, W2.g, tol(dtype, True) * 4)
    check("dgamma vs unfused", G.g, G2.g, tol(dtype, True) * 4)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("stride", [1, 2])
def test_basic_block(rts, dtype, stride):
    from multimodal_sc_amd import functional as F
    rt = rts[dtype]
    B, H, Cin = 4, 12, 64
    Cout = 64 if stride == 1 else 128
    x = rnd(B, H, H, Cin, seed=1)
    ws = {"conv1": rnd(Cout, 3, 3, Cin, seed=2, scale=0.05), "conv2": rnd(Cout, 3, 3, Cout, seed=3, scale=0.05)}
    if stride == 2:
        ws["dconv"] = rnd(Cout, 1, 1, Cin, seed=4, scale=0.1)
    vec = {k: (1 + 0.1 * rnd(Cout, seed=10 + i)) if k.endswith("g") else 0.1 * rnd(Cout, seed=20 + i)
           for i, k in enumerate(["bn1g", "bn1b", "bn2g", "bn2b", "dbng", "dbnb"])}
    xq = q(x, dtype).requires_grad_()
    tw = {k: q(v, dtype).requires_grad_() for k, v in ws.items()}
    tv = {k: v.clone().requires_grad_() for k, v in vec.items()}

    def cbn(inp, wk, gk, bk, s, pad):
        c = ste(TF.conv2d(inp, tw[wk].permute(0, 3, 1, 2), None, s, pad), dtype)
        return TF.batch_norm(c, None, None, tv[gk], tv[bk], True, 0.0, 1e-5)

    xin = xq.permute(0, 3, 1, 2)
    o = ste(torch.relu(cbn(xin, "conv1", "bn1g", "bn1b", stride, 1)), dtype)
    o = cbn(o, "conv2", "bn2g", "bn2b", 1, 1)
    idt = ste(cbn(xin, "dconv", "dbng", "dbnb", stride, 0), dtype) if stride == 2 else xin
    ref = torch.relu(o + idt).permute(0, 2, 3, 1)
    p = {k: FakeParam(rt, v, "conv") for k, v in ws.items()}
    p.update({k: FakeParam(rt, v, "vec") for k, v in vec.items() if stride == 2 or not k.startswith("db")})
    xd = dev(x, rt).requires_grad_()
    y = F.BasicBlockFn.apply(xd, rt, p, stride)
    check("y", y, ref, tol(dtype) * 2)
    dy = rnd(*ref.shape, seed=7)
    ref.backward(q(dy, dtype))
    y.backward(dev(dy, rt))
    check_grad("dx", xd.grad, xq.grad, tol(dtype) * 8, dtype, True)
    for k in ws:
        check_grad("d" + k, p[k].g, tw[k].grad, tol(dtype, True) * 16, dtype, True)
    for k in p:
        if k in vec:
            check_grad("d" + k, p[k].g, tv[k].grad, tol(dtype, True) * 16, dtype, True)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(4, 12, 64, 64, 1), (4, 12, 64, 128, 2), (3, 7, 256, 256, 1)])
def test_block_chain_premasked_gradient_is_bitwise_the_same(rts, dtype, shape):
    """Two BasicBlocks in a row.  With the pre-mask flags the SECOND block's dgrad epilogue (or split-K finish: the 7x7x256 case)
    applies the first block's output ReLU backward, and the first block skips its mask and the masked copy for its identity
    branch.  (dgrad + res) * [y > 0] is rounded once either way, so every gradient must be BIT-identical to the unflagged chain."""
    from multimodal_sc_amd import functional as F
    rt = rts[dtype]
    B, H, C0, C1, s2 = shape
    x = rnd(B, H, H, C0, seed=1)
    def params(cin, cout, stride, seed):
        ws = {"conv1": rnd(cout, 3, 3, cin, seed=seed, scale=0.05), "conv2": rnd(cout, 3, 3, cout, seed=seed + 1, scale=0.05)}
        if stride == 2:
            ws["dconv"] = rnd(cout, 1, 1, cin, seed=seed + 2, scale=0.1)
        vec = {k: (1 + 0.1 * rnd(cout, seed=seed + 10 + i)) if k.endswith("g") else 0.1 * rnd(cout, seed=seed + 20 + i)
               for i, k in enumerate(["bn1g", "bn1b", "bn2g", "bn2b"] + (["dbng", "dbnb"] if stride == 2 else []))}
        return ws, vec
    res = []
    dy = None
    for flags in (False, True):
        ps = []
        for cin, cout, st, seed in ((C0, C0, 1, 100), (C0, C1, s2, 200)):
            ws, vec = params(cin, cout, st, seed)
            p = {k: FakeParam(rt, v, "conv") for k, v in ws.items()}
            p.update({k: FakeParam(rt, v, "vec") for k, v in vec.items()})
            ps.append((p, st))
        xd = dev(x, rt).requires_grad_()
        h = F.BasicBlockFn.apply(xd, rt, ps[0][0], ps[0][1], False, flags)
        y = F.BasicBlockFn.apply(h, rt, ps[1][0], ps[1][1], flags, False)
        if dy is None:
            dy = dev(rnd(*y.shape, seed=7), rt)
        y.backward(dy)
        torch.cuda.synchronize()
        rt.join_side()
        torch.cuda.synchronize()
        res.append((y.detach().clone(), xd.grad.clone(), [pp.g.clone() for p, _ in ps for pp in p.values()]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert res[0][1].abs().max() > 0
    for a, b in zip(res[0][2], res[1][2]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", DTYPES)
def test_block_with_fused_bn_statistics(rts, dtype, monkeypatch):
    """A BasicBlock whose conv2 dgrad epilogue does bn1's backward statistics pass (MSC_BN_STAT_FUSE=1, opt-in): every gradient equals the
    default path's within the summation-order difference of the two per-channel sums."""
    from multimodal_sc_amd import functional as F
    rt = rts[dtype]
    B, H, C = 3, 14, 64
    x, dy = rnd(B, H, H, C, seed=1), rnd(B, H, H, C, seed=7)
    ws = {"conv1": rnd(C, 3, 3, C, seed=100, scale=0.05), "conv2": rnd(C, 3, 3, C, seed=101, scale=0.05)}
    vec = {k: (1 + 0.1 * rnd(C, seed=110 + i)) if k.endswith("g") else 0.1 * rnd(C, seed=120 + i) for i, k in enumerate(["bn1g", "bn1b", "bn2g", "bn2b"])}
    res = []
    for fuse in (False, True):
        monkeypatch.setattr(F, "_BN_STAT_FUSE", fuse)
        p = {k: FakeParam(rt, v, "conv") for k, v in ws.items()}
        p.update({k: FakeParam(rt, v, "vec") for k, v in vec.items()})
        xd = dev(x, rt).requires_grad_()
        y = F.BasicBlockFn.apply(xd, rt, p, 1)
        y.backward(dev(dy, rt))
        torch.cuda.synchronize()
        rt.join_side()
        torch.cuda.synchronize()
        res.append((xd.grad.float().clone(), {k: pp.g.clone() for k, pp in p.items()}))
    t = 1e-5 if dtype == "f32" else 2e-2
    check("dx", res[1][0], res[0][0], t)
    for k in res[0][1]:
        check(k, res[1][1][k], res[0][1][k], 1e-4 if dtype == "f32" else 2e-2)
    assert res[0][0].abs().max() > 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_maxpool(rts, dtype):
    from multimodal_sc_amd import functional as F
    rt = rts[dtype]
    x = rnd(2, 17, 17, 64, seed=1)
    xq = q(x, dtype).requires_grad_()
    ref = TF.max_pool2d(xq.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    xd = dev(x, rt).requires_grad_()
    y = F.MaxPool.apply(xd, rt, 3, 2, 1)
    check("y", y, ref, 0.0)
    dy = rnd(*ref.shape, seed=2)
    ref.backward(q(dy, dtype))
    y.backward(dev(dy, rt))
    check("dx", xd.grad, xq.grad, tol(dtype))


# ------------------------------------------------------------------------------------------------ LN / tokens / pools
@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm(rts, dtype):
    from multimodal_sc_amd import functional as F
    rt = rts[dtype]
    x, g, b = rnd(3, 50, 512, seed=1) * 2 + 0.5, 1 + 0.1 * rnd(512, seed=2), 0.1 * rnd(512, seed=3)
    xq, gq, bq = q(x, dtype).requires_grad_(), g.clone().requires_grad_(), b.clone().requires_grad_()
    ref = TF.layer_norm(xq, (512,), gq, bq, 1e-5)
    G, Bp = FakeParam(rt, g, "vec"), FakeParam(rt, b, "vec")
    xd = dev(x, rt).requires_grad_()
    y = F.LayerNorm.apply(xd, rt, G, Bp)
    check("y", y, ref, tol(dtype))
    dy = rnd(3, 50, 512, seed=4)
    ref.backward(q(dy, dtype))
    y.backward(dev(dy, rt))
    check("dx", xd.grad, xq.grad, tol(dtype) * 2)
    check("dg", G.g, gq.grad, tol(dtype, True) * 4)
    check("db", Bp.g, bq.grad, tol(dtype, True) * 4)


@pytest.mark.parametrize("dtype", DTYPES)
def test_tokens_meanpool(rts, dtype):
    from multimodal_sc_amd import functional as F
    rt = rts[dtype]
    B, D = 3, 512
    s0, s1, s2, pos = rnd(B, 4, D, seed=1), rnd(B, 9, D, seed=2), rnd(B, 1, D, seed=3), rnd(14, D, seed=4)
    t = [q(s, dtype).requires_grad_() for s in (s0, s1, s2)]
    pq = pos.clone().requires_grad_()
    tok = torch.cat(t, 1) + pq
    ref = tok.mean(1)
    Pp = FakeParam(rt, pos, "vec")
    d = [dev(s, rt).requires_grad_() for s in (s0, s1, s2)]
    tk = F.TokensAssemble.apply(d[0], d[1], d[2], rt, Pp)
    check("tokens", tk, tok, tol(dtype))
    y = F.MeanPool.apply(tk, rt)
    check("mean", y, ref, tol(dtype))
    dy = rnd(B, D, seed=5)
    ref.backward(q(dy, dtype))
    y.backward(dev(dy, rt))
    for i in range(3):
        check(f"ds{i}", d[i].grad, t[i].grad, tol(dtype))
    check("dpos", Pp.g, pq.grad, tol(dtype, True) * 2)


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("dtype,fused", [("f32", False), ("bf16", False), ("bf16", True)])
@pytest.mark.parametrize("T", [306, 49, 1, 512, 33])
def test_attention(rts, dtype, fused, T):
    """fused = the flash-style LDS-resident kernel (bf16); unfused = batched GEMM + softmax kernels (both dtypes).
    Large-magnitude logits (qkv ~ N(0,1), |q.k|/8 up to ~4) exercise the online-softmax rescale path; T=306/33 the
    ragged last key tile and padded query rows; T=1 a tile with a single valid key."""
    from multimodal_sc_amd import functional as F
    rt = rts[dtype]
    rt.fused_attention = fused
    B, h, hd = 2, 8, 64
    D = h * hd
    qkv = rnd(B * T, 3 * D, seed=1)
    qkv[:, :D] *= 2.0            # sharper softmax: the running maximum moves between key tiles
    do = rnd(B * T, D, seed=2)
    t = q(qkv, dtype).requires_grad_()
    v5 = t.view(B, T, 3, h, hd)
    qq, kk, vv = (v5[:, :, i].transpose(1, 2) for i in range(3))
    att = torch.softmax(qq @ kk.transpose(-1, -2) / math.sqrt(hd), -1)
    ref = (att @ vv).transpose(1, 2).reshape(B * T, D)
    qd = dev(qkv, rt)
    o, P = F.attention_fwd(rt, qd, B, T, h, hd)
    check("o", o, ref, tol(dtype))
    ref.backward(q(do, dtype))
    dqkv = F.attention_bwd(rt, dev(do, rt), qd, o, P, B, T, h, hd)
    rt.fused_attention = True
    check("dqkv", dqkv, t.grad, tol(dtype) * 4)


# ------------------------------------------------------------------------------------------------ channel
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", ["awgn", "rayleigh"])
def test_channel(rts, dtype, mode):
    import sys
    from multimodal_sc_amd import functional as F, lib as L
    from oracle.model import Channel
    rt = rts[dtype]
    B, k = 10, 256
    x, n, hfad = rnd(B, k, seed=1) * 3, rnd(B, k, seed=2), rnd(B, k, seed=3)
    snr = torch.linspace(-5, 15, B)
    xq = q(x, dtype).requires_grad_()
    ref = Channel(mode)(xq, snr, n, hfad)
    xd = dev(x, rt).requires_grad_()
    y = F.ChannelFn.apply(xd, rt, snr.to(rt.device), n.to(rt.device), hfad.to(rt.device), L.CH_AWGN if mode == "awgn" else L.CH_RAYLEIGH, 0, 0, None)
    # Rayleigh deep fades amplify noise by 1/|h|^2: compare relative to the output magnitude
    check("y", y, ref, tol(dtype))
    dy = rnd(B, k, seed=4)
    ref.backward(q(dy, dtype))
    y.backward(dev(dy, rt))
    check("dx", xd.grad, xq.grad, tol(dtype) * 2)


def test_channel_power_and_snr(rts):
    """known-answer checks: unit power after normalisation; measured SNR within 0.1 dB (Philox noise)."""
    from multimodal_sc_amd import lib as L
    rt = rts["f32"]
    B, k = 4096, 256
    x = (rnd(B, k, seed=1) * 5).to(rt.device)
    for snr_db in (-5.0, 0.0, 10.0):
        snr = torch.full((B,), snr_db, device=rt.device)
        y, scale = rt.f32(B, k), rt.f32(B)
        rt.call("msc_channel_fwd", x, None, None, snr, y, scale, B, k, L.CH_AWGN, 1234, 0, None, rt.dt)
        s = x * scale[:, None]
        assert abs(float(s.pow(2).mean()) - 1.0) < 1e-5
        meas = 10 * math.log10(1.0 / float((y - s).pow(2).mean()))
        assert abs(meas - snr_db) < 0.1, (meas, snr_db)
    # Rayleigh: E|h|^2 = 1 from the fading stream
    z = rt.f32(1 << 20)
    rt.call("msc_philox_normal", z, (1 << 20) // 4, 77, 1, 0)
    h2 = (z.view(-1, 2) / math.sqrt(2)).pow(2).sum(-1).mean()
    assert abs(float(h2) - 1.0) < 1e-2


def test_philox_matches_cpu_restatement(rts):
    from oracle.philox import philox_normal
    rt = rts["f32"]
    nq = 1000
    z = rt.f32(nq * 4)
    rt.call("msc_philox_normal", z, nq, 0x123456789ABCDEF, 1, 0)
    ref = torch.from_numpy(philox_normal(nq, 0x123456789ABCDEF, 1))
    check("philox", z, ref, 2e-5)
    # the draw counter (bumped per forward) and the rank word of the key select independent streams, none aliasing another
    for seed, draw in ((0x123456789ABCDEF, 5), (7, (3 << 32) | 9), ((1 << 32) | 7, 0)):
        rt.call("msc_philox_normal", z, nq, seed, 0, draw)
        check(f"philox seed={seed:#x} draw={draw:#x}", z, torch.from_numpy(philox_normal(nq, seed, 0, draw)), 2e-5)
    outs = []
    for seed, draw in ((0, 1), (1, 0), (1 << 32, 0), (0, 0)):      # (seed 0, draw 1) must differ from (seed 1, draw 0)
        rt.call("msc_philox_normal", z, nq, seed, 0, draw)
        outs.append(z.clone())
    for i in range(len(outs)):
        for j in range(i):
            assert not torch.equal(outs[i], outs[j]), (i, j)


def test_channel_noise_fresh_per_forward_and_per_rank(rts):
    """in-kernel noise: the draw counter feeds a Philox counter word, so consecutive forwards (rollouts, inference) and
    different DP ranks (key word) see different realisations; the same (seed, rank, draw) reproduces bit for bit."""
    from multimodal_sc_amd import lib as L
    rt = rts["f32"]
    B, k = 8, 256
    x = (rnd(B, k, seed=1) * 2).to(rt.device)
    snr = torch.full((B,), 5.0, device=rt.device)
    draw = torch.zeros(1, dtype=torch.int64, device=rt.device)

    def run(seed, hdraw=0):
        y, scale = rt.f32(B, k), rt.f32(B)
        rt.call("msc_channel_fwd", x, None, None, snr, y, scale, B, k, L.CH_RAYLEIGH, seed, hdraw, draw, rt.dt)
        return y

    a0 = run(3)
    rt.call("msc_counter_add", draw, 1)
    a1 = run(3)
    r1 = run((1 << 32) | 3)                  # rank 1, same draw
    draw.zero_()
    b0 = run(3)
    torch.cuda.synchronize()
    h1 = run(3, hdraw=1)                     # host-supplied draw 1 + device counter 0 == device counter 1
    torch.cuda.synchronize()
    assert torch.equal(a0, b0) and torch.equal(h1, a1)
    assert not torch.equal(a0, a1) and not torch.equal(a1, r1)
    assert (a0 - a1).abs().mean() > 0.1


# ------------------------------------------------------------------------------------------------ pillars
@pytest.mark.parametrize("dtype", DTYPES)
def test_pillar(rts, dtype):
    from multimodal_sc_amd import functional as F
    from oracle.model import BEVEncoder, MSCConfig
    rt = rts[dtype]
    cfg = MSCConfig(bev_grid=32, bev_range=12.8)
    B, Np = 2, 4096
    g = torch.Generator().manual_seed(0)
    pts = torch.cat([(torch.rand(B, Np, 2, generator=g) * 2 - 1) * 14.0, torch.rand(B, Np, 2, generator=g)], -1)  # some outside
    torch.manual_seed(4321)          # the point-MLP init must not depend on which tests ran before
    enc = BEVEncoder(cfg)
    ref = enc.pillar_grid(pts)
    dy = rnd(*ref.shape, seed=3)
    ref.backward(q(dy, dtype))
    W, Bp = FakeParam(rt, enc.point_mlp.weight.detach(), "vec"), FakeParam(rt, enc.point_mlp.bias.detach(), "vec")
    anchor = torch.zeros(1, device=rt.device, requires_grad=True)
    out = F.PillarFn.apply(pts.to(rt.device), anchor, rt, W, Bp, cfg.bev_grid, cfg.bev_range)
    check("grid", out, ref, tol(dtype))
    out.backward(dev(dy, rt))
    # Arg-max routing: when two points of a cell have features within an ulp of each other, the kernel (fmaf chain)
    # and torch (its own summation order) may crown different points; the gradient then moves between them.  Rare and
    # harmless, but a max-abs bound is the wrong yardstick for it -> relative Frobenius error.
    torch.cuda.synchronize()
    for name, got, want in (("dw", W.g, enc.point_mlp.weight.grad), ("db", Bp.g, enc.point_mlp.bias.grad)):
        g64, r64 = got.double().cpu(), want.double()
        rel = ((g64 - r64).norm() / r64.norm()).item()
        assert rel < 2e-2, f"{name}: relative Frobenius error {rel:.3e}"


@pytest.mark.parametrize("case", ["all_outside", "one_cell", "duplicates", "ragged"])
def test_pillar_edge_cases(rts, case):
    """collisions and empties of the BEV binning against the oracle (f32 mode, tight): every point outside the grid (empty grid, zero
    gradients), every point of a frame in ONE cell (a 3000-point bin), exact duplicates (tied maxima share the gradient evenly, as
    scatter_reduce(amax) does), and frames with very different numbers of valid points"""
    from multimodal_sc_amd import functional as F
    from oracle.model import BEVEncoder, MSCConfig
    rt = rts["f32"]
    cfg = MSCConfig(bev_grid=32, bev_range=12.8)
    B, Np = 3, 3000
    g = torch.Generator().manual_seed(11)
    pts = torch.cat([(torch.rand(B, Np, 2, generator=g) * 2 - 1) * 12.0, torch.rand(B, Np, 2, generator=g)], -1)
    if case == "all_outside":
        pts[..., 0] += 100.0
    elif case == "one_cell":
        pts[0, :, 0] = 0.1 + 0.5 * torch.rand(Np, generator=g)          # frame 0: one 0.8 m cell holds all 3000 points
        pts[0, :, 1] = 0.1 + 0.5 * torch.rand(Np, generator=g)
    elif case == "duplicates":
        pts[:, 1::2] = pts[:, 0::2]                                       # every point twice: every maximum is a two-way tie
    elif case == "ragged":
        pts[1, 100:, 0] = 1e3                                             # frame 1 keeps 100 points, frame 2 none
        pts[2, :, 1] = 1e6
    torch.manual_seed(99)
    enc = BEVEncoder(cfg)
    ref_in = pts
    ref = enc.pillar_grid(ref_in)
    dy = rnd(*ref.shape, seed=3)
    ref.backward(dy)
    W, Bp = FakeParam(rt, enc.point_mlp.weight.detach(), "vec"), FakeParam(rt, enc.point_mlp.bias.detach(), "vec")
    anchor = torch.zeros(1, device=rt.device, requires_grad=True)
    out = F.PillarFn.apply(ref_in.to(rt.device), anchor, rt, W, Bp, cfg.bev_grid, cfg.bev_range)
    check("grid", out, ref, 1e-5)
    out.backward(dy.to(rt.device))
    torch.cuda.synchronize()
    for name, got, want in (("dw", W.g, enc.point_mlp.weight.grad), ("db", Bp.g, enc.point_mlp.bias.grad)):
        g64, r64 = got.double().cpu(), want.double()
        assert torch.isfinite(g64).all(), name
        den = r64.norm().item()
        if den == 0.0:
            assert g64.abs().max().item() == 0.0, name
        else:
            rel = ((g64 - r64).norm() / den).item()
            assert rel < 2e-3, f"{case} {name}: relative Frobenius error {rel:.3e}"


def test_add2_f32(rts):
    """out = (accumulate ? out : 0) + (a + b): the fixed-order sum of the two fusion chains' LayerNorm parameter gradients"""
    rt = rts["bf16"]
    a, b, o = rnd(1024, seed=1).to(rt.device), rnd(1024, seed=2).to(rt.device), rnd(1024, seed=3).to(rt.device)
    o0 = o.clone()
    rt.call("msc_add2_f32", a, b, o, 1024, 1)
    torch.cuda.synchronize()
    assert torch.equal(o, o0 + (a + b))
    rt.call("msc_add2_f32", a, b, o, 1024, 0)
    torch.cuda.synchronize()
    assert torch.equal(o, a + b)


# ------------------------------------------------------------------------------------------------ heads / losses
def test_argmax_ties_bit_exact(rts):
    for dtype in DTYPES:
        rt = rts[dtype]
        x = torch.tensor([[1.0, 3.0, 3.0, 2.0], [5.0, 5.0, 5.0, 5.0], [-1.0, -2.0, -1.0, -3.0], [0.0, 0.0, 1.0, 1.0]])
        x = torch.nn.functional.pad(x, (0, 12), value=9.0)   # padded columns must be ignored (n=4, ld=16)
        big = rnd(1000, 16, seed=1)
        allx = torch.cat([x, big.to(torch.bfloat16).float()], 0)
        out = torch.empty(allx.shape[0], dtype=torch.int64, device=rt.device)
        rt.call("msc_argmax_rows", dev(allx, rt), out, allx.shape[0], 4, 16, rt.dt)
        assert torch.equal(out.cpu(), torch.argmax(allx[:, :4], -1))
        assert out[:4].tolist() == [1, 0, 0, 2]


def test_topk_rows_bit_exact(rts):
    """msc_topk_rows against oracle/rl.py topk_rows and the committed fixture (hand-built exact ties, signed zeros, +-inf, NaN,
    random rows on the bf16 grid): int64 indices AND values bit-exact for k = 1, 3, 8 in both storage types; k = 1 equals
    msc_argmax_rows; padded columns (n < ld) are ignored."""
    import os
    import numpy as np
    from oracle import rl as R
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "topk.npz"), allow_pickle=False)
    x = torch.from_numpy(G["x"])
    rows, n = x.shape
    xp = torch.nn.functional.pad(x, (0, 16 - n), value=99.0)
    for dtype in DTYPES:
        rt = rts[dtype]
        xd = dev(xp, rt)
        for k in (1, 3, 8):
            idx = torch.full((rows, k), -7, dtype=torch.int64, device=rt.device)
            val = torch.full((rows, k), -7.0, device=rt.device)
            rt.call("msc_topk_rows", xd, idx, val, rows, n, k, 16, rt.dt)
            torch.cuda.synchronize()
            assert np.array_equal(idx.cpu().numpy(), G[f"idx{k}"]), (dtype, k)
            assert np.array_equal(val.cpu().numpy(), G[f"val{k}"], equal_nan=True), (dtype, k)
            oi, ov = R.topk_rows(G["x"], k)
            assert np.array_equal(idx.cpu().numpy(), oi)
        am = torch.empty(rows, dtype=torch.int64, device=rt.device)
        rt.call("msc_argmax_rows", xd, am, rows, n, 16, rt.dt)
        i1 = torch.empty(rows, 1, dtype=torch.int64, device=rt.device)
        rt.call("msc_topk_rows", xd, i1, None, rows, n, 1, 16, rt.dt)           # out_val is optional
        torch.cuda.synchronize()
        assert torch.equal(am, i1[:, 0])
    # argument validation: k > 8, k > n, n > 32 are refused
    rt = rts["f32"]
    from multimodal_sc_amd import lib as L
    for bad in ((9, 9), (4, 5), (40, 2)):
        assert L.lib.msc_topk_rows(1, 1, None, 4, bad[0], bad[1], 64, 0, None) == -1


@pytest.mark.parametrize("dtype", DTYPES)
def test_ppo_loss(rts, dtype):
    from oracle.model import MSCConfig, ppo_loss
    rt = rts[dtype]
    cfg = MSCConfig()
    B, A = 64, 9
    head = rnd(B, 16, seed=1)
    g = torch.Generator().manual_seed(2)
    act = torch.randint(0, A, (B,), generator=g)
    old = -math.log(A) + 0.3 * torch.randn(B, generator=g)
    adv, ret = torch.randn(B, generator=g), torch.randn(B, generator=g)
    hq = q(head, dtype).requires_grad_()
    tot, pl, vl, ent = ppo_loss(hq[:, :A], hq[:, A], act, old, adv, ret, cfg)
    tot.backward()
    terms = rt.f32(4)
    hd_ = dev(head, rt)
    dh = torch.empty_like(hd_)
    rt.call("msc_ppo_loss", hd_, act.to(rt.device), old.to(rt.device), adv.to(rt.device), ret.to(rt.device), terms, dh, B, A, 16,
            cfg.clip_eps, cfg.c_value, cfg.c_entropy, rt.dt)
    check("terms", terms, torch.stack([tot, pl, vl, ent]), 1e-5)
    check("dhead", dh, hq.grad, 1e-5 if dtype == "f32" else 1e-2 * float(hq.grad.abs().max()))


def test_ppo_hand_computed(rts):
    """4-sample hand-computed clipped surrogate: uniform logits -> ratio known in closed form."""
    rt = rts["f32"]
    A, B = 3, 4
    head = torch.zeros(B, 16)
    head[:, A] = torch.tensor([1.0, 0.0, -1.0, 2.0])
    lp = -math.log(3.0)
    old = torch.tensor([lp, lp - math.log(1.5), lp + math.log(2.0), lp])     # ratios 1, 1.5, 0.5, 1
    adv = torch.tensor([1.0, 2.0, -1.0, -3.0])
    ret = torch.tensor([0.0, 0.0, 0.0, 0.0])
    # surr = min(r*A, clip(r,.8,1.2)*A): 1, min(3,2.4)=2.4, min(-.5,-.8)=-.8, -3  -> policy = -mean = 0.1
    terms, dh = rt.f32(4), torch.empty(B, 16, device=rt.device)
    rt.call("msc_ppo_loss", head.to(rt.device), torch.zeros(B, dtype=torch.int64, device=rt.device), old.to(rt.device), adv.to(rt.device),
            ret.to(rt.device), terms, dh, B, A, 16, 0.2, 0.5, 0.0, rt.dt)
    t = terms.cpu()
    assert abs(float(t[1]) - 0.1) < 1e-6
    assert abs(float(t[2]) - 0.5 * (1 + 0 + 1 + 4) / 4) < 1e-6
    assert abs(float(t[3]) - math.log(3.0)) < 1e-6


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("G,P", [(2, 32), (3, 12)])      # 8-pixel vector path / element-wise path (P % 8 != 0)
def test_recon_loss_psnr(rts, dtype, G, P):
    rt = rts[dtype]
    B = 3
    img = torch.rand(B, 3, G * P, G * P, generator=torch.Generator().manual_seed(1))
    pat = torch.rand(B * G * G, P * P * 3, generator=torch.Generator().manual_seed(2))
    pq = q(pat, dtype)
    rec = pq.view(B, G, G, P, P, 3).permute(0, 5, 1, 3, 2, 4).reshape(B, 3, G * P, G * P)
    sse_ref = (rec - img).pow(2).flatten(1).sum(1)
    sse, ws = rt.f32(B), rt.f32(B * 32)
    pd = dev(pat, rt)
    dp = torch.empty_like(pd)
    rt.call("msc_recon_loss", pd, img.to(rt.device), sse, dp, ws, B, G, P, 0.25, rt.dt)
    check("sse", sse, sse_ref, 1e-5)
    out_img = rt.f32(B, 3, G * P, G * P)
    rt.call("msc_unpatchify", pd, out_img, B, G, P, rt.dt)
    check("unpatchify", out_img, rec, 0.0)
    dref = 0.25 * (rec - img).view(B, 3, G, P, G, P).permute(0, 2, 4, 3, 5, 1).reshape(B * G * G, P * P * 3)
    check("drecon", dp, dref, tol(dtype))
    # the caller's image as a contiguous view at a 4-byte storage offset (eager mode passes the user's tensor as it is): the 16-byte vector
    # path must not be taken on it — same results through the element-wise loop
    store = torch.empty(img.numel() + 1, device=rt.device)
    img_off = store[1:].view_as(img)
    img_off.copy_(img)
    assert img_off.data_ptr() % 16 == 4 and img_off.is_contiguous()
    sse_o, dp_o = rt.f32(B), torch.empty_like(pd)
    rt.call("msc_recon_loss", pd, img_off, sse_o, dp_o, ws, B, G, P, 0.25, rt.dt)
    check("sse, image at a 4-byte offset", sse_o, sse_ref, 1e-5)
    check("drecon, image at a 4-byte offset", dp_o, dref, tol(dtype))
    # PSNR known answers: uniform error eps -> 20 log10(1/eps)
    eps = 0.1
    pat2 = (img + eps).view(B, 3, G, P, G, P).permute(0, 2, 4, 3, 5, 1).reshape(B * G * G, P * P * 3)
    rt2 = rts["f32"]
    sse2 = rt2.f32(B)
    rt2.call("msc_recon_loss", pat2.to(rt2.device), img.to(rt2.device), sse2, None, rt2.f32(B * 32), B, G, P, 0.0, rt2.dt)
    psnr = 10 * torch.log10(3 * (G * P) ** 2 / sse2.cpu())
    assert (psnr - 20 * math.log10(1 / eps)).abs().max() < 1e-3


# ------------------------------------------------------------------------------------------------ optimizer
def test_adam_matches_torch(rts):
    rt = rts["bf16"]
    n = 10000
    p0, g1, g2 = rnd(n, seed=1), rnd(n, seed=2), rnd(n, seed=3)
    ref = p0.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    p, m, v = p0.to(rt.device), torch.zeros(n, device=rt.device), torch.zeros(n, device=rt.device)
    sh = torch.empty(n, dtype=torch.bfloat16, device=rt.device)
    for step, g in enumerate((g1, g2), 1):
        ref.grad = g.clone()
        opt.step()
        rt.call("msc_adam_step", p, (g * 4).to(rt.device), m, v, sh, n, 1e-2, 0.9, 0.999, 1e-8, 0.0, step, None, 0.25, rt.dt)
    check("p", p, ref, 1e-6)
    assert torch.equal(sh.cpu(), p.cpu().to(torch.bfloat16))


# ------------------------------------------------------------------------------------------------ fp8 (e4m3) codec GEMMs
@pytest.mark.parametrize("dtype", DTYPES)
def test_fp8_quantize_matches_oracle_grid(rts, dtype):
    """msc_fp8_quantize against the oracle's written-out e4m3 rounding: the scale and EVERY de-quantised value bit-exact
    (fp32 input; bf16 mode: de-quantised values are exactly representable fp8 x scale products rounded once to bf16)."""
    from oracle import model as O
    rt = rts[dtype]
    for n, sc in [(64 * 512, 1.0), (512 * 512, 0.03), (1028, 300.0)]:
        x = rnd(n, seed=n, scale=sc)
        x[5], x[6] = 0.0, -x.abs().max() * 1.0          # exact zero and the (negative) maximum
        xin = q(x, dtype)
        xd = dev(x, rt)
        qb = torch.empty(n, dtype=torch.uint8, device=rt.device)
        dq = rt.empty(n)
        s = torch.empty(1, device=rt.device)
        rt.call("msc_fp8_quantize", xd, qb, dq, s, n, rt.dt, rt.dt)
        torch.cuda.synchronize()
        amax = xin.abs().max()
        assert s.item() == (amax / 448.0).item()
        ref = O.fp8_qdq(xin)
        ref = ref if dtype == "f32" else ref.to(torch.bfloat16).float()
        assert torch.equal(dq.float().cpu(), ref), (dq.float().cpu() - ref).abs().max()
        # the bytes decode (OCP e4m3fn) to the same grid values
        dec = qb.cpu().view(torch.float8_e4m3fn).float() * s.item()
        assert torch.equal(dec if dtype == "f32" else dec.to(torch.bfloat16).float(), ref)
    z = torch.zeros(64, device=rt.device, dtype=rt.tdtype)
    qb, dq, s = torch.empty(64, dtype=torch.uint8, device=rt.device), rt.empty(64), torch.empty(1, device=rt.device)
    rt.call("msc_fp8_quantize", z, qb, dq, s, 64, rt.dt, rt.dt)
    torch.cuda.synchronize()
    assert s.item() == 1.0 and not qb.any() and not dq.any()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,act", [(64, 512, 512, 1), (64, 256, 512, 0), (37, 96, 256, 1), (512, 512, 256, 0)])
def test_linear_fp8_fwd_bwd(rts, dtype, M, N, K, act):
    """LinearFP8 (quantise + fp8 MFMA + straight-through backward) against oracle.FP8Linear on identical weights"""
    from multimodal_sc_amd import functional as F
    from oracle import model as O
    rt = rts[dtype]
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3, scale=0.1)
    dy = rnd(M, N, seed=4)
    lin = O.FP8Linear(K, N)
    with torch.no_grad():
        lin.weight.copy_(w)
        lin.bias.copy_(b)
    xq = q(x, dtype).requires_grad_()
    pre = lin(xq)
    ref = torch.relu(pre) if act == 1 else pre
    W, Bp = FakeParam(rt, w, "linear"), FakeParam(rt, b, "vec")
    xd = dev(x, rt).requires_grad_()
    y = F.LinearFP8.apply(xd, rt, W, Bp, act, True)
    # the MFMA sums exact products of e4m3 values and scales once; the oracle multiplies de-quantised fp32 operands
    check("y", y, ref, tol(dtype))
    ref.backward(q(dy, dtype))
    y.backward(dev(dy, rt))
    check_grad("dx", xd.grad, xq.grad, tol(dtype) * 2, dtype, act == 1)
    check_grad("dw", W.g, lin.weight.grad, tol(dtype, True) * 4, dtype, act == 1)
    check_grad("db", Bp.g, lin.bias.grad, tol(dtype, True) * 4, dtype, act == 1)


# ------------------------------------------------------------------------------------------------ PPO-loop kernels, wire format
def test_gae_bit_exact(rts):
    """msc_gae against oracle/rl.py: same fp32 recurrence, unfused -> bit-exact"""
    import numpy as np
    from oracle import rl as R
    rt = rts["f32"]
    T, N = 33, 130
    g = torch.Generator().manual_seed(5)
    rew, val = torch.randn(T, N, generator=g), torch.randn(T + 1, N, generator=g)
    done = (torch.rand(T, N, generator=g) < 0.1).float()
    adv, ret = torch.empty(T, N, device=rt.device), torch.empty(T, N, device=rt.device)
    rt.call("msc_gae", rew.to(rt.device), val.to(rt.device), done.to(rt.device), adv, ret, T, N, 0.99, 0.95)
    torch.cuda.synchronize()
    oadv, oret = R.gae(rew.numpy(), val.numpy(), done.numpy(), 0.99, 0.95)
    assert np.array_equal(adv.cpu().numpy(), oadv) and np.array_equal(ret.cpu().numpy(), oret)
    x = adv.clone()
    rt.call("msc_normalize", x, T * N, 1e-8)
    check("normalize", x, torch.from_numpy(R.normalize(oadv)), 1e-5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_sample_actions(rts, dtype):
    """inverse-CDF sampling driven by Philox: actions bit-exact against oracle/rl.py on the logits the kernel sees,
    log-probabilities within 1e-5, values passed through; the empirical distribution follows softmax(logits)."""
    import numpy as np
    from oracle import rl as R
    rt = rts[dtype]
    B, A, ld = 4096, 9, 16
    head = rnd(B, ld, seed=3, scale=1.5)
    hd_ = dev(head, rt)
    act = torch.empty(B, dtype=torch.int64, device=rt.device)
    logp, val = torch.empty(B, device=rt.device), torch.empty(B, device=rt.device)
    seen = q(head, dtype)
    for step in (0, 7, (1 << 33) + 5):
        rt.call("msc_sample_actions", hd_, act, logp, val, B, A, ld, 1234, None, step, rt.dt)
        torch.cuda.synchronize()
        oa, ol, margin = R.sample_actions(seen[:, :A].numpy(), 1234, step, return_margin=True)
        ga = act.cpu().numpy()
        # bit-exact, except that a uniform within a few ulps of a CDF boundary may fall on the neighbouring action
        # (expf on the GPU and np.exp can differ in the last bit); such rows are rare (~1e-6 of them) and must be adjacent
        diff = ga != oa
        assert diff.sum() <= 2 and np.all(margin[diff] < 1e-6) and np.all(np.abs(ga[diff] - oa[diff]) == 1), (diff.sum(), margin[diff])
        assert np.abs(logp.cpu().numpy() - ol)[~diff].max() < 1e-5
        assert torch.equal(val.cpu(), seen[:, A])
    same = torch.zeros(B, ld)
    same[:, :A] = torch.tensor([2.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -1.0])
    rt.call("msc_sample_actions", dev(same, rt), act, logp, val, B, A, ld, 99, None, 0, rt.dt)
    freq = torch.bincount(act.cpu(), minlength=A).float() / B
    assert (freq - torch.softmax(same[0, :A], 0)).abs().max() < 0.03


@pytest.mark.parametrize("dtype", DTYPES)
def test_symbols_wire_format(rts, dtype):
    """pack: bytes and scales bit-exact against oracle/wire.py (integer output); unpack o pack within half a step"""
    import numpy as np
    from oracle import wire as W
    from multimodal_sc_amd.wire import pack_symbols, unpack_symbols, frame_bytes
    rt = rts[dtype]
    B, k = 67, 256
    x = rnd(B, k, seed=8, scale=1.3)
    x[3] = 0.0
    x[4, 10] = 50.0
    qd, sd = pack_symbols(rt, dev(x, rt))
    y = unpack_symbols(rt, qd, sd)
    torch.cuda.synchronize()
    oq, os_ = W.pack(q(x, dtype).numpy())
    assert np.array_equal(qd.cpu().numpy(), oq) and np.array_equal(sd.cpu().numpy(), os_)
    ref = torch.from_numpy(W.unpack(oq, os_))
    check("unpack", y, ref, tol(dtype))
    assert (ref - q(x, dtype)).abs().max() <= 0.5 * os_.max() * 1.001
    assert frame_bytes(k) == 260


# ------------------------------------------------------------------------------------------------ full-size properties
def test_full_size_invariants_c3_batch64(rts):
    """BASELINE.json's full sizes (C3, 64 frames) are too large for the CPU oracle in a test, so the kernels are held to
    size-independent properties there — all EXACT in bf16:
      * attention with V == 1 returns 1 for every (token, channel) (softmax rows sum to one), its dV sums d(out);
      * a linear layer is homogeneous: f(2x) == 2 f(x) bit for bit (no bias), and its weight gradient scales the same way;
      * a 3x3 convolution with a centred delta kernel is the identity on the image-encoder layer-1 map;
      * the power-normalised channel output has unit average power per frame at infinite SNR."""
    from multimodal_sc_amd import functional as F
    from multimodal_sc_amd import lib as L
    rt = rts["bf16"]
    dev_ = rt.device
    B, T, h, hd = 64, 306, 8, 64
    D = h * hd
    g = torch.Generator().manual_seed(11)
    qkv = (torch.randn(B * T, 3 * D, generator=g)).to(torch.bfloat16)
    qkv[:, 2 * D:] = 1.0
    o, P = F.attention_fwd(rt, qkv.to(dev_), B, T, h, hd)
    torch.cuda.synchronize()
    assert torch.equal(o.float().cpu(), torch.ones(B * T, D))
    # linear homogeneity at M = 64 * 306 rows
    M, N, K = B * T, 1536, 512
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev_)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev_)
    y1, y2 = rt.empty(M, N), rt.empty(M, N)
    rt.call("msc_linear_fwd", x, w, None, None, y1, None, M, N, K, L.ACT_NONE, rt.dt)
    rt.call("msc_linear_fwd", x * 2, w, None, None, y2, None, M, N, K, L.ACT_NONE, rt.dt)
    torch.cuda.synchronize()
    assert torch.equal(y2.float(), 2 * y1.float())
    dw1, dw2 = torch.empty(N, K, device=dev_), torch.empty(N, K, device=dev_)
    wsb = L.query("msc_wgrad_workspace_bytes", N, K, M)
    ws = torch.empty(wsb // 4, device=dev_)
    rt.call("msc_linear_wgrad", x, y1, dw1, None, ws, wsb, M, N, K, 0, rt.dt)
    rt.call("msc_linear_wgrad", x * 2, y1, dw2, None, ws, wsb, M, N, K, 0, rt.dt)
    torch.cuda.synchronize()
    assert torch.equal(dw2, 2 * dw1)
    # delta-kernel convolution == identity (64 x 56 x 56 x 64)
    Bc, H, C = 64, 56, 64
    xi = torch.randn(Bc, H, H, C, generator=g).to(torch.bfloat16).to(dev_)
    wd = torch.zeros(C, 3, 3, C)
    wd[torch.arange(C), 1, 1, torch.arange(C)] = 1.0
    yo = rt.empty(Bc, H, H, C)
    nrows = L.query("msc_conv2d_fwd_stat_rows", Bc * H * H, C, rt.dt)
    part = torch.empty(nrows * 2 * C, device=dev_)
    rt.call("msc_conv2d_fwd", xi, wd.to(torch.bfloat16).to(dev_), yo, part, Bc, H, H, C, C, 3, 3, 1, 1, None, 0, rt.dt)
    torch.cuda.synchronize()
    assert torch.equal(yo, xi)
    # the fused BN statistics of that launch are the column sums of the input
    st = part.view(nrows, 2, C)
    ref_sum = xi.float().sum((0, 1, 2))
    assert (st[:, 0].sum(0) - ref_sum).abs().max() <= 2e-3 * xi.float().abs().sum((0, 1, 2)).max()
    # channel: unit power at (practically) infinite SNR
    k = 256
    sym = torch.randn(B, k, generator=g).to(torch.bfloat16).to(dev_) * 3
    yc, sc = rt.empty(B, k), torch.empty(B, device=dev_)
    snr = torch.full((B,), 200.0, device=dev_)
    rt.call("msc_channel_fwd", sym, None, None, snr, yc, sc, B, k, L.CH_AWGN, 7, 0, None, rt.dt)
    torch.cuda.synchronize()
    pw = yc.float().pow(2).mean(1)
    assert (pw - 1).abs().max() < 1e-2
