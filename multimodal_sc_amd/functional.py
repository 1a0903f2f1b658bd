"""autograd.Function wrappers over the C-ABI kernels (include/msc.h).

Activations flow through torch.autograd (tensors of the runtime storage dtype); parameter gradients do NOT:
every backward writes dW/db straight into the flat fp32 gradient buffer through ParamRef.g, so autograd never
allocates or adds parameter gradients.  Residual-branch gradients are folded into the producing kernel
(conv dgrad `res`, LayerNorm backward `dadd`) instead of separate add kernels.
"""
from __future__ import annotations

import math
import os

import torch

from . import lib as L
from .runtime import ParamRef, Runtime

BN_EPS = 1e-5
LN_EPS = 1e-5


def _cdiv(a, b):
    return (a + b - 1) // b


# ------------------------------------------------------------------------------------------------ conv + BN (+ReLU)
def conv_bn_fwd(rt: Runtime, x, res, conv: ParamRef, gamma: ParamRef, beta: ParamRef, stride, pad, relu):
    B, H, W, Cin = x.shape
    Cout, KH, KW, _ = conv.shape
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    M = B * Ho * Wo
    nblk = L.query("msc_conv2d_fwd_stat_rows", M, Cout, rt.dt)
    y_raw = rt.empty(B, Ho, Wo, Cout)
    partial = rt.workspace("bn_partial", nblk * 2 * Cout * 4)
    wsb = L.query("msc_conv2d_splitk_workspace_bytes", M, Cout, KH * KW * Cin, rt.dt)     # > 0: chip-underfilling launch -> split-K
    ws = rt.workspace("conv_splitk", wsb) if wsb else None
    rt.call("msc_conv2d_fwd", x, conv.c, y_raw, partial, B, H, W, Cin, Cout, KH, KW, stride, pad, ws, wsb, rt.dt)
    stats = rt.f32(4, Cout)                      # mean, rstd (kept for backward), scale, shift (folded affine)
    mean, rstd = stats[0], stats[1]
    rt.call("msc_bn_finalize", partial, nblk, Cout, M, BN_EPS, gamma.p, beta.p, mean, rstd, stats[2], stats[3])
    y = rt.empty(B, Ho, Wo, Cout)
    rt.call("msc_bn_apply", y_raw, stats[2], stats[3], res, y, M, Cout, int(relu), rt.dt)
    if rt.taps is not None and relu:
        rt.taps[conv.name] = y
    return y, y_raw, stats


_STEM_WGRAD_MAIN = os.environ.get("MSC_STEM_WGRAD_MAIN", "1") != "0"
_MASK_FROM_X = os.environ.get("MSC_BN_MASK_FROM_X", "1") != "0"   # A/B knob: 0 = BatchNorm backward reads y for every ReLU mask
_FUSED_STEM = os.environ.get("MSC_FUSED_STEM", "1") != "0"    # A/B knob: 0 = separate bn_apply / max-pool kernels in the stem
_STEM_S2D = os.environ.get("MSC_STEM_S2D", "1") != "0"       # A/B knob: 0 = the stem convolution over NHWC with RGB padded to 8 channels
_FUSED_BIAS = os.environ.get("MSC_FUSED_BIAS", "1") != "0"    # A/B knob: 0 = bias gradients by separate column-sum launches
# fusion-layer GELU: fc1's forward epilogue stores gelu'(u) (it shares the erf evaluation's exponential) where it used to store u, and
# fc2's dgrad epilogue multiplies by it instead of evaluating erf + exp per element (A/B knob: 0 = store u, evaluate in the backward)
# 1 (opt-in): conv2's dgrad epilogue does the statistics pass of bn1's backward in the 8 residual blocks (msc_conv2d_dgrad_bnstat).  Built for
# the round-2 review (item 3a), correct — and +0.03 ms in the step (profiles/r3_bn_stat_fuse_ab.txt): the heavier epilogue costs the dgrad
# more than the 8 reduction launches it removes returned
_BN_STAT_FUSE = os.environ.get("MSC_BN_STAT_FUSE", "0") != "0"
_STEM_HOOK_AT = os.environ.get("MSC_STEM_HOOK_AT", "wgrad")       # experiment knob: "start" = before the stem's BatchNorm / pooling backward
_GELU_DSAVE = os.environ.get("MSC_GELU_DSAVE", "1") != "0"
_GELU_FWD, _GELU_BWD = (L.ACT_GELU_DSAVE, L.ACT_MUL_REF) if _GELU_DSAVE else (L.ACT_GELU, L.ACT_GELU)


def conv_bn_bwd(rt: Runtime, dy, x, y_raw, y, stats, conv: ParamRef, gamma: ParamRef, beta: ParamRef, stride, pad,
                relu, want_dres, need_dx, dx_res=None, had_res=False, dx_relu_ref=None, dy_partials=None, x_bn=None):
    """-> (dx or None, dres or None).  dx_res: extra gradient w.r.t. x added inside the dgrad epilogue.
    dx_relu_ref: x itself when x is the output of a ReLU whose backward belongs to the PRODUCER of x: applied in the dgrad epilogue.
    stats: [4,C] mean, rstd, scale, shift of the forward.  had_res: the forward added a residual before the ReLU (the mask
    then has to come from y; otherwise it is rebuilt from y_raw*scale+shift and y is not read).
    x_bn = (x_raw, x_stats): x is the output of a BatchNorm + ReLU (no residual) over x_raw — this layer's dgrad epilogue then also does
    the statistics pass of THAT BatchNorm's backward; the partial sums come back as a third result (partial, rows) for the caller to hand
    to the producer's conv_bn_bwd as dy_partials (which then runs no reduction of its own)."""
    B, H, W, Cin = x.shape
    Cout, KH, KW, _ = conv.shape
    M = y_raw.numel() // Cout
    d_raw = rt.empty_like(y_raw)
    dres = rt.empty_like(y_raw) if want_dres else None
    ws = rt.workspace("bn_bwd", L.query("msc_bn_bwd_workspace_bytes", Cout))
    xmask = relu and not had_res and _MASK_FROM_X
    if dy_partials is not None:
        assert xmask and not want_dres
        part, nblk = dy_partials
        rt.call("msc_bn_bwd_from_partials", dy, y_raw, stats[0], stats[1], gamma.p, stats[2], stats[3], part, nblk, d_raw, gamma.g, beta.g, ws,
                M, Cout, rt.accumulate, rt.dt)
    else:
        rt.call("msc_bn_bwd", dy, y_raw, y, stats[0], stats[1], gamma.p, stats[2] if xmask else None, stats[3] if xmask else None,
                d_raw, dres, gamma.g, beta.g, ws, M, Cout, int(relu), rt.accumulate, rt.dt)
    wsb = L.query("msc_wgrad_workspace_bytes", Cout, KH * KW * Cin, M)
    wsw = rt.workspace("wgrad", wsb, per_stream=False)     # only the side stream touches it
    def wgrad():
        # The stem (need_dx False) is the LAST node of the backward: nothing is left on the launching stream for its weight
        # gradient to hide behind, so it runs there, beside the side stream's remaining layer-1 weight gradients, instead
        # of queueing behind them (MSC_STEM_WGRAD_MAIN=0: old behaviour).
        if need_dx or not _STEM_WGRAD_MAIN:
            rt.call_side("msc_conv2d_wgrad", x, d_raw, conv.g, wsw, wsb, B, H, W, Cin, Cout, KH, KW, stride, pad, rt.accumulate, rt.dt)
        else:
            wsm = rt.workspace("wgrad_main", wsb)
            rt.call("msc_conv2d_wgrad", x, d_raw, conv.g, wsm, wsb, B, H, W, Cin, Cout, KH, KW, stride, pad, rt.accumulate, rt.dt)

    wgrad()
    dx = None
    if need_dx:
        dx = rt.empty_like(x)
        wsd = L.query("msc_conv2d_splitk_workspace_bytes", B * H * W, Cin, KH * KW * Cout, rt.dt) if stride == 1 else 0
        ws = rt.workspace("conv_splitk", wsd) if wsd else None
        if x_bn is not None:
            assert dx_relu_ref is None and stride == 1
            nblk = L.query("msc_conv2d_dgrad_stat_rows", B * H * W, Cin, rt.dt)
            part = rt.f32(nblk * 2 * Cin)
            rt.call("msc_conv2d_dgrad_bnstat", d_raw, conv.t, dx_res, dx, x_bn[0], x_bn[1], part, B, H, W, Cin, Cout, KH, KW, stride, pad, ws, wsd, rt.dt)
            return dx, dres, (part, nblk)
        rt.call("msc_conv2d_dgrad", d_raw, conv.t, dx_res, dx_relu_ref, dx, B, H, W, Cin, Cout, KH, KW, stride, pad, ws, wsd, rt.dt)
    return (dx, dres, None) if x_bn is not None else (dx, dres)


class ConvBNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, rt, conv, gamma, beta, stride, pad, relu, need_dx):
        # anchor: any tensor with requires_grad=True; keeps this node in the graph when x needs no gradient
        y, y_raw, stats = conv_bn_fwd(rt, x, None, conv, gamma, beta, stride, pad, relu)
        ctx.save_for_backward(x, y_raw, y, stats)
        ctx.cfg = (rt, conv, gamma, beta, stride, pad, relu, need_dx)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y_raw, y, stats = ctx.saved_tensors
        rt, conv, gamma, beta, stride, pad, relu, need_dx = ctx.cfg
        dx, _ = conv_bn_bwd(rt, dy.contiguous(), x, y_raw, y, stats, conv, gamma, beta, stride, pad, relu, False, need_dx)
        return (dx,) + (None,) * 9


class StemFn(torch.autograd.Function):
    """ImageEncoder stem as ONE autograd node: conv (implicit GEMM + fused BN statistics) -> BN finalize -> BN + ReLU + max-pool
    in one pass (msc_bn_relu_pool_fwd).  The post-activation map (103 MB at 64 frames) is never written; the backward gathers
    the pooled gradient through the arg-max ids and rebuilds the ReLU mask from the conv output (msc_bn_relu_pool_bwd), then
    runs the weight gradient.  The stem's input is the image: no data gradient."""

    @staticmethod
    def forward(ctx, x, anchor, rt, conv, gamma, beta, stride, pad, pK, pS, pP, w_s2d=None):
        # w_s2d given: x is the image in space-to-depth form [B,H/2,W/2,16] and the 7x7/2 convolution runs as a 4x4/1 one over it
        # (msc_stem_s2d_conv_*: K = 256 instead of 392, same sums); else x is NHWC with RGB padded to 8 channels
        B, H, W, Cin = x.shape
        Cout, KH, KW, _ = conv.shape
        Ho, Wo = (H, W) if w_s2d is not None else ((H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1)
        M = B * Ho * Wo
        nblk = L.query("msc_conv2d_fwd_stat_rows", M, Cout, rt.dt)
        y_raw = rt.empty(B, Ho, Wo, Cout)
        partial = rt.workspace("bn_partial", nblk * 2 * Cout * 4)
        if w_s2d is not None:
            rt.call("msc_stem_s2d_conv_fwd", x, w_s2d, y_raw, partial, B, H, W, Cout, rt.dt)
        else:
            wsb = L.query("msc_conv2d_splitk_workspace_bytes", M, Cout, KH * KW * Cin, rt.dt)
            ws = rt.workspace("conv_splitk", wsb) if wsb else None
            rt.call("msc_conv2d_fwd", x, conv.c, y_raw, partial, B, H, W, Cin, Cout, KH, KW, stride, pad, ws, wsb, rt.dt)
        stats = rt.f32(4, Cout)
        rt.call("msc_bn_finalize", partial, nblk, Cout, M, BN_EPS, gamma.p, beta.p, stats[0], stats[1], stats[2], stats[3])
        Hp, Wp = (Ho + 2 * pP - pK) // pS + 1, (Wo + 2 * pP - pK) // pS + 1
        y = rt.empty(B, Hp, Wp, Cout)
        idx = rt.alloc((B, Hp, Wp, Cout), torch.uint8)
        xsel = rt.empty(B, Hp, Wp, Cout)      # conv output at each window's arg-max: the backward's statistics pass reads this, not y_raw
        rt.call("msc_bn_relu_pool_fwd", y_raw, stats[2], stats[3], y, idx, xsel, B, Ho, Wo, Cout, pK, pS, pP, rt.dt)
        if rt.taps is not None:   # parity tests compare ReLU masks: materialise the activation the fused pass never stores
            act = rt.empty(B, Ho, Wo, Cout)
            rt.call("msc_bn_apply", y_raw, stats[2], stats[3], None, act, M, Cout, 1, rt.dt)
            rt.taps[conv.name] = act
        ctx.save_for_backward(x, y_raw, stats, idx, xsel)
        ctx.cfg = (rt, conv, gamma, beta, stride, pad, pK, pS, pP, w_s2d is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y_raw, stats, idx, xsel = ctx.saved_tensors
        rt, conv, gamma, beta, stride, pad, pK, pS, pP, s2d = ctx.cfg
        if rt.on_stem_backward is not None and _STEM_HOOK_AT == "start":
            rt.on_stem_backward()
        B, H, W, Cin = x.shape
        Cout, KH, KW, _ = conv.shape
        _, Ho, Wo, _ = y_raw.shape
        d_raw = rt.empty_like(y_raw)
        ws = rt.workspace("bn_bwd", L.query("msc_bn_bwd_workspace_bytes", Cout))
        rt.call("msc_bn_relu_pool_bwd", dy.contiguous(), idx, xsel, y_raw, stats[0], stats[1], gamma.p, stats[2], stats[3], d_raw, gamma.g, beta.g,
                ws, B, Ho, Wo, Cout, pK, pS, pP, rt.accumulate, rt.dt)
        # last node of the backward: every other encoder parameter's gradient has its producer enqueued.  The Trainer puts their optimizer update
        # onto the side stream here — behind the BatchNorm / pooling backward above (two memory-bound passes that would share HBM with it), beside
        # the stem's weight gradient below
        if rt.on_stem_backward is not None and _STEM_HOOK_AT != "start":
            rt.on_stem_backward()
        if s2d:
            # (the stem is the last node of the backward: its weight gradient runs on the launching stream, see conv_bn_bwd)
            wsb = L.query("msc_wgrad_workspace_bytes", Cout, 256, B * Ho * Wo)
            dws = rt.f32(Cout * 256)
            rt.call("msc_stem_s2d_conv_wgrad", x, d_raw, dws, rt.workspace("wgrad_main", wsb), wsb, B, H, W, Cout, 0, rt.dt)
            rt.call("msc_stem_wgrad_unpack", dws, conv.g, Cout, rt.accumulate)
            return (None,) * 12
        wsb = L.query("msc_wgrad_workspace_bytes", Cout, KH * KW * Cin, B * Ho * Wo)
        if _STEM_WGRAD_MAIN:     # last node of the backward: nothing left on the launching stream to hide behind (see conv_bn_bwd)
            rt.call("msc_conv2d_wgrad", x, d_raw, conv.g, rt.workspace("wgrad_main", wsb), wsb, B, H, W, Cin, Cout, KH, KW, stride, pad,
                    rt.accumulate, rt.dt)
        else:
            rt.call_side("msc_conv2d_wgrad", x, d_raw, conv.g, rt.workspace("wgrad", wsb, per_stream=False), wsb, B, H, W, Cin, Cout, KH, KW,
                         stride, pad, rt.accumulate, rt.dt)
        return (None,) * 12


class BasicBlockFn(torch.autograd.Function):
    """ResNet BasicBlock as ONE autograd node: conv-bn-relu, conv-bn, (+1x1 conv-bn identity), add, relu."""

    @staticmethod
    def forward(ctx, x, rt, p, stride, mask_in=False, dy_masked=False):
        # p: dict of ParamRefs: conv1,bn1g,bn1b,conv2,bn2g,bn2b[,dconv,dbng,dbnb]
        # mask_in: x is the previous block's output (post-ReLU) and that block expects an already masked gradient: this block's
        #   dgrad epilogue applies relu'(x).  dy_masked: the next block does the same for THIS block's output, so the gradient that
        #   arrives is dz = dy * [y > 0] already: bn2's backward needs no mask (y is not read) and the identity branch's gradient
        #   is the incoming tensor itself (no masked copy is written)
        o1, o1_raw, st1 = conv_bn_fwd(rt, x, None, p["conv1"], p["bn1g"], p["bn1b"], stride, 1, True)
        has_ds = "dconv" in p
        if has_ds:
            idt, idt_raw, std = conv_bn_fwd(rt, x, None, p["dconv"], p["dbng"], p["dbnb"], stride, 0, False)
        else:
            idt = x
        y, o2_raw, st2 = conv_bn_fwd(rt, o1, idt, p["conv2"], p["bn2g"], p["bn2b"], 1, 1, True)
        saved = [x, o1_raw, o1, st1, o2_raw, y, st2]
        if has_ds:
            saved += [idt_raw, std]
        ctx.save_for_backward(*saved)
        ctx.cfg = (rt, p, stride, has_ds, mask_in, dy_masked)
        return y

    @staticmethod
    def backward(ctx, dy):
        rt, p, stride, has_ds, mask_in, dy_masked = ctx.cfg
        sv = ctx.saved_tensors
        x, o1_raw, o1, st1, o2_raw, y, st2 = sv[:7]
        dy = dy.contiguous()
        # conv2's dgrad produces dy of bn1 (+ ReLU, no residual): its epilogue does bn1's backward statistics pass (MSC_BN_STAT_FUSE=0: a
        # reduction kernel of bn1's own re-reads do1 and o1_raw)
        fuse = (o1_raw, st1) if (_BN_STAT_FUSE and _MASK_FROM_X) else None
        part1 = None
        if dy_masked:
            # the consumer's dgrad epilogue already applied this block's output ReLU: BatchNorm backward without a mask, identity = dy
            r = conv_bn_bwd(rt, dy, o1, o2_raw, None, st2, p["conv2"], p["bn2g"], p["bn2b"], 1, 1, False, False, True, x_bn=fuse)
            do1, didt = r[0], dy
        else:
            # conv2/bn2: dres = gradient of the identity branch (post-ReLU mask, which needs y: the residual was added before the ReLU)
            r = conv_bn_bwd(rt, dy, o1, o2_raw, y, st2, p["conv2"], p["bn2g"], p["bn2b"], 1, 1, True, True, True, had_res=True, x_bn=fuse)
            do1, didt = r[0], r[1]
        if fuse is not None:
            part1 = r[2]
        if has_ds:
            idt_raw, std = sv[7:]
            dx_id, _ = conv_bn_bwd(rt, didt, x, idt_raw, None, std, p["dconv"], p["dbng"], p["dbnb"], stride, 0, False, False, True)
        else:
            dx_id = didt
        dx, _ = conv_bn_bwd(rt, do1, x, o1_raw, o1, st1, p["conv1"], p["bn1g"], p["bn1b"], stride, 1, True, False, True, dx_res=dx_id,
                            dx_relu_ref=x if mask_in else None, dy_partials=part1)
        return dx, None, None, None, None, None


class MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rt, K, S, P):
        B, H, W, C = x.shape
        Ho, Wo = (H + 2 * P - K) // S + 1, (W + 2 * P - K) // S + 1
        y = rt.empty(B, Ho, Wo, C)
        idx = rt.alloc((B, Ho, Wo, C), torch.uint8)
        rt.call("msc_maxpool_fwd", x, y, idx, B, H, W, C, K, S, P, rt.dt)
        ctx.save_for_backward(idx)
        ctx.cfg = (rt, x.shape, K, S, P)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        rt, shp, K, S, P = ctx.cfg
        B, H, W, C = shp
        dx = rt.empty(*shp)
        rt.call("msc_maxpool_bwd", dy.contiguous(), idx, dx, B, H, W, C, K, S, P, rt.dt)
        return dx, None, None, None, None


# ------------------------------------------------------------------------------------------------ dense layers
def linear_fwd(rt: Runtime, x2d, w: ParamRef, b: ParamRef, act=L.ACT_NONE, res=None, want_pre=False):
    M, K = x2d.shape
    N = w.shape[0]
    y = rt.empty(M, N)
    pre = rt.empty(M, N) if want_pre else None
    rt.call("msc_linear_fwd", x2d, w.c, b.p if b is not None else None, res, y, pre, M, N, K, act, rt.dt)
    return y, pre


def linear_bwd(rt: Runtime, dy2d, x2d, ref, w: ParamRef, b: ParamRef, act, need_dx=True, bias_done=False,
               prev_act=L.ACT_NONE, prev_ref=None, prev_bias: ParamRef = None, res=None):
    """dy2d: gradient w.r.t. the layer output (after activation). ref: y (relu/tanh/sigmoid) or pre (gelu).
    prev_act/prev_ref/prev_bias: fuse the backward of the activation that FED this layer (and the bias gradient of the
    layer that produced it) into the dgrad epilogue — the returned dx is then the gradient w.r.t. that pre-activation
    and prev_bias.g is filled; the caller passes act=ACT_NONE, bias_done=True to the previous layer's linear_bwd.
    res: a gradient w.r.t. the layer INPUT arriving over another branch, added inside the dgrad epilogue (before prev_act')."""
    M, N = dy2d.shape
    K = x2d.shape[1]
    if act != L.ACT_NONE:
        d = rt.empty_like(dy2d)
        rt.call("msc_act_bwd", dy2d, ref, d, M * N, act, rt.dt)
    else:
        d = dy2d
    def side_work():
        # weight gradient and (same launch: the row sums of dy ride on one extra MFMA per fragment) bias gradient
        wsb = L.query("msc_wgrad_workspace_bytes", N, K, M)
        wsw = rt.workspace("wgrad", wsb, per_stream=False)     # only the side stream touches it
        db = b.g if (b is not None and not bias_done) else None
        if db is not None and not _FUSED_BIAS:
            ws = rt.workspace("colsum", L.query("msc_colsum_workspace_bytes", M, N), per_stream=False)
            rt.call_side("msc_colsum", d, db, ws, M, N, rt.accumulate, rt.dt)
            db = None
        rt.call_side("msc_linear_wgrad", x2d, d, w.g, db, wsw, wsb, M, N, K, rt.accumulate, rt.dt)

    # (enqueueing the dgrad BEFORE the side-stream launches was measured: 0.5 % slower — the side stream then also
    # waits for the dgrad)
    side_work()
    dx = None
    if need_dx:
        dx = rt.empty(M, K)
        part = None
        if prev_bias is not None:
            nblk = L.query("msc_linear_dgrad_colsum_rows", M, K, rt.dt)
            part = rt.workspace("dgrad_colsum", nblk * K * 4)
        rt.call("msc_linear_dgrad", d, w.t, res, prev_ref, prev_act, part, dx, M, N, K, rt.dt)
        if prev_bias is not None:
            rt.call("msc_colsum_finalize", part, nblk, K, prev_bias.g, rt.accumulate)
    return dx


class Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, rt, w, b, act, need_dx):
        x2d = x.reshape(-1, x.shape[-1])
        y, pre = linear_fwd(rt, x2d, w, b, act, None, want_pre=(act == L.ACT_GELU))
        ctx.save_for_backward(x2d, pre if act == L.ACT_GELU else (y if act != L.ACT_NONE else None))
        ctx.cfg = (rt, w, b, act, need_dx, x.shape)
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2d, ref = ctx.saved_tensors
        rt, w, b, act, need_dx, shp = ctx.cfg
        dx = linear_bwd(rt, dy.reshape(-1, dy.shape[-1]).contiguous(), x2d, ref, w, b, act, need_dx)
        return (dx.view(shp) if dx is not None else None), None, None, None, None, None, None


class MLP2(torch.autograd.Function):
    """Two dense layers with an activation in between as ONE autograd node: y = act2(W2 act1(W1 x + b1) + b2).
    The backward of act1 rides in the epilogue of layer 2's dgrad GEMM (prev_act), so the pair costs no activation-backward
    launch; used for the codec / head pairs, whose 64-row kernels are launch-latency-bound."""

    @staticmethod
    def forward(ctx, x, rt, w1, b1, act1, w2, b2, act2):
        x2d = x.reshape(-1, x.shape[-1])
        h, hpre = linear_fwd(rt, x2d, w1, b1, act1, None, want_pre=(act1 == L.ACT_GELU))
        y, ypre = linear_fwd(rt, h, w2, b2, act2, None, want_pre=(act2 == L.ACT_GELU))
        ctx.save_for_backward(x2d, h, hpre if act1 == L.ACT_GELU else None, ypre if act2 == L.ACT_GELU else (y if act2 != L.ACT_NONE else None))
        ctx.cfg = (rt, w1, b1, act1, w2, b2, act2, x.shape)
        return y.view(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2d, h, hpre, yref = ctx.saved_tensors
        rt, w1, b1, act1, w2, b2, act2, shp = ctx.cfg
        dy2d = dy.reshape(-1, dy.shape[-1]).contiguous()
        ref1 = hpre if act1 == L.ACT_GELU else h
        dh = linear_bwd(rt, dy2d, h, yref, w2, b2, act2, prev_act=act1, prev_ref=ref1 if act1 != L.ACT_NONE else None)
        dx = linear_bwd(rt, dh, x2d, None, w1, b1, L.ACT_NONE)
        return dx.view(shp), None, None, None, None, None, None, None


class HeadsFn(torch.autograd.Function):
    """Everything that reads the decoded feature z, as ONE autograd node with two outputs: the policy/value head
    (fc + tanh -> packed [logits | value] rows) and the reconstruction head (fc1 + ReLU -> per-cell fc2 + sigmoid).  z has two
    consumers; as separate nodes autograd would sum their two gradients with an add kernel of its own.  Here the head branch's
    gradient rides into the reconstruction branch's dgrad epilogue as `res` (one fp32 sum, one rounding, no extra launch), and the
    two activation backwards in front of the second layers ride in those layers' dgrad epilogues (prev_act)."""

    @staticmethod
    def forward(ctx, z, rt, hw1, hb1, hw2, hb2, rw1, rb1, rw2, rb2, cells, recon_hidden):
        z2d = z.reshape(-1, z.shape[-1])
        B = z2d.shape[0]
        h, _ = linear_fwd(rt, z2d, hw1, hb1, L.ACT_TANH)
        head, _ = linear_fwd(rt, h, hw2, hb2, L.ACT_NONE)
        r1, _ = linear_fwd(rt, z2d, rw1, rb1, L.ACT_RELU)
        r1v = r1.view(B * cells, recon_hidden)
        patches, _ = linear_fwd(rt, r1v, rw2, rb2, L.ACT_SIGMOID)
        ctx.save_for_backward(z2d, h, r1v, patches)
        ctx.cfg = (rt, hw1, hb1, hw2, hb2, rw1, rb1, rw2, rb2, z.shape)
        ctx.set_materialize_grads(False)
        return head, patches

    @staticmethod
    def backward(ctx, dhead, dpat):
        z2d, h, r1v, patches = ctx.saved_tensors
        rt, hw1, hb1, hw2, hb2, rw1, rb1, rw2, rb2, shp = ctx.cfg
        B = z2d.shape[0]
        dz = None
        if dhead is not None:
            dh = linear_bwd(rt, dhead.contiguous(), h, None, hw2, hb2, L.ACT_NONE, prev_act=L.ACT_TANH, prev_ref=h)
            dz = linear_bwd(rt, dh, z2d, None, hw1, hb1, L.ACT_NONE, need_dx=True)
        if dpat is not None:
            dr1 = linear_bwd(rt, dpat.contiguous(), r1v, patches, rw2, rb2, L.ACT_SIGMOID, prev_act=L.ACT_RELU, prev_ref=r1v)
            dz = linear_bwd(rt, dr1.view(B, -1), z2d, None, rw1, rb1, L.ACT_NONE, res=dz)
        return (dz.view(shp) if dz is not None else None,) + (None,) * 11


class LinearFP8(torch.autograd.Function):
    """Semantic encoder / decoder GEMM on OCP e4m3 operands (BASELINE.json configs[4]); oracle: oracle.model.FP8Linear.
    Forward: per-tensor quantise x (activations) and W (fp32 master), contraction on the fp8 MFMA, fused bias + activation.
    Backward: straight-through on the de-quantised operands x^, W^ through the ordinary kernels
    (dx = d W^ via the mn-major-B contraction — no transposed copy of W^ is made; dW = d^T x^)."""

    @staticmethod
    def forward(ctx, x, rt, w, b, act, need_dx):
        x2d = x.reshape(-1, x.shape[-1])
        M, K = x2d.shape
        N = w.shape[0]
        dev = rt.device
        xq = rt.alloc((M, K), torch.uint8)
        wq = rt.alloc((N, K), torch.uint8)
        xh, wh = rt.empty(M, K), rt.empty(N, K)
        sc = rt.f32(2)
        rt.call("msc_fp8_quantize", x2d, xq, xh, sc[0:1], M * K, rt.dt, rt.dt)
        rt.call("msc_fp8_quantize", w.p, wq, wh, sc[1:2], N * K, L.MSC_F32, rt.dt)
        y = rt.empty(M, N)
        rt.call("msc_linear_fp8_fwd", xq, wq, sc[0:1], sc[1:2], b.p if b is not None else None, y, M, N, K, act, rt.dt)
        ctx.save_for_backward(xh, wh, y if act != L.ACT_NONE else None)
        ctx.cfg = (rt, w, b, act, need_dx, x.shape)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        xh, wh, ref = ctx.saved_tensors
        rt, w, b, act, need_dx, shp = ctx.cfg
        dy2d = dy.reshape(-1, dy.shape[-1]).contiguous()
        M, N = dy2d.shape
        K = xh.shape[1]
        if act != L.ACT_NONE:
            d = rt.empty_like(dy2d)
            rt.call("msc_act_bwd", dy2d, ref, d, M * N, act, rt.dt)
        else:
            d = dy2d
        wsb = L.query("msc_wgrad_workspace_bytes", N, K, M)
        wsw = rt.workspace("wgrad", wsb, per_stream=False)
        rt.call_side("msc_linear_wgrad", xh, d, w.g, b.g if b is not None else None, wsw, wsb, M, N, K, rt.accumulate, rt.dt)
        dx = None
        if need_dx:
            dx = rt.empty(M, K)
            # dx[M][K] = d[M][N] . W^[N][K]: A k-major (reduction over N), B stored [reduction][column] = mn-major
            rt.call("msc_gemm_batched", d, wh, dx, M, K, N, N, K, K, 1, 0, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, 0, rt.dt)
        return (dx.view(shp) if dx is not None else None), None, None, None, None, None


def layernorm_fwd(rt, x2d, g: ParamRef, b: ParamRef):
    R, D = x2d.shape
    y = rt.empty_like(x2d)
    mean, rstd = rt.f32(R), rt.f32(R)
    rt.call("msc_layernorm_fwd", x2d, g.p, b.p, y, mean, rstd, R, D, LN_EPS, rt.dt)
    return y, mean, rstd


def layernorm_bwd(rt, dy2d, x2d, mean, rstd, g: ParamRef, b: ParamRef, dadd=None):
    R, D = x2d.shape
    dx = rt.empty_like(x2d)
    # the per-row-block partials of dgamma / dbeta get a buffer of their own (not a shared workspace): their reduction runs on the
    # side stream, off the critical chain — the parameter gradients are needed by the optimizer only
    part = rt.f32(L.query("msc_layernorm_bwd_workspace_bytes", R, D) // 4)
    rt.call("msc_layernorm_bwd", dy2d, x2d, mean, rstd, g.p, dadd, dx, None, None, part, R, D, rt.accumulate, rt.dt)
    rt.call_side("msc_layernorm_bwd_params", part, R, None, 0, D, g.g, b.g, rt.accumulate)
    return dx


class LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rt, g, b):
        x2d = x.reshape(-1, x.shape[-1])
        y, mean, rstd = layernorm_fwd(rt, x2d, g, b)
        ctx.save_for_backward(x2d, mean, rstd)
        ctx.cfg = (rt, g, b, x.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2d, mean, rstd = ctx.saved_tensors
        rt, g, b, shp = ctx.cfg
        dx = layernorm_bwd(rt, dy.reshape(-1, dy.shape[-1]).contiguous(), x2d, mean, rstd, g, b)
        return dx.view(shp), None, None, None


# ------------------------------------------------------------------------------------------------ attention
def fused_attention_ok(rt: Runtime, T, hd):
    """the flash-style kernel covers bf16, head_dim 64, T <= 512; the exact-fp32 parity mode takes the GEMM+softmax path"""
    return rt.dt == L.MSC_BF16 and hd == 64 and T <= 512 and rt.fused_attention


def attention_fwd(rt: Runtime, qkv, B, T, h, hd):
    """qkv: [B*T, 3*h*hd] (q|k|v, heads inside). -> o [B*T, h*hd], aux (lse [B*h,Tpad] fused | P [B*h,T,Tp] unfused)"""
    D = h * hd
    if fused_attention_ok(rt, T, hd):
        o = rt.empty(B * T, D)
        lse = rt.f32(B * h, L.query("msc_attention_tpad", T))
        rt.call("msc_attention_fwd", qkv, o, lse, B, T, h, hd, 1.0 / math.sqrt(hd), rt.dt)
        return o, lse
    Tp = _cdiv(T, 64) * 64
    S = rt.workspace("attn_scores", B * h * T * Tp * 4)
    q, k, v = qkv, qkv[:, D:], qkv[:, 2 * D:]
    ld = 3 * D
    scale = 1.0 / math.sqrt(hd)
    rt.call("msc_gemm_batched", q, k, S, T, T, hd, ld, ld, Tp, 1, 1, B * h, h, T * ld, hd, T * ld, hd, h * T * Tp, T * Tp,
            scale, 1, rt.dt)
    P = rt.empty(B * h, T, Tp)
    rt.call("msc_softmax_fwd", S, P, B * h * T, T, Tp, rt.dt)
    o = rt.empty(B * T, D)
    rt.call("msc_gemm_batched", P, v, o, T, hd, T, Tp, ld, D, 1, 0, B * h, h, h * T * Tp, T * Tp, T * ld, hd, T * D, hd,
            1.0, 0, rt.dt)
    return o, P


def attention_bwd(rt: Runtime, do, qkv, o, P, B, T, h, hd):
    D = h * hd
    if fused_attention_ok(rt, T, hd):
        lse = P
        dqkv = rt.empty(B * T, 3 * D)
        delta = rt.workspace("attn_delta", 2 * lse.numel() * 4)       # [2][B*h][Tpad]: -delta and -lse/c (see msc.h)
        rt.call("msc_attention_bwd", qkv, o, do, lse, delta, dqkv, B, T, h, hd, 1.0 / math.sqrt(hd), rt.dt)
        return dqkv
    Tp = P.shape[-1]
    ld = 3 * D
    q, k, v = qkv, qkv[:, D:], qkv[:, 2 * D:]
    scale = 1.0 / math.sqrt(hd)
    dqkv = rt.empty(B * T, 3 * D)
    dq, dk, dv = dqkv, dqkv[:, D:], dqkv[:, 2 * D:]
    nb = B * h
    # dV[key][d] = sum_q P[q][key] dO[q][d]
    rt.call("msc_gemm_batched", P, do, dv, T, hd, T, Tp, D, ld, 0, 0, nb, h, h * T * Tp, T * Tp, T * D, hd, T * ld, hd, 1.0, 0, rt.dt)
    # dP[q][key] = sum_d dO[q][d] V[key][d]
    dP = rt.workspace("attn_scores", nb * T * Tp * 4)
    rt.call("msc_gemm_batched", do, v, dP, T, T, hd, D, ld, Tp, 1, 1, nb, h, T * D, hd, T * ld, hd, h * T * Tp, T * Tp, 1.0, 1, rt.dt)
    dS = rt.empty_like(P)
    rt.call("msc_softmax_bwd", dP, P, dS, nb * T, T, Tp, rt.dt)
    # dQ[q][d] = scale * sum_key dS[q][key] K[key][d]
    rt.call("msc_gemm_batched", dS, k, dq, T, hd, T, Tp, ld, ld, 1, 0, nb, h, h * T * Tp, T * Tp, T * ld, hd, T * ld, hd, scale, 0, rt.dt)
    # dK[key][d] = scale * sum_q dS[q][key] Q[q][d]
    rt.call("msc_gemm_batched", dS, q, dk, T, hd, T, Tp, ld, ld, 0, 0, nb, h, h * T * Tp, T * Tp, T * ld, hd, T * ld, hd, scale, 0, rt.dt)
    return dqkv


# MSC_FUSION_CHAINS: 1 = the whole batch as one chain of launches, 2 / force = two half-batch chains.  Default (round 4): ONE chain while
# the dense linears run on the persistent ping-pong kernel (knob gemm_pp): that kernel owns a CU (136-160 KB of LDS) and overlaps its own
# load / MFMA / store phases, so two concurrent half-size launches only take turns (same box: 7.71 vs 7.52 ms); two chains with the
# tiled kernel (gemm_pp = 0), whose lock-step phases the second chain interleaves (round 2: 8.10 -> 7.95 ms).
_FUSION_CHAINS = (os.environ["MSC_FUSION_CHAINS"] != "1") if "MSC_FUSION_CHAINS" in os.environ else (L.lib.msc_get_knob(b"gemm_pp") == 0)
_CHAIN_MIN_ROWS = 0 if os.environ.get("MSC_FUSION_CHAINS") == "force" else 11264   # token rows (B*T) from which two chains pay


def _two_chains_ok(rt: Runtime, B, T, hd):
    """The fusion layers run the two halves of the batch as two chains of launches on two streams (the BEV-branch stream is idle
    during the fusion).  Every kernel of a layer is per-sample, so nothing changes numerically; the workgroups of ONE launch run
    their load / MFMA / store phases in lock-step, two concurrent half-size launches interleave them (tools/two_chain_probe.py:
    the four forward linears of a layer 255 -> 207 us).  Weight gradients still run ONCE over the whole batch (the halves are
    row ranges of the same buffers)."""
    # below ~36 frames of 306 tokens the half-size kernels are too short to pay for twice the launches (same box, one chain vs two:
    # 16 frames 4.07 vs 6.03 ms, 32: 5.44 vs 5.58, 40: 6.17 vs 6.11, 48: 6.78 vs 6.67, 64: 8.10 vs 7.95); tests force it with MSC_FUSION_CHAINS=force
    big = B * T >= _CHAIN_MIN_ROWS
    return _FUSION_CHAINS and _FUSED_BIAS and B >= 2 and big and rt.branch_streams and rt.side_active() and fused_attention_ok(rt, T, hd)


def _fusion_fwd_two_chains(ctx, x, rt, p, h):
    B, T, D = x.shape
    hd, M, Fh = D // h, B * T, p["fc1w"].shape[0]
    main = rt.stream
    s2, ev_ab, ev_ba, _ = rt.chain_b()
    x2d = x.reshape(M, D)
    h1, qkv, o, x1, h2 = rt.empty(M, D), rt.empty(M, 3 * D), rt.empty(M, D), rt.empty(M, D), rt.empty(M, D)
    upre, u, x2 = rt.empty(M, Fh), rt.empty(M, Fh), rt.empty(M, D)
    st = rt.f32(4, M)                                              # mu1, rs1, mu2, rs2
    Tpad = L.query("msc_attention_tpad", T)
    lse = rt.f32(B * h, Tpad)
    scale = 1.0 / math.sqrt(hd)
    saving = bool(ctx.needs_input_grad[0])      # (torch.is_grad_enabled() is always False inside Function.forward)
    if p.get("first", True) or not saving:
        # first layer: the tokens were assembled on the main stream.  Inference / rollout (nothing saved): EVERY layer orders chain B
        # behind the main stream, because this layer's buffers come from the caching allocator and may be blocks the main stream
        # freed a layer ago — chain B must not write them while chain A of the previous layer still reads them
        L.call("msc_stream_wait_stream", s2, main, ev_ab)
    # Raw device addresses of the chains' row ranges (plain integer arithmetic: slicing ~60 tensor views per layer costs the host
    # more than the launches themselves, and the host paces the launch-dense fusion backward)
    es = x2d.element_size()
    a_x, a_h1, a_qkv, a_o, a_x1, a_h2 = x2d.data_ptr(), h1.data_ptr(), qkv.data_ptr(), o.data_ptr(), x1.data_ptr(), h2.data_ptr()
    a_upre, a_u, a_x2, a_st, a_lse = upre.data_ptr(), u.data_ptr(), x2.data_ptr(), st.data_ptr(), lse.data_ptr()
    ln1g, ln1b, ln2g, ln2b = p["ln1g"].p.data_ptr(), p["ln1b"].p.data_ptr(), p["ln2g"].p.data_ptr(), p["ln2b"].p.data_ptr()
    wq, bq, wp, bp = p["qkvw"].c.data_ptr(), p["qkvb"].p.data_ptr(), p["projw"].c.data_ptr(), p["projb"].p.data_ptr()
    w1, b1, w2, b2 = p["fc1w"].c.data_ptr(), p["fc1b"].p.data_ptr(), p["fc2w"].c.data_ptr(), p["fc2b"].p.data_ptr()
    Bh = B // 2
    # (stream, first row, rows, first (frame, head) row of lse, frames)
    chains = ((main, 0, Bh * T, 0, Bh), (s2, Bh * T, (B - Bh) * T, Bh * h, B - Bh))
    call, dt = L.call, rt.dt
    rD, r3D, rF = D * es, 3 * D * es, Fh * es
    # the two chains' launches are enqueued alternately, kernel by kernel
    for c, r0, n, _, _ in chains:
        call("msc_layernorm_fwd", a_x + r0 * rD, ln1g, ln1b, a_h1 + r0 * rD, a_st + r0 * 4, a_st + (M + r0) * 4, n, D, LN_EPS, dt, c)
    for c, r0, n, _, _ in chains:
        call("msc_linear_fwd", a_h1 + r0 * rD, wq, bq, None, a_qkv + r0 * r3D, None, n, 3 * D, D, L.ACT_NONE, dt, c)
    for c, r0, n, l0, nb in chains:
        call("msc_attention_fwd", a_qkv + r0 * r3D, a_o + r0 * rD, a_lse + l0 * Tpad * 4, nb, T, h, hd, scale, dt, c)
    for c, r0, n, _, _ in chains:
        call("msc_linear_fwd", a_o + r0 * rD, wp, bp, a_x + r0 * rD, a_x1 + r0 * rD, None, n, D, D, L.ACT_NONE, dt, c)
    for c, r0, n, _, _ in chains:
        call("msc_layernorm_fwd", a_x1 + r0 * rD, ln2g, ln2b, a_h2 + r0 * rD, a_st + (2 * M + r0) * 4, a_st + (3 * M + r0) * 4, n, D, LN_EPS, dt, c)
    for c, r0, n, _, _ in chains:
        call("msc_linear_fwd", a_h2 + r0 * rD, w1, b1, None, a_u + r0 * rF, a_upre + r0 * rF, n, Fh, D, _GELU_FWD, dt, c)
    for c, r0, n, _, _ in chains:
        call("msc_linear_fwd", a_u + r0 * rF, w2, b2, a_x1 + r0 * rD, a_x2 + r0 * rD, None, n, D, Fh, L.ACT_NONE, dt, c)
    if p.get("last", True):                                       # the final LayerNorm reads the whole batch on the main stream
        L.call("msc_stream_wait_stream", main, s2, ev_ba)
    if not saving:
        # inference / rollout collection: nothing keeps these buffers alive for a backward, and the caching allocator would hand
        # them back to the main stream while the second chain still uses them
        s2t = rt.branch_stream()
        for t in (x2d, st, h1, qkv, lse, o, x1, h2, upre, u, x2):
            t.record_stream(s2t)
    ctx.save_for_backward(x2d, st, h1, qkv, lse, o, x1, h2, upre, u)
    ctx.cfg = (rt, p, h, (B, T, D))
    ctx.two = True
    ctx.side_was_active = rt.side_active()      # the backward's weight gradients must take the same route (asserted there)
    return x2.view(B, T, D)


def _fusion_bwd_two_chains(ctx, dy):
    rt, p, h, (B, T, D) = ctx.cfg
    hd, M, Fh = D // h, B * T, p["fc1w"].shape[0]
    x2d, st, h1, qkv, lse, o, x1, h2, upre, u = ctx.saved_tensors
    if not (rt.side_active() and ctx.side_was_active):
        # call_side() would fall back to the launching stream and run the whole-batch weight gradient without waiting for chain B
        raise RuntimeError("the two-chain fusion backward needs the side stream it was recorded with (rt.profile / MSC_SIDE_STREAM "
                           "changed between forward and backward)")
    main = rt.stream
    s2, ev_ab, ev_ba, ev_bs = rt.chain_b()
    dx2 = dy.reshape(M, D).contiguous()
    dpre, dh2, dx1, do = rt.empty(M, Fh), rt.empty(M, D), rt.empty(M, D), rt.empty(M, D)
    dqkv, dh1, dx = rt.empty(M, 3 * D), rt.empty(M, D), rt.empty(M, D)
    delta = rt.f32(2 * lse.numel())                              # per chain [2][frames*h][Tpad]: -delta | -lse/c
    Tpad = lse.shape[1]
    # per-row-block partials of the LayerNorm parameter gradients: [ln1 | ln2][chain] buffers, reduced on the side stream
    lnp = [[rt.f32(L.query("msc_layernorm_bwd_workspace_bytes", n, D) // 4) for n in ((B // 2) * T, (B - B // 2) * T)] for _ in range(2)]
    scale = 1.0 / math.sqrt(hd)
    if p.get("last", True):                                       # dy comes from the final LayerNorm's backward on the main stream
        L.call("msc_stream_wait_stream", s2, main, ev_ab)
    Bh = B // 2
    chains = ((main, 0, Bh * T, 0, Bh, 0), (s2, Bh * T, (B - Bh) * T, Bh * h, B - Bh, 1))
    es = x2d.element_size()
    rD, r3D, rF = D * es, 3 * D * es, Fh * es
    a_x, a_h1, a_qkv, a_o, a_x1, a_h2 = x2d.data_ptr(), h1.data_ptr(), qkv.data_ptr(), o.data_ptr(), x1.data_ptr(), h2.data_ptr()
    a_upre, a_st, a_lse, a_delta = upre.data_ptr(), st.data_ptr(), lse.data_ptr(), delta.data_ptr()
    a_dx2, a_dpre, a_dh2, a_dx1, a_do = dx2.data_ptr(), dpre.data_ptr(), dh2.data_ptr(), dx1.data_ptr(), do.data_ptr()
    a_dqkv, a_dh1, a_dx = dqkv.data_ptr(), dh1.data_ptr(), dx.data_ptr()
    ln1g, ln2g = p["ln1g"].p.data_ptr(), p["ln2g"].p.data_ptr()
    tq, tp, t1, t2 = p["qkvw"].t.data_ptr(), p["projw"].t.data_ptr(), p["fc1w"].t.data_ptr(), p["fc2w"].t.data_ptr()
    call, dt = L.call, rt.dt

    def wgrad(xf, df, w, b):
        # ONE weight (+ bias) gradient over the whole batch on the side stream, which waits for both chains
        N, K = w.shape
        wsb = L.query("msc_wgrad_workspace_bytes", N, K, M)
        rt.call_side("msc_linear_wgrad", xf, df, w.g, b.g, rt.workspace("wgrad", wsb, per_stream=False), wsb, M, N, K, rt.accumulate, rt.dt)

    def ln_sum(k, g, b):
        # ONE fixed-order reduction over both chains' partial rows, on the side stream (which waits for both chains)
        rt.call_side("msc_layernorm_bwd_params", lnp[k][0], chains[0][2], lnp[k][1], chains[1][2], D, g.g, b.g, rt.accumulate)

    rt.side_also_waits = (s2, ev_bs)
    try:
        wgrad(u, dx2, p["fc2w"], p["fc2b"])
        # fc2's dgrad epilogue applies GELU' (the derivative fc1's forward epilogue stored; MSC_GELU_DSAVE=0: from the pre-activation)
        for c, r0, n, _, _, _ in chains:
            call("msc_linear_dgrad", a_dx2 + r0 * rD, t2, None, a_upre + r0 * rF, _GELU_BWD, None, a_dpre + r0 * rF, n, D, Fh, dt, c)
        wgrad(h2, dpre, p["fc1w"], p["fc1b"])
        for c, r0, n, _, _, _ in chains:
            call("msc_linear_dgrad", a_dpre + r0 * rF, t1, None, None, L.ACT_NONE, None, a_dh2 + r0 * rD, n, Fh, D, dt, c)
        for c, r0, n, _, _, ci in chains:
            call("msc_layernorm_bwd", a_dh2 + r0 * rD, a_x1 + r0 * rD, a_st + (2 * M + r0) * 4, a_st + (3 * M + r0) * 4, ln2g, a_dx2 + r0 * rD,
                 a_dx1 + r0 * rD, None, None, lnp[1][ci], n, D, 0, dt, c)
        wgrad(o, dx1, p["projw"], p["projb"])
        ln_sum(1, p["ln2g"], p["ln2b"])
        for c, r0, n, _, _, _ in chains:
            call("msc_linear_dgrad", a_dx1 + r0 * rD, tp, None, None, L.ACT_NONE, None, a_do + r0 * rD, n, D, D, dt, c)
        for c, r0, n, l0, nb, _ in chains:
            call("msc_attention_bwd", a_qkv + r0 * r3D, a_o + r0 * rD, a_do + r0 * rD, a_lse + l0 * Tpad * 4, a_delta + 2 * l0 * Tpad * 4,
                 a_dqkv + r0 * r3D, nb, T, h, hd, scale, dt, c)
        wgrad(h1, dqkv, p["qkvw"], p["qkvb"])
        for c, r0, n, _, _, _ in chains:
            call("msc_linear_dgrad", a_dqkv + r0 * r3D, tq, None, None, L.ACT_NONE, None, a_dh1 + r0 * rD, n, 3 * D, D, dt, c)
        for c, r0, n, _, _, ci in chains:
            call("msc_layernorm_bwd", a_dh1 + r0 * rD, a_x + r0 * rD, a_st + r0 * 4, a_st + (M + r0) * 4, ln1g, a_dx1 + r0 * rD, a_dx + r0 * rD,
                 None, None, lnp[0][ci], n, D, 0, dt, c)
        ln_sum(0, p["ln1g"], p["ln1b"])
    finally:
        rt.side_also_waits = None
    if p.get("first", True):                                      # the token gradient is consumed on the main stream
        L.call("msc_stream_wait_stream", main, s2, ev_ba)
    # everything the second chain reads must outlive its kernels: released with the side stream's operands at join_side()
    (rt.capture_keep if torch.cuda.is_current_stream_capturing() else rt._side_keep).append(
        (x2d, st, h1, qkv, lse, o, x1, h2, upre, u, dx2, dpre, dh2, dx1, do, dqkv, dh1, delta, lnp))
    return dx.view(B, T, D), None, None, None


class FusionLayerFn(torch.autograd.Function):
    """One pre-LN transformer block as ONE autograd node (LN -> QKV -> attention -> proj(+x) -> LN -> FFN(+x))."""

    @staticmethod
    def forward(ctx, x, rt, p, h):
        B, T, D = x.shape
        hd = D // h
        if _two_chains_ok(rt, B, T, hd):
            return _fusion_fwd_two_chains(ctx, x, rt, p, h)
        ctx.two = False
        x2d = x.reshape(B * T, D)
        h1, mu1, rs1 = layernorm_fwd(rt, x2d, p["ln1g"], p["ln1b"])
        qkv, _ = linear_fwd(rt, h1, p["qkvw"], p["qkvb"])
        o, P = attention_fwd(rt, qkv, B, T, h, hd)
        x1, _ = linear_fwd(rt, o, p["projw"], p["projb"], L.ACT_NONE, res=x2d)
        h2, mu2, rs2 = layernorm_fwd(rt, x1, p["ln2g"], p["ln2b"])
        u, upre = linear_fwd(rt, h2, p["fc1w"], p["fc1b"], _GELU_FWD, want_pre=True)
        x2, _ = linear_fwd(rt, u, p["fc2w"], p["fc2b"], L.ACT_NONE, res=x1)
        ctx.save_for_backward(x2d, mu1, rs1, h1, qkv, P, o, x1, mu2, rs2, h2, upre, u)
        ctx.cfg = (rt, p, h, (B, T, D))
        return x2.view(B, T, D)

    @staticmethod
    def backward(ctx, dy):
        if ctx.two:
            return _fusion_bwd_two_chains(ctx, dy)
        rt, p, h, (B, T, D) = ctx.cfg
        hd = D // h
        x2d, mu1, rs1, h1, qkv, P, o, x1, mu2, rs2, h2, upre, u = ctx.saved_tensors
        dx2 = dy.reshape(B * T, D).contiguous()
        # fc2's dgrad epilogue applies GELU' (from the saved pre-activation); fc1's bias gradient = row sums of dpre, which
        # fc1's weight-gradient GEMM produces on the side (no column-sum epilogue + finalize launch on the main chain)
        dpre = linear_bwd(rt, dx2, u, None, p["fc2w"], p["fc2b"], L.ACT_NONE, prev_act=_GELU_BWD, prev_ref=upre)
        dh2 = linear_bwd(rt, dpre, h2, None, p["fc1w"], p["fc1b"], L.ACT_NONE)
        dx1 = layernorm_bwd(rt, dh2, x1, mu2, rs2, p["ln2g"], p["ln2b"], dadd=dx2)       # + residual branch
        do = linear_bwd(rt, dx1, o, None, p["projw"], p["projb"], L.ACT_NONE)
        dqkv = attention_bwd(rt, do, qkv, o, P, B, T, h, hd)
        dh1 = linear_bwd(rt, dqkv, h1, None, p["qkvw"], p["qkvb"], L.ACT_NONE)
        dx = layernorm_bwd(rt, dh1, x2d, mu1, rs1, p["ln1g"], p["ln1b"], dadd=dx1)
        return dx.view(B, T, D), None, None, None


class GradTap(torch.autograd.Function):
    """Autograd boundary of the two-phase backward: forwards x (already detached from its producer's graph) and, in the backward,
    puts the gradient that arrives into slots[i] — the very tensor the upstream kernel wrote, no copy — and lets nothing flow on.
    (A detached leaf + .grad would go through AccumulateGrad, which is free to CLONE the gradient with a torch kernel into
    allocator memory: a launch the recorded step list does not contain, so a replayed step would read a stale gradient.)"""

    @staticmethod
    def forward(ctx, x, anchor, slots, i):
        ctx.slots, ctx.i = slots, i
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        ctx.slots[ctx.i] = dy
        return None, None, None, None


# ------------------------------------------------------------------------------------------------ tokens / pooling
class TokensAssemble(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s0, s1, s2, rt, pos: ParamRef):
        B, n0, D = s0.shape
        n1 = s1.shape[1] if s1 is not None else 0
        n2 = s2.shape[1] if s2 is not None else 0
        out = rt.empty(B, n0 + n1 + n2, D)
        rt.call("msc_tokens_assemble", s0, s1, s2, pos.p, out, B, n0, n1, n2, D, rt.dt)
        ctx.cfg = (rt, pos, B, n0, n1, n2, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        rt, pos, B, n0, n1, n2, D = ctx.cfg
        dout = dout.contiguous()
        d0 = rt.empty(B, n0, D)
        d1 = rt.empty(B, n1, D) if n1 else None
        d2 = rt.empty(B, n2, D) if n2 else None
        rt.call("msc_tokens_split", dout, d0, d1, d2, B, n0, n1, n2, D, rt.dt)
        cols = (n0 + n1 + n2) * D
        ws = rt.workspace("colsum", L.query("msc_colsum_workspace_bytes", B, cols), per_stream=False)
        rt.call_side("msc_colsum", dout, pos.g, ws, B, cols, rt.accumulate, rt.dt)
        return d0, d1, d2, None, None


class MeanPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rt):
        B, T, D = x.shape
        y = rt.empty(B, D)
        rt.call("msc_meanpool_fwd", x, y, B, T, D, rt.dt)
        ctx.cfg = (rt, B, T, D)
        return y

    @staticmethod
    def backward(ctx, dy):
        rt, B, T, D = ctx.cfg
        dx = rt.empty(B, T, D)
        rt.call("msc_meanpool_bwd", dy.contiguous(), dx, B, T, D, rt.dt)
        return dx, None


# ------------------------------------------------------------------------------------------------ channel
class ChannelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rt, snr_db, noise, fading, mode, seed, draw=0, draw_dev=None):
        B, k = x.shape
        y = rt.empty(B, k)
        scale = rt.f32(B)
        rt.call("msc_channel_fwd", x, noise, fading, snr_db, y, scale, B, k, mode, int(seed), int(draw), draw_dev, rt.dt)
        ctx.save_for_backward(x, scale)
        ctx.rt = rt
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale = ctx.saved_tensors
        rt = ctx.rt
        dx = rt.empty_like(x)
        rt.call("msc_channel_bwd", dy.contiguous(), x, scale, dx, x.shape[0], x.shape[1], rt.dt)
        return dx, None, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------ LiDAR pillars
class PillarFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, anchor, rt, w: ParamRef, b: ParamRef, G, rng):
        B, Np, _ = points.shape
        C = w.shape[0]
        cell = 2.0 * rng / G
        # bins of the frame's points by BEV cell (written by the forward, walked again by the backward)
        sorted_pts = rt.f32(B * Np, 4)
        cell_start = rt.alloc((L.query("msc_pillar_index_bytes", B, G) // 4,), torch.int32)
        rank = rt.alloc((B * Np,), torch.int32)
        out = rt.empty(B, G, G, C)
        rt.call("msc_pillar_fwd", points, w.p, b.p, sorted_pts, cell_start, rank, out, B, Np, G, C, -rng, cell, rt.dt)
        ctx.save_for_backward(sorted_pts, cell_start)
        ctx.cfg = (rt, w, b, G, B)
        return out

    @staticmethod
    def backward(ctx, dgrid):
        sorted_pts, cell_start = ctx.saved_tensors
        rt, w, b, G, B = ctx.cfg
        ws = rt.workspace("pillar", L.query("msc_pillar_bwd_workspace_bytes"), per_stream=False)
        rt.call_side("msc_pillar_bwd", sorted_pts, cell_start, w.p, b.p, dgrid.contiguous(), w.g, b.g, ws, B, G, w.shape[0],
                     rt.accumulate, rt.dt)
        return None, None, None, None, None, None, None
