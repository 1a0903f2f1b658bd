"""Error behaviour of the C-ABI (include/msc.h: 'return value: MSC_OK or a negative MSC_ERR_*; never throws, never prints'): invalid
arguments and unsupported shapes are refused with the documented code BEFORE anything is launched (the output buffers keep their
sentinel), and the Python binding turns the code into a RuntimeError that names the entry point — the product path fails loudly."""
import pytest
import torch

pytestmark = pytest.mark.gpu

BAD, UNSUP = "code -1", "code -3"


@pytest.fixture(scope="module")
def rt():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from multimodal_sc_amd.runtime import Runtime
    return Runtime("bf16", "cuda:0")


def _t(rt, *shape, dtype=None, fill=7.0):
    return torch.full(shape, fill, device=rt.device, dtype=dtype or rt.tdtype)


def _refused(rt, code, name, *args, outputs=()):
    before = [o.clone() for o in outputs]
    with pytest.raises(RuntimeError, match=rf"{name} failed: .*\({code}\)"):
        rt.call(name, *args)
    torch.cuda.synchronize()
    for o, b in zip(outputs, before):
        assert torch.equal(o, b), f"{name}: an output was written although the call was refused"


def test_linear_family_refuses_bad_shapes(rt):
    x, w, b, y = _t(rt, 64, 72), _t(rt, 128, 72), _t(rt, 128, dtype=torch.float32), _t(rt, 64, 128)
    _refused(rt, BAD, "msc_linear_fwd", x, w, b, None, y, None, 64, 128, 0, 0, rt.dt, outputs=(y,))            # K = 0
    _refused(rt, BAD, "msc_linear_fwd", x, w, b, None, y, None, 64, 128, 70, 0, rt.dt, outputs=(y,))           # K not a multiple of 8 (bf16)
    _refused(rt, BAD, "msc_linear_fwd", x, w, b, None, y, None, 64, 128, 72, 9, rt.dt, outputs=(y,))           # unknown activation
    _refused(rt, BAD, "msc_linear_fwd", None, w, b, None, y, None, 64, 128, 72, 0, rt.dt, outputs=(y,))        # x == NULL
    _refused(rt, BAD, "msc_linear_fwd", x, w, b, None, y, None, 64, 128, 72, 0, 7, outputs=(y,))               # unknown dtype
    dx = _t(rt, 64, 72)
    _refused(rt, BAD, "msc_linear_dgrad", y, w, None, None, 5, None, dx, 64, 128, 72, rt.dt, outputs=(dx,))    # the forward-only GELU mode as dact
    _refused(rt, BAD, "msc_linear_dgrad", y, w, None, None, 0, None, dx, 64, 126, 72, rt.dt, outputs=(dx,))    # N not a multiple of 8


def test_attention_refuses_what_the_fused_kernel_does_not_cover(rt):
    B, T, H = 2, 40, 2
    qkv, o = _t(rt, B * T, 3 * H * 64), _t(rt, B * T, H * 64)
    lse = torch.zeros(B * H * 64, device=rt.device)
    _refused(rt, UNSUP, "msc_attention_fwd", qkv, o, lse, B, T, H, 32, 0.125, rt.dt, outputs=(o,))              # head_dim != 64
    _refused(rt, UNSUP, "msc_attention_fwd", qkv, o, lse, B, T, H, 64, 0.125, 0, outputs=(o,))                  # fp32: the unfused path's job
    _refused(rt, UNSUP, "msc_attention_fwd", qkv, o, lse, B, 600, H, 64, 0.125, rt.dt, outputs=(o,))            # T beyond the LDS-resident panel
    _refused(rt, BAD, "msc_attention_fwd", qkv, o, lse, B, T, H, 64, 0.0, rt.dt, outputs=(o,))                  # scale must be positive
    _refused(rt, BAD, "msc_attention_fwd", qkv, o, lse, 0, T, H, 64, 0.125, rt.dt, outputs=(o,))                # empty batch


def test_norm_family_refuses_bad_shapes(rt):
    rows = 16
    x, y = _t(rt, rows, 512), _t(rt, rows, 512)
    g, b = torch.ones(512, device=rt.device), torch.zeros(512, device=rt.device)
    mu, rs = torch.zeros(rows, device=rt.device), torch.zeros(rows, device=rt.device)
    _refused(rt, BAD, "msc_layernorm_fwd", x, g, b, y, mu, rs, rows, 300, 1e-5, rt.dt, outputs=(y,))            # D not a multiple of 512 (bf16)
    _refused(rt, BAD, "msc_layernorm_fwd", x, g, b, y, mu, rs, rows, 2048, 1e-5, rt.dt, outputs=(y,))           # D > 1024: the row no longer fits a wave's registers
    _refused(rt, BAD, "msc_layernorm_fwd", x, g, b, y, mu, rs, 0, 512, 1e-5, rt.dt, outputs=(y,))               # no rows
    sc, sh = torch.ones(60, device=rt.device), torch.zeros(60, device=rt.device)
    x2, y2 = _t(rt, 32, 60), _t(rt, 32, 60)
    _refused(rt, BAD, "msc_bn_apply", x2, sc, sh, None, y2, 32, 60, 1, rt.dt, outputs=(y2,))                    # C not a multiple of 8 (bf16)
    ws = torch.zeros(1 << 16, device=rt.device)
    dg, db = torch.zeros(64, device=rt.device), torch.zeros(64, device=rt.device)
    x3, dx3 = _t(rt, 32, 64), _t(rt, 32, 64)
    # a ReLU mask needs either the activation y or the folded affine (scale, shift): neither given
    _refused(rt, BAD, "msc_bn_bwd", x3, x3, None, dg, dg, dg, None, None, dx3, None, dg, db, ws, 32, 64, 1, 0, rt.dt, outputs=(dx3,))
    # scale without shift
    _refused(rt, BAD, "msc_bn_bwd", x3, x3, None, dg, dg, dg, dg, None, dx3, None, dg, db, ws, 32, 64, 1, 0, rt.dt, outputs=(dx3,))


def test_conv_family_refuses_bad_shapes(rt):
    B, H, W = 2, 8, 8
    x, w = _t(rt, B, H, W, 24), _t(rt, 64, 3, 3, 24)
    y = _t(rt, B, H, W, 64)
    part = torch.zeros(4096, device=rt.device)
    # implicit-GEMM gather: the input channel count is a power of two (the filter-tap index is a shift)
    _refused(rt, BAD, "msc_conv2d_fwd", x, w, y, part, B, H, W, 24, 64, 3, 3, 1, 1, None, 0, rt.dt, outputs=(y,))
    x16, w16 = _t(rt, B, H, W, 16), _t(rt, 64, 3, 3, 16)
    _refused(rt, BAD, "msc_conv2d_fwd", x16, w16, y, part, B, H, W, 16, 64, 3, 3, 0, 1, None, 0, rt.dt, outputs=(y,))   # stride 0
    _refused(rt, BAD, "msc_conv2d_fwd", x16, w16, y, part, B, H, W, 16, 64, 9, 9, 1, 0, None, 0, rt.dt, outputs=(y,))   # filter larger than the padded image
    # BatchNorm-statistics dgrad: row-major tiled launches only
    dy, wt, dx = _t(rt, B, 4, 4, 64), _t(rt, 16, 3, 3, 64), _t(rt, B, H, W, 16)
    stats = torch.zeros(4, 16, device=rt.device)
    _refused(rt, UNSUP, "msc_conv2d_dgrad_bnstat", dy, wt, None, dx, x16, stats, part, B, H, W, 16, 64, 3, 3, 2, 1, None, 0, rt.dt, outputs=(dx,))


def test_small_kernels_refuse_bad_arguments(rt):
    head = torch.zeros(4, 16, device=rt.device, dtype=rt.tdtype)
    idx = torch.full((4, 3), -1, device=rt.device, dtype=torch.int64)
    val = torch.full((4, 3), -1.0, device=rt.device)
    _refused(rt, BAD, "msc_topk_rows", head, idx, val, 4, 16, 9, 16, rt.dt, outputs=(idx, val))                 # k > 8
    _refused(rt, BAD, "msc_topk_rows", head, idx, val, 4, 2, 3, 16, rt.dt, outputs=(idx, val))                  # k > n
    img = torch.zeros(2, 3, 9, 9, device=rt.device)
    out = _t(rt, 2, 4, 4, 16)
    _refused(rt, BAD, "msc_image_to_s2d", img, out, 2, 9, 9, rt.dt, outputs=(out,))                             # odd image size
