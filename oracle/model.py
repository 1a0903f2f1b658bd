"""oracle/model.py — BUILD-AUTHORED CPU oracle (PyTorch fp32) for the multimodal semantic-communication path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package (multimodal_sc_amd/) never imports it.

PARITY UNPINNED BY THE REFERENCE: /root/reference holds only README.md (2 lines, SURVEY.md finding zero), so
there is no upstream code, fixture or golden vector to follow or to pin this oracle against.  Every block below
restates the *published textbook definition* named in SURVEY.md §2 / BASELINE.json `north_star`:
  ImageEncoder      ResNet-18 trunk (He et al. 2016, layer table)              -> [B,512,H/32,W/32]
  BEVEncoder        per-point MLP 4->64 + ReLU, scatter-max into a GxG BEV grid, 3x (conv3x3/2 + BN + ReLU)
  StateEncoder      MLP 8 -> 512 -> 512
  FusionTransformer pre-LN transformer encoder over concat(camera, LiDAR, ego) tokens + learned positions
  SemanticEncoder   MLP 512->512->k, average-power normalisation
  Channel           AWGN / Rayleigh with perfect-CSI equalisation, SNR in dB
  SemanticDecoder   MLP k->512->512
  PolicyValueHead   MLP -> |A| logits + scalar value; action = arg-max (lowest index on ties)
  ReconHead         linear patch decoder (MAE-style) -> image in [0,1]; PSNR = 10 log10(1/MSE)
  ppo_loss          PPO clipped surrogate + 0.5*c_v*(V-R)^2 - c_e*entropy (Schulman et al. 2017)
Hyper-parameters are the frozen build decisions of SURVEY.md §7 (recorded in DESIGN.md).
BatchNorm uses batch statistics (training mode), eps 1e-5, no running averages.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class MSCConfig:
    img_size: int = 224
    use_lidar: bool = True
    use_state: bool = True
    n_points: int = 16384
    bev_grid: int = 128
    bev_range: float = 51.2          # grid covers [-range, range) in x and y
    point_feat: int = 64
    state_dim: int = 8
    d_model: int = 512
    n_heads: int = 8
    n_layers: int = 4
    ffn: int = 2048
    n_symbols: int = 256             # real symbols (= 128 complex)
    n_actions: int = 9
    head_hidden: int = 256
    patch: int = 32
    recon_hidden: int = 64
    channel: str = "awgn"            # "awgn" | "rayleigh"
    codec_fp8: bool = False          # semantic encoder/decoder GEMMs on OCP e4m3 operands (configs[4])
    clip_eps: float = 0.2
    c_value: float = 0.5
    c_entropy: float = 0.01
    recon_weight: float = 1.0

    @property
    def img_tokens(self) -> int:
        return (self.img_size // 32) ** 2

    @property
    def bev_tokens(self) -> int:
        return (self.bev_grid // 8) ** 2 if self.use_lidar else 0

    @property
    def n_tokens(self) -> int:
        return self.img_tokens + self.bev_tokens + (1 if self.use_state else 0)

    @property
    def recon_grid(self) -> int:
        return self.img_size // self.patch


def named_config(name: str) -> MSCConfig:
    """The BASELINE.json configs[i] as named presets (C0..C4)."""
    name = name.upper()
    if name == "C0":
        return MSCConfig(img_size=32, use_lidar=False, use_state=False)
    if name == "C1":
        return MSCConfig(img_size=224, use_lidar=False, use_state=False)
    if name == "C2":
        return MSCConfig(img_size=224, use_lidar=True, use_state=False)
    if name == "C3":
        return MSCConfig(img_size=224, use_lidar=True, use_state=True)
    if name == "C4":
        return MSCConfig(img_size=224, use_lidar=True, use_state=True, channel="rayleigh", codec_fp8=True)
    raise ValueError(name)


# ---- optional emulation of the product's bf16 storage points (tests only) ---------------------------------------
# The HIP path stores every activation tensor in bf16 (fp32 accumulation inside each kernel).  With _ROUND set, the
# oracle rounds (straight-through) at the same storage points, so a bf16 run can be compared against "fp32 math with
# the same roundings" instead of against pure fp32 — deep BN/ReLU stacks amplify rounding noise, and this separates
# that inherent sensitivity from implementation error.  Default: None = pure fp32 oracle.
_ROUND = None


def _r(t):
    return t if _ROUND is None else _ROUND(t)


class emulate_bf16_storage:
    def __enter__(self):
        global _ROUND
        _ROUND = lambda t: t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())
        return self

    def __exit__(self, *a):
        global _ROUND
        _ROUND = None


def _bn(x, w, b):
    return F.batch_norm(_r(x), None, None, w, b, True, 0.0, 1e-5)


# ---- optional taps on the conv-stack ReLUs (tests only): with RELU_TAPS a dict, every ReLU that follows a BatchNorm
# records its PRE-activation under the name of the convolution that feeds it (the product's parameter name), so a test can
# locate elements whose sign differs between the two implementations (tests/test_model_gpu.py).  Default None = no-op.
RELU_TAPS = None


def _relu(x, tag):
    if RELU_TAPS is not None:
        RELU_TAPS[tag] = x.detach()
    return F.relu(x)


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        self.tag = ""        # set by ImageEncoder: "image_encoder.layer<i>.<j>"
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, 0, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x
        out = _r(_relu(_bn(self.conv1(x), self.bn1.weight, self.bn1.bias), self.tag + ".conv1.weight"))
        out = _bn(self.conv2(out), self.bn2.weight, self.bn2.bias)
        if self.downsample is not None:
            idt = _r(_bn(self.downsample[0](x), self.downsample[1].weight, self.downsample[1].bias))
        return _r(_relu(out + idt, self.tag + ".conv2.weight"))


class ImageEncoder(nn.Module):
    """ResNet-18 trunk: 7x7/2 conv-64, 3x3/2 max-pool, 4 stages x 2 BasicBlocks (64/128/256/512)."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        chans = [64, 64, 128, 256, 512]
        for i in range(1, 5):
            stride = 1 if i == 1 else 2
            setattr(self, f"layer{i}", nn.Sequential(BasicBlock(chans[i - 1], chans[i], stride), BasicBlock(chans[i], chans[i], 1)))
            for j, blk in enumerate(getattr(self, f"layer{i}")):
                blk.tag = f"image_encoder.layer{i}.{j}"

    def forward(self, img):
        x = _r(_relu(_bn(self.conv1(_r(img)), self.bn1.weight, self.bn1.bias), "image_encoder.conv1.weight"))
        x = F.max_pool2d(x, 3, 2, 1)
        for i in range(1, 5):
            x = getattr(self, f"layer{i}")(x)
        return x  # [B,512,h,w]


class BEVEncoder(nn.Module):
    def __init__(self, cfg: MSCConfig):
        super().__init__()
        self.cfg = cfg
        self.point_mlp = nn.Linear(4, cfg.point_feat)
        c = cfg.point_feat
        self.conv1 = nn.Conv2d(c, 128, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(128)
        self.conv2 = nn.Conv2d(128, 256, 3, 2, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(256)
        self.conv3 = nn.Conv2d(256, cfg.d_model, 3, 2, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cfg.d_model)

    def pillar_grid(self, points):
        """points [B,N,4] -> grid [B,G,G,C] (NHWC): max of relu(point_mlp) over the points of each cell, 0 if empty."""
        cfg = self.cfg
        B, N, _ = points.shape
        G = cfg.bev_grid
        cell = torch.tensor(2.0 * cfg.bev_range / G, dtype=torch.float32)
        lo = torch.tensor(-cfg.bev_range, dtype=torch.float32)
        feat = F.relu(self.point_mlp(points))                        # [B,N,C]
        pf = points.float()                                          # cell assignment is defined in fp32 arithmetic
        fx = torch.floor((pf[..., 0] - lo) / cell)
        fy = torch.floor((pf[..., 1] - lo) / cell)
        valid = (fx >= 0) & (fx < G) & (fy >= 0) & (fy < G)
        idx = (fy.clamp(0, G - 1) * G + fx.clamp(0, G - 1)).long()   # [B,N]
        feat = feat * valid.unsqueeze(-1)                            # invalid points contribute 0 (= empty)
        grid = torch.zeros(B, G * G, cfg.point_feat, dtype=feat.dtype)
        grid = grid.scatter_reduce(1, idx.unsqueeze(-1).expand(-1, -1, cfg.point_feat), feat, reduce="amax", include_self=True)
        return grid.view(B, G, G, cfg.point_feat)

    def forward(self, points):
        g = _r(self.pillar_grid(points)).permute(0, 3, 1, 2)         # NCHW for torch convs
        x = _r(_relu(_bn(self.conv1(g), self.bn1.weight, self.bn1.bias), "bev_encoder.conv1.weight"))
        x = _r(_relu(_bn(self.conv2(x), self.bn2.weight, self.bn2.bias), "bev_encoder.conv2.weight"))
        x = _r(_relu(_bn(self.conv3(x), self.bn3.weight, self.bn3.bias), "bev_encoder.conv3.weight"))
        return x  # [B,512,G/8,G/8]


class StateEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.fc1 = nn.Linear(cfg.state_dim, cfg.d_model)
        self.fc2 = nn.Linear(cfg.d_model, cfg.d_model)

    def forward(self, s):
        return _r(self.fc2(_r(F.relu(self.fc1(_r(s))))))


class Attention(nn.Module):
    def __init__(self, d, h):
        super().__init__()
        self.h = h
        self.qkv = nn.Linear(d, 3 * d)
        self.proj = nn.Linear(d, d)

    def forward(self, x):
        B, T, D = x.shape
        qkv = _r(self.qkv(x)).view(B, T, 3, self.h, D // self.h)
        q, k, v = qkv[:, :, 0].transpose(1, 2), qkv[:, :, 1].transpose(1, 2), qkv[:, :, 2].transpose(1, 2)
        att = _r(torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D // self.h), dim=-1))
        o = _r((att @ v).transpose(1, 2).reshape(B, T, D))
        return self.proj(o)


class MLP(nn.Module):
    def __init__(self, d, f):
        super().__init__()
        self.fc1 = nn.Linear(d, f)
        self.fc2 = nn.Linear(f, d)

    def forward(self, x):
        return self.fc2(_r(F.gelu(_r(self.fc1(x)))))


class FusionLayer(nn.Module):
    def __init__(self, d, h, f):
        super().__init__()
        self.ln1 = nn.LayerNorm(d)
        self.attn = Attention(d, h)
        self.ln2 = nn.LayerNorm(d)
        self.mlp = MLP(d, f)

    def forward(self, x):
        x = _r(x + self.attn(_r(self.ln1(x))))
        return _r(x + self.mlp(_r(self.ln2(x))))


class FusionTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.pos_embed = nn.Parameter(torch.zeros(cfg.n_tokens, cfg.d_model))
        nn.init.normal_(self.pos_embed, std=0.02)
        self.layers = nn.ModuleList([FusionLayer(cfg.d_model, cfg.n_heads, cfg.ffn) for _ in range(cfg.n_layers)])
        self.ln_f = nn.LayerNorm(cfg.d_model)

    def forward(self, tokens):
        x = _r(tokens + self.pos_embed)
        for l in self.layers:
            x = l(x)
        return _r(self.ln_f(x))


def e4m3_rne(v: torch.Tensor) -> torch.Tensor:
    """Round fp32 values to the OCP e4m3 grid (4 exponent bits, bias 7, 3 mantissa bits, max 448, subnormal step 2^-9),
    nearest-even, saturating.  Written out with exact power-of-two arithmetic (no dependence on a library cast)."""
    a = v.abs().clamp(max=448.0)
    _, ex = torch.frexp(a)                         # a = m * 2^ex, m in [0.5, 1)
    e = (ex - 1).clamp(min=-6)                     # subnormals share the exponent of the smallest normal
    step = torch.ldexp(torch.ones_like(a), e - 3)
    return torch.sign(v) * torch.round(a / step) * step      # torch.round: half to even


def fp8_qdq(t: torch.Tensor) -> torch.Tensor:
    """per-tensor e4m3 quantise / de-quantise: scale = amax/448, q = rne(t * (448/amax)), result q * scale"""
    amax = t.detach().abs().max().to(torch.float32)
    if float(amax) == 0.0:
        return t * 0.0
    scale = amax / 448.0
    inv = 448.0 / amax
    return e4m3_rne(t.to(torch.float32) * inv) * scale


class FP8Linear(nn.Linear):
    """y = x^ W^^T + b with x^, W^ the e4m3 quantise/de-quantise of x and W; backward straight-through on x^ and W^
    (dx = dy W^, dW = dy^T x^), which is what multimodal_sc_amd.functional.LinearFP8 computes."""

    def forward(self, x):
        xh = x + (fp8_qdq(x) - x).detach()
        wh = self.weight + (fp8_qdq(self.weight) - self.weight).detach()
        return F.linear(xh, wh, self.bias)


def _codec_linear(cfg, i, o):
    return FP8Linear(i, o) if cfg.codec_fp8 else nn.Linear(i, o)


class SemanticEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.fc1 = _codec_linear(cfg, cfg.d_model, cfg.d_model)
        self.fc2 = _codec_linear(cfg, cfg.d_model, cfg.n_symbols)

    def forward(self, f):
        return _r(self.fc2(_r(F.relu(self.fc1(f)))))     # un-normalised symbols; Channel applies the power constraint


def power_normalize(x):
    k = x.shape[-1]
    return x * torch.sqrt(k / (x.pow(2).sum(-1, keepdim=True) + 1e-12))


class Channel(nn.Module):
    """y = h*s + n with unit-power s; AWGN: h=1.  Rayleigh: h~CN(0,1), output equalised by perfect CSI.
    noise / fading are explicit standard-normal tensors [B,k] (Rayleigh pairs consecutive reals as re,im)."""

    def __init__(self, mode="awgn"):
        super().__init__()
        self.mode = mode

    def forward(self, x, snr_db, noise, fading=None):
        s = power_normalize(x)
        sigma = torch.pow(torch.tensor(10.0), -snr_db / 20.0).view(-1, 1)
        if self.mode == "awgn":
            return s + sigma * noise
        B, k = s.shape
        sc = s.view(B, k // 2, 2)
        n = (sigma * noise).view(B, k // 2, 2)
        h = fading.view(B, k // 2, 2) * (1.0 / math.sqrt(2.0))
        hr, hi, nr, ni = h[..., 0], h[..., 1], n[..., 0], n[..., 1]
        den = 1.0 / (hr * hr + hi * hi + 1e-9)
        out_r = sc[..., 0] + (nr * hr + ni * hi) * den
        out_i = sc[..., 1] + (ni * hr - nr * hi) * den
        return torch.stack([out_r, out_i], -1).view(B, k)


class SemanticDecoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.fc1 = _codec_linear(cfg, cfg.n_symbols, cfg.d_model)
        self.fc2 = _codec_linear(cfg, cfg.d_model, cfg.d_model)

    def forward(self, y):
        return _r(self.fc2(_r(F.relu(self.fc1(y)))))


class PolicyValueHead(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.fc = nn.Linear(cfg.d_model, cfg.head_hidden)
        self.policy = nn.Linear(cfg.head_hidden, cfg.n_actions)
        self.value = nn.Linear(cfg.head_hidden, 1)

    def forward(self, z):
        h = _r(torch.tanh(self.fc(z)))
        return _r(self.policy(h)), _r(self.value(h)).squeeze(-1)


class ReconHead(nn.Module):
    """z -> G*G cells x hidden -> ReLU -> per-cell linear to a PxPx3 patch -> sigmoid."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        g = cfg.recon_grid
        self.fc1 = nn.Linear(cfg.d_model, g * g * cfg.recon_hidden)
        self.fc2 = nn.Linear(cfg.recon_hidden, cfg.patch * cfg.patch * 3)

    def forward(self, z):
        cfg = self.cfg
        B, g, P = z.shape[0], cfg.recon_grid, cfg.patch
        h = _r(F.relu(self.fc1(z))).view(B * g * g, cfg.recon_hidden)
        p = _r(torch.sigmoid(self.fc2(h))).view(B, g, g, P, P, 3)          # (gy,gx,py,px,c)
        return p.permute(0, 5, 1, 3, 2, 4).reshape(B, 3, g * P, g * P)      # NCHW


class MultimodalSC(nn.Module):
    def __init__(self, cfg: MSCConfig):
        super().__init__()
        self.cfg = cfg
        self.image_encoder = ImageEncoder()
        if cfg.use_lidar:
            self.bev_encoder = BEVEncoder(cfg)
        if cfg.use_state:
            self.state_encoder = StateEncoder(cfg)
        self.fusion = FusionTransformer(cfg)
        self.semantic_encoder = SemanticEncoder(cfg)
        self.channel = Channel(cfg.channel)
        self.semantic_decoder = SemanticDecoder(cfg)
        self.head = PolicyValueHead(cfg)
        self.recon = ReconHead(cfg)

    def forward(self, image, points=None, state=None, snr_db=None, noise=None, fading=None):
        cfg = self.cfg
        B = image.shape[0]
        toks = [self.image_encoder(image).flatten(2).transpose(1, 2)]          # [B,hw,512]
        if cfg.use_lidar:
            toks.append(self.bev_encoder(points).flatten(2).transpose(1, 2))
        if cfg.use_state:
            toks.append(self.state_encoder(state).unsqueeze(1))
        x = self.fusion(torch.cat(toks, 1))
        f = _r(x.mean(1))
        sym = self.semantic_encoder(f)
        y = _r(self.channel(sym, snr_db, noise, fading))
        z = self.semantic_decoder(y)
        logits, value = self.head(z)
        recon = self.recon(z)
        return {"logits": logits, "value": value, "recon": recon, "symbols": sym, "rx": y, "feature": f, "z": z}


def select_action(logits):
    """arg-max, lowest index on exact ties (torch.argmax on CPU returns the first maximal index)."""
    return torch.argmax(logits, dim=-1)


def select_topk(logits, k):
    """the k best actions, order (logit descending, index ascending on exact ties) — oracle/rl.py topk_rows"""
    from .rl import topk_rows
    idx, val = topk_rows(logits.detach().numpy(), k)
    return torch.from_numpy(idx), torch.from_numpy(val)


def psnr(recon, image):
    mse = (recon - image).pow(2).flatten(1).mean(1)
    return 10.0 * torch.log10(1.0 / mse)


def ppo_loss(logits, value, actions, old_logp, adv, ret, cfg: MSCConfig):
    logp_all = torch.log_softmax(logits, -1)
    logp = logp_all.gather(1, actions.view(-1, 1)).squeeze(1)
    ratio = torch.exp(logp - old_logp)
    surr = torch.minimum(ratio * adv, torch.clamp(ratio, 1 - cfg.clip_eps, 1 + cfg.clip_eps) * adv)
    pl = -surr.mean()
    vl = 0.5 * (value - ret).pow(2).mean()
    ent = -(logp_all.exp() * logp_all).sum(-1).mean()
    return pl + cfg.c_value * vl - cfg.c_entropy * ent, pl, vl, ent


def total_loss(out, image, cfg: MSCConfig, ppo=None):
    """ppo = dict(actions, old_logp, adv, ret) or None (reconstruction only)."""
    mse = (out["recon"] - image).pow(2).mean()
    loss = cfg.recon_weight * mse
    terms = {"mse": mse}
    if ppo is not None:
        t, pl, vl, ent = ppo_loss(out["logits"], out["value"], ppo["actions"], ppo["old_logp"], ppo["adv"], ppo["ret"], cfg)
        loss = loss + t
        terms.update(ppo_total=t, policy=pl, value=vl, entropy=ent)
    terms["loss"] = loss
    return loss, terms


def synthetic_batch(cfg: MSCConfig, B: int, seed: int = 0, snr_db: float = 10.0):
    """SURVEY.md §8d synthetic inputs (seeded torch.Generator on the CPU).  The generator lives in the product package
    (multimodal_sc_amd/synthetic.py: no model code) so bench.py's GPU leg needs nothing from oracle/; both sides of every
    parity test draw identical frames from it."""
    from multimodal_sc_amd.synthetic import synthetic_batch as gen
    return gen(cfg, B, seed, snr_db)


def build_oracle(cfg: MSCConfig, seed: int = 0) -> MultimodalSC:
    torch.manual_seed(seed)
    return MultimodalSC(cfg).float()
