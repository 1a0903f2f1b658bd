"""MultimodalSC on MI355X: the Python model surface (names from BASELINE.json `north_star`) over the HIP kernels.

The upstream reference ships no code (SURVEY.md finding zero), so the "reference Python API" this mirrors is the
module/state_dict naming of the build-authored oracle (oracle/model.py): same keys, same forward signature
`model(image, points, state, snr_db, noise, fading)`.  Conv weights are held OHWI (channels-last) and the
policy/value output layers are packed into one [16,hidden] matrix; load_oracle_state_dict/export_state_dict
convert.  Nothing in this file runs math on the CPU or through torch ops: every layer is a C-ABI kernel call (tensors with two
consumers are handled inside one autograd node, so autograd never launches an add of its own: functional.HeadsFn).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass

import torch

from . import functional as F
from . import lib as L
from .runtime import ParamStore, Runtime


@dataclass
class MSCConfig:
    img_size: int = 224
    use_lidar: bool = True
    use_state: bool = True
    n_points: int = 16384
    bev_grid: int = 128
    bev_range: float = 51.2
    point_feat: int = 64
    state_dim: int = 8
    d_model: int = 512
    n_heads: int = 8
    n_layers: int = 4
    ffn: int = 2048
    n_symbols: int = 256
    n_actions: int = 9
    head_hidden: int = 256
    patch: int = 32
    recon_hidden: int = 64
    channel: str = "awgn"
    codec_fp8: bool = False    # semantic encoder/decoder GEMMs on OCP e4m3 operands (configs[4])
    clip_eps: float = 0.2
    c_value: float = 0.5
    c_entropy: float = 0.01
    recon_weight: float = 1.0

    @property
    def img_tokens(self):
        return (self.img_size // 32) ** 2

    @property
    def bev_tokens(self):
        return (self.bev_grid // 8) ** 2 if self.use_lidar else 0

    @property
    def n_tokens(self):
        return self.img_tokens + self.bev_tokens + (1 if self.use_state else 0)

    @property
    def recon_grid(self):
        return self.img_size // self.patch


def named_config(name: str) -> MSCConfig:
    name = name.upper()
    table = {
        "C0": dict(img_size=32, use_lidar=False, use_state=False),
        "C1": dict(img_size=224, use_lidar=False, use_state=False),
        "C2": dict(img_size=224, use_lidar=True, use_state=False),
        "C3": dict(img_size=224, use_lidar=True, use_state=True),
        "C4": dict(img_size=224, use_lidar=True, use_state=True, channel="rayleigh", codec_fp8=True),
    }
    return MSCConfig(**table[name])


_PREMASK = os.environ.get("MSC_BLOCK_PREMASK", "0") == "1"    # opt-in (measured +-0 in the step): 1 = the NEXT block's dgrad epilogue applies a block's output ReLU backward
HEAD_PAD = 16   # policy logits (n_actions) + value (1) packed and zero-padded to 16 output rows
IMG_CPAD = 8    # RGB padded to 8 channels so every implicit-GEMM K chunk is a full 16-byte vector


def _resnet_specs():
    specs = [("image_encoder.conv1.weight", (64, 7, 7, IMG_CPAD), "conv"),
             ("image_encoder.bn1.weight", (64,), "vec"), ("image_encoder.bn1.bias", (64,), "vec")]
    chans = [64, 64, 128, 256, 512]
    for li in range(1, 5):
        for bi in range(2):
            cin = chans[li - 1] if bi == 0 else chans[li]
            cout = chans[li]
            pre = f"image_encoder.layer{li}.{bi}"
            specs += [(f"{pre}.conv1.weight", (cout, 3, 3, cin), "conv"), (f"{pre}.bn1.weight", (cout,), "vec"),
                      (f"{pre}.bn1.bias", (cout,), "vec"), (f"{pre}.conv2.weight", (cout, 3, 3, cout), "conv"),
                      (f"{pre}.bn2.weight", (cout,), "vec"), (f"{pre}.bn2.bias", (cout,), "vec")]
            if bi == 0 and li > 1:
                specs += [(f"{pre}.downsample.0.weight", (cout, 1, 1, cin), "conv"),
                          (f"{pre}.downsample.1.weight", (cout,), "vec"), (f"{pre}.downsample.1.bias", (cout,), "vec")]
    return specs


def param_specs(cfg: MSCConfig):
    d = cfg.d_model
    specs = _resnet_specs()
    if cfg.use_lidar:
        specs += [("bev_encoder.point_mlp.weight", (cfg.point_feat, 4), "vec"), ("bev_encoder.point_mlp.bias", (cfg.point_feat,), "vec")]
        cc = [cfg.point_feat, 128, 256, d]
        for i in range(1, 4):
            specs += [(f"bev_encoder.conv{i}.weight", (cc[i], 3, 3, cc[i - 1]), "conv"),
                      (f"bev_encoder.bn{i}.weight", (cc[i],), "vec"), (f"bev_encoder.bn{i}.bias", (cc[i],), "vec")]
    if cfg.use_state:
        specs += [("state_encoder.fc1.weight", (d, cfg.state_dim), "linear"), ("state_encoder.fc1.bias", (d,), "vec"),
                  ("state_encoder.fc2.weight", (d, d), "linear"), ("state_encoder.fc2.bias", (d,), "vec")]
    specs += [("fusion.pos_embed", (cfg.n_tokens, d), "vec")]
    for i in range(cfg.n_layers):
        pre = f"fusion.layers.{i}"
        specs += [(f"{pre}.ln1.weight", (d,), "vec"), (f"{pre}.ln1.bias", (d,), "vec"),
                  (f"{pre}.attn.qkv.weight", (3 * d, d), "linear"), (f"{pre}.attn.qkv.bias", (3 * d,), "vec"),
                  (f"{pre}.attn.proj.weight", (d, d), "linear"), (f"{pre}.attn.proj.bias", (d,), "vec"),
                  (f"{pre}.ln2.weight", (d,), "vec"), (f"{pre}.ln2.bias", (d,), "vec"),
                  (f"{pre}.mlp.fc1.weight", (cfg.ffn, d), "linear"), (f"{pre}.mlp.fc1.bias", (cfg.ffn,), "vec"),
                  (f"{pre}.mlp.fc2.weight", (d, cfg.ffn), "linear"), (f"{pre}.mlp.fc2.bias", (d,), "vec")]
    specs += [("fusion.ln_f.weight", (d,), "vec"), ("fusion.ln_f.bias", (d,), "vec")]
    specs += [("semantic_encoder.fc1.weight", (d, d), "linear"), ("semantic_encoder.fc1.bias", (d,), "vec"),
              ("semantic_encoder.fc2.weight", (cfg.n_symbols, d), "linear"), ("semantic_encoder.fc2.bias", (cfg.n_symbols,), "vec"),
              ("semantic_decoder.fc1.weight", (d, cfg.n_symbols), "linear"), ("semantic_decoder.fc1.bias", (d,), "vec"),
              ("semantic_decoder.fc2.weight", (d, d), "linear"), ("semantic_decoder.fc2.bias", (d,), "vec"),
              ("head.fc.weight", (cfg.head_hidden, d), "linear"), ("head.fc.bias", (cfg.head_hidden,), "vec"),
              ("head.out.weight", (HEAD_PAD, cfg.head_hidden), "linear"), ("head.out.bias", (HEAD_PAD,), "vec")]
    g = cfg.recon_grid
    specs += [("recon.fc1.weight", (g * g * cfg.recon_hidden, d), "linear"), ("recon.fc1.bias", (g * g * cfg.recon_hidden,), "vec"),
              ("recon.fc2.weight", (cfg.patch * cfg.patch * 3, cfg.recon_hidden), "linear"),
              ("recon.fc2.bias", (cfg.patch * cfg.patch * 3,), "vec")]
    return specs


class MultimodalSC:
    """Camera + LiDAR/BEV + ego-state encoders -> fusion transformer -> semantic encoder -> channel -> decoder ->
    policy/value head + reconstruction head, every layer a hand-written gfx950 kernel."""

    def __init__(self, cfg: MSCConfig, dtype: str = "bf16", device: str = "cuda:0", seed: int = 0):
        assert cfg.n_actions + 1 <= HEAD_PAD
        self.cfg = cfg
        self.rt = Runtime(dtype, device)
        self.params = ParamStore(self.rt, param_specs(cfg))
        self.anchor = torch.zeros(1, device=self.rt.device, requires_grad=True)
        # trunk layers below the backward boundary of backward_lower (0 or 4 = no boundary).  Single GPU: none — since the pillar front
        # end became cheap the BEV-branch backward is enqueued quickly and the boundary only costs (same-box: 8.25 -> 8.12 ms).
        # Data parallel: 3, which is what gives the gradient exchange its mid piece (Trainer sets it when MSC_IMG_CUT is not given).
        self._img_cut_auto = "MSC_IMG_CUT" not in os.environ
        self._img_cut = int(os.environ.get("MSC_IMG_CUT", "4"))
        self._img_cut_out = self._img_cut_grad = None
        self._bev_slot = None
        # in-kernel channel noise (throughput mode): Philox key = (channel_seed, channel_rank), counter word = draw_dev, a
        # device-side counter bumped after EVERY forward (training step, rollout or inference alike), so no two forwards
        # and no two DP ranks see the same noise realisation; Trainer sets channel_rank from torch.distributed
        self.channel_seed = 0
        self.channel_rank = 0
        self.draw_dev = torch.zeros(1, dtype=torch.int64, device=self.rt.device)   # used when rt.host_counters is off (hipGraph)
        self._draw_host = 0                                                        # forwards so far (eager launches)
        self._init_params(seed)

    # ------------------------------------------------------------------ parameters
    def _init_params(self, seed):
        g = torch.Generator().manual_seed(seed)
        for r in self.params.refs.values():
            if r.name.endswith("bn1.weight") or r.name.endswith("bn2.weight") or r.name.endswith("bn3.weight") or \
               r.name.endswith("downsample.1.weight") or ".ln" in r.name and r.name.endswith("weight") or r.name.endswith("ln_f.weight"):
                v = torch.ones(r.shape)
            elif r.name.endswith("bias"):
                v = torch.zeros(r.shape)
            elif r.name == "fusion.pos_embed":
                v = torch.randn(r.shape, generator=g) * 0.02
            else:
                fan_in = 1
                for s in r.shape[1:]:
                    fan_in *= s
                v = torch.randn(r.shape, generator=g) * math.sqrt(2.0 / fan_in) * 0.7
            r.p.copy_(v)
        self.params["image_encoder.conv1.weight"].p[..., 3:].zero_()
        A = self.cfg.n_actions
        self.params["head.out.weight"].p[A + 1:].zero_()
        self.params.refresh_compute_copies()

    def load_oracle_state_dict(self, sd):
        """sd: state_dict of oracle.model.MultimodalSC (or any dict with the same keys/shapes, OIHW convs)."""
        self.rt.join_side()      # a pending side-stream optimizer update must not overwrite what is loaded here
        A = self.cfg.n_actions
        for r in self.params.refs.values():
            if r.name == "head.out.weight":
                v = torch.zeros(r.shape)
                v[:A] = sd["head.policy.weight"]
                v[A] = sd["head.value.weight"][0]
            elif r.name == "head.out.bias":
                v = torch.zeros(r.shape)
                v[:A] = sd["head.policy.bias"]
                v[A] = sd["head.value.bias"][0]
            elif r.kind == "conv":
                w = sd[r.name].permute(0, 2, 3, 1)                       # OIHW -> OHWI
                v = torch.zeros(r.shape)
                v[..., : w.shape[-1]] = w
            else:
                v = sd[r.name]
            r.p.copy_(v.to(torch.float32).reshape(r.shape))
        self.params.refresh_compute_copies()

    def export_state_dict(self, grads: bool = False):
        """Back to the oracle's key names / OIHW layout (parameters, or their gradients with grads=True)."""
        self.rt.join_side()      # weight gradients and the upper half of the optimizer update run on the side stream
        A = self.cfg.n_actions
        out = {}
        for r in self.params.refs.values():
            t = (r.g if grads else r.p).detach().float().cpu()
            if r.name == "head.out.weight":
                out["head.policy.weight"], out["head.value.weight"] = t[:A].clone(), t[A:A + 1].clone()
            elif r.name == "head.out.bias":
                out["head.policy.bias"], out["head.value.bias"] = t[:A].clone(), t[A:A + 1].clone()
            elif r.kind == "conv":
                w = t.permute(0, 3, 1, 2)
                out[r.name] = (w[:, :3] if r.name == "image_encoder.conv1.weight" else w).contiguous()
            else:
                out[r.name] = t.clone()
        return out

    # ------------------------------------------------------------------ blocks
    def _block_params(self, pre, has_ds):
        P = self.params
        d = {"conv1": P[f"{pre}.conv1.weight"], "bn1g": P[f"{pre}.bn1.weight"], "bn1b": P[f"{pre}.bn1.bias"],
             "conv2": P[f"{pre}.conv2.weight"], "bn2g": P[f"{pre}.bn2.weight"], "bn2b": P[f"{pre}.bn2.bias"]}
        if has_ds:
            d.update(dconv=P[f"{pre}.downsample.0.weight"], dbng=P[f"{pre}.downsample.1.weight"], dbnb=P[f"{pre}.downsample.1.bias"])
        return d

    def image_encoder(self, image):
        """image: fp32 NCHW [B,3,H,W] in [0,1] -> tokens [B,(H/32)*(W/32),512]"""
        rt, P = self.rt, self.params
        B, C, H, W = image.shape
        if F._FUSED_STEM and F._STEM_S2D and C == 3 and H % 2 == 0 and W % 2 == 0:
            # the image re-blocked 2x2 -> 12 (16) channels: the 7x7/2 stem convolution as a 4x4/1 one (K 392 -> 256), then
            # BN + ReLU + max-pool in one pass
            x = rt.empty(B, H // 2, W // 2, 16)
            rt.call("msc_image_to_s2d", image, x, B, H, W, rt.dt)
            x = F.StemFn.apply(x, self.anchor, rt, P["image_encoder.conv1.weight"], P["image_encoder.bn1.weight"],
                               P["image_encoder.bn1.bias"], 2, 3, 3, 2, 1, P.stem_w)
        elif F._FUSED_STEM:
            x = rt.empty(B, H, W, IMG_CPAD)
            rt.call("msc_image_to_nhwc", image, x, B, C, H, W, IMG_CPAD, rt.dt)
            # conv -> BN -> ReLU -> max-pool with the BN/ReLU/pool part in one pass (the post-activation map is never written)
            x = F.StemFn.apply(x, self.anchor, rt, P["image_encoder.conv1.weight"], P["image_encoder.bn1.weight"],
                               P["image_encoder.bn1.bias"], 2, 3, 3, 2, 1)
        else:
            x = rt.empty(B, H, W, IMG_CPAD)
            rt.call("msc_image_to_nhwc", image, x, B, C, H, W, IMG_CPAD, rt.dt)
            x = F.ConvBNAct.apply(x, self.anchor, rt, P["image_encoder.conv1.weight"], P["image_encoder.bn1.weight"],
                                  P["image_encoder.bn1.bias"], 2, 3, True, False)
            x = F.MaxPool.apply(x, rt, 3, 2, 1)
        self._img_cut_out = self._img_cut_grad = None
        for li in range(1, 5):
            if li == self._img_cut + 1 and self.cfg.use_lidar and rt.branch_streams and x.requires_grad:
                # autograd boundary inside the trunk: backward_lower() runs the layers above it FIRST, so the main stream
                # has work queued while the host enqueues the whole BEV-branch backward (see backward_lower)
                self._img_cut_out = x
                self._img_cut_grad = [None]
                x = F.GradTap.apply(x.detach(), self.anchor, self._img_cut_grad, 0)
            for bi in range(2):
                has_ds = bi == 0 and li > 1
                # the ReLU backward of a block's output is applied by the NEXT block's dgrad epilogue (functional.BasicBlockFn):
                # every block but the first masks its input's gradient, every block but the last receives a masked one
                first, last = (li, bi) == (1, 0), (li, bi) == (4, 1)
                x = F.BasicBlockFn.apply(x, rt, self._block_params(f"image_encoder.layer{li}.{bi}", has_ds), 2 if has_ds else 1,
                                         not first and _PREMASK, not last and _PREMASK)
        return x.view(B, -1, x.shape[-1])

    def bev_encoder(self, points):
        rt, P, cfg = self.rt, self.params, self.cfg
        g = F.PillarFn.apply(points, self.anchor, rt, P["bev_encoder.point_mlp.weight"], P["bev_encoder.point_mlp.bias"],
                             cfg.bev_grid, cfg.bev_range)
        x = g
        for i in range(1, 4):
            x = F.ConvBNAct.apply(x, self.anchor, rt, P[f"bev_encoder.conv{i}.weight"], P[f"bev_encoder.bn{i}.weight"],
                                  P[f"bev_encoder.bn{i}.bias"], 2, 1, True, True)
        return x.view(x.shape[0], -1, x.shape[-1])

    def state_encoder(self, state):
        rt, P = self.rt, self.params
        s = rt.empty(*state.shape)
        rt.call("msc_cast", state, s, state.numel(), rt.dt)
        h = F.Linear.apply(s, self.anchor, rt, P["state_encoder.fc1.weight"], P["state_encoder.fc1.bias"], L.ACT_RELU, False)
        return F.Linear.apply(h, None, rt, P["state_encoder.fc2.weight"], P["state_encoder.fc2.bias"], L.ACT_NONE, True)

    def fusion(self, img_tok, bev_tok, st_tok):
        rt, P, cfg = self.rt, self.params, self.cfg
        x = F.TokensAssemble.apply(img_tok, bev_tok, st_tok, rt, P["fusion.pos_embed"])
        for i in range(cfg.n_layers):
            pre = f"fusion.layers.{i}"
            p = {"ln1g": P[f"{pre}.ln1.weight"], "ln1b": P[f"{pre}.ln1.bias"], "qkvw": P[f"{pre}.attn.qkv.weight"],
                 "qkvb": P[f"{pre}.attn.qkv.bias"], "projw": P[f"{pre}.attn.proj.weight"], "projb": P[f"{pre}.attn.proj.bias"],
                 "ln2g": P[f"{pre}.ln2.weight"], "ln2b": P[f"{pre}.ln2.bias"], "fc1w": P[f"{pre}.mlp.fc1.weight"],
                 "fc1b": P[f"{pre}.mlp.fc1.bias"], "fc2w": P[f"{pre}.mlp.fc2.weight"], "fc2b": P[f"{pre}.mlp.fc2.bias"],
                 "first": i == 0, "last": i == cfg.n_layers - 1}      # (where the two half-batch chains meet the main stream)
            x = F.FusionLayerFn.apply(x, rt, p, cfg.n_heads)
        return F.LayerNorm.apply(x, rt, P["fusion.ln_f.weight"], P["fusion.ln_f.bias"])

    def _lin(self, x, name, act=L.ACT_NONE):
        P = self.params
        if self.cfg.codec_fp8 and name.startswith("semantic_"):
            return F.LinearFP8.apply(x, self.rt, P[f"{name}.weight"], P[f"{name}.bias"], act, True)
        return F.Linear.apply(x, None, self.rt, P[f"{name}.weight"], P[f"{name}.bias"], act, True)

    def _mlp2(self, x, n1, act1, n2, act2):
        """two dense layers as one autograd node (functional.MLP2); the e4m3 codec keeps its per-layer nodes"""
        P = self.params
        if (self.cfg.codec_fp8 and n1.startswith("semantic_")) or os.environ.get("MSC_MLP2", "1") == "0":   # (A/B knob)
            return self._lin(self._lin(x, n1, act1), n2, act2)
        return F.MLP2.apply(x, self.rt, P[f"{n1}.weight"], P[f"{n1}.bias"], act1, P[f"{n2}.weight"], P[f"{n2}.bias"], act2)

    # ------------------------------------------------------------------ forward
    def forward(self, image, points=None, state=None, snr_db=None, noise=None, fading=None):
        """Inputs are fp32 device tensors.  noise/fading None -> in-kernel Philox (throughput mode).
        Returns dict: head [B,16] (logits 0..A-1, value at A), patches [B*G*G, P*P*3], symbols, rx, feature, z."""
        cfg, rt = self.cfg, self.rt
        B = image.shape[0]
        if cfg.use_lidar and rt.branch_streams:
            # camera and LiDAR branches are independent until the fusion: run the BEV branch on its own HIP stream so
            # the two stacks of (often chip-underfilling) conv kernels overlap; autograd replays the backward of each
            # node on the stream its forward ran on, so the backward overlaps the same way.
            main, s2 = torch.cuda.current_stream(rt.device), rt.branch_stream()
            rt.stream_wait(s2.cuda_stream, main.cuda_stream)
            with torch.cuda.stream(s2):
                bev_tok = self.bev_encoder(points)
            img_tok = self.image_encoder(image)
            rt.stream_wait(main.cuda_stream, s2.cuda_stream)
            if not torch.cuda.is_current_stream_capturing():
                bev_tok.record_stream(main)
        else:
            img_tok = self.image_encoder(image)
            bev_tok = self.bev_encoder(points) if cfg.use_lidar else None
        st_tok = self.state_encoder(state).view(B, 1, cfg.d_model) if cfg.use_state else None
        # autograd boundary between the encoders and everything downstream: the backward runs in two phases
        # (upper: heads/codec/transformer; lower: encoders) so the trainer can start the all-reduce of the upper
        # gradients before the encoder backward (DP overlap), also across hipGraph replays
        self._enc_out = [t for t in (img_tok, bev_tok, st_tok) if t is not None]
        self._bev_slot = 1 if bev_tok is not None else None
        self._enc_grads = [None] * len(self._enc_out)       # filled by the taps' backward (phase 1), consumed by backward_lower
        it = iter([F.GradTap.apply(t.detach(), self.anchor, self._enc_grads, i) for i, t in enumerate(self._enc_out)])
        img_tok = next(it)
        bev_tok = next(it) if bev_tok is not None else None
        st_tok = next(it) if st_tok is not None else None
        rt.join_side()     # a side-stream Adam update of the transformer / codec / head parameters (Trainer) must have landed
        x = self.fusion(img_tok, bev_tok, st_tok)
        f = F.MeanPool.apply(x, rt)
        sym = self._mlp2(f, "semantic_encoder.fc1", L.ACT_RELU, "semantic_encoder.fc2", L.ACT_NONE)
        mode = L.CH_AWGN if cfg.channel == "awgn" else L.CH_RAYLEIGH
        key = ((self.channel_rank & 0xFFFFFFFF) << 32) | (self.channel_seed & 0xFFFFFFFF)
        inkernel = noise is None or (mode == L.CH_RAYLEIGH and fading is None)
        if rt.host_counters:       # eager launches: the draw number is a host argument (no counter kernel on the chain)
            rx = F.ChannelFn.apply(sym, rt, snr_db, noise, fading, mode, key, self._draw_host, None)
            self._draw_host += int(inkernel)
        else:                      # hipGraph replay: the number must live on the device
            rx = F.ChannelFn.apply(sym, rt, snr_db, noise, fading, mode, key, 0, self.draw_dev)
            if inkernel:
                # off the latency-bound codec stretch: the side stream waits for the channel kernel above, and every forward joins the side
                # stream (before the fusion) ahead of its own channel kernel
                rt.call_side("msc_counter_add", self.draw_dev, 1)
        z = self._mlp2(rx, "semantic_decoder.fc1", L.ACT_RELU, "semantic_decoder.fc2", L.ACT_NONE)
        # policy/value head + reconstruction head: one autograd node (z has two consumers; functional.HeadsFn sums their
        # gradients inside a dgrad epilogue instead of an add kernel)
        g, P = cfg.recon_grid, self.params
        head, patches = F.HeadsFn.apply(z, rt, P["head.fc.weight"], P["head.fc.bias"], P["head.out.weight"], P["head.out.bias"],
                                        P["recon.fc1.weight"], P["recon.fc1.bias"], P["recon.fc2.weight"], P["recon.fc2.bias"],
                                        g * g, cfg.recon_hidden)
        return {"head": head, "patches": patches, "symbols": sym, "rx": rx, "feature": f, "z": z}

    __call__ = forward

    # ------------------------------------------------------------------ outputs in the oracle's format
    def logits_value(self, out):
        A = self.cfg.n_actions
        h = out["head"].float()
        return h[:, :A], h[:, A]

    def select_action(self, out):
        """arg-max over the action logits, lowest index on ties (int64, bit-exact integer output)."""
        rt = self.rt
        B = out["head"].shape[0]
        act = torch.empty(B, dtype=torch.int64, device=rt.device)
        rt.call("msc_argmax_rows", out["head"], act, B, self.cfg.n_actions, HEAD_PAD, rt.dt)
        return act

    def select_topk(self, out, k: int):
        """the k best actions per frame, order (logit descending, index ascending on exact ties): int64 [B,k] (bit-exact integer
        output) and their logits fp32 [B,k]; k = 1 is select_action."""
        rt = self.rt
        B = out["head"].shape[0]
        idx = torch.empty(B, k, dtype=torch.int64, device=rt.device)
        val = rt.f32(B, k)
        rt.call("msc_topk_rows", out["head"], idx, val, B, self.cfg.n_actions, k, HEAD_PAD, rt.dt)
        return idx, val

    def recon_image(self, out):
        rt, cfg = self.rt, self.cfg
        B = out["head"].shape[0]
        img = rt.f32(B, 3, cfg.img_size, cfg.img_size)
        rt.call("msc_unpatchify", out["patches"], img, B, cfg.recon_grid, cfg.patch, rt.dt)
        return img

    # ------------------------------------------------------------------ loss + backward
    def mid_offset(self):
        """element offset in the flat buffers of the first parameter whose gradient is final after the first two phases of
        backward_lower (top of the camera trunk, BEV branch, state encoder), or None when the step has no such split"""
        if not (self.cfg.use_lidar and self.rt.branch_streams and 1 <= self._img_cut <= 3):
            return None
        return self.params.refs[f"image_encoder.layer{self._img_cut + 1}.0.conv1.weight"].off

    def backward_lower(self, on_mid=None):
        """phase 2 of the backward: encoders (ResNet-18, BEV, state) from the gradients phase 1 left at the boundary.
        on_mid: callback fired once everything whose parameters lie at or above mid_offset() has been ENQUEUED (top of the
        trunk + state encoder on the main stream, BEV branch on its stream, their weight gradients on the side stream) —
        the trainer starts the all-reduce of that slice there, under the rest of the trunk's backward."""
        rt = self.rt
        outs, grads = self._enc_out, list(self._enc_grads)
        if self._bev_slot is not None and rt._branch is not None:
            # Order of enqueueing (the host, not the GPU, paces this part):
            #  1. the top of the camera trunk (above the boundary image_encoder() left) on the main stream — ~0.5 ms of GPU
            #     work the main stream chews on while
            #  2. the host enqueues the whole BEV-branch backward on ITS stream.  autograd alone would replay the branch
            #     whose forward was recorded first (BEV) LAST — a ~1 ms tail on an otherwise idle chip.  The branch stream
            #     is made current for that call: the engine ends a backward by making the CALLER's stream wait for every
            #     stream it used, which would stall the main stream for the whole BEV backward.
            #  3. the rest of the camera trunk and the state encoder.
            mid_ready = False
            if self._img_cut_grad is not None:
                # (the state encoder — two tiny layers whose parameters sit above mid_offset() — rides along)
                rest = [j for j in range(len(outs)) if j not in (0, self._bev_slot)]
                torch.autograd.backward([outs[0]] + [outs[j] for j in rest], [grads[0]] + [grads[j] for j in rest])
                outs = [self._img_cut_out, outs[self._bev_slot]]
                grads = [self._img_cut_grad[0], grads[self._bev_slot]]
                self._img_cut_out = self._img_cut_grad = None
                mid_ready = True
            i = 1 if mid_ready else self._bev_slot
            main, s2 = torch.cuda.current_stream(rt.device), rt._branch
            rt.stream_wait(s2.cuda_stream, main.cuda_stream)
            if not torch.cuda.is_current_stream_capturing():
                grads[i].record_stream(s2)
            with torch.cuda.stream(s2):
                torch.autograd.backward([outs[i]], [grads[i]])
            outs, grads = outs[:i] + outs[i + 1:], grads[:i] + grads[i + 1:]
            if mid_ready and on_mid is not None:
                on_mid()
        torch.autograd.backward(outs, grads)
        self._enc_out = self._enc_grads = None
        rt.join_side()      # the weight-gradient GEMMs ran on the side stream: order the main stream behind them
        if rt._branch is not None:
            rt.stream_wait(rt.stream, rt._branch.cuda_stream)

    def loss_and_grads(self, out, image, ppo=None, backward=True, lower=True):
        """Fused loss kernels produce the loss terms AND d(loss)/d(head), d(loss)/d(patches); autograd then runs
        the kernel backward chain from those two gradients.  Returns dict of fp32 device scalars."""
        rt, cfg = self.rt, self.cfg
        B = image.shape[0]
        g, P = cfg.recon_grid, cfg.patch
        n_el = 3 * cfg.img_size * cfg.img_size
        sse = rt.f32(B)
        dpat = rt.empty_like(out["patches"])
        ws = rt.workspace("recon_loss", L.query("msc_recon_loss_workspace_bytes", B))
        rt.call("msc_recon_loss", out["patches"], image, sse, dpat, ws, B, g, P, 2.0 * cfg.recon_weight / (B * n_el), rt.dt)
        terms = {"sse": sse}
        dhead = None
        if ppo is not None:
            t = rt.f32(4)
            dhead = rt.empty_like(out["head"])
            rt.call("msc_ppo_loss", out["head"], ppo["actions"], ppo["old_logp"], ppo["adv"], ppo["ret"], t, dhead, B,
                    cfg.n_actions, HEAD_PAD, cfg.clip_eps, cfg.c_value, cfg.c_entropy, rt.dt)
            terms["ppo"] = t
        if backward:
            if dhead is not None:
                torch.autograd.backward([out["patches"], out["head"]], [dpat, dhead])
            else:
                torch.autograd.backward([out["patches"]], [dpat])
            # upper gradients (flat buffer from fusion.pos_embed on) are final once the side stream is joined
            if rt.on_upper_grads_ready is not None:
                rt.join_side()
                rt.on_upper_grads_ready()
            if lower:
                self.backward_lower()
        return terms

    def metrics(self, terms):
        """Host-side scalars (synchronises): mse, psnr (mean over samples), loss, PPO terms."""
        cfg = self.cfg
        n_el = 3 * cfg.img_size * cfg.img_size
        sse = terms["sse"].double().cpu()
        mse = float(sse.sum() / (sse.numel() * n_el))
        psnr = float((10.0 * torch.log10(n_el / sse)).mean())
        m = {"mse": mse, "psnr": psnr, "loss": cfg.recon_weight * mse}
        if "ppo" in terms:
            t = terms["ppo"].cpu()
            m.update(ppo_total=float(t[0]), policy=float(t[1]), value=float(t[2]), entropy=float(t[3]))
            m["loss"] += float(t[0])
        return m
