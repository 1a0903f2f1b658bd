"""End-to-end parity: the HIP model (through the C-ABI) against the build-authored CPU oracle on identical weights
and seeded inputs — forward outputs, decoder PSNR, policy actions (bit-exact), loss terms and EVERY parameter gradient.

Tolerances:
  f32 mode : forward outputs / PSNR (dB) within 1e-3 (north_star: "within 1e-3"), actions bit-exact, loss terms 1e-4.
             Gradients: training-mode BatchNorm over a handful of samples is ill-conditioned, so the fp32 oracle itself
             carries rounding error; the criterion is therefore anchored on an fp64 run of the oracle:
                 err(gpu, fp64) <= 5 * err(fp32-oracle, fp64) + 5e-4 * max|grad|        for every parameter tensor.
  bf16 mode: PSNR within 0.05 dB, logits within 0.08 absolute; actions must match wherever the oracle's top-2 logit
             margin exceeds 0.1 (bf16 cannot resolve smaller margins after ~60 layers); gradient direction: cosine
             similarity > 0.90 per large tensor against the oracle run with the same bf16 storage roundings, and
             never further from it than from pure fp32 (test_bf16_end_to_end_parity_c3 explains the chaos floor).
"""
import copy
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _to(d, batch, dtype=None):
    out = {}
    for k, v in batch.items():
        if isinstance(v, dict):
            out[k] = _to(d, v, dtype)
        elif torch.is_tensor(v):
            out[k] = v.to(d, dtype) if (dtype is not None and v.is_floating_point()) else v.to(d)
        else:
            out[k] = v
    return out


def _oracle_run(om, batch, ocfg, ppo=True):
    from oracle import model as O
    for p in om.parameters():
        p.grad = None
    out = om(batch["image"], batch.get("points"), batch.get("state"), batch["snr_db"], batch["noise"], batch["fading"])
    loss, terms = O.total_loss(out, batch["image"], ocfg, batch["ppo"] if ppo else None)
    loss.backward()
    return out, {k: float(v.detach()) for k, v in terms.items()}


PAIR_INFO = {}      # seed chosen by the last _run_pair and the distance of its PPO ratios from the clip boundaries (recorded by the tests)


def _clip_class(logits, ppo, clip_eps):
    """per sample: -1 / 0 / +1 = the PPO surrogate takes the clipped branch below 1-eps / the unclipped branch / clipped above 1+eps"""
    lp = torch.log_softmax(logits.detach().float().cpu(), -1).gather(1, ppo["actions"].view(-1, 1)).squeeze(1)
    r = torch.exp(lp - ppo["old_logp"])
    up = (ppo["adv"] > 0) & (r > 1 + clip_eps)
    dn = (ppo["adv"] < 0) & (r < 1 - clip_eps)
    return up.int() - dn.int(), r


def _run_pair(name, B, dtype, ppo=True, with_f64=False, snr_db=10.0, taps=False):
    from multimodal_sc_amd.model import MultimodalSC, named_config
    from oracle import model as O
    ocfg, cfg = O.named_config(name), named_config(name)
    om = O.build_oracle(ocfg, seed=1)
    # The PPO surrogate min(r A, clip(r) A) is discontinuous in its gradient where the probability ratio r crosses 1 +- clip_eps:
    # a sample sitting ON the boundary (seed 2 at 15 dB has r = 1.2003) is clipped in one implementation and not in the
    # other for any rounding difference, which moves the whole gradient (norm ratio 1.9), not an implementation error.
    # The batch seed is therefore the first of 2, 3, .. whose oracle ratios all keep `clip_margin` from both boundaries.
    clip_margin = 0.03 if ocfg.codec_fp8 else 2e-3      # e4m3 rounding flips move C4's logits by up to ~1e-2
    for seed in range(2, 8):
        batch = O.synthetic_batch(ocfg, B, seed=seed, snr_db=snr_db)
        otaps = {} if taps else None
        O.RELU_TAPS = otaps
        try:
            oout, oterms = _oracle_run(om, batch, ocfg, ppo)
        finally:
            O.RELU_TAPS = None
        if not ppo:
            break
        lp = torch.log_softmax(oout["logits"].detach(), -1).gather(1, batch["ppo"]["actions"].view(-1, 1)).squeeze(1)
        ratio = torch.exp(lp - batch["ppo"]["old_logp"])
        margin = float(min((ratio - (1 - ocfg.clip_eps)).abs().min(), (ratio - (1 + ocfg.clip_eps)).abs().min()))
        PAIR_INFO.update(seed=seed, clip_margin=margin, required=clip_margin)
        if margin >= clip_margin:
            break
    else:
        raise AssertionError("no batch seed keeps the PPO ratios off the clip boundaries")
    g64 = None
    if with_f64:
        om64 = copy.deepcopy(om).double()
        _oracle_run(om64, _to("cpu", batch, torch.float64), ocfg, ppo)
        g64 = {k: p.grad for k, p in om64.named_parameters()}
    m = MultimodalSC(cfg, dtype=dtype, device="cuda:0")
    m.load_oracle_state_dict(om.state_dict())
    if taps:
        m.rt.taps = {}
    gb = _to(m.rt.device, batch)
    out = m(gb["image"], gb.get("points"), gb.get("state"), gb["snr_db"], gb["noise"], gb["fading"])
    terms = m.loss_and_grads(out, gb["image"], gb["ppo"] if ppo else None)
    torch.cuda.synchronize()
    if taps:
        return om, oout, oterms, m, out, terms, batch, g64, otaps
    return om, oout, oterms, m, out, terms, batch, g64


def _conv_order(cfg):
    """forward order of the conv+BN(+ReLU) layers of each encoder branch, by the weight name that tags their ReLU"""
    img = ["image_encoder.conv1.weight"]
    for li in range(1, 5):
        for bi in range(2):
            img += [f"image_encoder.layer{li}.{bi}.conv1.weight", f"image_encoder.layer{li}.{bi}.conv2.weight"]
    bev = [f"bev_encoder.conv{i}.weight" for i in (1, 2, 3)] if cfg.use_lidar else []
    return img, bev


def _layer_tag_of_param(k):
    """parameter name -> tag of the ReLU its layer feeds (None outside the conv stacks)"""
    if k.startswith("bev_encoder.point_mlp"):
        return "bev_encoder.conv1.weight"          # upstream of the whole BEV conv stack
    if k.startswith("bev_encoder."):
        i = k.split(".")[1][-1]
        return f"bev_encoder.conv{i}.weight"
    if not k.startswith("image_encoder."):
        return None
    parts = k.split(".")
    if parts[1] in ("conv1", "bn1"):
        return "image_encoder.conv1.weight"
    blk = ".".join(parts[:3])
    which = parts[3]
    if which in ("conv1", "bn1"):
        return blk + ".conv1.weight"
    return blk + ".conv2.weight"                   # conv2, bn2 and the downsample branch all end in the block-output ReLU


def _relu_flips(m, otaps):
    """elements whose ReLU mask differs between the HIP path and the oracle, per ReLU: {tag: (count, max |pre-activation|
    of the oracle at those elements relative to the tensor's max)}.  Such an element is a pre-activation within fp32
    rounding of zero that landed on the other side in the two summation orders."""
    flips = {}
    for tag, pre in otaps.items():
        y = m.rt.taps[tag].float().cpu()                       # NHWC, post-ReLU
        pre = pre.permute(0, 2, 3, 1)
        diff = (y > 0) != (pre > 0)
        n = int(diff.sum())
        if n:
            flips[tag] = (n, float(pre[diff].abs().max() / pre.abs().max()))
    return flips


F32_CASES = [("C0", 8, 10.0, "1"), ("C1", 2, 10.0, "1"), ("C2", 2, 10.0, "1"), ("C3", 2, 10.0, "1"), ("C3", 2, 10.0, "0"),
             ("C4", 2, 10.0, "1"), ("C4", 2, -5.0, "1"), ("C4", 2, 5.0, "1"), ("C4", 2, 15.0, "1")]


@pytest.mark.parametrize("name,B,snr_db,splitk", F32_CASES)
def test_f32_end_to_end_parity(cuda, knob, name, B, snr_db, splitk):
    """Every named single-GPU configuration (C0..C4; C4 over the SNR sweep of configs[4]) against the fp32 oracle.
    splitk = knob conv_splitk: C3 also runs with the split-K convolutions switched off — both runs are held to the SAME
    criterion, so nothing about it depends on split-K.
    C4 = Rayleigh channel + e4m3 semantic encoder/decoder GEMMs: an activation that sits within fp32 rounding of an
    e4m3 rounding boundary may quantise one grid step (6 %) apart in the two implementations, so C4 compares the
    codec-dependent tensors at 2e-2 instead of 1e-3 (one flipped element moves a 512-term dot product by ~4e-3) (PSNR and the bit-exact action check are unchanged).
    Gradient criterion (all but C4): every parameter tensor within 5x the fp32 oracle's own error against its fp64 run.
    A tensor may miss that bound ONLY if a ReLU-mask flip is PROVEN in its own layer or downstream of it: an element of a
    conv-stack ReLU whose mask differs between the two implementations, with an oracle pre-activation below 1e-5 of the
    tensor's maximum (i.e. zero to fp32 rounding) — at most 2 such elements per ReLU.  At 2 frames one flipped element of
    layer4 (98 samples per channel) moves every gradient upstream of it by ~1 %; those tensors are then held to 2 %
    relative Frobenius error and the whole gradient to cosine > 0.9995.  No proven flip -> no loosening."""
    from multimodal_sc_amd.model import named_config
    from oracle import model as O
    knob("conv_splitk", int(splitk))
    om, oout, oterms, m, out, terms, batch, g64, otaps = _run_pair(name, B, "f32", with_f64=True, snr_db=snr_db, taps=True)
    logits, value = m.logits_value(out)
    t = 2e-2 if name == "C4" else 1e-3
    gslack = 2e-2 if name == "C4" else 5e-4
    assert (logits.cpu() - oout["logits"]).abs().max() < t
    assert (value.cpu() - oout["value"]).abs().max() < t
    assert (out["symbols"].float().cpu() - oout["symbols"]).abs().max() < t
    # Rayleigh: equalised by 1/|h|^2; the noise term scales with sigma = 10^(-snr/20), so low SNR widens the absolute bound
    rx_t = t * (10 if name == "C4" else 1) * max(1.0, 10 ** (-snr_db / 20.0))
    assert (out["rx"].float().cpu() - oout["rx"]).abs().max() < rx_t
    assert (m.recon_image(out).cpu() - oout["recon"]).abs().max() < t
    met = m.metrics(terms)
    opsnr = O.psnr(oout["recon"], batch["image"]).mean().item()
    assert abs(met["psnr"] - opsnr) < 1e-3, (met["psnr"], opsnr)                      # decoder-PSNR parity
    assert torch.equal(m.select_action(out).cpu(), O.select_action(oout["logits"]))   # actions bit-exact
    # top-k action indices (north_star "index/top-k"): bit-exact wherever the oracle's ranking is decided by more than the
    # output tolerance (adjacent sorted logits at least 2t apart); the kernel itself is held bit-exact on ties in test_kernels_gpu
    oi, ov = O.select_topk(oout["logits"], 3)
    gi, _ = m.select_topk(out, 3)
    srt = oout["logits"].detach().sort(-1, descending=True).values
    clear = ((srt[:, :3] - srt[:, 1:4]) > 2 * t).all(1)
    assert torch.equal(gi.cpu()[clear], oi[clear])        # (C4's 2e-2 tolerance can leave no clear row among 2 frames: its top-k
    assert name == "C4" or clear.any()                    #  check with rows that ARE clear is test_c4_topk_end_to_end_8_frames)
    assert torch.equal(gi[:, 0], m.select_action(out))
    lt = 5e-3 if name == "C4" else 1e-4        # C4: one e4m3 grid step on a codec activation moves a value-loss term by ~3e-3
    for k in ("mse", "policy", "value", "entropy", "loss"):
        assert abs(met[k] - oterms[k]) < lt * max(1.0, abs(oterms[k])), (k, met[k], oterms[k])
    # ---- ReLU-mask flips: located, counted, and shown to be zero-to-rounding pre-activations
    assert set(otaps) == set(m.rt.taps), (sorted(otaps), sorted(m.rt.taps))
    flips = _relu_flips(m, otaps)
    for tag, (n, mag) in flips.items():
        assert n <= 2 and mag < 1e-5, ("ReLU masks differ beyond rounding", tag, n, mag)
    img_order, bev_order = _conv_order(named_config(name))
    last = {}                                          # branch -> index of the LAST (closest to the loss) flipped ReLU
    for br, order in (("img", img_order), ("bev", bev_order)):
        idx = [order.index(tg) for tg in flips if tg in order]
        last[br] = max(idx) if idx else -1

    def flip_downstream_of(k):
        tag = _layer_tag_of_param(k)
        if tag is None:
            return False
        br, order = ("bev", bev_order) if tag.startswith("bev") else ("img", img_order)
        return order.index(tag) <= last[br]
    gg = m.export_state_dict(grads=True)
    bad, loosened = [], []
    for k, p in om.named_parameters():
        ref64 = g64[k]
        if name == "C4":
            # e4m3 rounding decisions differ between ANY two runs whose activations differ in the last fp32 bits (the fp32
            # oracle itself sits ~1 % from its fp64 run here), so element-wise bounds are meaningless; the codec kernels are
            # held to tight bounds in test_kernels_gpu.py, and the whole-model gradient is checked for direction and norm
            a, b = gg[k].double().flatten(), ref64.flatten()
            cos = (a @ b / (a.norm() * b.norm()).clamp_min(1e-300)).item()
            nr = (a.norm() / b.norm().clamp_min(1e-300)).item()
            if b.norm() > 0 and (cos < 0.97 or abs(nr - 1) > 0.1):
                bad.append((k, cos, nr))
            continue
        den = ref64.abs().max().item()
        err = (gg[k].double() - ref64).abs()
        e_gpu = err.max().item()
        e_f32 = (p.grad.double() - ref64).abs().max().item()
        bound = 5 * e_f32 + gslack * den + 1e-12
        if e_gpu > bound:
            relf = (err.norm() / ref64.norm().clamp_min(1e-300)).item()
            if flip_downstream_of(k) and relf <= 2e-2:
                loosened.append(k)
            else:
                bad.append((k, e_gpu / max(den, 1e-30), e_f32 / max(den, 1e-30), relf, "flip proven" if flip_downstream_of(k) else "NO flip"))
    # the clipped / unclipped classification of every PPO sample must be the oracle's (the batch seed keeps the ratios off the boundaries)
    cg, _ = _clip_class(logits, batch["ppo"], named_config(name).clip_eps)
    co, _ = _clip_class(oout["logits"], batch["ppo"], named_config(name).clip_eps)
    assert torch.equal(cg, co), (cg.tolist(), co.tolist())
    line = (f"[{name} B={B} snr={snr_db} splitk={splitk}] batch seed {PAIR_INFO.get('seed')} (min |ratio - (1 +- eps)| = "
            f"{PAIR_INFO.get('clip_margin', float('nan')):.4f}, required {PAIR_INFO.get('required')}); "
            f"relu flips (tag: count, |pre|/max): {flips}; tensors on the loose bound: {len(loosened)}")
    print(line)
    rec = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(rec):                       # kept as evidence next to the test log (profiles/r2_relu_flips.txt)
        with open(os.path.join(rec, "relu_flips.txt"), "a") as fh:
            fh.write(line + "\n")
    assert not bad, (bad[:8], flips)
    if loosened:
        a = torch.cat([gg[k].double().flatten() for k in g64])
        b = torch.cat([g64[k].flatten() for k in g64])
        assert (a @ b / (a.norm() * b.norm())).item() > 0.9995, loosened[:8]


def test_c4_topk_end_to_end_8_frames(cuda):
    """configs[4] top-k action indices end to end (f32 mode), non-vacuously: the 2-frame case above accepts "no row is clear" for C4,
    whose a-priori output tolerance (2e-2: one e4m3 grid step on a codec activation) is wider than the spread of the logits at random
    initialisation.  Here the yardstick is the MEASURED logit error of the run: wherever the oracle's first four sorted logits are more
    than twice that error apart, the ranking of the first three actions is decided and the top-3 indices must be equal; at 8 frames
    such rows exist at both ends of the SNR sweep (asserted)."""
    from oracle import model as O
    for snr in (10.0, -5.0):
        om, oout, oterms, m, out, terms, batch, _ = _run_pair("C4", 8, "f32", snr_db=snr)
        logits, _ = m.logits_value(out)
        err = (logits.cpu() - oout["logits"]).abs().max().item()
        assert err < 2e-2, err
        oi, _ = O.select_topk(oout["logits"], 3)
        gi, _ = m.select_topk(out, 3)
        srt = oout["logits"].detach().sort(-1, descending=True).values
        clear = ((srt[:, :3] - srt[:, 1:4]) > 2 * err + 1e-6).all(1)
        print(f"C4 f32 snr={snr}: max|dlogit| {err:.2e}, rows with a decided top-3: {int(clear.sum())} of 8")
        assert clear.any(), (err, srt)
        assert torch.equal(gi.cpu()[clear], oi[clear])
        assert torch.equal(gi[:, 0], m.select_action(out))
        top2 = (srt[:, 0] - srt[:, 1]) > 2 * err + 1e-6
        assert torch.equal(m.select_action(out).cpu()[top2], O.select_action(oout["logits"])[top2])


def test_bf16_forward_parity_c3_batch64(cuda):
    """The benchmarked mode at the benchmarked size: C3, 64 frames, bf16, explicit noise — forward against the fp32 oracle's
    forward on the host cores.  PSNR within 0.05 dB, logits within 0.08, actions equal wherever the oracle's top-2 margin
    exceeds 0.1; top-3 indices equal wherever the first four sorted logits are 0.16 apart."""
    from multimodal_sc_amd.model import MultimodalSC, named_config
    from oracle import model as O
    ocfg, cfg = O.named_config("C3"), named_config("C3")
    om = O.build_oracle(ocfg, seed=1)
    batch = O.synthetic_batch(ocfg, 64, seed=2, snr_db=10.0)
    with torch.no_grad():
        oout = om(batch["image"], batch.get("points"), batch.get("state"), batch["snr_db"], batch["noise"], batch["fading"])
    m = MultimodalSC(cfg, dtype="bf16", device="cuda:0")
    m.load_oracle_state_dict(om.state_dict())
    gb = _to(m.rt.device, batch)
    with torch.no_grad():
        out = m(gb["image"], gb.get("points"), gb.get("state"), gb["snr_db"], gb["noise"], gb["fading"])
        logits, value = m.logits_value(out)
        rec = m.recon_image(out)
    torch.cuda.synchronize()
    dl = (logits.float().cpu() - oout["logits"]).abs().max().item()
    psnr_g = O.psnr(rec.float().cpu(), batch["image"]).mean().item()
    psnr_o = O.psnr(oout["recon"], batch["image"]).mean().item()
    print(f"C3 bf16 B=64: max|dlogit| {dl:.4f} psnr {psnr_g:.4f} oracle {psnr_o:.4f}")
    assert dl < 0.08, dl
    assert abs(psnr_g - psnr_o) < 0.05, (psnr_g, psnr_o)
    # actions: equal wherever the oracle's top-2 margin exceeds 0.1 (the stated bf16 criterion; may be few rows at random init) AND
    # wherever it exceeds twice the measured logit error (never vacuous)
    top2 = oout["logits"].topk(2, -1).values
    for margin in (0.1, 2 * dl + 1e-6):
        sure = (top2[:, 0] - top2[:, 1]) > margin
        assert torch.equal(m.select_action(out).cpu()[sure], O.select_action(oout["logits"])[sure])
    assert sure.any()
    srt = oout["logits"].sort(-1, descending=True).values
    clear = ((srt[:, :3] - srt[:, 1:4]) > 2 * dl + 1e-6).all(1)
    oi, _ = O.select_topk(oout["logits"], 3)
    gi, _ = m.select_topk(out, 3)
    assert torch.equal(gi.cpu()[clear], oi[clear])


def test_bf16_forward_parity_c0(cuda):
    from oracle import model as O
    om, oout, oterms, m, out, terms, batch, _ = _run_pair("C0", 8, "bf16")
    logits, _ = m.logits_value(out)
    assert (logits.cpu() - oout["logits"]).abs().max() < 0.08
    met = m.metrics(terms)
    assert abs(met["psnr"] - O.psnr(oout["recon"], batch["image"]).mean().item()) < 0.05
    assert torch.isfinite(m.params.G).all()


def test_bf16_forward_parity_c4(cuda):
    """configs[4] in the mode it is benchmarked in — bf16 storage, e4m3 quantise-from-bf16 codec GEMMs, Rayleigh channel with
    explicit noise/fading — end to end against the fp32 oracle (which applies the same e4m3 grid to its fp32 activations).
    PSNR as C3 (0.05 dB).  Logits: a bf16-rounded activation near an e4m3 rounding boundary lands one grid step (6 %) away from
    the oracle's, so the bound is 0.15 instead of C3's 0.08 (measured max 0.05-0.1 over the SNR sweep); actions must match
    wherever the oracle's top-2 margin exceeds that bound; gradients finite, direction checked on the large tensors downstream
    of the codec (upstream the BN/ReLU chaos floor of test_bf16_end_to_end_parity_c3 applies unchanged)."""
    from oracle import model as O
    for snr in (-5.0, 10.0):
        om, oout, oterms, m, out, terms, batch, _ = _run_pair("C4", 8, "bf16", snr_db=snr)
        logits, value = m.logits_value(out)
        dl = (logits.cpu() - oout["logits"]).abs().max().item()
        met = m.metrics(terms)
        opsnr = O.psnr(oout["recon"], batch["image"]).mean().item()
        print(f"C4 bf16 snr={snr}: max|dlogit| {dl:.4f} psnr {met['psnr']:.4f} oracle {opsnr:.4f}")
        assert dl < 0.15, dl
        assert abs(met["psnr"] - opsnr) < 0.05, (met["psnr"], opsnr)
        top2 = oout["logits"].topk(2, -1).values
        sure = (top2[:, 0] - top2[:, 1]) > 0.3
        assert torch.equal(m.select_action(out).cpu()[sure], O.select_action(oout["logits"])[sure])
        assert torch.isfinite(m.params.G).all() and m.params.G.abs().max() > 0
        gg = m.export_state_dict(grads=True)
        cs = torch.nn.functional.cosine_similarity
        for k in ("recon.fc1.weight", "recon.fc2.weight", "semantic_decoder.fc2.weight", "head.fc.weight"):
            c = cs(gg[k].flatten().double(), dict(om.named_parameters())[k].grad.flatten().double(), dim=0).item()
            assert c > 0.9, (k, c)


def test_bf16_end_to_end_parity_c3(cuda):
    """bf16 storage vs the oracle.  A BN/ReLU stack at random init amplifies ANY perturbation of its activations
    (the fp32 oracle with bf16 roundings merely inserted at the storage points already drops to cosine ~0.85-0.9
    against itself on the stem gradients), so gradient DIRECTION is checked against the oracle run with
    oracle.model.emulate_bf16_storage() and bf16-rounded GEMM weights — same math, same rounding points — where the
    HIP path must stay above the chaos floor measured below; the cosine against the pure-fp32 oracle is printed for
    the record.  Forward quantities are checked against the pure-fp32 oracle."""
    from oracle import model as O
    om, oout, oterms, m, out, terms, batch, _ = _run_pair("C3", 8, "bf16")
    logits, value = m.logits_value(out)
    assert (logits.cpu() - oout["logits"]).abs().max() < 0.08
    met = m.metrics(terms)
    opsnr = O.psnr(oout["recon"], batch["image"]).mean().item()
    assert abs(met["psnr"] - opsnr) < 0.05, (met["psnr"], opsnr)
    top2 = oout["logits"].topk(2, -1).values
    sure = (top2[:, 0] - top2[:, 1]) > 0.1
    assert torch.equal(m.select_action(out).cpu()[sure], O.select_action(oout["logits"])[sure])
    g32 = {k: p.grad.clone() for k, p in om.named_parameters()}
    # the emulated run: GEMM/conv weights as the bf16 compute copies, activations rounded at the storage points
    ome = copy.deepcopy(om)
    with torch.no_grad():
        for k, p in ome.named_parameters():
            if p.dim() >= 2 and k not in ("fusion.pos_embed", "bev_encoder.point_mlp.weight"):
                p.copy_(p.to(torch.bfloat16).float())
    with O.emulate_bf16_storage():
        _oracle_run(ome, batch, O.named_config("C3"), True)
    gg = m.export_state_dict(grads=True)
    rows = []
    for k, p in ome.named_parameters():
        if p.numel() >= 4096 and p.grad.abs().max() > 1e-7:
            cs = torch.nn.functional.cosine_similarity
            rows.append((cs(gg[k].flatten().double(), p.grad.flatten().double(), dim=0).item(),
                         cs(gg[k].flatten().double(), g32[k].flatten().double(), dim=0).item(), k))
    rows.sort()
    print("gradient cosines (vs bf16-emulating oracle, vs pure-fp32 oracle):", [(round(a, 4), round(b, 4), k) for a, b, k in rows])
    # Measured on CPU (oracle only): two bf16-emulating runs whose INPUT differs by 1e-6 relative already disagree at
    # cosine 0.92 on the stem gradients (bf16 rounding decisions flip and cascade through the BN/ReLU stack), so no two
    # correct bf16 implementations with different fp32 summation order can agree better than that.  The bound below is
    # that chaos floor; implementation errors are caught by the per-kernel tests (tight) and the f32 end-to-end test.
    assert rows[0][0] > 0.90, rows[:8]
    assert all(a > b - 0.02 for a, b, _ in rows), [r for r in rows if r[0] <= r[1] - 0.02]


def test_fusion_two_chains_no_grad_forward(cuda, monkeypatch):
    """inference / rollout forwards at a batch large enough for the two fusion chains (nothing is saved for a backward there: the
    buffers the second chain uses are protected by record_stream) give the same outputs as the training-mode forward, repeatedly"""
    from multimodal_sc_amd import functional as F
    from multimodal_sc_amd.model import MultimodalSC, named_config
    from oracle import model as O
    monkeypatch.setattr(F, "_FUSION_CHAINS", True)        # (the default is one chain while the linears run on the ping-pong kernel)
    m = MultimodalSC(named_config("C3"), dtype="bf16", device="cuda:0", seed=3)
    gb = _to(m.rt.device, O.synthetic_batch(O.named_config("C3"), 48, seed=5))
    args = (gb["image"], gb.get("points"), gb.get("state"), gb["snr_db"], gb["noise"], gb["fading"])
    ref = m(*args)
    ref = (ref["head"].clone(), ref["patches"].clone())
    for _ in range(3):
        with torch.no_grad():
            out = m(*args)
            junk = [torch.full((1 << 22,), float("nan"), device=m.rt.device) for _ in range(8)]     # churn the allocator
        torch.cuda.synchronize()
        assert torch.equal(out["head"], ref[0]) and torch.equal(out["patches"], ref[1])
        del junk


def test_fusion_two_chains_equal_one_chain(cuda, monkeypatch):
    """The fusion layers run the two halves of the batch as two chains of launches on two streams.  Every kernel of a layer is
    per-sample and the weight gradients still run once over the whole batch, so against the one-chain schedule the outputs and
    every gradient are BIT-identical, except the LayerNorm parameter gradients, whose two half sums are added in a different order."""
    from multimodal_sc_amd import functional as F
    from multimodal_sc_amd.model import MultimodalSC, named_config
    from oracle import model as O
    res = {}
    monkeypatch.setattr(F, "_CHAIN_MIN_ROWS", 0)          # (the default only splits batches of >= ~37 frames)
    for two in (True, False):
        monkeypatch.setattr(F, "_FUSION_CHAINS", two)
        m = MultimodalSC(named_config("C3"), dtype="bf16", device="cuda:0", seed=3)
        gb = _to(m.rt.device, O.synthetic_batch(O.named_config("C3"), 5, seed=5))      # odd batch: chains of 2 and 3 frames
        out = m(gb["image"], gb.get("points"), gb.get("state"), gb["snr_db"], gb["noise"], gb["fading"])
        m.loss_and_grads(out, gb["image"], gb["ppo"])
        torch.cuda.synchronize()
        res[two] = (out["head"].clone(), out["patches"].clone(), m.params.G.clone(), m)
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    m = res[True][3]
    for name, ref in m.params.refs.items():
        a, b = res[True][2][ref.off:ref.off + ref.numel], res[False][2][ref.off:ref.off + ref.numel]
        if name.startswith("fusion.layers.") and (".ln1." in name or ".ln2." in name):
            assert (a - b).abs().max().item() <= 1e-5 * max(1e-6, b.abs().max().item()), name
        else:
            assert torch.equal(a, b), name


@pytest.mark.parametrize("name,B", [("C3", 2), ("C3", 64), ("C1", 64), ("C2", 64), ("C4", 64)])
def test_determinism_bitwise(cuda, name, B):
    """run the same step twice: every output and every gradient must be bit-identical (no float atomics anywhere) — also
    at the full benchmark batch (64 frames), where split-K slices, three streams and every tile shape are in play.
    C1 (camera only: T=49 tokens, no BEV stream, single-stream schedule), C2 (no ego-state token) and C4 (Rayleigh channel +
    e4m3 codec GEMMs quantised from bf16 activations — the mode C4 would be benchmarked in) run here as named configurations
    at 64 bf16 frames per GPU; their outputs must also be finite."""
    from multimodal_sc_amd.model import MultimodalSC, named_config
    from oracle import model as O
    m = MultimodalSC(named_config(name), dtype="bf16", device="cuda:0", seed=3)
    gb = _to(m.rt.device, O.synthetic_batch(O.named_config(name), B, seed=5))
    res = []
    for _ in range(2):
        out = m(gb["image"], gb.get("points"), gb.get("state"), gb["snr_db"], gb["noise"], gb["fading"])
        m.loss_and_grads(out, gb["image"], gb["ppo"])
        torch.cuda.synchronize()
        res.append((out["head"].clone(), out["patches"].clone(), m.params.G.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[0][2], res[1][2])
    assert torch.isfinite(res[0][2]).all() and torch.isfinite(res[0][0].float()).all()
    assert res[0][2].abs().max() > 0


def test_inkernel_noise_differs_between_forwards_and_ranks(cuda):
    """throughput mode (noise=None): every forward draws fresh channel noise (device-side draw counter), and two models
    that differ only in channel_rank (what Trainer sets from the DP rank) draw different noise for the same frames."""
    from multimodal_sc_amd.model import MultimodalSC, named_config
    from oracle import model as O
    cfg = named_config("C0")
    batch = O.synthetic_batch(O.named_config("C0"), 8, seed=5)
    rx = []
    for rank in (0, 1):
        m = MultimodalSC(cfg, dtype="f32", device="cuda:0", seed=3)
        m.channel_rank = rank
        d = m.rt.device
        with torch.no_grad():
            for _ in range(2):
                out = m(batch["image"].to(d), None, None, batch["snr_db"].to(d), None, None)
                rx.append(out["rx"].clone())
    torch.cuda.synchronize()
    assert not torch.equal(rx[0], rx[1]), "two consecutive forwards saw the same noise"
    assert not torch.equal(rx[0], rx[2]) and not torch.equal(rx[1], rx[3]), "rank 0 and rank 1 drew the same noise"


def test_training_reduces_loss(cuda):
    """a few fused-Adam steps on a fixed batch must reduce the loss (optimizer + compute-copy refresh wired correctly)."""
    from multimodal_sc_amd.model import MultimodalSC, named_config
    from oracle import model as O
    m = MultimodalSC(named_config("C0"), dtype="bf16", device="cuda:0", seed=3)
    batch = O.synthetic_batch(O.named_config("C0"), 8, seed=5)
    d = m.rt.device
    img, snr = batch["image"].to(d), batch["snr_db"].to(d)
    losses = []
    for _ in range(8):
        out = m(img, None, None, snr, None, None)
        terms = m.loss_and_grads(out, img, None)
        losses.append(m.metrics(terms)["mse"])
        m.params.adam_step(lr=2e-3)
    assert losses[-1] < losses[0] * 0.9, losses


@pytest.mark.parametrize("name,B,chains", [("C3", 2, True), ("C3", 40, False), ("C1", 3, False), ("C4", 4, True)])
def test_launch_replay_matches_eager(cuda, monkeypatch, name, B, chains):
    """Trainer.step through the C-side launch replay (default: two eager warm-up steps, one recorded step, then replays of the
    recorded launch list from csrc/replay.hip) must leave parameters, Adam moments and loss terms BIT-identical to the eager
    Python loop (MSC_REPLAY=0): same kernels, same streams, same order; step and noise-draw numbers come from device counters.
    Also: a second batch of the same shape goes through the static input buffers (copied by step(), or written there by the caller:
    Trainer.input_buffers), and a batch of another shape records its own program without disturbing the first."""
    from multimodal_sc_amd import functional as F
    from multimodal_sc_amd.model import MultimodalSC, named_config
    from multimodal_sc_amd.trainer import Trainer
    from oracle import model as O
    monkeypatch.setattr(F, "_FUSION_CHAINS", bool(chains))
    if chains:
        monkeypatch.setattr(F, "_CHAIN_MIN_ROWS", 0)      # the fusion's two half-batch chains are part of what is recorded
    ocfg = O.named_config(name)
    batches = [_to("cuda:0", O.synthetic_batch(ocfg, B, seed=5 + i)) for i in range(2)]
    other = _to("cuda:0", O.synthetic_batch(ocfg, B + 1, seed=9))
    for b in batches + [other]:
        b["noise"] = b["fading"] = None            # in-kernel Philox noise keyed by the device-side draw counter
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MSC_REPLAY", mode)
        m = MultimodalSC(named_config(name), dtype="bf16", device="cuda:0", seed=3)
        tr = Trainer(m, lr=1e-3)
        assert tr.use_replay == (mode == "1")
        sse = []
        for i in range(7):
            feed = batches[i % 2] if i != 4 else other
            if mode == "1" and i >= 5:
                # a frame source that writes into the trainer's own input buffers: no staging copy, same result
                own = tr.input_buffers(feed)
                assert own is not None and own["image"].data_ptr() != feed["image"].data_ptr()
                for k in ("image", "points", "state", "snr_db"):
                    if torch.is_tensor(feed.get(k)):
                        own[k].copy_(feed[k])
                for k, v in feed["ppo"].items():
                    own["ppo"][k].copy_(v)
                feed = own
            terms = tr.step(feed)
            sse.append(terms["sse"].clone())
        torch.cuda.synchronize()
        if mode == "1":
            progs = [st["prog"] for st in tr._programs.values()]      # least recently used first: the one-off shape, then the replayed one
            rec = [p for p in progs if p is not None]
            assert len(progs) == 2 and len(rec) == 1, "the launch list was not recorded"
            assert rec[0].count() > 100
        res[mode] = (m.params.P.clone(), m.params.M.clone(), m.params.V.clone(), sse)
    for a, b in zip(res["0"][:3], res["1"][:3]):
        assert torch.equal(a, b)
    for a, b in zip(res["0"][3], res["1"][3]):
        assert torch.equal(a, b)
    assert torch.isfinite(res["1"][0]).all()


def test_launch_replay_recorded_from_a_clean_side_stream(cuda, monkeypatch):
    """The recorded step must carry its main<-side join whatever the host thought about the side stream when recording started:
    export_state_dict() between the warm-up steps and the recorded step leaves the side stream clean (host flag cleared), and every
    LATER replay follows a step whose optimizer tail / weight transposes are still running there.  Replay must stay bit-identical to
    the eager loop under the same sequence of calls (round-3 advisor finding on Runtime.join_side)."""
    from multimodal_sc_amd.model import MultimodalSC, named_config
    from multimodal_sc_amd.trainer import Trainer
    from oracle import model as O
    ocfg = O.named_config("C3")
    batch = _to("cuda:0", O.synthetic_batch(ocfg, 2, seed=5))
    batch["noise"] = batch["fading"] = None
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MSC_REPLAY", mode)
        m = MultimodalSC(named_config("C3"), dtype="bf16", device="cuda:0", seed=3)
        tr = Trainer(m, lr=1e-3)
        for i in range(6):
            if i == 2:
                m.export_state_dict()            # joins + synchronises: the step recorded next starts from a clean side stream
                assert not m.rt._side_dirty
            tr.step(batch)
        torch.cuda.synchronize()
        if mode == "1":
            progs = [st["prog"] for st in tr._programs.values()]
            assert progs and progs[0] is not None
            # the list contains the main<-side wait of the forward (a stream_wait whose waiter is the main stream and whose signaler is the side stream)
            names = [n for n, _ in progs[0].records] if hasattr(progs[0], "records") else None
            assert names is None or "msc_stream_wait_stream" in names
        res[mode] = (m.params.P.clone(), m.params.M.clone(), m.params.V.clone())
    for a, b in zip(res["0"], res["1"]):
        assert torch.equal(a, b)


def test_launch_replay_programs_are_bounded_under_a_per_step_schedule(cuda, monkeypatch):
    """The learning rate is a baked argument of the recorded optimizer launch: a schedule that changes it every step makes a new program
    key every step.  Those keys never reach a recording; the trainer must stay on the eager loop, keep at most MAX_PROGRAMS entries and
    say so once (round-3 advisor finding on Trainer._programs)."""
    import warnings
    from multimodal_sc_amd.model import MultimodalSC, named_config
    from multimodal_sc_amd.trainer import Trainer
    from oracle import model as O
    monkeypatch.setenv("MSC_REPLAY", "1")
    batch = _to("cuda:0", O.synthetic_batch(O.named_config("C0"), 2, seed=5))
    batch["noise"] = batch["fading"] = None
    m = MultimodalSC(named_config("C0"), dtype="bf16", device="cuda:0", seed=3)
    tr = Trainer(m, lr=1e-3)
    monkeypatch.setattr(Trainer, "MAX_PROGRAMS", 3)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        for i in range(7):
            tr.lr = 1e-3 * (1.0 - 0.1 * i)
            tr.step(batch)
    torch.cuda.synchronize()
    assert len(tr._programs) <= 3
    assert all(st["prog"] is None for st in tr._programs.values())
    assert sum("launch replay" in str(w.message) for w in seen) == 1
    # a constant rate afterwards records and replays as usual, and the oldest entries make room for it
    for _ in range(4):
        tr.step(batch)
    torch.cuda.synchronize()
    assert len(tr._programs) <= 3 and any(st["prog"] is not None for st in tr._programs.values())
    assert torch.isfinite(m.params.P).all()


def test_encoder_update_beside_the_stem_backward_is_bit_identical(cuda, monkeypatch):
    """Single-rank step: the optimizer update of the encoder parameters (all but the stem's own) is enqueued on the side stream when the stem — the
    last node of the backward — reaches its weight gradient (Trainer._adam_encoders_early / Params.adam_early), instead of on the main stream
    after the backward.  Same kernels on the same gradients: parameters and optimizer state must equal the late update bit for bit, eager and
    replayed."""
    from multimodal_sc_amd.model import MultimodalSC, named_config
    from multimodal_sc_amd.trainer import Trainer
    from oracle import model as O
    batch = _to("cuda:0", O.synthetic_batch(O.named_config("C3"), 2, seed=5))
    batch["noise"] = batch["fading"] = None
    res = {}
    for early in ("0", "1"):
        monkeypatch.setenv("MSC_ADAM_EARLY", early)
        m = MultimodalSC(named_config("C3"), dtype="bf16", device="cuda:0", seed=3)
        tr = Trainer(m, lr=1e-3)
        assert tr.adam_early == (early == "1")
        for _ in range(5):
            tr.step(batch)
        torch.cuda.synchronize()
        assert m.params._early is None          # (the host-side step_count stops at the recorded step: replayed steps count on the device)
        res[early] = (m.params.P.clone(), m.params.M.clone(), m.params.V.clone(), m.params.C.clone(), m.params.TR.clone())
    for a, b in zip(res["0"], res["1"]):
        assert torch.equal(a, b)


def test_hipgraph_replay_matches_eager(cuda, monkeypatch):
    """Trainer.step captured into hipGraphs (opt-in, MSC_HIP_GRAPH=1: two eager warm-up steps, then replays) must leave
    bit-identical parameters to the eager launch path — same kernels, same order, step counter / noise seed read from device memory."""
    from multimodal_sc_amd import functional as F
    from multimodal_sc_amd.model import MultimodalSC, named_config
    from multimodal_sc_amd.trainer import Trainer
    from oracle import model as O
    monkeypatch.setattr(F, "_FUSION_CHAINS", True)
    monkeypatch.setattr(F, "_CHAIN_MIN_ROWS", 0)          # the fusion's two half-batch chains are part of what is captured
    monkeypatch.setenv("MSC_REPLAY", "0")                 # eager Python loop vs hipGraph (the launch replay has its own test above)
    res = {}
    for tag in ("0", "0b", "1"):
        mode = tag[0]
        monkeypatch.setenv("MSC_HIP_GRAPH", mode)
        m = MultimodalSC(named_config("C3"), dtype="bf16", device="cuda:0", seed=3)
        tr = Trainer(m, lr=1e-3)
        gb = _to(m.rt.device, O.synthetic_batch(O.named_config("C3"), 2, seed=5))
        gb["noise"] = gb["fading"] = None          # in-kernel Philox noise keyed by the device-side step counter
        for _ in range(6):
            terms = tr.step(gb)
        torch.cuda.synchronize()
        assert (tr._graphs is not None) == (mode == "1"), "graph path not taken / taken unexpectedly"
        res[tag] = (m.params.P.clone(), terms["sse"].clone())
    assert torch.equal(res["0"][0], res["0b"][0]), "eager multi-stream steps are not reproducible"
    assert torch.equal(res["0"][0], res["1"][0])
    assert torch.equal(res["0"][1], res["1"][1])
    assert torch.isfinite(res["1"][0]).all()


def test_ppo_loop_collect_and_update(cuda):
    """PPO loop around the path: rollout with sampled actions, GAE on the device, minibatch epochs through
    Trainer.step.  Checks bookkeeping (shapes, finite, returns = adv_raw + values), reproducibility (same seed -> same
    actions) and that the updates move the policy toward the rewarded action of the synthetic environment."""
    from multimodal_sc_amd.model import MultimodalSC, MSCConfig
    from multimodal_sc_amd.ppo import PPOTrainer, RolloutBuffer, SyntheticDrivingEnv
    cfg = MSCConfig(img_size=32, use_lidar=False, use_state=True)
    T, N = 8, 16

    def run(seed):
        torch.manual_seed(0)
        m = MultimodalSC(cfg, dtype="bf16", device="cuda:0")
        ppo = PPOTrainer(m, lr=3e-4, seed=seed)
        env = SyntheticDrivingEnv(cfg, N, "cuda:0", episode_len=4, seed=1)
        buf = RolloutBuffer(T, N, "cuda:0")
        ppo.collect(env, buf)
        return m, ppo, env, buf

    m, ppo, env, buf = run(11)
    _, _, _, buf2 = run(11)
    torch.cuda.synchronize()
    assert torch.equal(buf.actions, buf2.actions) and torch.equal(buf.logp, buf2.logp)
    assert buf.actions.min() >= 0 and buf.actions.max() < cfg.n_actions
    for t_ in (buf.logp, buf.values, buf.adv, buf.ret):
        assert torch.isfinite(t_).all()
    assert abs(buf.adv.mean().item()) < 1e-4 and abs(buf.adv.std().item() - 1) < 1e-3
    logs = ppo.update(buf, epochs=2, minibatch=64, generator=torch.Generator().manual_seed(3))
    assert len(logs) == 2 * (T * N // 64) and all(torch.isfinite(torch.tensor(l["loss"])) for l in logs)


def test_hip_path_against_committed_golden_vectors(cuda):
    """The HIP path (f32 mode) against tests/golden/oracle_c0_b8.npz — the frozen outputs of the oracle's C0 run — and the
    small-kernel vectors (GAE, sampled actions, wire bytes, e4m3 grid) bit-exactly.  No oracle code runs on this side
    except the seeded input generator."""
    import os
    import numpy as np
    from multimodal_sc_amd.model import MultimodalSC, named_config
    from multimodal_sc_amd.wire import pack_symbols
    from oracle import model as O
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    G = np.load(os.path.join(gdir, "oracle_c0_b8.npz"), allow_pickle=False)
    om = O.build_oracle(O.named_config("C0"), seed=1)             # weights only (seeded init), no forward
    batch = O.synthetic_batch(O.named_config("C0"), 8, seed=2)
    m = MultimodalSC(named_config("C0"), dtype="f32", device="cuda:0")
    m.load_oracle_state_dict(om.state_dict())
    gb = _to(m.rt.device, batch)
    out = m(gb["image"], None, None, gb["snr_db"], gb["noise"], gb["fading"])
    terms = m.loss_and_grads(out, gb["image"], gb["ppo"])
    torch.cuda.synchronize()
    logits, _ = m.logits_value(out)
    assert np.abs(logits.detach().cpu().numpy() - G["logits"]).max() < 1e-3
    assert np.abs(out["symbols"].detach().float().cpu().numpy() - G["symbols"]).max() < 1e-3
    met = m.metrics(terms)
    assert abs(met["psnr"] - float(G["psnr"].mean())) < 1e-3
    assert np.array_equal(m.select_action(out).cpu().numpy(), G["action"])
    for k, want in zip(("mse", "policy", "value", "entropy", "loss"), G["terms"]):
        assert abs(met[k] - want) < 1e-4 * max(1.0, abs(want)), (k, met[k], want)
    gg = m.export_state_dict(grads=True)
    norms = np.array([float(gg[str(k)].double().norm()) for k in G["grad_names"]])
    assert np.allclose(norms, G["grad_norm"], rtol=3e-2, atol=1e-7)

    S = np.load(os.path.join(gdir, "small_kernels.npz"), allow_pickle=False)
    rt, dev = m.rt, m.rt.device
    T, N = S["gae_rew"].shape
    adv, ret = torch.empty(T, N, device=dev), torch.empty(T, N, device=dev)
    rt.call("msc_gae", torch.from_numpy(S["gae_rew"]).to(dev), torch.from_numpy(S["gae_val"]).to(dev),
            torch.from_numpy(S["gae_done"]).to(dev), adv, ret, T, N, 0.99, 0.95)
    head = torch.zeros(64, 16)
    head[:, :9] = torch.from_numpy(S["samp_logits"])
    act = torch.empty(64, dtype=torch.int64, device=dev)
    logp, val = torch.empty(64, device=dev), torch.empty(64, device=dev)
    rt.call("msc_sample_actions", head.to(dev), act, logp, val, 64, 9, 16, 1234, None, 17, rt.dt)
    q, s = pack_symbols(rt, torch.from_numpy(S["wire_x"]).to(dev))
    n8 = S["e4m3_in"].shape[0] // 4 * 4
    x8 = torch.from_numpy(S["e4m3_in"][:n8]).to(dev) * (448.0 / float(np.abs(S["e4m3_in"][:n8]).max()))   # amax == 448 -> scale 1
    qb, dq, sc = torch.empty(n8, dtype=torch.uint8, device=dev), torch.empty(n8, device=dev), torch.empty(1, device=dev)
    rt.call("msc_fp8_quantize", x8, qb, dq, sc, n8, rt.dt, rt.dt)
    torch.cuda.synchronize()
    assert np.array_equal(adv.cpu().numpy(), S["gae_adv"]) and np.array_equal(ret.cpu().numpy(), S["gae_ret"])
    assert np.array_equal(act.cpu().numpy(), S["samp_action"]) and np.abs(logp.cpu().numpy() - S["samp_logp"]).max() < 1e-5
    assert np.array_equal(q.cpu().numpy(), S["wire_q"]) and np.array_equal(s.cpu().numpy(), S["wire_scale"])
    assert sc.item() == 1.0 and np.array_equal(dq.cpu().numpy(), O.e4m3_rne(x8.cpu()).numpy())


def test_loaded_library_was_built_from_checked_out_sources(cuda):
    """lib.msc_build_id() (embedded by csrc/Makefile) == the hash of the sources in this checkout, recomputed here: the .so that
    ran these tests on the GPU box is the one HEAD's kernels produce."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _buildid import source_build_id
    from multimodal_sc_amd import lib as L
    assert L.lib.msc_build_id().decode() == source_build_id()


@pytest.mark.parametrize("name,B", [("C0", 8), ("C3", 2)])
def test_checkpoint_round_trip_product_to_oracle(cuda, name, B):
    """SURVEY §8(f) row 4, the direction the other tests do not take: parameters INITIALISED AND UPDATED by the product
    (own seeded init, two fused-Adam steps) -> export_state_dict() -> oracle.load_state_dict(strict=False: the oracle also holds
    BatchNorm running buffers the product does not keep) -> the oracle's forward equals the product's (f32 mode, 1e-3; actions
    bit-exact), and loading the export back into a fresh product model reproduces the flat parameter buffer bit for bit."""
    from multimodal_sc_amd.model import MultimodalSC, named_config
    from oracle import model as O
    ocfg, cfg = O.named_config(name), named_config(name)
    m = MultimodalSC(cfg, dtype="f32", device="cuda:0", seed=7)
    batch = O.synthetic_batch(ocfg, B, seed=3)
    gb = _to(m.rt.device, batch)
    for _ in range(2):
        out = m(gb["image"], gb.get("points"), gb.get("state"), gb["snr_db"], gb["noise"], gb["fading"])
        m.loss_and_grads(out, gb["image"], gb["ppo"])
        m.params.adam_step(lr=1e-3)
    sd = m.export_state_dict()
    om = O.build_oracle(ocfg, seed=99)                    # different init: everything must come from the export
    missing, unexpected = om.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(("running_" in k or "num_batches" in k) for k in missing), missing
    oout = om(batch["image"], batch.get("points"), batch.get("state"), batch["snr_db"], batch["noise"], batch["fading"])
    out = m(gb["image"], gb.get("points"), gb.get("state"), gb["snr_db"], gb["noise"], gb["fading"])
    logits, value = m.logits_value(out)
    assert (logits.cpu() - oout["logits"]).abs().max() < 1e-3 and (value.cpu() - oout["value"]).abs().max() < 1e-3
    assert (m.recon_image(out).cpu() - oout["recon"]).abs().max() < 1e-3
    assert torch.equal(m.select_action(out).cpu(), O.select_action(oout["logits"]))
    m2 = MultimodalSC(cfg, dtype="f32", device="cuda:0", seed=1)
    m2.load_oracle_state_dict(sd)
    torch.cuda.synchronize()
    assert torch.equal(m2.params.P, m.params.P)


def test_ppo_clip_classification_fixed_seed(cuda):
    """A FIXED batch (C0, 8 frames, batch seed 2, no seed search): the clipped-above / unclipped / clipped-below classification of
    every PPO sample computed from the HIP path's logits equals the oracle's, both branches occur, and the policy-loss term agrees."""
    from multimodal_sc_amd.model import MultimodalSC, named_config
    from oracle import model as O
    ocfg, cfg = O.named_config("C0"), named_config("C0")
    om = O.build_oracle(ocfg, seed=1)
    batch = O.synthetic_batch(ocfg, 8, seed=2)
    batch["ppo"]["old_logp"] = batch["ppo"]["old_logp"] + torch.tensor([0.5, -0.5, 0.3, -0.3, 0.0, 0.0, 0.25, -0.25])   # spread the ratios
    oout, oterms = _oracle_run(om, batch, ocfg, True)
    m = MultimodalSC(cfg, dtype="f32", device="cuda:0")
    m.load_oracle_state_dict(om.state_dict())
    gb = _to(m.rt.device, batch)
    out = m(gb["image"], None, None, gb["snr_db"], gb["noise"], gb["fading"])
    terms = m.loss_and_grads(out, gb["image"], gb["ppo"])
    logits, _ = m.logits_value(out)
    cg, rg = _clip_class(logits, batch["ppo"], cfg.clip_eps)
    co, ro = _clip_class(oout["logits"], batch["ppo"], cfg.clip_eps)
    margin = float(torch.minimum((ro - 0.8).abs(), (ro - 1.2).abs()).min())
    print(f"fixed seed 2: ratios {[round(float(x), 4) for x in ro]}, classes {co.tolist()}, min distance to a clip boundary {margin:.4f}")
    assert margin > 1e-3                                   # (a property of the fixed batch; the classification below is the check)
    assert torch.equal(cg, co)
    assert (co != 0).any() and (co == 0).any()
    assert abs(m.metrics(terms)["policy"] - oterms["policy"]) < 1e-4
