#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the build-authored oracle (oracle/) in this container.

The upstream reference holds no code, tests or fixtures (README only), so there is nothing of the reference's to record;
these vectors pin the ORACLE itself (any later edit that changes its arithmetic shows up as a diff against them) and give
the HIP path a second, frozen target besides the live oracle run.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model as O          # noqa: E402
from oracle import philox, rl, wire    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def model_case(name, B, seed_model=1, seed_batch=2):
    torch.set_num_threads(1)           # one summation order, whatever the machine
    cfg = O.named_config(name)
    om = O.build_oracle(cfg, seed=seed_model)
    batch = O.synthetic_batch(cfg, B, seed=seed_batch)
    out = om(batch["image"], batch.get("points"), batch.get("state"), batch["snr_db"], batch["noise"], batch["fading"])
    loss, terms = O.total_loss(out, batch["image"], cfg, batch["ppo"])
    loss.backward()
    g = {k: p.grad for k, p in om.named_parameters()}
    names = sorted(g)
    rec = {
        "logits": out["logits"].detach().numpy(), "value": out["value"].detach().numpy(),
        "symbols": out["symbols"].detach().numpy(), "rx": out["rx"].detach().numpy(),
        "psnr": O.psnr(out["recon"], batch["image"]).detach().numpy(),
        "action": O.select_action(out["logits"]).numpy(),
        "terms": np.array([float(terms[k].detach()) for k in ("mse", "policy", "value", "entropy", "loss")], np.float64),
        # every parameter gradient by its L2 norm and its first 16 elements (the full tensors would be 40 MB)
        "grad_names": np.array(names),
        "grad_norm": np.array([float(g[k].double().norm()) for k in names], np.float64),
        "grad_head": np.stack([np.pad(g[k].flatten()[:16].numpy(), (0, max(0, 16 - g[k].numel()))) for k in names]),
    }
    return rec


def topk_case():
    """top-k action selection (north_star: 'bit-exact for index/top-k action selection'): hand-built rows with exact ties, signed
    zeros, infinities and NaN, plus random rows on the bf16 grid (so the same vectors serve both storage types)."""
    rng = np.random.default_rng(11)
    hand = np.array([[1, 3, 3, 2, 3, 0, 0, 0, 0], [5, 5, 5, 5, 5, 5, 5, 5, 5], [-1, -2, -1, -3, -1, -2, -9, -9, -9],
                     [0.0, -0.0, 0.0, -0.0, 1, 1, -1, -1, 0.0], [np.inf, 1, -np.inf, np.inf, 2, 2, -np.inf, 0, 0],
                     [np.nan, 1, np.nan, 1, 0, -np.inf, np.nan, 2, 2], [np.nan] * 9], np.float32)
    rand = torch.from_numpy(rng.normal(size=(200, 9)).astype(np.float32)).to(torch.bfloat16).float().numpy()
    coarse = np.round(rng.normal(size=(200, 9)) * 2).astype(np.float32) / 2        # many exact ties
    x = np.concatenate([hand, rand, coarse], 0)
    rec = {"x": x}
    for k in (1, 3, 8):
        idx, val = rl.topk_rows(x, k)
        rec[f"idx{k}"], rec[f"val{k}"] = idx, val
    return rec


def main():
    if "--only-new" in sys.argv:       # adds the fixtures of later rounds without touching the frozen earlier files
        np.savez_compressed(os.path.join(HERE, "topk.npz"), **topk_case())
        print("wrote topk.npz")
        return
    np.savez_compressed(os.path.join(HERE, "topk.npz"), **topk_case())
    np.savez_compressed(os.path.join(HERE, "oracle_c0_b8.npz"), **model_case("C0", 8))
    r = philox.philox4x32_10(np.array([0, 1, 0xFFFFFFFF], np.uint32), np.array([0, 2, 0xFFFFFFFF], np.uint32),
                             np.array([0, 3, 0xFFFFFFFF], np.uint32), np.array([0, 4, 0xFFFFFFFF], np.uint32), 0x0, 0x0)
    rng = np.random.default_rng(5)
    rew, val = rng.normal(size=(12, 5)).astype(np.float32), rng.normal(size=(13, 5)).astype(np.float32)
    done = (rng.random((12, 5)) < 0.2).astype(np.float32)
    adv, ret = rl.gae(rew, val, done, 0.99, 0.95)
    lg = rng.normal(size=(64, 9)).astype(np.float32)
    act, logp = rl.sample_actions(lg, 1234, 17)
    sym = rng.normal(size=(6, 32)).astype(np.float32)
    q, s = wire.pack(sym)
    v8 = np.concatenate([rng.normal(size=512).astype(np.float32) * 3, np.array([448, -448, 0, 2 ** -9, 1.0625, 1.1875], np.float32)])
    np.savez_compressed(os.path.join(HERE, "small_kernels.npz"), philox=np.stack(r), normals=philox.philox_normal(4, 42, 1),
                        gae_rew=rew, gae_val=val, gae_done=done, gae_adv=adv, gae_ret=ret,
                        samp_logits=lg, samp_action=act, samp_logp=logp, wire_x=sym, wire_q=q, wire_scale=s,
                        e4m3_in=v8, e4m3_out=O.e4m3_rne(torch.from_numpy(v8)).numpy())
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
