"""Seeded synthetic frames of the benchmark shape: camera image U[0,1) [B,3,H,W], LiDAR cloud [B,N,4]
(x,y U[-range,range), z U[-3,1), intensity U[0,1)), ego state N(0,1) [B,S], SNR, explicit channel noise / fading, and a
random PPO minibatch.  Pure torch.Generator code on the CPU — no model, no oracle.  bench.py's GPU leg, the tests and the
CPU oracle all draw their inputs from here, so both sides of every parity check see identical frames."""
from __future__ import annotations

import math

import torch


def synthetic_batch(cfg, B: int, seed: int = 0, snr_db: float = 10.0):
    """cfg: any object with img_size, use_lidar, n_points, bev_range, use_state, state_dim, n_symbols, n_actions."""
    g = torch.Generator().manual_seed(seed)
    b = {"image": torch.rand(B, 3, cfg.img_size, cfg.img_size, generator=g)}
    if cfg.use_lidar:
        xy = (torch.rand(B, cfg.n_points, 2, generator=g) * 2 - 1) * cfg.bev_range
        z = torch.rand(B, cfg.n_points, 1, generator=g) * 4 - 3
        it = torch.rand(B, cfg.n_points, 1, generator=g)
        b["points"] = torch.cat([xy, z, it], -1)
    if cfg.use_state:
        b["state"] = torch.randn(B, cfg.state_dim, generator=g)
    b["snr_db"] = torch.full((B,), float(snr_db))
    b["noise"] = torch.randn(B, cfg.n_symbols, generator=g)
    b["fading"] = torch.randn(B, cfg.n_symbols, generator=g)
    b["ppo"] = {
        "actions": torch.randint(0, cfg.n_actions, (B,), generator=g),
        "old_logp": -math.log(cfg.n_actions) + 0.1 * torch.randn(B, generator=g),
        "adv": torch.randn(B, generator=g),
        "ret": torch.randn(B, generator=g),
    }
    return b
