"""Trainer.step: forward -> fused losses -> kernel backward chain -> DP gradient all-reduce (RCCL) -> fused Adam.

Data parallelism: one process per GPU, frames sharded over ranks (BatchNorm statistics stay per-GPU).  The only
exchange step of the path is the gradient sum over the flat fp32 gradient buffer G, done with torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests) in two large pieces so that most of it hides
under compute:
  upper  G[fusion.pos_embed:]  (transformer, semantic codec, heads: ~56 MB) — final as soon as the backward leaves
         the fusion transformer; its all-reduce is enqueued from inside the backward (Runtime.on_upper_grads_ready)
         and runs on RCCL's stream under the whole encoder (ResNet-18 + BEV) backward that follows;
  lower  G[:fusion.pos_embed]  (image / BEV / state encoders: ~52 MB) — reduced after the backward.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): few, large messages let RCCL drive all links; small buckets
would be latency-bound per link.  MSC_DP_OVERLAP=0 falls back to one non-overlapped bucketed all-reduce.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .model import MultimodalSC


def bucket_bounds(total: int, n_buckets: int, align: int = 64):
    """Split [0,total) into <= n_buckets contiguous ranges whose boundaries are multiples of `align`."""
    n_buckets = max(1, n_buckets)
    per = (total + n_buckets - 1) // n_buckets
    per = (per + align - 1) // align * align
    out, lo = [], 0
    while lo < total:
        hi = min(total, lo + per)
        out.append((lo, hi))
        lo = hi
    return out


def allreduce_flat(flat: torch.Tensor, n_buckets: int = 4, group=None, async_op: bool = True):
    """Sum `flat` over the process group in large contiguous buckets; returns the work handles."""
    works = []
    for lo, hi in bucket_bounds(flat.numel(), n_buckets):
        works.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=async_op))
    return works


class Trainer:
    def __init__(self, model: MultimodalSC, lr: float = 1e-4, n_buckets: int = 4, use_ppo: bool = True):
        self.model = model
        self.lr = lr
        self.n_buckets = n_buckets
        self.use_ppo = use_ppo
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.overlap = os.environ.get("MSC_DP_OVERLAP", "1") != "0"
        self.split = model.params.refs["fusion.pos_embed"].off

    def forward_backward(self, batch):
        m = self.model
        out = m(batch["image"], batch.get("points"), batch.get("state"), batch["snr_db"], batch.get("noise"), batch.get("fading"))
        terms = m.loss_and_grads(out, batch["image"], batch.get("ppo") if self.use_ppo else None)
        return out, terms

    def step(self, batch):
        """One optimisation step on this rank's shard of the batch; returns the (device-side) loss terms."""
        m = self.model
        G = m.params.G
        works = []
        if self.world > 1 and self.overlap:
            def upper_ready():
                m.rt.join_side()     # the upper weight gradients were produced on the side stream
                works.append(dist.all_reduce(G[self.split:], op=dist.ReduceOp.SUM, async_op=True))
            m.rt.on_upper_grads_ready = upper_ready
        try:
            _, terms = self.forward_backward(batch)
        finally:
            m.rt.on_upper_grads_ready = None
        if self.world > 1:
            if works:
                works.append(dist.all_reduce(G[: self.split], op=dist.ReduceOp.SUM, async_op=True))
            else:
                works = allreduce_flat(G, self.n_buckets)
            for w in works:
                w.wait()
        m.params.adam_step(lr=self.lr, grad_scale=1.0 / self.world)
        return terms
