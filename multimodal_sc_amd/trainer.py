"""Trainer.step: forward -> fused losses -> kernel backward chain -> DP gradient all-reduce (RCCL) -> fused Adam.

Launch-bound host loop -> hipGraphs.  One training step is ~450 kernel launches on three HIP streams (main chain,
weight-gradient side stream, LiDAR/BEV branch); issuing them from Python costs ~7 ms of host time against ~9 ms of GPU
time, which both caps the step rate and starves the side streams of work to overlap.  After two eager warm-up steps
(workspaces sized, lazy initialisation done) the step is captured once into hipGraphs and replayed:
    world == 1 : ONE graph  = forward + losses + backward (both phases) + fused Adam
    world  > 1 : graph A    = forward + losses + upper backward (heads / codec / transformer)
                 all-reduce of the upper gradients enqueued on RCCL's stream (async)
                 graph B    = lower backward (ResNet-18 / BEV / state encoders) — overlaps that all-reduce
                 all-reduce of the lower gradients, wait, graph C = fused Adam
Step counter and channel-noise seed live in device memory (msc_counter_add), so a replay is a NEW step.  Inputs are
static device buffers (a different batch is copied into them).  If capture fails the trainer falls back to eager
launches.
MEASURED (MI355X, ROCm 7.2, C3 @ 64 frames): replay 10.61 ms/step vs eager 9.66 ms/step, bit-identical results — the
replay runs the three captured streams' branches no more concurrently than a single stream does (its time equals the
eager time with the side streams disabled), while the eager host loop (7.2 ms of enqueue per step) still keeps ahead
of the GPU.  The graph path is therefore OPT-IN (MSC_HIP_GRAPH=1); the default is eager multi-stream launches.

Data parallelism: one process per GPU, frames sharded over ranks (BatchNorm statistics stay per-GPU).  The only
exchange step of the path is the gradient sum over the flat fp32 gradient buffer G, with torch.distributed (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests) in two large pieces:
  upper  G[fusion.pos_embed:]  (transformer, semantic codec, heads: ~56 MB) — final after backward phase 1; its
         all-reduce runs under the whole encoder backward (phase 2);
  mid    G[layer4:fusion.pos_embed]  (top of the camera trunk, BEV branch, state encoder: ~40 MB) — final once the first
         two phases of the encoder backward are done; reduced under the rest of the trunk's backward;
  rest   G[:layer4]  (stem .. layer3: ~11 MB) — the only piece left exposed after the backward.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): few, large messages let RCCL drive all links; small buckets
would be latency-bound per link.  MSC_DP_OVERLAP=0 falls back to one non-overlapped bucketed all-reduce.
"""
from __future__ import annotations

import os
import warnings

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


def _flatten(batch, prefix=""):
    for k, v in batch.items():
        if isinstance(v, dict):
            yield from _flatten(v, prefix + k + ".")
        elif torch.is_tensor(v):
            yield prefix + k, v


class Trainer:
    def __init__(self, model: MultimodalSC, lr: float = 1e-4, n_buckets: int = 4, use_ppo: bool = True):
        self.model = model
        self.lr = lr
        self.n_buckets = n_buckets
        self.use_ppo = use_ppo
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.overlap = os.environ.get("MSC_DP_OVERLAP", "1") != "0"
        self.split = model.params.refs["fusion.pos_embed"].off
        self.use_graph = os.environ.get("MSC_HIP_GRAPH", "0") == "1"
        self.dp_pieces = int(os.environ.get("MSC_DP_PIECES", "3"))    # 2: upper + encoders; 3: upper + mid + rest of the trunk
        self.adam_overlap = os.environ.get("MSC_ADAM_OVERLAP", "1") != "0" and not self.use_graph
        self._seen = 0
        self._graphs = None       # (gA, gB, gC) once captured
        self._static = None       # the batch whose tensors the graphs read
        self._terms = None

    # ------------------------------------------------------------------ pieces of a step
    def _forward_upper(self, batch):
        m = self.model
        out = m(batch["image"], batch.get("points"), batch.get("state"), batch["snr_db"], batch.get("noise"), batch.get("fading"))
        two_phase = self.world > 1 and self.overlap
        terms = m.loss_and_grads(out, batch["image"], batch.get("ppo") if self.use_ppo else None, lower=not two_phase)
        if two_phase:
            m.rt.join_side()          # upper weight gradients (side stream) are final behind this point
        return out, terms

    def forward_backward(self, batch):
        """forward + complete backward (no collective) — used by the eager path, bench.py's instrumented pass, tests"""
        m = self.model
        out = m(batch["image"], batch.get("points"), batch.get("state"), batch["snr_db"], batch.get("noise"), batch.get("fading"))
        terms = m.loss_and_grads(out, batch["image"], batch.get("ppo") if self.use_ppo else None)
        return out, terms

    def _allreduce_mid(self, mid):
        """all-reduce of G[mid:split] (top of the camera trunk, BEV branch, state encoder: ~41 of the 52 MB of encoder
        gradients) started while the rest of the trunk's backward is still being enqueued.  Its producers run on three
        streams (main, BEV branch, side).  The collective is issued with the BEV-branch stream current — that stream has
        nothing left to do once its backward is enqueued — after making it wait for the other two, so the main stream is
        not held up and NO further stream is created: with five or more streams active the HIP runtime on this stack
        serialises badly (measured: a third side stream took the step from 9.1 to 16 ms)."""
        m, rt = self.model, self.model.rt
        s2, main = rt._branch, torch.cuda.current_stream(rt.device)
        s2.wait_stream(main)
        if rt.side is not None:
            s2.wait_stream(rt.side)
        with torch.cuda.stream(s2):
            return dist.all_reduce(m.params.G[mid:self.split], op=dist.ReduceOp.SUM, async_op=True)

    def _adam(self):
        # the upper half of the parameters (transformer / codec / heads) is updated on the side stream: that update then
        # overlaps the encoder trunks of the next forward instead of sitting on the critical chain (MSC_ADAM_OVERLAP=0: off)
        self.model.params.adam_step(lr=self.lr, grad_scale=1.0 / self.world, overlap_from=self.split if self.adam_overlap else None)

    def _eager_step(self, batch):
        m, G = self.model, self.model.params.G
        if self.world == 1:
            _, terms = self.forward_backward(batch)
        elif self.overlap:
            _, terms = self._forward_upper(batch)
            works = [dist.all_reduce(G[self.split:], op=dist.ReduceOp.SUM, async_op=True)]
            mid = m.mid_offset() if self.dp_pieces >= 3 else None
            m.backward_lower(on_mid=(lambda: works.append(self._allreduce_mid(mid))) if mid is not None else None)
            lo_end = mid if (mid is not None and len(works) == 2) else self.split
            works.append(dist.all_reduce(G[:lo_end], op=dist.ReduceOp.SUM, async_op=True))
            for w in works:
                w.wait()
        else:
            _, terms = self.forward_backward(batch)
            for w in allreduce_flat(G, self.n_buckets):
                w.wait()
        self._adam()
        return terms

    # ------------------------------------------------------------------ hipGraph capture / replay
    def _capture(self, batch):
        m = self.model
        self._static = batch
        torch.cuda.synchronize(m.rt.device)
        gA, gB, gC = torch.cuda.CUDAGraph(), None, None
        pool = None
        if self.world == 1:
            with torch.cuda.graph(gA):
                _, terms = self.forward_backward(batch)
                self._adam()
        else:
            two_phase = self.overlap
            with torch.cuda.graph(gA):
                if two_phase:
                    _, terms = self._forward_upper(batch)
                else:
                    _, terms = self.forward_backward(batch)
            pool = gA.pool()
            if two_phase:
                gB = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gB, pool=pool):
                    m.backward_lower()
            gC = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gC, pool=pool):
                self._adam()
        self._graphs = (gA, gB, gC)
        self._terms = terms
        m.rt.capture_keep.clear()
        # capture only RECORDS the launches: the step this call stands for still has to run
        return self._replay(batch)

    def _replay(self, batch):
        if batch is not self._static:     # a new batch: copy it into the static input buffers the graphs read
            dst = dict(_flatten(self._static))
            for k, v in _flatten(batch):
                if dst[k].data_ptr() != v.data_ptr():
                    dst[k].copy_(v, non_blocking=True)
        gA, gB, gC = self._graphs
        G = self.model.params.G
        gA.replay()
        if self.world > 1:
            if gB is not None:
                works = [dist.all_reduce(G[self.split:], op=dist.ReduceOp.SUM, async_op=True)]
                gB.replay()
                works.append(dist.all_reduce(G[: self.split], op=dist.ReduceOp.SUM, async_op=True))
            else:
                works = allreduce_flat(G, self.n_buckets)
            for w in works:
                w.wait()
            gC.replay()
        return self._terms

    def step(self, batch):
        """One optimisation step on this rank's shard of the batch; returns the (device-side) loss terms."""
        if not self.use_graph or self.model.rt.profile is not None:
            return self._eager_step(batch)
        if self._graphs is None:
            self._seen += 1
            if self._seen <= 2:
                return self._eager_step(batch)
            try:
                return self._capture(batch)
            except Exception as e:  # noqa: BLE001 — any capture problem: stay on the eager path, loudly
                warnings.warn(f"hipGraph capture failed ({type(e).__name__}: {e}); continuing with eager launches")
                self.use_graph = False
                self._graphs = None
                torch.cuda.synchronize(self.model.rt.device)
                return self._eager_step(batch)
        return self._replay(batch)
