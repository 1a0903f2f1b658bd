"""Trainer.step: forward -> fused losses -> kernel backward chain -> DP gradient all-reduce (RCCL) -> fused Adam.

Host loop: one training step is ~300 C-ABI calls (~450 kernel launches) on three HIP streams (main chain, weight-gradient
side stream, LiDAR/BEV branch).  Default = LAUNCH REPLAY (MSC_REPLAY=0 turns it off): two eager warm-up steps, then one eager
step is RECORDED (every C-ABI call with its argument values, stream handle and event waits; activations come from a per-step
bump arena, so the addresses repeat) and every later step re-issues that list from C (csrc/replay.hip) — one ctypes call per
segment instead of ~12 us of Python per launch; segments end where the host must act (the DP all-reduces).  Same kernels, same
streams, same order: bit-identical to the eager loop (tests/test_model_gpu.py::test_launch_replay_matches_eager).  A step with a
different batch signature or learning rate records its own program.  Tensors a step returns (loss terms) live in the arena and
are valid until the next step starts.
hipGraph capture of the step is implemented and bit-identical to eager, but OPT-IN (MSC_HIP_GRAPH=1) because it measured
SLOWER on this stack (MI355X, ROCm 7.2, C3 @ 64 frames: replay 10.6 ms/step vs eager 9.7): the replay does not run the
captured branches concurrently.  With the knob set, after two eager warm-up steps the step is captured once and replayed:
    world == 1 : ONE graph  = forward + losses + backward (both phases) + fused Adam
    world  > 1 : graph A    = forward + losses + upper backward (heads / codec / transformer)
                 all-reduce of the upper gradients (async), graph B = encoder backward under it,
                 all-reduce of the encoder gradients, wait, graph C = fused Adam
Step counter and channel-noise draw counter live in device memory (msc_counter_add), so a replay is a NEW step; inputs are
static device buffers.  If capture fails the trainer falls back to eager launches, loudly.

Data parallelism: one process per GPU, frames sharded over ranks (BatchNorm statistics stay per-GPU).  The only
exchange step of the path is the gradient sum over the flat fp32 gradient buffer G, with torch.distributed (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests) in three large pieces (GradExchange below):
  upper  G[fusion.pos_embed:]  (transformer, semantic codec, heads: ~56 MB) — final after backward phase 1; its
         all-reduce runs under the whole encoder backward (phase 2);
  mid    G[layer4:fusion.pos_embed]  (top of the camera trunk, BEV branch, state encoder: ~40 MB) — final once the first
         two phases of the encoder backward are done; reduced under the rest of the trunk's backward;
  rest   G[:layer4]  (stem .. layer3: ~11 MB) — the only piece left exposed after the backward.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): few, large messages let RCCL drive all links; small buckets
would be latency-bound per link.  MSC_DP_OVERLAP=0 falls back to one non-overlapped bucketed all-reduce.
"""
from __future__ import annotations

import collections
import os
import warnings

import torch
import torch.distributed as dist

from .model import MultimodalSC
from .replay import Program, Recorder
from .runtime import Arena


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


class GradExchange:
    """Slice bounds and issue order of the DP gradient sum over the flat buffer G — the only exchange step of the path.
    Kept free of model and stream details so the schedule that ships can be driven by a fake backward on CPU tensors
    (tests/test_dp_cpu.py, gloo): every rank must issue the same collectives over the same slices in the same order,
    and the slices must tile [0, total) exactly once.
        pieces >= 3 and mid is not None :  upper [split,total) -> mid [mid,split) -> rest [0,mid)
        otherwise                        :  upper [split,total) -> rest [0,split)
    `log` records (name, lo, hi) per issued collective."""

    def __init__(self, total: int, split: int, mid=None, pieces: int = 3, group=None):
        assert 0 < split <= total and (mid is None or 0 < mid < split)
        self.total, self.split, self.group = total, split, group
        self.mid = mid if pieces >= 3 else None
        self.log = []
        self._issued_mid = False

    def bounds(self, name):
        if name == "upper":
            return self.split, self.total
        if name == "mid":
            assert self.mid is not None
            return self.mid, self.split
        assert name == "rest"
        return 0, (self.mid if self._issued_mid else self.split)

    def start(self, G, name):
        """enqueue the asynchronous all-reduce (sum) of one piece on the CURRENT stream's ordering; returns the work handle"""
        lo, hi = self.bounds(name)
        if name == "mid":
            self._issued_mid = True
        self.log.append((name, lo, hi))
        return dist.all_reduce(G[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def begin_step(self):
        self.log.clear()
        self._issued_mid = False

    def covered(self):
        """True when the pieces issued since begin_step() tile [0, total) exactly once"""
        iv = sorted((lo, hi) for _, lo, hi in self.log)
        return bool(iv) and iv[0][0] == 0 and iv[-1][1] == self.total and all(a[1] == b[0] for a, b in zip(iv, iv[1:]))


def _flatten(batch, prefix=""):
    for k, v in batch.items():
        if isinstance(v, dict):
            yield from _flatten(v, prefix + k + ".")
        elif torch.is_tensor(v):
            yield prefix + k, v


class Trainer:
    MAX_PROGRAMS = 8          # recorded launch lists kept (least recently used dropped)

    def __init__(self, model: MultimodalSC, lr: float = 1e-4, n_buckets: int = 4, use_ppo: bool = True):
        self.model = model
        self.lr = lr
        self.n_buckets = n_buckets
        self.use_ppo = use_ppo
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # data-parallel code path (two-phase backward, GradExchange, replay segments cut at the collectives).  MSC_DP_FORCE=1 takes it
        # with a ONE-rank process group too: the only way RCCL itself can execute this schedule on a one-GPU box (tests/test_dp_gpu.py)
        self.dp = self.world > 1 or (os.environ.get("MSC_DP_FORCE") == "1" and dist.is_available() and dist.is_initialized())
        if self.world > 1:
            model.channel_rank = dist.get_rank()      # every rank draws its own channel noise (Philox key word)
        self.overlap = os.environ.get("MSC_DP_OVERLAP", "1") != "0"
        self.split = model.params.refs["fusion.pos_embed"].off
        self.use_graph = os.environ.get("MSC_HIP_GRAPH", "0") == "1"
        if self.use_graph:
            model.rt.host_counters = False     # a replayed launch sequence needs its step / noise-draw numbers on the device
        self.dp_pieces = int(os.environ.get("MSC_DP_PIECES", "3"))    # 2: upper + encoders; 3: upper + mid + rest of the trunk
        if self.dp and getattr(model, "_img_cut_auto", False) and self.overlap and self.dp_pieces >= 3 and not self.use_graph:
            # boundary inside the camera trunk: the exchange's mid piece starts there.  Only the schedule that issues a mid piece
            # pays for the boundary (one GPU: 8.12 -> 8.25 ms); two-piece, non-overlapped and hipGraph DP keep the uncut trunk
            model._img_cut = 3
        self.adam_overlap = os.environ.get("MSC_ADAM_OVERLAP", "1") != "0" and not self.use_graph
        # single-rank step: the encoder half of the update (minus the stem's own parameters) on the side stream beside the stem's backward
        self.adam_early = os.environ.get("MSC_ADAM_EARLY", "1") != "0" and self.adam_overlap and not self.dp
        self.use_replay = os.environ.get("MSC_REPLAY", "1") != "0" and not self.use_graph
        self.replay_warmup = int(os.environ.get("MSC_REPLAY_WARMUP", "2"))   # eager (arena) steps before the recorded one
        if self.use_replay:
            model.rt.host_counters = False     # as for hipGraph replay: step / noise-draw numbers on the device
        self._main = None
        if os.environ.get("MSC_MAIN_PRIO"):      # experiment knob: +1 = highest priority of the device for the main chain
            from . import lib as L
            with torch.cuda.device(model.rt.device):
                h = L.query("msc_stream_create", int(os.environ["MSC_MAIN_PRIO"]))
            self._main = torch.cuda.ExternalStream(h, device=model.rt.device)
        self._arena = None        # runtime.Arena of this trainer's steps (replay mode)
        # (batch signature, lr, loss mode) -> {"seen": eager steps so far, "prog": Program, "static": batch, "terms": ...}; least recently
        # used first, at most MAX_PROGRAMS entries: the learning rate is a baked argument of the recorded optimizer launch, so a per-step
        # schedule makes a new key every step — those never reach a recording (the step stays eager, said once) and must not pile up;
        # a piecewise schedule keeps its few programs, each pinning its own ~55 MB input clone
        self._programs = collections.OrderedDict()
        self._lru_warned = False
        self._rec = None          # active Recorder while a step is being recorded
        self.exchange = None      # GradExchange, built on the first multi-rank step (needs the model's mid offset)
        self._seen = 0
        self._graphs = None       # (gA, gB, gC) once captured
        self._static = None       # the batch whose tensors the graphs read
        self._terms = None

    # ------------------------------------------------------------------ pieces of a step
    def _forward_upper(self, batch):
        m = self.model
        out = m(batch["image"], batch.get("points"), batch.get("state"), batch["snr_db"], batch.get("noise"), batch.get("fading"))
        two_phase = self.dp and self.overlap
        terms = m.loss_and_grads(out, batch["image"], batch.get("ppo") if self.use_ppo else None, lower=not two_phase)
        if two_phase:
            m.rt.join_side()          # upper weight gradients (side stream) are final behind this point
        return out, terms

    def forward_backward(self, batch, adam_early=False):
        """forward + complete backward (no collective) — used by the eager path, bench.py's instrumented pass, tests.
        adam_early (single-rank step; the caller runs _adam() next): when the last node of the backward — the ImageEncoder stem — starts
        enqueueing, every other encoder gradient has its producer enqueued; the update of those parameters goes onto the side stream right
        there (behind their weight gradients) and runs beside the stem's backward instead of on the main stream after it."""
        m = self.model
        out = m(batch["image"], batch.get("points"), batch.get("state"), batch["snr_db"], batch.get("noise"), batch.get("fading"))
        m.rt.on_stem_backward = self._adam_encoders_early if adam_early else None
        try:
            terms = m.loss_and_grads(out, batch["image"], batch.get("ppo") if self.use_ppo else None)
        finally:
            m.rt.on_stem_backward = None
        return out, terms

    def _adam_encoders_early(self):
        m, rt = self.model, self.model.rt
        first = m.params.refs.get("image_encoder.layer1.0.conv1.weight")
        if rt.side is None or first is None or first.off >= self.split:
            return
        if rt._branch is not None:      # the BEV branch's gradients / its data gradients' weight reads (call_side orders the side stream behind main)
            rt.stream_wait(rt._side_raw, rt._branch.cuda_stream)
        m.params.adam_early(first.off, self.split, lr=self.lr, grad_scale=1.0 / self.world)

    def _allreduce_mid(self):
        """all-reduce of G[mid:split] (top of the camera trunk, BEV branch, state encoder: ~41 of the 52 MB of encoder
        gradients) started while the rest of the trunk's backward is still being enqueued.  Its producers run on three
        streams (main, BEV branch, side).  The collective is issued with the BEV-branch stream current — that stream has
        nothing left to do once its backward is enqueued — after making it wait for the other two, so the main stream is
        not held up and NO further stream is created: with five or more streams active the HIP runtime on this stack
        serialises badly (measured: a third side stream took the step from 9.1 to 16 ms)."""
        m, rt = self.model, self.model.rt
        s2 = rt._branch
        rt.stream_wait(s2.cuda_stream, rt.stream)
        if rt.side is not None:
            rt.stream_wait(s2.cuda_stream, rt._side_raw)
        self._cut("mid")
        return self._act_mid()

    def _act_mid(self):
        with torch.cuda.stream(self.model.rt._branch):
            return self.exchange.start(self.model.params.G, "mid")

    def _cut(self, label):
        """end of a replay segment: the host acts here (label names the action that follows); no-op unless a step is being recorded"""
        if self._rec is not None:
            self._rec.cut(label)

    def _adam(self):
        # the upper half of the parameters (transformer / codec / heads) is updated on the side stream: that update then
        # overlaps the encoder trunks of the next forward instead of sitting on the critical chain (MSC_ADAM_OVERLAP=0: off)
        self.model.params.adam_step(lr=self.lr, grad_scale=1.0 / self.world, overlap_from=self.split if self.adam_overlap else None)

    def _eager_step(self, batch):
        m, G = self.model, self.model.params.G
        if not self.dp:
            _, terms = self.forward_backward(batch, adam_early=self.adam_early)
        elif self.overlap:
            _, terms = self._forward_upper(batch)
            if self.exchange is None:
                self.exchange = GradExchange(G.numel(), self.split, m.mid_offset(), self.dp_pieces)
            ex = self.exchange
            ex.begin_step()
            self._cut("upper")
            works = [ex.start(G, "upper")]
            # on_mid fires only when the encoder backward really ran its first two phases separately (C1 has no BEV
            # branch: two pieces); "rest" then ends at mid, else at split — GradExchange.bounds tracks which
            m.backward_lower(on_mid=(lambda: works.append(self._allreduce_mid())) if ex.mid is not None else None)
            self._cut("rest")
            self._act_rest(works)
        else:
            _, terms = self.forward_backward(batch)
            self._cut("flat")
            self._act_flat()
        self._adam()
        return terms

    def _act_rest(self, works):
        ex = self.exchange
        works.append(ex.start(self.model.params.G, "rest"))
        assert ex.covered(), ex.log
        for w in works:
            w.wait()

    def _act_flat(self):
        for w in allreduce_flat(self.model.params.G, self.n_buckets):
            w.wait()

    # ------------------------------------------------------------------ launch replay (default host loop)
    @staticmethod
    def _signature(batch, prefix=""):
        sig = []
        for k in sorted(batch):
            v = batch[k]
            if isinstance(v, dict):
                sig += Trainer._signature(v, prefix + k + ".")
            elif torch.is_tensor(v):
                sig.append((prefix + k, tuple(v.shape), str(v.dtype), v.data_ptr() % 16))
            else:
                sig.append((prefix + k, repr(v)))        # None (in-kernel noise) vs a tensor changes the launch arguments
        return tuple(sig) if not prefix else sig

    def _replay_step(self, batch):
        rt = self.model.rt
        key = (self._signature(batch), float(self.lr), self.use_ppo)
        st = self._programs.get(key)
        if st is None:
            st = self._programs[key] = {"seen": 0, "prog": None, "static": None, "terms": None}
            while len(self._programs) > self.MAX_PROGRAMS:
                _, old = self._programs.popitem(last=False)      # the Program's destructor frees the C-side list and its kept tensors
                if old["prog"] is None and not self._lru_warned:
                    self._lru_warned = True
                    warnings.warn("launch replay: a (batch shape, learning rate, loss mode) key was dropped before it was recorded — with a "
                                  "learning rate that changes every step the trainer stays on the eager host loop", RuntimeWarning)
        else:
            self._programs.move_to_end(key)
        if self._arena is None:
            self._arena = Arena(rt.device)
        if st["prog"] is not None:
            return self._run_program(batch, st)
        self._arena.reset()
        rt.arena = self._arena
        try:
            st["seen"] += 1
            if st["seen"] <= self.replay_warmup:  # warm-up: workspaces reach their size, streams and events exist
                return self._eager_step(batch)
            # the recorded launches read the batch from buffers the trainer owns (the caller's tensors stay the caller's)
            static = self._clone_batch(batch)
            rt.ws_grew = False
            with Recorder() as rec:
                self._rec = rec
                try:
                    terms = self._eager_step(static)
                finally:
                    self._rec = None
            if rt.ws_grew:                       # a workspace moved during the step: its old address is in the list — record again
                return terms
            st["prog"], st["static"], st["terms"] = Program(rec), static, terms
            return terms
        finally:
            rt.arena = None

    @staticmethod
    def _clone_batch(batch):
        return {k: (Trainer._clone_batch(v) if isinstance(v, dict) else (v.clone() if torch.is_tensor(v) else v)) for k, v in batch.items()}

    def input_buffers(self, batch):
        """The trainer-owned input tensors the recorded launch list of `batch`'s shape reads (same nesting and keys as `batch`), or None
        while no list is recorded for it (the first steps of a shape; eager and hipGraph modes).  A data source that WRITES its frames
        into these tensors and passes them to step() pays no staging copy: step() copies only tensors that are not already these."""
        if not self.use_replay:
            return None
        st = self._programs.get((self._signature(batch), float(self.lr), self.use_ppo))
        return st["static"] if st is not None and st["prog"] is not None else None

    def _run_program(self, batch, st):
        # the batch goes into the static input buffers the recorded launches read (55 MB device-to-device at 64 frames: ~60 us of copy
        # kernels) — unless the caller filled those buffers itself (input_buffers)
        dst = dict(_flatten(st["static"]))
        for k, v in _flatten(batch):
            if v.data_ptr() != dst[k].data_ptr():
                dst[k].copy_(v, non_blocking=True)
        prog = st["prog"]
        self.model.rt._side_dirty = True         # the step leaves work on the side stream (upper half of the optimizer update)
        if prog.n_segments == 1:
            prog.run(-1)
            return st["terms"]
        works = []
        for i, label in enumerate(prog.labels):
            prog.run(i)
            if label == "upper":
                self.exchange.begin_step()
                works = [self.exchange.start(self.model.params.G, "upper")]
            elif label == "mid":
                works.append(self._act_mid())
            elif label == "rest":
                self._act_rest(works)
            elif label == "flat":
                self._act_flat()
        return st["terms"]

    # ------------------------------------------------------------------ hipGraph capture / replay
    def _capture(self, batch):
        m = self.model
        self._static = batch
        torch.cuda.synchronize(m.rt.device)
        gA, gB, gC = torch.cuda.CUDAGraph(), None, None
        pool = None
        if not self.dp:
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
        if self.dp:
            if gB is not None:
                ex = GradExchange(G.numel(), self.split, None, 2)
                works = [ex.start(G, "upper")]
                gB.replay()
                works.append(ex.start(G, "rest"))
            else:
                works = allreduce_flat(G, self.n_buckets)
            for w in works:
                w.wait()
            gC.replay()
        return self._terms

    def step(self, batch):
        """One optimisation step on this rank's shard of the batch; returns the (device-side) loss terms."""
        if self._main is None:
            return self._step(batch)
        # MSC_MAIN_PRIO: the step's main chain on a stream of its own priority instead of the caller's (default) stream
        rt = self.model.rt
        cur = torch.cuda.current_stream(rt.device)
        rt.stream_wait(self._main.cuda_stream, cur.cuda_stream)
        with torch.cuda.stream(self._main):
            out = self._step(batch)
        rt.stream_wait(cur.cuda_stream, self._main.cuda_stream)
        return out

    def _step(self, batch):
        if self.model.rt.profile is not None or self.model.rt.taps is not None:
            return self._eager_step(batch)
        if self.use_replay:
            return self._replay_step(batch)
        if not self.use_graph:
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
