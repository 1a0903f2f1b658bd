"""Runtime context + flat parameter store for the MI355X path.

Memory layout in HBM (one process per GPU):
  P   fp32 master parameters, one flat buffer (views per tensor, 64-element aligned)
  G   fp32 gradients, same offsets (filled directly by the wgrad / reduction kernels, no autograd adds)
  C   compute copies in the storage dtype (bf16 mode: written by the fused Adam kernel; fp32 mode: C is P)
  TR  transposed compute copies used by the dgrad GEMMs ([Cin][KH*KW][Cout] for convs, [in][out] for linears)
  M,V Adam moments (fp32)
Flat buffers make the DP all-reduce a handful of large RCCL calls and the optimizer one kernel.
"""
from __future__ import annotations

import os

import torch

from . import lib as L


class ParamRef:
    """Handle the kernels use: .p fp32 master, .c compute copy, .t transposed compute copy (or None), .g fp32 grad."""
    __slots__ = ("name", "shape", "kind", "off", "numel", "p", "c", "t", "g")

    def __init__(self, name, shape, kind, off):
        self.name, self.shape, self.kind, self.off = name, tuple(shape), kind, off
        n = 1
        for s in shape:
            n *= s
        self.numel = n
        self.p = self.c = self.t = self.g = None


_raw_stream = torch._C._cuda_getCurrentRawStream


class Arena:
    """Per-step bump allocator for activations: 2 GiB chunks of HBM (never freed), a bump pointer that Trainer.step resets at the
    start of every step.  The same sequence of allocations therefore returns the same device addresses step after step — what
    lets a recorded launch list be replayed (multimodal_sc_amd/replay.py) — and nothing is handed back and re-used inside a
    step, so no buffer of a step can be re-used while another stream still reads it.  Sized for 288 GB of HBM: a 64-frame C3
    step keeps every activation of the step (~10 GB) instead of recycling them."""
    CHUNK = 1 << 31
    ALIGN = 256

    def __init__(self, device):
        self.device = device
        self.chunks = []
        self.ci = self.off = 0
        self.peak = 0

    def reset(self):
        self.ci = self.off = 0

    def alloc(self, shape, dtype):
        n = dtype.itemsize
        for s in shape:
            n *= s
        n_al = (max(n, 1) + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        while True:
            if self.ci == len(self.chunks):
                self.chunks.append(torch.empty(max(self.CHUNK, n_al), dtype=torch.uint8, device=self.device))
            c = self.chunks[self.ci]
            if self.off + n_al <= c.numel():
                break
            self.ci += 1
            self.off = 0
        t = c[self.off:self.off + n].view(dtype).view(shape)
        self.off += n_al
        self.peak = max(self.peak, self.ci * self.CHUNK + self.off)
        return t


class Runtime:
    def __init__(self, dtype: str = "bf16", device: str = "cuda:0"):
        assert dtype in ("bf16", "f32")
        self.dt = L.MSC_BF16 if dtype == "bf16" else L.MSC_F32
        self.tdtype = torch.bfloat16 if dtype == "bf16" else torch.float32
        self.vec = 8 if dtype == "bf16" else 4
        self.device = torch.device(device)
        self._dev_index = self.device.index if self.device.index is not None else 0
        self.accumulate = 0          # 1: parameter-gradient kernels add into G instead of overwriting
        self._ws = {}
        self._ws_retired = []        # outgrown workspaces (never freed: see workspace())
        self.side_also_waits = None  # (raw stream, event): call_side() also orders the side stream behind that stream
        self._chain = None           # chain_b(): raw BEV-branch stream + the three events of the fusion's second chain
        self.profile = None          # list -> Runtime.call records (name, scalar args, start event, end event)
        self.arena = None            # Arena while a Trainer step runs in replay mode (else torch's caching allocator)
        self.ws_grew = False         # a workspace was (re)allocated since the flag was cleared (a recorded step must not see that)
        self._wait_events = {}       # (waiter, signaler) raw stream handles -> reusable event of stream_wait()
        self.host_counters = True    # step / noise-draw numbers as host arguments; Trainer turns it off for hipGraph replay,
                                     # where they must be device-side counters (one tiny kernel each per step)
        self.taps = None             # dict -> conv_bn_fwd records the post-ReLU output of every conv+BN+ReLU by weight name (tests)
        self.fused_attention = True  # bf16: flash-style kernel; False forces the batched-GEMM + softmax path (tests)
        self.on_upper_grads_ready = None   # callback fired in backward once G[fusion.pos_embed:] is final (DP overlap)
        self.on_stem_backward = None       # callback fired when the LAST node of the backward (the ImageEncoder stem) starts enqueueing (Trainer)
        self.side = None                   # second HIP stream for the weight-gradient GEMMs (call_side / join_side)
        self._side_dirty = False
        self._side_keep = []
        self.capture_keep = []
        self._branch = None
        self.branch_streams = os.environ.get("MSC_BRANCH_STREAMS", "1") != "0"
        self.wgrad_on_side = os.environ.get("MSC_SIDE_STREAM", "1") != "0"
        L.lib.load()                # fail loudly, now, if the HIP extension is missing

    @property
    def stream(self):
        """raw hipStream_t of torch's current stream (the C-level query: this runs once per kernel launch)"""
        return _raw_stream(self._dev_index)

    def alloc(self, shape, dtype):
        """every per-step device buffer of the path comes from here: the step's bump arena when one is active (replay mode),
        else torch's caching allocator"""
        if self.arena is not None:
            return self.arena.alloc(tuple(shape), dtype)
        return torch.empty(tuple(shape), dtype=dtype, device=self.device)

    def empty(self, *shape, dtype=None):
        return self.alloc(shape, dtype or self.tdtype)

    def f32(self, *shape):
        return self.alloc(shape, torch.float32)

    def empty_like(self, t):
        return self.alloc(t.shape, t.dtype)

    def stream_wait(self, waiter: int, signaler: int):
        """`waiter` waits for everything enqueued so far on `signaler` (raw hipStream_t handles): event record + stream wait
        through the C-ABI, so the dependency is part of a recorded launch list (torch's Stream.wait_stream would not be)"""
        if waiter == signaler:
            return
        ev = self._wait_events.get((waiter, signaler))
        if ev is None:
            ev = self._wait_events[(waiter, signaler)] = L.query("msc_event_create")
            if not ev:
                raise RuntimeError("msc_event_create failed")
        L.call("msc_stream_wait_stream", waiter, signaler, ev)

    def workspace(self, name: str, nbytes: int, per_stream: bool = True) -> torch.Tensor:
        """Persistent fp32 scratch, grown on demand.  One buffer per (name, HIP stream): the encoder branches run on
        different streams and must not share reduction scratch.  per_stream=False: buffers only ever touched from the
        weight-gradient side stream."""
        n = (int(nbytes) + 3) // 4
        # per_stream=False is only safe while call_side() really routes to the side stream: when it falls back to the
        # launching stream (MSC_SIDE_STREAM=0, instrumented pass) the main and BEV-branch streams would share the scratch
        key = (name, _raw_stream(self._dev_index) if (per_stream or not self.side_active()) else 0)
        t = self._ws.get(key)
        if t is None or t.numel() < n:
            if t is not None:
                # a recorded launch list (replay program, hipGraph) may hold the old buffer's address: it stays allocated, so a
                # program recorded at a smaller shape keeps working in its own (sufficient) scratch
                self._ws_retired.append(t)
            t = torch.empty(max(n, 1), dtype=torch.float32, device=self.device)
            self._ws[key] = t
            self.ws_grew = True
        return t

    def branch_stream(self):
        """second compute stream for the LiDAR/BEV branch (independent of the camera branch until the fusion)"""
        if self._branch is None:
            self._branch = torch.cuda.Stream(self.device)
        return self._branch

    def call(self, name, *args):
        if self.profile is None:
            return L.call(name, *args, self.stream)
        # instrumented pass (bench.py): HIP events on the launch stream around every C-ABI call
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = L.call(name, *args, self.stream)
        e1.record()
        self.profile.append((name, tuple(a for a in args if not hasattr(a, "data_ptr") and a is not None), e0, e1))
        return rc

    def side_active(self):
        """True when call_side() enqueues on the side stream (False: it falls back to the launching stream)"""
        return self.wgrad_on_side and self.profile is None

    def call_side(self, name, *args):
        """Weight-gradient GEMMs are off the backward's critical path (nothing downstream reads them until the
        optimizer), and many of them cannot fill 256 CUs on their own: enqueue them on a second HIP stream so the
        hardware co-schedules them with the dgrad / normalisation chain on the main stream.  join_side() orders the main
        stream behind them.  The instrumented (profile) pass launches them on the calling stream (bench.py also turns the
        BEV-branch stream off for that pass, so every timed kernel runs alone)."""
        if not self.side_active():
            return self.call(name, *args)
        if self.side is None:
            self.side = self._make_side_stream()
            self._side_raw = self.side.cuda_stream
            self._side_event = L.query("msc_event_create")
            if not self._side_event:
                raise RuntimeError("msc_event_create failed")
        # side stream waits for everything enqueued so far on the launching stream (one reusable event, one C call)
        L.call("msc_stream_wait_stream", self._side_raw, _raw_stream(self._dev_index), self._side_event)
        if self.side_also_waits is not None:      # operands produced on a second chain's stream (raw handle, event)
            L.call("msc_stream_wait_stream", self._side_raw, self.side_also_waits[0], self.side_also_waits[1])
        rc = L.call(name, *args, self._side_raw)
        # The operands are read on the side stream but owned (for the caching allocator) by the launching stream: keep
        # them alive until join_side() has ordered the launching streams behind the side stream.  Cheaper than
        # record_stream per operand, and also what a hipGraph capture needs (Trainer clears capture_keep after capture).
        (self.capture_keep if torch.cuda.is_current_stream_capturing() else self._side_keep).append(args)
        self._side_dirty = True
        return rc

    # ---- second chain (FusionLayerFn): half of the batch runs on the BEV-branch stream, idle during the fusion
    def chain_b(self):
        """(raw handle of the BEV-branch stream, event main->b, event b->main, event b->side), created on first use"""
        if self._chain is None:
            s2 = self.branch_stream()
            evs = [L.query("msc_event_create") for _ in range(3)]
            if not all(evs):
                raise RuntimeError("msc_event_create failed")
            self._chain = (s2.cuda_stream, *evs)
        return self._chain

    def call_on(self, raw_stream, name, *args):
        return L.call(name, *args, raw_stream)

    def _make_side_stream(self):
        """The side stream runs at the LOWEST HIP priority: its kernels fill CUs the critical chain leaves idle instead
        of competing with it (MSC_SIDE_PRIO=0 keeps the default priority)."""
        level = int(os.environ.get("MSC_SIDE_PRIO", "-1"))
        if level == 0:
            return torch.cuda.Stream(self.device)
        with torch.cuda.device(self.device):
            h = L.query("msc_stream_create", level)
        if not h:
            raise RuntimeError("msc_stream_create failed")
        return torch.cuda.ExternalStream(h, device=self.device)

    def join_side(self):
        # While a step is being RECORDED (replay.Recorder active) the wait is emitted unconditionally: the recorded program is
        # replayed after steps whose side stream is always busy (upper-half Adam, weight transposes, draw-counter bump), whatever
        # the host flag said about the step that happened to be recorded (a clean side stream after export_state_dict /
        # load_oracle_state_dict / a no-grad forward would otherwise bake "no wait" into every later replay: a silent race on
        # P / C / TR between the side stream's optimizer tail and the fusion, codec and heads of the next step).
        if self.side is not None and (self._side_dirty or L._REC is not None):
            self.stream_wait(self.stream, self._side_raw)
            self._side_dirty = False
            self._side_keep.clear()


_TR_LO_SIDE = os.environ.get("MSC_TR_LO_SIDE", "1") != "0"


class ParamStore:
    ALIGN = 64

    def __init__(self, rt: Runtime, specs):
        """specs: list of (name, shape, kind) with kind in {'conv','linear','vec'}."""
        self.rt = rt
        self.refs = {}
        off = 0
        for name, shape, kind in specs:
            r = ParamRef(name, shape, kind, off)
            self.refs[name] = r
            off += (r.numel + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.total = off
        dev = rt.device
        self.P = torch.zeros(off, dtype=torch.float32, device=dev)
        self.G = torch.zeros(off, dtype=torch.float32, device=dev)
        self.M = torch.zeros(off, dtype=torch.float32, device=dev)
        self.V = torch.zeros(off, dtype=torch.float32, device=dev)
        self.C = self.P if rt.dt == L.MSC_F32 else torch.zeros(off, dtype=rt.tdtype, device=dev)
        self.TR = torch.zeros(off, dtype=rt.tdtype, device=dev)
        for r in self.refs.values():
            sl = slice(r.off, r.off + r.numel)
            r.p = self.P[sl].view(r.shape)
            r.g = self.G[sl].view(r.shape)
            r.c = self.C[sl].view(r.shape)
            r.t = self.TR[sl] if r.kind in ("conv", "linear") else None
        # ImageEncoder stem weights in the space-to-depth layout [Cout][4][4][16] (compute dtype), refreshed with the other compute copies
        self.stem = self.refs.get("image_encoder.conv1.weight")
        self.stem_w = torch.zeros(self.stem.shape[0] * 256, dtype=rt.tdtype, device=dev) if self.stem is not None else None
        self.step_count = 0
        self._early = None         # (lo, hi, hyper-parameter tuple) of this step's adam_early() call
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev)   # device-side step counter (hipGraph replay)
        self._ptable = None
        self._ptable_split = None

    def __getitem__(self, name) -> ParamRef:
        return self.refs[name]

    def _tables(self, split, mid=None):
        """descriptor tables of the weight transposes (one launch instead of one per weight): all / below split / above split
        (+ with mid: "rest" = below mid, "mid" = [mid, split))"""
        if self._ptable is None or self._ptable_split != (split, mid):
            def build(lo, hi):
                rows, blk = [], 0
                for r in self.refs.values():
                    if r.kind == "conv":
                        d0, d1, d2 = r.shape[0], r.shape[1] * r.shape[2], r.shape[3]
                    elif r.kind == "linear":
                        d0, d1, d2 = r.shape[0], 1, r.shape[1]
                    else:
                        continue
                    if not (lo <= r.off < hi):
                        continue
                    rows.append([r.off, r.off, d0, d1, d2, blk])
                    blk += ((d2 + 31) // 32) * ((d0 + 31) // 32) * d1
                t = torch.tensor(rows, dtype=torch.int64, device=self.rt.device) if rows else None
                return t, blk
            self._ptable = {"all": build(0, self.total)}
            if split:
                self._ptable["lo"], self._ptable["hi"] = build(0, split), build(split, self.total)
                if mid:
                    self._ptable["rest"], self._ptable["mid"] = build(0, mid), build(mid, split)
            self._ptable_split = (split, mid)
        return self._ptable

    def refresh_compute_copies(self, cast: bool = True):
        """P -> C (bf16 mode, unless Adam already wrote the shadow) and the transposed copies TR."""
        rt = self.rt
        if cast and rt.dt != L.MSC_F32:
            rt.call("msc_cast", self.P, self.C, self.total, rt.dt)
        t, blk = self._tables(*(self._ptable_split or (None, None)))["all"]
        rt.call("msc_permute3_cast_batched", self.P, self.TR, t, t.shape[0], blk, rt.dt)
        self._refresh_stem()

    def _refresh_stem(self):
        if self.stem is not None:
            self.rt.call("msc_stem_weight_s2d", self.stem.p, self.stem_w, self.stem.shape[0], self.rt.dt)

    def _adam_hp(self, lr, betas, eps, weight_decay, grad_scale):
        rt = self.rt
        self.step_count += 1
        if rt.host_counters:
            return (float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), self.step_count, None, float(grad_scale), rt.dt)
        rt.call("msc_counter_add", self.step_dev, 1)      # the kernels read the step from device memory, so the same launches replay correctly
        return (float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), 0, self.step_dev, float(grad_scale), rt.dt)

    def adam_early(self, lo, hi, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        """Update of the slice [lo, hi) of the flat buffers on the side stream, behind the weight-gradient GEMMs that produced its gradients,
        while the launching stream still has backward work of OTHER parameters to do.  The caller has ordered the side stream behind every
        stream that wrote gradients of the slice or still reads its weights.  adam_step() of the same step then updates [0, lo) and
        [hi, total) only (hi must be its overlap_from).  Returns False (nothing done) where adam_step would not split either."""
        rt = self.rt
        if not (rt.wgrad_on_side and rt.profile is None and not torch.cuda.is_current_stream_capturing()):
            return False
        hp = self._adam_hp(lr, betas, eps, weight_decay, grad_scale)
        shadow = None if rt.dt == L.MSC_F32 else self.C[lo:hi]
        rt.call_side("msc_adam_step", self.P[lo:hi], self.G[lo:hi], self.M[lo:hi], self.V[lo:hi], shadow, hi - lo, *hp)
        t, blk = self._tables(hi, lo)["mid"]
        if t is not None:
            rt.call_side("msc_permute3_cast_batched", self.P, self.TR, t, t.shape[0], blk, rt.dt)
        self._early = (lo, hi, hp)
        return True

    def adam_step(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0, overlap_from=None):
        """Fused Adam (+ bf16 shadow) over the flat buffers, then the transposed compute copies.
        overlap_from = element offset: parameters from there on (fusion transformer, codec, heads) are updated on the side
        stream, where the update overlaps the camera/BEV trunks of the NEXT forward; MultimodalSC.forward joins the side
        stream before the fusion reads them.  Not used while a hipGraph is being captured or in the instrumented pass."""
        rt = self.rt
        early, self._early = self._early, None
        hp = early[2] if early else self._adam_hp(lr, betas, eps, weight_decay, grad_scale)
        split = overlap_from if (overlap_from and rt.wgrad_on_side and rt.profile is None
                                 and not torch.cuda.is_current_stream_capturing()) else None
        if early:
            assert split == early[1], "adam_early() updated [lo, split): the step's adam_step must split at the same offset"
        if not split:
            rt.call("msc_adam_step", self.P, self.G, self.M, self.V, None if rt.dt == L.MSC_F32 else self.C, self.total, *hp)
            self.refresh_compute_copies(cast=False)
            return
        lo_end = early[0] if early else split          # [lo_end, split) was updated by adam_early()
        tabs = self._tables(split, early[0] if early else (self._ptable_split or (None, None))[1])
        shadow = lambda a, b: None if rt.dt == L.MSC_F32 else self.C[a:b]
        rt.call("msc_adam_step", self.P[:lo_end], self.G[:lo_end], self.M[:lo_end], self.V[:lo_end], shadow(0, lo_end), lo_end, *hp)
        t, blk = tabs["rest"] if early else tabs["lo"]
        if t is not None:
            # the transposed copies feed the dgrads only: off the step boundary's critical path, onto the side stream (which waits for the
            # update above; the next forward joins the side stream before the fusion, long before the first dgrad).  MSC_TR_LO_SIDE=0: main
            (rt.call_side if _TR_LO_SIDE else rt.call)("msc_permute3_cast_batched", self.P, self.TR, t, t.shape[0], blk, rt.dt)
        self._refresh_stem()
        n_hi = self.total - split
        rt.call_side("msc_adam_step", self.P[split:], self.G[split:], self.M[split:], self.V[split:], shadow(split, self.total), n_hi, *hp)
        t, blk = tabs["hi"]
        if t is not None:
            rt.call_side("msc_permute3_cast_batched", self.P, self.TR, t, t.shape[0], blk, rt.dt)
