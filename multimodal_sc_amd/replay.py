"""Recording and C-side replay of one training step's launch list (csrc/replay.hip; include/msc.h "Launch replay").

An eager step is a fixed sequence of C-ABI calls; Python spends ~12 us on each (autograd node, marshalling, ctypes), which
makes the step host-bound below ~40 frames per GPU.  Recorder hooks lib.call during ONE eager step and keeps, per call, the
entry point and the raw argument values (device addresses, sizes, stream handles, events).  Program hands them to the C side,
which re-issues a whole segment from one ctypes call.  Conditions the Trainer guarantees:
  * every per-step buffer comes from the step's bump arena (runtime.Arena): the same addresses every step; buffers that do not
    (the caller's input batch, anything torch allocated) are kept alive by the program for as long as it exists;
  * step / noise-draw numbers are device-side counters (Runtime.host_counters = False), persistent workspaces did not grow
    during the recorded step, and every cross-stream dependency of the step is an msc_stream_wait_stream call (Runtime.stream_wait)
    — including those torch.autograd would otherwise supply.
"""
from __future__ import annotations

import ctypes

from . import lib as L


class Recorder:
    """context manager: records every lib.call between __enter__ and __exit__; cut(label) ends a segment (Python acts there)"""

    def __init__(self):
        self.recs = []          # (name, converted args) | ("__cut__", label)
        self.keep = []          # tensors seen as arguments: alive as long as the program
        self.labels = []

    def add(self, name, conv, raw):
        # A host-blocking wait never belongs in a replayed list (every replay would stall the host on a pending H2D copy); calls of
        # threads that opted out (lib.no_record(): the frame stager's filler thread) are dropped in lib.call.  NOT a "recording
        # thread only" filter: autograd runs the backward of a step on its own engine thread, and those launches ARE the step.
        if name == "msc_event_synchronize":
            return
        self.recs.append((name, conv))
        for a in raw:
            if hasattr(a, "data_ptr"):
                self.keep.append(a)

    def cut(self, label):
        self.recs.append(("__cut__", label))

    def __enter__(self):
        if L._REC is not None:
            raise RuntimeError("a launch recording is already active")
        L._REC = self
        return self

    def __exit__(self, *exc):
        L._REC = None
        return False


def _streams_of(name, conv):
    """streams a record is ordered on: a kernel launch -> its stream (last argument); msc_stream_wait_stream -> waiter AND signaler
    (it must keep its place in both streams' sequences: it captures what the signaler has been given so far); the split event
    calls (record now / wait later, paired through the event) -> None = ordered against everything"""
    if name == "msc_stream_wait_stream":
        return (conv[0] or 0, conv[1] or 0)
    if name in ("msc_event_record", "msc_stream_wait_event", "msc_event_synchronize"):
        return None
    return (conv[-1] or 0,)


def reorder(recs, policy):
    """A different ENQUEUE order of the same launches: per-stream order and every cross-stream dependency are kept (so the GPU
    sees the same per-stream programs and waits), only the order in which the host hands them to the streams changes.  The
    eager loop's order is what Python produced — e.g. the whole BEV-branch backward before the first launch of the camera
    trunk's backward; a replayed list can feed the streams evenly.  policy: "recorded" | "rr" (round-robin one launch at a time
    among the streams with an emittable launch) | "main" (the main stream first; others only when it waits for them)."""
    if policy == "recorded":
        return recs
    out, seg = [], []

    def flush():
        if not seg:
            return
        order = []                     # streams in order of first appearance: the main stream launches first in every segment
        queues = {}
        glob = []                      # indices of records ordered against everything
        for i, (name, conv) in enumerate(seg):
            st = _streams_of(name, conv)
            if st is None:
                glob.append(i)
                continue
            for s in st:
                if s not in queues:
                    queues[s] = []
                    order.append(s)
                queues[s].append(i)
        head = {s: 0 for s in order}
        done = [False] * len(seg)
        gi = 0
        emitted = 0
        rr = 0

        def emittable(i):
            if gi < len(glob) and i > glob[gi]:
                return False           # a globally ordered record comes first
            st = _streams_of(*seg[i])
            return all(queues[s][head[s]] == i for s in st)

        while emitted < len(seg):
            if gi < len(glob) and all(done[j] for j in range(glob[gi])):
                i = glob[gi]
                gi += 1
            else:
                cand = [s for s in order if head[s] < len(queues[s]) and emittable(queues[s][head[s]])]
                assert cand, "reorder: no emittable record (cycle?)"
                if policy == "main":
                    s = cand[0]
                else:
                    s = cand[rr % len(cand)]
                    rr += 1
                i = queues[s][head[s]]
                for t in _streams_of(*seg[i]):
                    head[t] += 1
            done[i] = True
            out.append(seg[i])
            emitted += 1
        seg.clear()

    for r in recs:
        if r[0] == "__cut__":
            flush()
            out.append(r)
        else:
            seg.append(r)
    flush()
    return out


class Program:
    """the recorded launch list on the C side; run(segment) re-issues it"""

    def __init__(self, rec: Recorder):
        lib = L.lib.load()
        self._lib = lib
        self.handle = int(lib.msc_replay_create())
        if not self.handle:
            raise RuntimeError("msc_replay_create failed")
        self.keep = rec.keep
        self.labels = []            # labels[i] = the host action after segment i (None after the last)
        self.names = []
        ids = {}
        import os
        self.order = os.environ.get("MSC_REPLAY_ORDER", "recorded")
        ordered = reorder(rec.recs, self.order)
        if os.environ.get("MSC_REPLAY_DUMP"):       # debugging aid: the launch list as text (entry point, streams, scalar arguments)
            with open(os.environ["MSC_REPLAY_DUMP"], "a") as fh:
                fh.write(f"# program of {len(ordered)} records, order={self.order}\n")
                for name, conv in ordered:
                    if name == "__cut__":
                        fh.write(f"---- cut: {conv}\n")
                    else:
                        st = _streams_of(name, conv)
                        fh.write(f"{name:34s} streams={[hex(s) for s in st] if st else 'all'} scalars={[a for a in conv[:-1] if isinstance(a, (int, float)) and abs(a) < 1 << 32]}\n")
        for name, conv in ordered:
            if name == "__cut__":
                lib.msc_replay_cut(self.handle)
                self.labels.append(conv)
                continue
            fid = ids.get(name)
            if fid is None:
                fid = ids[name] = lib.msc_replay_fn_id(name.encode())
                if fid < 0:
                    raise RuntimeError(f"{name} is not a replayable entry point")
            argtypes = L.lib.protos[name][1]
            n = len(conv)
            ia, fa = (ctypes.c_int64 * n)(), (ctypes.c_double * n)()
            for j, (v, t) in enumerate(zip(conv, argtypes)):
                if t is ctypes.c_float:
                    fa[j] = float(v)
                elif v is not None:
                    iv = int(v)
                    ia[j] = iv - (1 << 64) if iv >= (1 << 63) else iv
            rc = lib.msc_replay_add(self.handle, fid, ia, fa, n)
            if rc != 0:
                raise RuntimeError(f"msc_replay_add({name}) failed: code {rc}")
            self.names.append(name)
        self.labels.append(None)
        self.n_segments = len(self.labels)

    def run(self, segment: int = -1):
        rc = self._lib.msc_replay_run(self.handle, segment)
        if rc != 0:
            i = int(self._lib.msc_replay_last_failed(self.handle))
            name = self.names[i] if 0 <= i < len(self.names) else "?"
            raise RuntimeError(f"replay of segment {segment} failed at record {i} ({name}): {self._lib.msc_strerror(rc).decode()} (code {rc})")

    def count(self, segment: int = -1) -> int:
        return int(self._lib.msc_replay_count(self.handle, segment))

    def __del__(self):
        try:
            if self.handle:
                self._lib.msc_replay_destroy(self.handle)
                self.handle = 0
        except Exception:  # noqa: BLE001 — interpreter shutdown
            pass
