"""Host-resident frame source for the path ("driving-environment / dataset adapter").

The reference names the domain ("Autonomous Driving", /root/reference/README.md:2) and ships no simulator binding or dataset
code, and no simulator is installed here; what a training loop needs from either is the same: camera + LiDAR + ego frames that
live in HOST memory or on disk and must reach HBM without stalling the step.  This module is that adapter:

  FrameLog        a recorded drive on disk, raw little-endian arrays + meta.json (no pickle):
                      camera.u8  [n][H][W][3]   RGB as a camera / decoder delivers it (pixel-interleaved bytes)
                      lidar.f32  [n][P][4]      x, y, z, intensity
                      ego.f32    [n][S]         ego state
                      action.i64 [n]            the logged (expert) action — the reward signal of RecordedDrivingEnv
                  opened with numpy.memmap (streams from the page cache; nothing is loaded whole).
  FrameStager     pinned host ring (a filler thread copies the next batches' frames out of the log) -> msc_memcpy_h2d_async into
                  one of two device staging sets ON THE WEIGHT-GRADIENT SIDE STREAM (an existing stream: the step uses three and a
                  fifth active stream serialises this stack; the side stream is idle while the forward runs, which is
                  when the copy engine moves batch k+1) -> at the next step, on the main stream: wait for that copy's event,
                  msc_frames_u8_to_nchw decodes the bytes to the fp32 planes the path reads.  A camera frame crosses PCIe as bytes
                  (150 KB instead of 602 KB as fp32): 26 MB per 64-frame batch with the 16384-point cloud.
  RecordedDrivingEnv   the reset()/step() environment interface of ppo.py over a FrameLog: N cursors into the drive, observations
                  through a FrameStager, reward 1 where the policy's action equals the logged one.
"""
from __future__ import annotations

import json
import os
import queue
import threading

import numpy as np
import torch

from . import lib as L

_FILES = {"camera": ("camera.u8", np.uint8), "lidar": ("lidar.f32", np.float32), "ego": ("ego.f32", np.float32),
          "action": ("action.i64", np.int64)}


class FrameLog:
    def __init__(self, path: str):
        self.path = path
        self.meta = json.load(open(os.path.join(path, "meta.json")))
        n, H, W = self.meta["n"], self.meta["H"], self.meta["W"]
        self.n = n
        shapes = {"camera": (n, H, W, 3), "lidar": (n, self.meta["n_points"], 4), "ego": (n, self.meta["state_dim"]), "action": (n,)}
        self.arrays = {}
        for k, (fn, dt) in _FILES.items():
            f = os.path.join(path, fn)
            if os.path.exists(f) and all(s > 0 for s in shapes[k]):
                self.arrays[k] = np.memmap(f, dtype=dt, mode="r", shape=shapes[k])

    @staticmethod
    def write_synthetic(path: str, cfg, n: int, seed: int = 0, chunk: int = 64):
        """a seeded synthetic drive of n frames in the on-disk format (camera bytes uniform, cloud as synthetic.synthetic_batch)"""
        os.makedirs(path, exist_ok=True)
        rng = np.random.default_rng(seed)
        H = W = cfg.img_size
        P = cfg.n_points if cfg.use_lidar else 0
        S = cfg.state_dim if cfg.use_state else 0
        meta = {"n": n, "H": H, "W": W, "n_points": P, "state_dim": S, "n_actions": cfg.n_actions, "seed": seed}
        fh = {k: open(os.path.join(path, fn), "wb") for k, (fn, _) in _FILES.items()}
        for lo in range(0, n, chunk):
            c = min(chunk, n - lo)
            fh["camera"].write(rng.integers(0, 256, size=(c, H, W, 3), dtype=np.uint8).tobytes())
            if P:
                xy = (rng.random((c, P, 2), dtype=np.float32) * 2 - 1) * np.float32(cfg.bev_range)
                z = rng.random((c, P, 1), dtype=np.float32) * 4 - 3
                it = rng.random((c, P, 1), dtype=np.float32)
                fh["lidar"].write(np.concatenate([xy, z, it], -1).astype(np.float32).tobytes())
            if S:
                fh["ego"].write(rng.standard_normal((c, S)).astype(np.float32).tobytes())
            fh["action"].write(rng.integers(0, cfg.n_actions, size=(c,), dtype=np.int64).tobytes())
        for f in fh.values():
            f.close()
        json.dump(meta, open(os.path.join(path, "meta.json"), "w"))
        return FrameLog(path)


class FrameStager:
    """batches of `batch` frames from a FrameLog (or any dict of host arrays with the same keys) -> device, one step ahead.

        stager = FrameStager(rt, source, batch)            # rt: the model's Runtime (its side stream carries the copies)
        for k in range(steps):
            b = stager.next()          # device tensors of batch k (the main stream waits for its copy; the bytes are decoded there)
            stager.prefetch()          # H2D of batch k+1 on the side stream, BEFORE the step's own side-stream work is enqueued
            trainer.step({...b...})

    index_fn(k) -> the `batch` frame indices of batch k (default: consecutive windows, wrapping).
    extra_fn(k) -> dict of small host arrays (float32 / int64) staged with batch k (e.g. the PPO minibatch of the step)."""

    RING = 3        # pinned slots: one being copied, one filled and waiting, one being filled

    def __init__(self, rt, source, batch: int, index_fn=None, extra_fn=None):
        self.rt, self.B = rt, batch
        self.src = source.arrays if isinstance(source, FrameLog) else source
        self.n = len(self.src["camera"])
        self.index_fn = index_fn or (lambda k: (np.arange(batch) + k * batch) % self.n)
        self.extra_fn = extra_fn
        dev = rt.device
        tdt = {np.dtype(np.uint8): torch.uint8, np.dtype(np.float32): torch.float32, np.dtype(np.int64): torch.int64}
        self.keys = [k for k in ("camera", "lidar", "ego", "action") if k in self.src]
        spec = {k: ((batch,) + tuple(self.src[k].shape[1:]), np.dtype(self.src[k].dtype)) for k in self.keys}
        self.extra_keys = []
        if extra_fn is not None:
            for k, v in extra_fn(0).items():
                spec["x_" + k] = (tuple(v.shape), np.dtype(v.dtype))
                self.extra_keys.append(k)
        self.names = list(spec)
        self.pinned = [{k: torch.empty(s, dtype=tdt[d]).pin_memory() for k, (s, d) in spec.items()} for _ in range(self.RING)]
        self.pinned_np = [{k: t.numpy() for k, t in slot.items()} for slot in self.pinned]
        self.stage = [{k: torch.empty(s, dtype=tdt[d], device=dev) for k, (s, d) in spec.items()} for _ in range(2)]
        _, H, W, _ = spec["camera"][0]
        self.image = [torch.empty(batch, 3, H, W, dtype=torch.float32, device=dev) for _ in range(2)]
        self.events = [L.query("msc_event_create") for _ in range(self.RING)]      # one per pinned slot: "its copy has run"
        if not all(self.events):
            raise RuntimeError("msc_event_create failed")
        self.bytes_per_batch = sum(t.numel() * t.element_size() for t in self.pinned[0].values())
        self._filled = queue.Queue()       # (batch index, slot) ready for H2D
        self._free = queue.Queue()         # slots the filler may overwrite (after their last copy has completed)
        for s in range(self.RING):
            self._free.put(s)
        self._stop = False
        self._issued = 0                   # batches whose H2D has been enqueued
        self._taken = 0                    # batches handed out by next()
        self._slot_of = {}                 # batch index -> pinned slot that carries it
        self._thread = threading.Thread(target=self._fill_loop, daemon=True)
        self._thread.start()
        self.prefetch()                    # batch 0

    def _fill_loop(self):
        torch.cuda.set_device(self.rt.device)        # (HIP's current device is per host thread)
        L.no_record()                                # this thread's calls run beside the step, never inside a recorded launch list
        k = 0
        while not self._stop:
            try:
                slot = self._free.get(timeout=0.05)
            except queue.Empty:
                continue
            L.call("msc_event_synchronize", self.events[slot])      # the slot's previous copy must have left the pinned buffer
            idx = np.asarray(self.index_fn(k))
            pn = self.pinned_np[slot]
            i0 = int(idx[0])
            run = len(idx) > 0 and i0 + len(idx) <= self.n and bool(np.all(np.diff(idx) == 1))
            for key in self.keys:
                if run:          # a window of consecutive frames: one block copy (a gather of 26 MB per batch is what paces this thread)
                    np.copyto(pn[key], self.src[key][i0:i0 + len(idx)])
                else:
                    np.take(self.src[key], idx, axis=0, out=pn[key])
            if self.extra_fn is not None:
                for kk, v in self.extra_fn(k).items():
                    np.copyto(pn["x_" + kk], v)
            self._filled.put((k, slot))
            k += 1

    def side_stream(self):
        rt = self.rt
        if rt.side is None:                # the step creates it on first use; a stager started before the first step does it here
            rt.side = rt._make_side_stream()
            rt._side_raw = rt.side.cuda_stream
            rt._side_event = L.query("msc_event_create")
        return rt._side_raw

    def prefetch(self):
        """enqueue the H2D copy of the next batch on the side stream, behind everything the main stream has been given so far
        (the device staging set it overwrites was last read two batches ago, on streams the main stream has since joined)"""
        k, slot = self._filled.get()
        assert k == self._issued, (k, self._issued)
        rt, side = self.rt, self.side_stream()
        rt.stream_wait(side, rt.stream)
        st = self.stage[k % 2]
        for name in self.names:
            t = self.pinned[slot][name]
            L.call("msc_memcpy_h2d_async", st[name], t.data_ptr(), t.numel() * t.element_size(), side)
        L.call("msc_event_record", self.events[slot], side)
        self._slot_of[k] = slot
        self._issued += 1

    def next(self, image_out=None):
        """-> dict(image fp32 [B,3,H,W] in [0,1], frames_u8, points, state, action, extra{...}): device tensors of the next batch,
        valid until the batch after the next one is taken.  image_out: decode the camera bytes into this tensor instead of the stager's
        own (e.g. Trainer.input_buffers(...)["image"]: the step then has no staging copy of the largest input)"""
        k = self._taken
        assert k < self._issued, "call prefetch() once per next()"
        rt, st, slot = self.rt, self.stage[k % 2], self._slot_of.pop(k)
        L.call("msc_stream_wait_event", rt.stream, self.events[slot])
        self._free.put(slot)               # (the filler waits for the copy's event before it writes the slot again)
        B, H, W, _ = st["camera"].shape
        img = self.image[k % 2]
        if image_out is not None:
            if tuple(image_out.shape) != tuple(img.shape) or image_out.dtype != img.dtype or not image_out.is_contiguous():
                raise ValueError(f"image_out must be a contiguous {tuple(img.shape)} {img.dtype} tensor")
            img = image_out
        L.call("msc_frames_u8_to_nchw", st["camera"], img, B, H, W, rt.stream)
        self._taken += 1
        out = {"image": img, "frames_u8": st["camera"]}
        for src_key, name in (("lidar", "points"), ("ego", "state"), ("action", "action")):
            if src_key in st:
                out[name] = st[src_key]
        if self.extra_keys:
            out["extra"] = {k2: st["x_" + k2] for k2 in self.extra_keys}
        return out

    def close(self):
        self._stop = True
        self._thread.join(timeout=2.0)


class RecordedDrivingEnv:
    """ppo.py's environment interface over a recorded drive: n_envs cursors spaced evenly through the log advance one frame per
    step(); observations arrive through a FrameStager (the next observation's copy overlaps the policy forward of the current one);
    reward = 1 where the action equals the logged action of the frame it was taken on; episodes end every `episode_len` steps."""

    def __init__(self, log: FrameLog, rt, n_envs: int, episode_len: int = 16):
        self.log, self.rt, self.N, self.episode_len = log, rt, n_envs, episode_len
        self.starts = (np.arange(n_envs) * (log.n // n_envs)).astype(np.int64)
        self.stager = None
        self.t = 0

    def _make(self):
        if self.stager is not None:
            self.stager.close()
        self.stager = FrameStager(self.rt, self.log, self.N, index_fn=lambda k: (self.starts + k) % self.log.n)

    def _obs(self):
        b = self.stager.next()
        self.stager.prefetch()
        self._logged = b["action"]
        return {k: b[k] for k in ("image", "points", "state") if k in b}

    def reset(self):
        self.t = 0
        self._make()
        return self._obs()

    def step(self, actions):
        reward = (actions == self._logged).float()
        self.t += 1
        done = torch.full((self.N,), 1.0 if self.t % self.episode_len == 0 else 0.0, device=self.rt.device)
        return self._obs(), reward, done
