#!/usr/bin/env python3
"""Where does a replayed step first differ from the eager loop?  Runs the same steps in three host-loop modes (eager python loop;
eager with the arena + device counters but never recording; launch replay) and reports, per step, the first parameter tensors
whose values differ from the eager run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd import functional as F
from multimodal_sc_amd.model import MultimodalSC, named_config
from multimodal_sc_amd.synthetic import synthetic_batch
from multimodal_sc_amd.trainer import Trainer

name, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if "--chains" in sys.argv:
    F._CHAIN_MIN_ROWS = 0
cfg = named_config(name)


def dev(b):
    out = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items() if k != "ppo"}
    out["ppo"] = {k: v.cuda() for k, v in b["ppo"].items()}
    out["noise"] = out["fading"] = None
    return out


batches = [dev(synthetic_batch(cfg, B, seed=5 + i)) for i in range(2)]


def run(mode):
    os.environ["MSC_REPLAY"] = "0" if mode == "eager" else "1"
    os.environ["MSC_REPLAY_WARMUP"] = "1000000" if mode == "arena" else "2"
    m = MultimodalSC(cfg, dtype="bf16", device="cuda:0", seed=3)
    tr = Trainer(m, lr=1e-3)
    snaps = []
    for i in range(steps):
        tr.step(batches[i % 2])
        torch.cuda.synchronize()
        snaps.append((m.params.P.clone(), m.params.G.clone()))
    return m, snaps


ref_m, ref = run("eager")
for mode in ("arena", "replay"):
    m, got = run(mode)
    for i in range(steps):
        for which, a, b in (("G", ref[i][1], got[i][1]), ("P", ref[i][0], got[i][0])):
            if not torch.equal(a, b):
                bad = [n for n, r in m.params.refs.items() if not torch.equal(a[r.off:r.off + r.numel], b[r.off:r.off + r.numel])]
                d = (a - b).abs().max().item()
                print(f"[{mode}] step {i}: {which} differs in {len(bad)} tensors (max |d| {d:.3e}); first: {bad[:4]} last: {bad[-3:]}")
                break
        else:
            continue
        break
    else:
        print(f"[{mode}] all {steps} steps bit-identical to the eager loop")
