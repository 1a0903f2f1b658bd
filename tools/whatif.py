#!/usr/bin/env python3
"""What-if timing of the training step: SKIP the C-ABI calls whose name matches a regex (results are WRONG by construction —
this only answers "how much of the step does that work cost as scheduled", i.e. what a perfect speed-up of it could buy).
Usage: whatif.py [regex ...]     e.g.  whatif.py 'wgrad'  'msc_bn_bwd|msc_bn_apply'  'msc_pillar'
Prints ms/step (median of 3 x 30 steps) for the baseline and for each regex, all in one process on one box."""
import os, re, sys, statistics, time
os.environ.setdefault("MSC_REPLAY", "0")     # these tools hook the eager Python loop (a replayed step bypasses lib.call)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.model import MultimodalSC, named_config
from multimodal_sc_amd.synthetic import synthetic_batch
from multimodal_sc_amd.trainer import Trainer
from multimodal_sc_amd import lib as L

cfg = named_config(os.environ.get("CFG", "C3"))
B = int(os.environ.get("B", "64"))
m = MultimodalSC(cfg, "bf16", "cuda:0")
tr = Trainer(m, lr=0.0)      # parameters never move: skipped kernels cannot poison later steps (NaN data also changes power/clocks)
cb = synthetic_batch(cfg, B, seed=1)
b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in cb.items() if k != "ppo"}
b["ppo"] = {k: v.cuda() for k, v in cb["ppo"].items()}
b["noise"] = b["fading"] = None
orig = L.call
skip = [None]


OPT = re.compile("msc_adam_step|msc_permute3_cast_batched|msc_stem_weight_s2d")
skip_opt = [False]


def call(name, *a):
    if skip[0] is not None and skip[0].search(name):
        return 0
    if skip_opt[0] and OPT.search(name):      # no optimizer update while something upstream is skipped: garbage gradients must not reach the parameters
        return 0
    return orig(name, *a)


L.call = call


def timed(n=30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(5):
    tr.step(b)
torch.cuda.synchronize()
ps = m.params
snap = (ps.P.clone(), ps.M.clone(), ps.V.clone())


def restore():
    # lr = 0 does not protect the parameters from a skipped kernel's garbage: 0 * NaN = NaN, and a step on NaN data draws less power and runs at a
    # higher clock (the baseline after a skipped dgrad was 0.45 ms "faster") — every measurement starts from the same finite state
    torch.cuda.synchronize()
    m.rt.join_side()
    torch.cuda.synchronize()
    ps.P.copy_(snap[0]); ps.M.copy_(snap[1]); ps.V.copy_(snap[2])
    ps.refresh_compute_copies()
    torch.cuda.synchronize()


def measure(rx, no_opt):
    restore()
    skip[0] = None
    skip_opt[0] = no_opt
    tr.step(b)
    skip[0] = re.compile(rx) if rx else None
    return statistics.median([timed() for _ in range(3)])


print("# baseline and variant of a row both run WITHOUT the optimizer update (parameters stay finite: NaN data draws less power and clocks higher), except the row that skips the update itself")
for rx in sys.argv[1:]:
    targets_opt = bool(OPT.search(rx.split("|")[0])) or "adam" in rx
    b0 = measure(None, not targets_opt)      # a fresh baseline before every variant: the box drifts by several 0.1 ms over a minute
    v = measure(rx, not targets_opt)
    print(f"skip {rx:44s} {v:7.3f} ms/step   baseline {b0:7.3f}   delta {v - b0:+.3f}")
