#!/usr/bin/env python3
"""bench.py — frames/s of the multimodal semantic-communication training step on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W
N>1: one rank per GPU.  Started under torch.distributed.run (WORLD_SIZE set) this process IS a rank; started bare
(`python bench.py --gpus 8`) it spawns `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
process before anything touches the GPU, relays rank 0's JSON line and exits with the child's return code.

A "step" = one forward + backward + gradient all-reduce + fused-Adam update over a batch of 64 synthetic frames PER
GPU (weak scaling) of the workload BASELINE.json's metric is quoted on: camera 3x224x224 + 16384-point LiDAR cloud +
ego state, cross-modal fusion transformer, semantic codec over AWGN at 10 dB, PPO policy/value head (config "C3";
C1/C2 selectable).  bf16 storage / bf16 MFMA with fp32 accumulation.  Inputs are resident in HBM before timing.
Rank 0 prints ONE JSON line with the contract fields plus:
  roofline     — the MFMA contraction kernel (gemm_kernel: implicit-GEMM convs, linears, attention GEMMs): algorithmic
                 FLOPs per step / summed duration of its launches per step, measured with HIP events on the launch
                 stream in an instrumented step, against the dense bf16 MFMA peak (2.5 PFLOP/s).
  cpu_baseline — the build-authored PyTorch-CPU oracle timed on this box's host cores on a bounded sample.
  parity       — (N = 1) decoder-PSNR / logit / action deltas of the benchmarked model against the oracle on the same parameters and
                 frames with explicit noise: the second half of BASELINE.json's metric.
  alt_geometry — the same step at 32 frames per GPU (configs[3]'s shard size), labelled, beside the 64-per-GPU `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16


PMC_HBM_SUMMARY = "profiles/r4_pmc_hbm_bench_c3.csv"


def gemm_source_sha():
    import hashlib
    h = hashlib.sha256()
    for f in ("gemm.hip", "gemm_pp.hip", "gemm_skinny.hip", "conv_c64.hip", "conv_c64w.hip", "conv_stem.hip", "gemm_plan.h"):        # the kernels and their launch planning (tile choice, split counts)
        h.update(open(os.path.join(ROOT, "multimodal_sc_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_hbm_bytes_per_gemm_launch(n_launches_instrumented=0):
    """HBM bytes per launch of the contraction kernel from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE, separate passes over this same command; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM;
    tools/pmc_hbm_summary.py).  Hardware counters cannot be read from inside the timed process, so the figure comes from the
    profile file — and ONLY while that file was taken on the kernel source that is being benchmarked: the summary's
    sidecar (<csv>.meta.json, written by tools/pmc_hbm_summary.py) records the sha256 of csrc/gemm.hip at profiling time;
    if it is missing or differs from the current source the figure is stale and traffic is reported as null."""
    import csv
    path = os.path.join(ROOT, PMC_HBM_SUMMARY)
    meta = path + ".meta.json"
    if not (os.path.exists(path) and os.path.exists(meta)):
        return None, "no PMC summary for this build"
    if json.load(open(meta)).get("gemm_hip_sha256_16") != gemm_source_sha():
        return None, f"{PMC_HBM_SUMMARY} is stale (csrc/gemm.hip or gemm_plan.h changed since the PMC pass)"
    b = n = 0.0
    for r in csv.DictReader(open(path)):
        if any(k in r["kernel"] for k in ("gemm_kernel<unsigned short", "gemm_pp_kernel<", "gemm_skinny_kernel<", "conv_c64_kernel<", "conv_c64_wgrad_kernel", "stem_wgrad_kernel")):
            b += float(r["fetch_bytes_corrected_x2"]) + float(r["write_bytes"])
            n += int(r["calls"])
    # The profiled run is the normal multi-stream step, where the fusion layers launch every dense GEMM twice (two half-batch
    # chains); the roofline line counts launches of the single-chain instrumented pass.  With the number of profiled steps in
    # the sidecar the figure is bytes per STEP divided by that pass's launch count (so it compares with
    # algorithmic_bytes_per_launch); without it, bytes per profiled launch.
    steps = json.load(open(meta)).get("steps_profiled")
    if steps and n_launches_instrumented:
        return b / steps / n_launches_instrumented, PMC_HBM_SUMMARY + " (committed profile of this source; bytes per step / launches of the instrumented pass)"
    return (b / n if n else None), PMC_HBM_SUMMARY + " (committed profile of this source)"


def gemm_flops(name, s):
    """algorithmic FLOPs of one C-ABI GEMM-family call from its scalar arguments (see Runtime.call)."""
    def conv(B, H, W, Cin, Cout, KH, KW, st, p):
        Ho, Wo = (H + 2 * p - KH) // st + 1, (W + 2 * p - KW) // st + 1
        cin = 3 if Cin == 8 else Cin           # the RGB stem is zero-padded to 8 channels; count the real 3
        return 2.0 * B * Ho * Wo * Cout * KH * KW * cin
    if name in ("msc_conv2d_fwd", "msc_conv2d_dgrad"):
        return conv(*s[:9])
    if name == "msc_conv2d_wgrad":
        return conv(*s[1:10])
    if name == "msc_stem_s2d_conv_fwd":        # scalars: B, Hs, Ws, Cout, dtype — the real 7x7x3 taps
        return 2.0 * s[0] * s[1] * s[2] * s[3] * 147
    if name == "msc_stem_s2d_conv_wgrad":      # scalars: ws_bytes, B, Hs, Ws, Cout, accumulate, dtype
        return 2.0 * s[1] * s[2] * s[3] * s[4] * 147
    if name == "msc_linear_fwd":
        return 2.0 * s[0] * s[1] * s[2]
    if name == "msc_linear_dgrad":      # scalars: dact, M, N, K, dtype
        return 2.0 * s[1] * s[2] * s[3]
    if name == "msc_linear_wgrad":
        return 2.0 * s[1] * s[2] * s[3]
    if name == "msc_gemm_batched":
        return 2.0 * s[0] * s[1] * s[2] * s[8]
    return None


def gemm_bytes(name, s):
    """algorithmic HBM bytes of the same calls: every operand read once, the output written once (bf16 activations /
    compute weights, fp32 weight gradients); what `roofline.traffic` is to be compared with."""
    def conv(B, H, W, Cin, Cout, KH, KW, st, p, wbytes):
        Ho, Wo = (H + 2 * p - KH) // st + 1, (W + 2 * p - KW) // st + 1
        return 2.0 * B * H * W * Cin + 2.0 * B * Ho * Wo * Cout + wbytes * Cout * KH * KW * Cin
    if name in ("msc_conv2d_fwd", "msc_conv2d_dgrad"):
        return conv(*s[:9], 2.0)
    if name == "msc_conv2d_wgrad":
        return conv(*s[1:10], 4.0)
    if name == "msc_stem_s2d_conv_fwd":
        return 2.0 * s[0] * s[1] * s[2] * 16 + 2.0 * s[0] * s[1] * s[2] * s[3] + 2.0 * s[3] * 256
    if name == "msc_stem_s2d_conv_wgrad":
        return 2.0 * s[1] * s[2] * s[3] * 16 + 2.0 * s[1] * s[2] * s[3] * s[4] + 4.0 * s[4] * 256
    if name == "msc_linear_fwd":
        M, N, K = s[0], s[1], s[2]
        return 2.0 * (M * K + N * K + M * N)
    if name == "msc_linear_dgrad":
        M, N, K = s[1], s[2], s[3]
        return 2.0 * (M * K + N * K + M * N)
    if name == "msc_linear_wgrad":
        M, N, K = s[1], s[2], s[3]
        return 2.0 * (M * K + M * N) + 4.0 * N * K
    if name == "msc_gemm_batched":
        M, N, K, z = s[0], s[1], s[2], s[8]
        return 2.0 * z * (M * K + N * K + M * N)
    return None


def cpu_baseline(cfg_name, seconds_budget=80.0):
    """oracle fwd+bwd (fp32, all host threads) on a bounded sample of the same workload -> frames/s.
    Procedure (BASELINE.md "CPU-baseline plan"): 2 warm-up steps + 5 timed steps at one batch size, MEDIAN step time.
    The batch is 32 frames when one probing step at 16 frames says 7 steps of 32 fit the time budget, else 16 (a small
    batch cannot feed a many-core host, which would understate the baseline)."""
    import statistics
    import torch
    from oracle import model as O
    cfg = O.named_config(cfg_name)
    m = O.build_oracle(cfg, seed=0)

    def make_step(B):
        batch = O.synthetic_batch(cfg, B, seed=0)

        def step():
            for p in m.parameters():
                p.grad = None
            out = m(batch["image"], batch.get("points"), batch.get("state"), batch["snr_db"], batch["noise"], batch["fading"])
            loss, _ = O.total_loss(out, batch["image"], cfg, batch["ppo"])
            loss.backward()
        return step

    B = 16
    step = make_step(B)
    step()                                   # allocator / thread-pool warm-up (not counted as one of the two below)
    t0 = time.time()
    step()
    t16 = time.time() - t0
    if 2.0 * t16 * 7 < seconds_budget:       # 2 warm-ups + 5 timed steps at 32 frames fit the budget
        B = 32
        step = make_step(B)
    n_warm, n_timed, times = 2, 5, []
    for _ in range(n_warm):
        step()
    for _ in range(n_timed):
        t0 = time.time()
        step()
        times.append(time.time() - t0)
    med = statistics.median(times)
    return {"value": B / med, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "step_s_median": med, "step_s_min": min(times), "step_s_max": max(times),
            "sample": f"oracle/model.py fwd+bwd, config {cfg_name}, batch {B}: {n_warm} warm-up + {n_timed} timed steps, median, fp32, "
                      f"{torch.get_num_threads()} threads of {os.cpu_count()} host cpus, no optimizer step"}


def parity_leg(cfg_name, model, frames=32):
    """The second half of BASELINE.json's metric ("+ decoder-PSNR parity vs CPU ref"), in the mode and on the model that was just
    benchmarked: the oracle (checker only; never on the timed path) receives the benchmarked model's CURRENT parameters, both run one
    forward on the same `frames` synthetic frames with EXPLICIT channel noise, and the line reports decoder-PSNR delta, max |logit|
    delta and whether the arg-max actions agree (all frames / frames whose oracle top-2 margin exceeds 0.1)."""
    import torch
    from oracle import model as O
    ocfg = O.named_config(cfg_name)
    om = O.build_oracle(ocfg, seed=0)
    missing = om.load_state_dict(model.export_state_dict(), strict=False)
    assert not missing.unexpected_keys and all(k.rsplit(".", 1)[-1] in ("running_mean", "running_var", "num_batches_tracked")
                                               for k in missing.missing_keys), missing      # (BatchNorm runs on batch statistics on both sides)
    batch = O.synthetic_batch(ocfg, frames, seed=4242)
    with torch.no_grad():
        oout = om(batch["image"], batch.get("points"), batch.get("state"), batch["snr_db"], batch["noise"], batch["fading"])
    d = model.rt.device
    g = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in batch.items() if k != "ppo"}
    with torch.no_grad():
        out = model(g["image"], g.get("points"), g.get("state"), g["snr_db"], g["noise"], g["fading"])
        logits, _ = model.logits_value(out)
        rec = model.recon_image(out)
    torch.cuda.synchronize()
    psnr_g = O.psnr(rec.float().cpu(), batch["image"]).mean().item()
    psnr_o = O.psnr(oout["recon"], batch["image"]).mean().item()
    ga, oa = model.select_action(out).cpu(), O.select_action(oout["logits"])
    top2 = oout["logits"].topk(2, -1).values
    margin = top2[:, 0] - top2[:, 1]
    sure = margin > 0.1
    differ = ga != oa
    # a frame whose two best oracle logits lie closer together than twice the bf16 logit error may legitimately pick the other one
    return {"frames": frames, "noise": "explicit (same tensors on both sides)", "psnr_db_gpu": psnr_g, "psnr_db_oracle": psnr_o,
            "psnr_abs_db": abs(psnr_g - psnr_o), "logit_max_abs": (logits.float().cpu() - oout["logits"]).abs().max().item(),
            "actions_equal": bool(torch.equal(ga, oa)), "actions_equal_where_margin_gt_0p1": bool(torch.equal(ga[sure], oa[sure])),
            "frames_with_margin_gt_0p1": int(sure.sum()), "frames_with_different_action": int(differ.sum()),
            "largest_oracle_margin_among_them": (margin[differ].max().item() if differ.any() else 0.0), "checker": "oracle/model.py (fp32, host cores) on the benchmarked model's parameters"}


def host_inputs_leg(args, cfg, model, trainer, dev_batch, ms_device_resident):
    """--host-inputs: the same step fed from a HOST-resident frame source (multimodal_sc_amd/frames.py): camera frames as uint8
    bytes + fp32 clouds + ego state in ordinary host memory -> pinned ring (filler thread) -> H2D of batch k+1 on the
    weight-gradient side stream while step k runs -> decoded on the main stream at the start of step k+1.  Reported beside
    `value` (which stays the device-resident number, as the contract says)."""
    import numpy as np
    import torch
    from multimodal_sc_amd.frames import FrameStager
    B, nb = args.batch, 8
    rng = np.random.default_rng(0)
    src = {"camera": rng.integers(0, 256, size=(nb * B, cfg.img_size, cfg.img_size, 3), dtype=np.uint8)}
    if cfg.use_lidar:
        src["lidar"] = dev_batch["points"].cpu().numpy().repeat(nb, 0)
    if cfg.use_state:
        src["ego"] = dev_batch["state"].cpu().numpy().repeat(nb, 0)
    ppo = {k: v.cpu().numpy() for k, v in dev_batch["ppo"].items()}
    st = FrameStager(model.rt, src, B, extra_fn=lambda k: ppo)

    own = [None]
    direct = os.environ.get("MSC_BENCH_OWN_INPUTS", "1") != "0"     # (0: A/B knob — decode into the stager's tensor, step() copies it)

    def step():
        b = st.next(image_out=own[0]["image"] if own[0] is not None else None)      # camera bytes decoded straight into the step's input
        st.prefetch()
        feed = {k: b[k] for k in ("image", "points", "state") if k in b}        # same keys as the device-resident batch
        feed.update(snr_db=dev_batch["snr_db"], noise=None, fading=None, ppo=dict(b["extra"]))
        terms = trainer.step(feed)
        if own[0] is None and direct:
            own[0] = trainer.input_buffers(feed)
        return terms

    for _ in range(max(args.warmup, nb + 2)):       # one whole pass over the source ring first (first touch of its pages is not the steady state)
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st.close()
    ms = 1e3 * dt / args.steps
    return {"value": B * args.steps / dt, "unit": "frames/s", "ms_per_step": ms, "vs_device_resident": ms_device_resident / ms,
            "h2d_bytes_per_batch": st.bytes_per_batch, "copy_stream": "weight-gradient side stream (hipMemcpyAsync from a pinned ring)",
            "source": f"host memory ring of {nb} batches: uint8 camera frames + fp32 clouds + ego state"}


def instrumented_pass(model, trainer, batch, torch):
    """ONE stream (no side stream — Runtime.call_side falls back while rt.profile is set — and no BEV-branch stream), HIP events on
    that stream around every C-ABI call: each kernel is timed running alone.  -> (per-call breakdown, gemm ms, flops, bytes, launches)"""
    rt = model.rt
    torch.cuda.synchronize()
    rt.join_side()
    rt.profile = []
    branch = rt.branch_streams
    rt.branch_streams = False
    try:
        trainer.forward_backward(batch)          # no collective here: only rank 0 runs the instrumented pass
        model.params.adam_step(lr=1e-4)
        torch.cuda.synchronize()
    finally:
        rt.branch_streams = branch
    by = {}
    gemm_t = gemm_f = gemm_b = 0.0
    n_gemm = 0
    for name, s, e0, e1 in rt.profile:
        ms = e0.elapsed_time(e1)
        d = by.setdefault(name, [0, 0.0])
        d[0] += 1
        d[1] += ms
        f = gemm_flops(name, s)
        if f is not None:
            gemm_t += ms
            gemm_f += f
            gemm_b += gemm_bytes(name, s)
            n_gemm += 1
    rt.profile = None
    return by, gemm_t, gemm_f, gemm_b, n_gemm


def instrumented_only(args, model, trainer, batch):
    import torch
    instrumented_pass(model, trainer, batch, torch)          # warm-up (workspaces, code objects)
    ts = []
    for _ in range(args.instrumented_only):
        _, gemm_t, gemm_f, _, n_gemm = instrumented_pass(model, trainer, batch, torch)
        ts.append(gemm_t)
    ts.sort()
    med = ts[len(ts) // 2]
    print(json.dumps({"mode": "instrumented-only", "passes": len(ts), "kernel": "gemm_kernel<bf16,*> family (C-ABI GEMM calls incl. their split-K helpers)",
                      "kernel_ms_per_pass_median": med, "kernel_ms_per_pass_all": ts, "launches_per_pass": n_gemm,
                      "algorithmic_gflop_per_pass": gemm_f / 1e9, "achieved_tflops": gemm_f / (med * 1e-3) / 1e12}))
    return 0


def allreduce_probe(model, trainer, dist, torch, reps=5):
    """N>1, outside the timed region: each piece of the gradient exchange alone (barrier, `reps` all-reduces, max over
    ranks) -> ms and algorithmic GB/s per rank, so a scaling run explains itself."""
    G = model.params.G
    ex = trainer.exchange
    pieces = [(n, lo, hi) for n, lo, hi in ex.log] if ex is not None and ex.log else [("flat", 0, G.numel())]
    out = {}
    for name, lo, hi in pieces:
        buf = torch.zeros(hi - lo, dtype=G.dtype, device=G.device)
        dist.all_reduce(buf)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=G.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item()) * 1e3
        out[name] = {"mbytes": (hi - lo) * 4 / 1e6, "ms": ms, "GB_per_s": (hi - lo) * 4 / 1e9 / (ms * 1e-3)}
    return out


def _flat(batch, prefix=""):
    """(key, tensor) pairs of a nested batch dict in a fixed order"""
    for k in sorted(batch):
        v = batch[k]
        if isinstance(v, dict):
            yield from _flat(v, prefix + k + ".")
        elif hasattr(v, "data_ptr"):
            yield prefix + k, v


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE: start the N ranks ourselves.  Runs BEFORE anything in this process
    touches the GPU (no torch.cuda call, no HIP call) and starts torch.distributed.run as a child — never an exec of this
    process; stdout of the child (rank 0's JSON line) is relayed line by line, the child's return code is ours."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def selftest_ranks(args):
    """--selftest-ranks: the launch path alone (rendezvous, a sum over ranks, one JSON line from rank 0) on the gloo backend,
    no GPU and no model — what tests/test_bench_cpu.py drives through the self-spawn above."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    assert world == args.gpus, (world, args.gpus)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    if rank == world - 1 and os.environ.get("MSC_SELFTEST_EXIT"):     # (test hook: a failing rank must fail the parent)
        return int(os.environ["MSC_SELFTEST_EXIT"])
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "rank_sum": float(t.item())}))
    if world > 1:
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--host-inputs", action="store_true",
                    help="also time the step fed from a host-resident frame source (pinned ring + H2D staging)")
    ap.add_argument("--alt-frames-per-gpu", type=int, default=32, metavar="F",
                    help="also time the step at F frames per GPU after the main measurement (BASELINE.json configs[3]: batch 256 sharded "
                         "8x = 32 frames per GPU); reported as `alt_geometry`, never as `value`; 0 = skip")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle parity block of the JSON line")
    ap.add_argument("--selftest-ranks", action="store_true", help="launch-path self-test on gloo: no GPU, no model")
    ap.add_argument("--instrumented-only", type=int, default=0, metavar="N",
                    help="profiling aid: run ONLY N single-stream instrumented passes (every kernel alone, HIP events around it) — "
                         "`rocprofv3 --kernel-trace --stats` of this command is the summary roofline.achieved must agree with")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, argv)
    if args.selftest_ranks:
        return selftest_ranks(args)

    import torch
    import torch.distributed as dist
    from multimodal_sc_amd.model import MultimodalSC, named_config
    from multimodal_sc_amd.synthetic import synthetic_batch     # seeded input generator of the product package: the GPU
    from multimodal_sc_amd.trainer import Trainer               # leg imports nothing from oracle/

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # MSC_BENCH_BACKEND=gloo rehearses the N>1 code path on a one-GPU box (all ranks share cuda:0, gradients summed
    # through host memory); the driver's real runs use the default "nccl" = RCCL, one rank per GPU.
    backend = os.environ.get("MSC_BENCH_BACKEND", "nccl")
    local = local % max(1, torch.cuda.device_count()) if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend)

    cfg = named_config(args.config)
    model = MultimodalSC(cfg, dtype=args.dtype, device=dev, seed=0)
    trainer = Trainer(model, lr=1e-4, use_ppo=True)
    B = args.batch
    cpu_batch = synthetic_batch(cfg, B, seed=1000 + rank)
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in cpu_batch.items() if k != "ppo"}
    batch["ppo"] = {k: v.to(dev) for k, v in cpu_batch["ppo"].items()}
    batch["noise"] = None      # throughput mode: in-kernel Philox noise
    batch["fading"] = None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.instrumented_only:
        return instrumented_only(args, model, trainer, batch)

    for _ in range(args.warmup):
        trainer.step(batch)
    # the synthetic batch lives in the trainer's own input buffers from here on (what a frame source writes into: no device-to-device
    # staging copy inside a step); None while no launch list is recorded (fewer warm-up steps than the recorder needs, MSC_REPLAY=0)
    own = trainer.input_buffers(batch) if os.environ.get("MSC_BENCH_OWN_INPUTS", "1") != "0" else None     # (0: A/B knob)
    if own is not None:
        for (k, d), (_, v) in zip(_flat(own), _flat(batch)):
            d.copy_(v)
        batch = own
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        terms = trainer.step(batch)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    met = model.metrics(terms)

    result = {
        "metric": "multimodal frames/sec (whole job) at batch 64 per GPU, training step fwd+bwd+allreduce+Adam",
        "value": world * B * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "inputs": "HBM-resident in the trainer's input buffers" if own is not None else "HBM-resident, copied into the trainer's input buffers every step",
        "config": {"workload": f"{args.config}: camera 3x{cfg.img_size}x{cfg.img_size}"
                               + (f" + {cfg.n_points}-pt LiDAR BEV" if cfg.use_lidar else "") + (" + ego-state" if cfg.use_state else "")
                               + f", fusion transformer T={cfg.n_tokens} d={cfg.d_model} L={cfg.n_layers}, {cfg.channel.upper()} 10 dB, PPO head",
                   "frames_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}"},
        "frames_per_s_per_gpu": B * args.steps / dt, "psnr_db": met["psnr"], "loss": met["loss"],
    }
    if world > 1:
        # data-parallel sanity (outside the timed region): every rank must hold bit-identical parameters
        chk = model.params.P.double().sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        result["dp_params_in_sync"] = bool((lo == hi).item())
        result["rccl_ranks"] = dist.get_world_size()
        result["dp_backend"] = backend
        result["dp_schedule"] = [list(x) for x in trainer.exchange.log] if trainer.exchange is not None else "flat buckets"
        result["allreduce_alone"] = allreduce_probe(model, trainer, dist, torch)

    if args.alt_frames_per_gpu and args.alt_frames_per_gpu != B:
        # BASELINE.json configs[3] geometry (32 frames per GPU: batch 256 over 8 GPUs) beside the 64-per-GPU weak-scaling number: the
        # same step at the other shard size, every rank, same barrier + max-over-ranks timing.  Labelled; never `value`.
        Fa = args.alt_frames_per_gpu
        cpu_alt = synthetic_batch(cfg, Fa, seed=2000 + rank)
        alt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in cpu_alt.items() if k != "ppo"}
        alt["ppo"] = {k: v.to(dev) for k, v in cpu_alt["ppo"].items()}
        alt["noise"] = None
        alt["fading"] = None
        for _ in range(max(args.warmup, 4)):          # (a batch of another shape records its own launch list: 2 eager + 1 recorded step)
            trainer.step(alt)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            trainer.step(alt)
        barrier()
        dta = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dta], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dta = float(t.item())
        result["alt_geometry"] = {"label": f"{Fa} frames per GPU (BASELINE.json configs[3] shards batch 256 over 8 GPUs = 32 per GPU)",
                                  "frames_per_gpu": Fa, "global_batch": Fa * world, "value": world * Fa * args.steps / dta, "unit": "frames/s",
                                  "ms_per_step": 1e3 * dta / args.steps, "frames_per_s_per_gpu": Fa * args.steps / dta}

    if args.host_inputs and world == 1:
        result["host_inputs"] = host_inputs_leg(args, cfg, model, trainer, batch, result["ms_per_step"])

    if rank == 0 and not args.no_roofline:
        # instrumented pass: every kernel alone on one stream, HIP events on that stream around every C-ABI call (see instrumented_pass);
        # profiles/r4_kernel_stats_instrumented.csv is `rocprofv3 --kernel-trace --stats` of `bench.py --instrumented-only 8`
        by, gemm_t, gemm_f, gemm_b, n_gemm = instrumented_pass(model, trainer, batch, torch)
        achieved = gemm_f / (gemm_t * 1e-3) / 1e12
        traffic, traffic_src = pmc_hbm_bytes_per_gemm_launch(n_gemm)
        result["build_id"] = __import__("multimodal_sc_amd.lib", fromlist=["lib"]).lib.msc_build_id().decode()
        result["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                              "traffic_source": traffic_src, "algorithmic_bytes_per_launch": gemm_b / max(n_gemm, 1),
                              "kernel": "gemm_kernel<bf16,*> + gemm_pp_kernel<*> + conv_c64_kernel<*> + conv_c64_wgrad_kernel + stem_wgrad_kernel + gemm_skinny_kernel<*> (the MFMA contraction kernels: convolutions, linears)",
                              "launches_per_step": n_gemm, "algorithmic_gflop_per_step": gemm_f / 1e9, "kernel_ms_per_step": gemm_t,
                              # the same FLOPs over the WHOLE timed step (checkable against the driver's clock around the run):
                              # what the step sees, with every other kernel, the stream overlap and the host loop included
                              "whole_step_tflops": gemm_f / (result["ms_per_step"] * 1e-3) / 1e12,
                              "whole_step_frac": gemm_f / (result["ms_per_step"] * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                              # how this figure relates to the committed rocprofv3 summary of the same single-stream pass
                              "rocprof_stats": "profiles/r4_kernel_stats_instrumented.csv = rocprofv3 --kernel-trace --stats of `bench.py --instrumented-only 8`; "
                                               "profiles/r4_stats_vs_bench.txt: the event brackets read ~12 % longer than the kernel durations they enclose "
                                               "(marker + dispatch latency of ~5 us per C-ABI call), so `achieved` is the conservative of the two"}
        result["breakdown_ms_per_step"] = {k: round(v[1], 3) for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])}

    if rank == 0 and world == 1 and not args.no_parity:
        result["parity"] = parity_leg(args.config, model)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.config)
        result["gpu_over_cpu"] = result["frames_per_s_per_gpu"] / result["cpu_baseline"]["value"]

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
