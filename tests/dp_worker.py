"""Child process of tests/test_dp_gpu.py: one data-parallel rank running Trainer.step on cuda:0.
usage: python dp_worker.py <rank> <world> <port> <out.pt> <config> <frames> <steps> [shard]
world == 1: a single-process run (no process group) of shard `shard` — the independent answer the DP runs are checked against;
with MSC_DP_FORCE=1 in the environment it initialises a ONE-rank process group on MSC_DP_BACKEND (nccl = RCCL) and the Trainer takes the
data-parallel code path (exit code 77 + "RCCL_UNAVAILABLE: ..." on stdout when the communicator cannot be created on this box).
Ranks share the one GPU of the test box and exchange gradients through gloo (host memory) — the same Trainer code path
the real runs take over RCCL, one rank per GPU.  Schedule knobs (MSC_DP_PIECES, MSC_DP_OVERLAP) come from the environment."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out, cfg_name, B, steps = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
    shard = int(sys.argv[8]) if len(sys.argv) > 8 else rank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    import torch
    import torch.distributed as dist
    from multimodal_sc_amd.model import MultimodalSC, named_config
    from multimodal_sc_amd.synthetic import synthetic_batch
    from multimodal_sc_amd.trainer import Trainer
    torch.cuda.set_device(0)
    forced = world == 1 and os.environ.get("MSC_DP_FORCE") == "1"
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif forced:
        backend = os.environ.get("MSC_DP_BACKEND", "nccl")
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
            else:
                dist.init_process_group(backend, rank=0, world_size=1)
            probe = torch.ones(1024, device="cuda:0")
            dist.all_reduce(probe)                      # the communicator is created lazily: make it exist (or fail) here
            torch.cuda.synchronize()
            assert float(probe.sum()) == 1024.0
        except Exception as e:  # noqa: BLE001 — reported to the test, which records the text and xfails
            print("RCCL_UNAVAILABLE:", type(e).__name__, str(e).replace("\n", " | ")[:1500], flush=True)
            sys.exit(77)
    cfg = named_config(cfg_name)
    m = MultimodalSC(cfg, dtype="bf16", device="cuda:0", seed=0)          # same seed: identical initial parameters
    tr = Trainer(m, lr=1e-3)
    m.channel_rank = shard
    cb = synthetic_batch(cfg, B, seed=100 + shard)                          # each rank its own shard of frames
    gb = {k: (v.to("cuda:0") if torch.is_tensor(v) else v) for k, v in cb.items() if k != "ppo"}
    gb["ppo"] = {k: v.to("cuda:0") for k, v in cb["ppo"].items()}
    for _ in range(steps):
        tr.step(gb)
    torch.cuda.synchronize()
    m.rt.join_side()
    torch.cuda.synchronize()
    log = list(tr.exchange.log) if tr.exchange is not None else []
    replayed = any(st["prog"] is not None for st in tr._programs.values())
    torch.save({"P": m.params.P.cpu(), "G": m.params.G.cpu(), "log": log, "replayed": replayed}, out)
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
