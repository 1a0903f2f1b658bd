"""Data-parallel exchange step on CPU: world_size-2 gloo processes run the same bucketed flat all-reduce the GPU
trainer runs over RCCL (multimodal_sc_amd/trainer.py), and the result must equal the single-process sum."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multimodal_sc_amd.trainer import allreduce_flat, bucket_bounds


def test_bucket_bounds_cover_exactly():
    for total in (1, 63, 64, 65, 1000, 28_491_978):
        for nb in (1, 3, 4, 7):
            b = bucket_bounds(total, nb)
            assert b[0][0] == 0 and b[-1][1] == total and len(b) <= nb
            assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
            assert all(lo % 64 == 0 for lo, _ in b)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    for w in allreduce_flat(flat, n_buckets=4):
        w.wait()
    torch.save(flat, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2(tmp_path):
    n, world = 100_003, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    ref = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert torch.allclose(got, ref, atol=1e-6), r
    # the optimizer folds 1/world into the Adam kernel's grad_scale: mean gradient = sum * (1/world)
    assert torch.allclose(ref / world, (ref * (1.0 / world)))
