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
        got = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"), weights_only=True)
        assert torch.allclose(got, ref, atol=1e-6), r
    # the optimizer folds 1/world into the Adam kernel's grad_scale: mean gradient = sum * (1/world)
    assert torch.allclose(ref / world, (ref * (1.0 / world)))


# ------------------------------------------------------------------------------------------------ the schedule that ships
def _exchange_worker(rank, world, port, out_dir, total, split, mid, pieces):
    """Drives multimodal_sc_amd.trainer.GradExchange — the object Trainer._eager_step uses for the default overlapped DP
    path — with a fake backward: each piece of G is 'produced' (filled from this rank's seeded gradient) just before its
    collective is issued, exactly the order the trainer follows (upper -> [mid] -> rest)."""
    from multimodal_sc_amd.trainer import GradExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = torch.randn(total, generator=torch.Generator().manual_seed(7 + rank))
    G = torch.full((total,), float("nan"))
    ex = GradExchange(total, split, mid, pieces)
    for step in range(2):                                   # two steps: begin_step() must reset the piece bookkeeping
        G.fill_(float("nan"))
        ex.begin_step()
        G[split:] = full[split:]
        works = [ex.start(G, "upper")]
        if ex.mid is not None:
            G[ex.mid:split] = full[ex.mid:split]
            works.append(ex.start(G, "mid"))
        lo, hi = ex.bounds("rest")
        G[lo:hi] = full[lo:hi]
        works.append(ex.start(G, "rest"))
        for w in works:
            w.wait()
        assert ex.covered(), ex.log
    logs = [None] * world
    dist.all_gather_object(logs, list(ex.log))
    assert all(l == logs[0] for l in logs), logs             # same collectives, same slices, same order on every rank
    torch.save({"G": G, "log": list(ex.log)}, os.path.join(out_dir, f"x{rank}.pt"))
    dist.destroy_process_group()


def test_overlapped_three_piece_exchange_world2(tmp_path):
    """N-rank result of the overlapped schedule == single-process sum (SURVEY.md §4 'DP all-reduce'), for the three
    shapes the trainer can take: 3 pieces (C2-C4), 2 pieces by knob (MSC_DP_PIECES=2), 2 pieces because the model has no
    mid offset (C1: camera only)."""
    total, split, mid, world = 100_032, 60_160, 20_480, 2
    ref = sum(torch.randn(total, generator=torch.Generator().manual_seed(7 + r)) for r in range(world))
    for pieces, m, want in ((3, mid, ["upper", "mid", "rest"]), (2, mid, ["upper", "rest"]), (3, None, ["upper", "rest"])):
        d = tmp_path / f"p{pieces}_{m}"
        d.mkdir()
        mp.spawn(_exchange_worker, args=(world, _free_port(), str(d), total, split, m, pieces), nprocs=world, join=True)
        for r in range(world):
            got = torch.load(os.path.join(str(d), f"x{r}.pt"), weights_only=True)
            assert [n for n, _, _ in got["log"]] == want
            assert torch.equal(got["G"], ref), (pieces, m, r)        # two ranks: a+b is exact, so bitwise


def test_exchange_bounds_follow_what_was_issued():
    from multimodal_sc_amd.trainer import GradExchange
    ex = GradExchange(1000, 600, 200, 3)
    assert ex.bounds("upper") == (600, 1000) and ex.bounds("mid") == (200, 600)
    assert ex.bounds("rest") == (0, 600)           # the mid piece was not issued (encoder backward ran as one phase)
    ex._issued_mid = True
    assert ex.bounds("rest") == (0, 200)
    assert GradExchange(1000, 600, 200, 2).mid is None
