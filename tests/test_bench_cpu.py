"""bench.py's launch path on CPU: `python bench.py --gpus N` with WORLD_SIZE unset must start its own ranks (a child
`python -m torch.distributed.run`, never an exec), relay rank 0's JSON line and return the child's exit code — the form
the driver uses for the scaling runs.  --selftest-ranks swaps the model for a gloo sum over ranks, so no GPU is needed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_self_spawns_ranks():
    r = _run(["--gpus", "2", "--selftest-ranks"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out == {"selftest": True, "n_gpus": 2, "rank_sum": 3.0}


def test_bench_single_rank_needs_no_launcher():
    r = _run(["--gpus", "1", "--selftest-ranks"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_propagates_child_failure():
    r = _run(["--gpus", "2", "--selftest-ranks"], {"MSC_SELFTEST_EXIT": "3"})
    assert r.returncode != 0


def test_bench_rank_mode_when_world_size_is_set():
    """under torch.distributed.run (WORLD_SIZE set) the process is a rank and must not spawn again"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-ranks"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert sum(l.startswith("{") for l in r.stdout.splitlines()) == 1
