"""`python bench.py --gpus N` typed without a launcher: the parent starts its N ranks as child processes
before it touches the GPU, relays rank 0's JSON line on its own stdout and exits with the children's status.

CPU part (no GPU here): the launcher itself through `--launch-selftest` (spawn, rendezvous on 127.0.0.1 over gloo, one all-reduce,
one line), and the status relay when the ranks fail (no GPU: every rank refuses, since the product path has no CPU fallback).
GPU part (-m gpu): the real N = 2 line on the one-GPU box with BENCH_DIST_BACKEND=gloo -- the two ranks share the card and the
exchange is staged through the host, i.e. plumbing and parity, never a measurement -- through the same entry."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(args, env_extra, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def json_lines(text):
    out = []
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                out.append(json.loads(ln))
            except ValueError:
                pass
    return out


@pytest.mark.parametrize("n", [2, 3])
def test_bench_starts_its_own_ranks(n):
    r = run_bench(["--gpus", str(n), "--launch-selftest"], {"BENCH_DIST_BACKEND": "gloo"}, 600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                       # ONE line, rank 0's
    rec = lines[0]
    assert rec["n_gpus"] == n and rec["rank_sum"] == n * (n + 1) // 2 and rec["spawned_by_bench"] is True


def test_bench_relays_the_status_of_failing_ranks():
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU check: on a GPU box the ranks would run")
    r = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--size", "34"], {"BENCH_DIST_BACKEND": "gloo"}, 600)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr and not json_lines(r.stdout)


def test_world_size_mismatch_is_refused():
    r = run_bench(["--gpus", "2", "--launch-selftest"], {"WORLD_SIZE": "1", "RANK": "0"}, 120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.gpu
def test_bench_gpus_2_as_typed_prints_one_parsed_line():
    r = run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--spinup", "2", "--size", "64"],
                  {"BENCH_DIST_BACKEND": "gloo", "BENCH_DIST_STRICT": "1"}, 900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    rec = lines[0]
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["value"] > 0
    assert rec["parity"]["rel_l2_vs_oracle"] <= 1e-10 and rec["parity"]["ranks"] == 2
    assert rec["config"]["parallelism"].startswith("slab2+all2all(C host") and "c_host_fallback" not in rec["config"]
