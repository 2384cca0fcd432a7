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


def test_watchdog_ends_a_hung_rank_with_a_diagnostic_and_a_nonzero_status():
    """VERDICT r5 item 5 / ADVICE r5: a rank that never joins a collective must not read as success.  Rank 1 of the launcher self-test
    sleeps instead of joining the all-reduce; the watchdog of the ranks names the phase on stderr and ends them with status 3."""
    r = run_bench(["--gpus", "2", "--launch-selftest"], {"BENCH_DIST_BACKEND": "gloo", "BENCH_SELFTEST_HANG_RANK": "1", "BENCH_WATCHDOG_S": "8"}, 300)
    assert r.returncode != 0
    assert "WATCHDOG" in r.stderr and "did not return within 8 s" in r.stderr and "status 3" in r.stderr
    assert not json_lines(r.stdout)


def test_roofline_object_of_the_multi_gpu_line_prices_the_step_not_a_launch():
    """for N > 1 the line must not report the single-GPU kernel's per-launch fields."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    one = b.roofline_record(256, 1, 20, 20 * 0.248, 3, 354.7e6)
    assert one["kernel"] == "cheb_sweep_vec4_kernel" and one["priced_per"] == "launch" and abs(one["avg_launch_us"] - 248.0 / 3) < 1e-9
    assert abs(one["frac"] - 112.0 * 256 ** 3 / 0.248e-3 / 8e12) < 1e-12 and one["hbm_real_frac"] is not None
    many = b.roofline_record(256, 8, 20, 20 * 0.060, 4, None)
    assert many["priced_per"] == "step of one GPU" and many["kernel"].startswith("slab route")
    assert many["avg_launch_us"] is None and many["algorithmic_bytes_per_launch"] is None and many["traffic"] is None
    assert many["launches_per_step"] == 4 and many["exchanges_per_step"] == 2 and abs(many["step_us"] - 60.0) < 1e-9
    # one GPU's share of the model bytes and of the three directions' flops over the whole step
    assert abs(many["achieved"] * 1e9 - 112.0 * 256 ** 3 / 8 / 60e-6) < 1.0
    assert abs(many["mfma_f64_frac"] - 3 * 254.0 ** 4 / 8 / 60e-6 / 78.6e12) < 1e-12


@pytest.mark.gpu
def test_bench_gpus_2_batched_leg_runs_under_the_gloo_rehearsal():
    """BENCH_BATCHED=1: the informational 4-vectors-per-exchange leg through the chebhip_comm callback transport (it used to need RCCL)."""
    r = run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--spinup", "2", "--size", "40"],
                  {"BENCH_DIST_BACKEND": "gloo", "BENCH_DIST_STRICT": "1", "BENCH_BATCHED": "1", "BENCH_DIST_TRANSPORT": "messages"}, 900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(json_lines(r.stdout)) == 1
    info = [ln for ln in r.stderr.splitlines() if "vectors per exchange" in ln]
    assert len(info) == 1, r.stderr[-2000:]
    rec = json.loads(info[0].split("exchange: ", 1)[1])
    assert rec.get("nrhs") == 4 and rec.get("vectors_agree") is True and rec["matvecs_per_s"] > 0, rec


@pytest.mark.gpu
def test_bench_gpus_2_local_transport_threads_print_one_parsed_line():
    """BENCH_DIST_TRANSPORT=local -- one process, N thread ranks (device r % count), the LOCAL transport with the
    peers' arrays read in place.  On the one-GPU box both ranks land on device 0: one parsed line with parity, labelled plumbing."""
    r = run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--spinup", "2", "--size", "64"], {"BENCH_DIST_TRANSPORT": "local"}, 900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    rec = lines[0]
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["value"] > 0
    assert rec["parity"]["rel_l2_vs_oracle"] <= 1e-10 and rec["parity"]["ranks"] == 2
    assert "direct-pull(LOCAL transport: 2 thread ranks" in rec["config"]["parallelism"] and "not a measurement" in rec["config"]["parallelism"]
    assert rec["roofline"]["priced_per"] == "step of one GPU" and rec["roofline"]["avg_launch_us"] is None


def test_local_transport_refuses_a_process_launcher():
    r = run_bench(["--gpus", "2", "--steps", "1"], {"BENCH_DIST_TRANSPORT": "local", "WORLD_SIZE": "2", "RANK": "0"}, 120)
    assert r.returncode != 0 and "ONE process" in r.stderr


@pytest.mark.gpu
def test_bench_gpus_2_takes_the_direct_route_among_processes_by_default():
    """Round 6: under a launcher (here: bench.py's own ranks, gloo, sharing the box's GPU) the default transport is the IPC direct
    route -- kept only after the node granted the mappings and two reduced-size matvecs equalled the oracle's; the batched leg runs on it."""
    r = run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--spinup", "2", "--size", "64"],
                  {"BENCH_DIST_BACKEND": "gloo", "BENCH_DIST_STRICT": "1", "BENCH_BATCHED": "1"}, 900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    rec = lines[0]
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["value"] > 0
    assert "direct-pull(IPC transport: one process per GPU" in rec["config"]["parallelism"] and "ipc_fallback" not in rec["config"]
    assert rec["parity"]["rel_l2_vs_oracle"] <= 1e-10 and rec["parity"]["P"] == 130 and rec["parity"]["also"]["P"] == 34
    assert rec["roofline"]["exchanges_per_step"] == 0 and rec["config"]["launches_per_step"] == 2
    info = [ln for ln in r.stderr.splitlines() if "vectors per exchange" in ln]
    assert len(info) == 1, r.stderr[-2000:]
    brec = json.loads(info[0].split("exchange: ", 1)[1])
    assert brec.get("nrhs") == 4 and brec.get("vectors_agree") is True, brec


@pytest.mark.gpu
def test_bench_gpus_2_as_typed_prints_one_parsed_line():
    r = run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--spinup", "2", "--size", "64"],
                  {"BENCH_DIST_BACKEND": "gloo", "BENCH_DIST_STRICT": "1", "BENCH_DIST_TRANSPORT": "messages"}, 900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    rec = lines[0]
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["value"] > 0
    assert rec["parity"]["rel_l2_vs_oracle"] <= 1e-10 and rec["parity"]["ranks"] == 2
    assert rec["config"]["parallelism"].startswith("slab2+all2all(C host") and "c_host_fallback" not in rec["config"]
    assert rec["roofline"]["priced_per"] == "step of one GPU" and rec["roofline"]["avg_launch_us"] is None
