"""bench.py --gpus N starts its own N ranks: the launch / rendezvous / collective / relay plumbing on
CPU (BENCH_DRYRUN=1: no device work, gloo), the same code path the GPU runs take with RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the CPU baseline leg runs in the dry run too (it never needs a GPU); a sample of a fraction of a second for the tests
SMALL_CPU_SAMPLE = dict(BENCH_CPU_BUDGET_S="0.2", BENCH_CPU_SCALAR_STEPS="200", BENCH_CPU_SCALAR_FLOOR="50")


def run_bench(*argv, env_extra=None, timeout=300):
    env = dict(os.environ, BENCH_DRYRUN="1", OMP_NUM_THREADS="1", **SMALL_CPU_SAMPLE)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=env,
                         timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout  # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_bench_launches_its_own_ranks(n):
    d = run_bench("--gpus", str(n), "--steps", "3", "--warmup", "1")
    assert d["dry_run"] is True and d["value"] is None
    assert d["n_gpus"] == n and d["ranks_seen"] == n  # as many ranks answered the all-gather as were asked for
    assert len(d["per_rank_env_steps_per_s"]) == n
    assert d["episodes"]["completed"] == sum(4 * (r + 1) for r in range(n))  # every rank's statistics arrived
    assert d["config"]["parallelism"] == f"env-shard x{n}" and d["scaling"] == "weak"
    assert d["metric"].startswith("env-steps/sec whole node, 65 536 QQubeSwingUpSim")
    assert "gloo" in d["collective"]
    # the CPU baseline is in the multi-rank line too: timed by the parent that starts the ranks
    cb = d["cpu_baseline"]
    assert cb and cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and cb["scalar_all_cores"]["value"] > 0
    assert cb["timed_by"].startswith("launcher parent")
    assert d["preroll"] == 0  # (dry run: no launches; the GPU line reports BENCH_PREROLL's 400)


def test_bench_single_rank_needs_no_launcher():
    d = run_bench("--steps", "2", "--warmup", "1")
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["collective"].startswith("none")
    assert d["cpu_baseline"]["timed_by"].startswith("rank 0") and "preroll" in d


def test_bench_as_a_rank_of_an_external_launcher():
    """the driver's form: torchrun sets RANK / WORLD_SIZE and bench.py is a rank, not a launcher"""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BENCH_DRYRUN="1", OMP_NUM_THREADS="1", **SMALL_CPU_SAMPLE)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["timed_by"].startswith("rank 0")


def test_usable_cores_and_byte_models():
    sys.path.insert(0, ROOT)
    import bench

    assert 1 <= bench.usable_cores() <= len(os.sched_getaffinity(0))
    qq = bench.DIMS["qq-su"]
    assert bench.bytes_single_step(qq) == 117
    assert abs(bench.bytes_fused_step(qq, 100, 1) - (32.125 + 113 / 100)) < 1e-9
    assert abs(bench.bytes_fused_step(qq, 100, 2) - (52.125 + 113 / 100)) < 1e-9
