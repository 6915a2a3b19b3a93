"""bench.py's multi-rank control flow on the CPU: `--gpus 2` with no launcher environment must start its two ranks itself,
rendezvous, time behind barriers, take the max over ranks and print exactly one JSON line from rank 0.
CGE_BENCH_DRYRUN=1 replaces the env handles by no-op stand-ins (there is no GPU here); everything around them is the real code."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)


def _check(p, n):
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["steps"] == 4 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["unit"] == "env-steps/s" and out["value"] > 0 and "DRY RUN" in out["data"]
    assert out["config"]["envs_per_gpu"] == 1 << 20 and out["config"]["parallelism"].startswith(f"env-sharded x{n}")
    return out


def test_gpus_2_launches_its_own_ranks():
    p = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1"], {"CGE_BENCH_DRYRUN": "1"})
    _check(p, 2)


def test_gpus_2_under_the_drivers_torchrun_line():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1"], {"CGE_BENCH_DRYRUN": "1"})
    _check(p, 2)


def test_a_failing_rank_fails_the_launcher():
    p = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--workload", "nonsense"], {"CGE_BENCH_DRYRUN": "1"})
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_without_a_gpu_the_real_path_refuses_loudly():
    p = _run([sys.executable, "bench.py", "--steps", "1", "--warmup", "0"], {})
    import torch
    if torch.cuda.device_count() == 0:
        assert p.returncode != 0 and "no CPU path" in (p.stderr + p.stdout)


def _check_split(p, n):
    """placement A of BASELINE config 5: the eight env types dealt round-robin over the ranks, total work fixed -> strong scaling"""
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "strong"
    assert out["unit"] == "env-steps/s" and out["value"] > 0 and "DRY RUN" in out["data"]
    assert out["config"]["workload"].startswith("hetero_split_131k")
    # rank 0 holds types 0, n, 2n, ... of the sorted type list
    types = sorted(["snake", "crypto", "traffic", "parking", "climate", "fleet", "manufacturing", "hospital"])
    assert out["config"]["env_types"] == types[0::n]
    return out


def test_hetero_split_one_type_per_rank_at_8():
    p = _run([sys.executable, "bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1", "--workload", "hetero_split_131k"],
             {"CGE_BENCH_DRYRUN": "1", "OMP_NUM_THREADS": "1"})
    _check_split(p, 8)


def test_hetero_split_uneven_dealing_at_3():
    """8 types over 3 ranks: ranks hold 3, 3 and 2 types — a rank with fewer types must still meet every barrier"""
    p = _run([sys.executable, "bench.py", "--gpus", "3", "--steps", "3", "--warmup", "1", "--workload", "hetero_split_131k"],
             {"CGE_BENCH_DRYRUN": "1", "OMP_NUM_THREADS": "1"})
    _check_split(p, 3)


def test_hetero_coresident_weak_scaling_at_2():
    p = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "hetero_131k"],
             {"CGE_BENCH_DRYRUN": "1", "OMP_NUM_THREADS": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "weak" and out["n_gpus"] == 2 and len(out["config"]["env_types"]) == 8


def test_episode_and_repeats_shape_the_line():
    """--episode: K := the workload's episode length; --repeats R: the K-step region is timed R times, `ms_per_step` is the median
    region and `spread` carries min / max / every sample."""
    p = _run([sys.executable, "bench.py", "--workload", "fleet_131k", "--episode", "--warmup", "1", "--repeats", "3", "--no-cpu-baseline"],
             {"CGE_BENCH_DRYRUN": "1", "OMP_NUM_THREADS": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["steps"] == 800 and out["spread"]["repeats"] == 3 and len(out["spread"]["all_ms_per_step"]) == 3
    assert out["spread"]["min_ms_per_step"] <= out["ms_per_step"] <= out["spread"]["max_ms_per_step"]
    assert sorted(out["spread"]["all_ms_per_step"])[1] == out["ms_per_step"]
