"""bench.py end to end on the GPU at a small size (snake_64k): the line the driver parses carries every field of the contract, the
roofline and cpu_baseline objects, the repeats' spread and the steady_state block; and the numbers are consistent with each other
(value = envs x K / wall, frac = achieved / peak, the steady state is not faster than physics allows).  A child process, as the
driver runs it; the pytest process itself may already hold the GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contracts_fields():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CGE_BENCH_DRYRUN")}
    p = subprocess.run([sys.executable, "bench.py", "--workload", "snake_64k", "--steps", "10", "--warmup", "2", "--repeats", "3", "--steady-steps", "100"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    n, K = 1 << 16, 10
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == K and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "i8" and d["data"] == "synthetic" and d["unit"] == "env-steps/s"
    assert d["config"]["workload"].startswith("snake_64k") and d["config"]["envs_per_gpu"] == n and "model" not in d["config"]
    assert d["value"] == pytest.approx(n * K / (d["ms_per_step"] * 1e-3 * K), rel=1e-9)
    sp = d["spread"]
    assert sp["repeats"] == 3 and len(sp["all_ms_per_step"]) == 3 and sp["min_ms_per_step"] <= d["ms_per_step"] <= sp["max_ms_per_step"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9) and 0.0 < r["frac"] < 1.0
    assert "snake::rollout_kernel" in r["kernel"] and r["avg_launch_us"] > 0
    assert r["traffic"] is None or r["traffic"] > 0
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_env_step"] * r["env_steps_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9, rel=1e-6)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "env-steps/s" and c["value"] > 0 and c["cores"] >= 1 and "orc_snake" in c["sample"]
    s = d["steady_state"]
    assert s["after_steps"] == 100 and s["value"] > 0 and 0.0 < s["frac"] < 1.0 and s["kernel"] == r["kernel"] and s["spread"]["repeats"] == 3
    a = d["api_step"]
    assert a["path"] == "step" and a["value"] > 0 and "snake::step_kernel" in a["roofline"]["kernel"]
    w = d["rollout_with_final_obs"]
    assert w["dropped_last_launch"] == 0 and w["delivered_last_launch"] > 0
