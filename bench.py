#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched step() hot path on MI355X, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload snake_1m] [--path step|rollout]

A "step" is one pass of the hot path over one batch: one `step()` of 1,048,576 SnakeEnv 10x10
instances per GPU (BASELINE.json configs[1]; weak scaling: per-GPU batch fixed, env indices sharded
contiguously, no collective on the data path).  Inputs (actions) are resident in HBM before the timed
region; every step writes the full (N,10,10) int8 observation, reward and flags, with fused auto-reset.
Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` for the
dominant kernel (HIP-event timed on the launch stream) and `cpu_baseline` (the oracle's C port of
the reference timed on this box's host cores; reported, not the target).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
# algorithmic bytes per env-step, SURVEY.md section 8d
ALGO_BYTES = {"snake": 145, "crypto": 2346, "traffic": 1134}
WORKLOADS = {
    "snake_1m": dict(env="snake", n_per_gpu=1 << 20, grid=10,
                     desc="SnakeEnv 10x10, 1,048,576 parallel envs per GPU, random actions, fused auto-reset"),
    "snake_64k": dict(env="snake", n_per_gpu=1 << 16, grid=10, desc="SnakeEnv 10x10, 65,536 envs per GPU (quick check)"),
    "crypto_1m": dict(env="crypto", n_per_gpu=1 << 20,
                      desc="crypto_trading_env discrete, 1,048,576 parallel envs per GPU, random actions, fused auto-reset"),
}
WORKLOADS["traffic_262k"] = dict(env="traffic", n_per_gpu=1 << 18,
                                 desc="traffic_management_env (9 intersections), 262,144 parallel envs per GPU, random actions, fused auto-reset")
KERNELS = {("snake", "step"): "cge::snake::step_kernel<10, 256, 1, 8>", ("snake", "rollout"): "cge::snake::rollout_kernel<10, 256, 1, 8>",
           ("crypto", "step"): "cge::crypto::step_kernel<false>", ("crypto", "rollout"): "cge::crypto::step_kernel<true>",
           ("traffic", "step"): "cge::traffic::step_kernel<false>", ("traffic", "rollout"): "cge::traffic::step_kernel<true>"}
N_ACTIONS = {"snake": 4, "crypto": 5, "traffic": 3}
DTYPE = {"snake": "i8", "crypto": "f64", "traffic": "int32"}


def cpu_baseline_snake(grid, budget_s=12.0):
    """Times the oracle (C port of snake_env.py, single-threaded per handle) on all host cores:
    one handle per thread, ctypes releases the GIL.  Bounded sample of the same workload."""
    import oracle
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))        # a 1-GPU box's CPU share is 16 cores
    n_each, k = 8192, 250
    # calibrate on one core, then size the sample to roughly budget_s of wall time
    o = oracle.SnakeOracle(n_each, grid, oracle.SAME_STEP)
    o.reset()
    t = time.perf_counter()
    o.rollout(k, 123, 0, 0)
    one = time.perf_counter() - t
    reps = max(1, min(int(budget_s / max(one, 1e-3)), 2000))
    handles = []
    for c in range(cores):
        h = oracle.SnakeOracle(n_each, grid, oracle.SAME_STEP)
        h.seed(np.arange(c * n_each, (c + 1) * n_each, dtype=np.uint64))
        h.reset()
        handles.append(h)

    def work(c):
        for r in range(reps):
            handles[c].rollout(k, 123, r * k, c * n_each)

    th = [threading.Thread(target=work, args=(c,)) for c in range(cores)]
    t = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t
    steps = cores * reps * n_each * k
    return dict(value=steps / dt, unit="env-steps/s", cores=cores, kind="port",
                sample=f"{cores} threads x {n_each} envs x {reps * k} steps of the same workload (hash actions, auto-reset), "
                       f"oracle/orc_snake.c; single-core rate {n_each * k / one:.3e}",
                reference_python_note="reference Python measured in the build container (8-core Xeon 2.6 GHz): "
                                      "3.4e5-4.2e5 steps/s/process, 1.81e6 over 8 processes (BASELINE.md section 2)")


def cpu_baseline_crypto(budget_s=12.0):
    import oracle
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    n_each, k = 256, 100
    o = oracle.CryptoOracle(n_each, "discrete", oracle.SAME_STEP)
    o.reset()
    t = time.perf_counter()
    o.rollout(k, 123, 0, 0)
    one = time.perf_counter() - t
    reps = max(1, min(int(budget_s / max(one, 1e-3)), 2000))
    handles = []
    for c in range(cores):
        h = oracle.CryptoOracle(n_each, "discrete", oracle.SAME_STEP)
        h.seed(np.arange(c * n_each, (c + 1) * n_each, dtype=np.uint64))
        h.reset()
        handles.append(h)

    def work(c):
        for r in range(reps):
            handles[c].rollout(k, 123, r * k, c * n_each)

    th = [threading.Thread(target=work, args=(c,)) for c in range(cores)]
    t = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t
    return dict(value=cores * reps * n_each * k / dt, unit="env-steps/s", cores=cores, kind="port",
                sample=f"{cores} threads x {n_each} envs x {reps * k} steps (hash actions, auto-reset, obs assembled every step), "
                       f"oracle/orc_crypto.c; single-core rate {n_each * k / one:.3e}",
                reference_python_note="reference Python in the build container: 1.64e3-1.68e3 steps/s/process (BASELINE.md section 2)")


def cpu_baseline_traffic(budget_s=12.0):
    import oracle
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    n_each, k = 1024, 100
    o = oracle.TrafficOracle(n_each, oracle.SAME_STEP)
    o.reset()
    t = time.perf_counter()
    o.rollout(k, 123, 0, 0)
    one = time.perf_counter() - t
    reps = max(1, min(int(budget_s / max(one, 1e-3)), 2000))
    handles = []
    for c in range(cores):
        h = oracle.TrafficOracle(n_each, oracle.SAME_STEP)
        h.seed(np.arange(c * n_each, (c + 1) * n_each, dtype=np.uint64))
        h.reset()
        handles.append(h)

    def work(c):
        for r in range(reps):
            handles[c].rollout(k, 123, r * k, c * n_each)

    th = [threading.Thread(target=work, args=(c,)) for c in range(cores)]
    t = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t
    return dict(value=cores * reps * n_each * k / dt, unit="env-steps/s", cores=cores, kind="port",
                sample=f"{cores} threads x {n_each} envs x {reps * k} steps (hash actions, auto-reset, obs assembled every step), "
                       f"oracle/orc_traffic.c (collapsed state, no O(V*I) sqrt loop); single-core rate {n_each * k / one:.3e}",
                reference_python_note="reference Python in the build container: 1.75e3-1.90e3 steps/s/process (BASELINE.md section 2)")


def pmc_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json, written by
    tools_profile_summary.py: 2*FETCH_SIZE + WRITE_SIZE, the gfx950 read-side correction of
    MI355X_MICROARCH.md section HBM).  None when no profile of this kernel has been committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(kernel)
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="snake_1m", choices=sorted(WORKLOADS))
    ap.add_argument("--path", default="step", choices=["step", "rollout"],
                    help="step: one C-ABI step() launch per step with HBM-resident actions (default); "
                         "rollout: the K steps fused in one launch (device-side action hash)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                     "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import custom_gymnasium_environments_amd as cge
    wl = WORKLOADS[args.workload]
    n = wl["n_per_gpu"]
    K, W = args.steps, args.warmup
    dev = torch.device("cuda", local_rank)
    if wl["env"] == "snake":
        env = cge.SnakeVectorEnv(n, grid_size=wl["grid"], device=dev, autoreset_mode="SameStep", env_index0=rank * n,
                                 reuse_buffers=True)
    elif wl["env"] == "crypto":
        env = cge.CryptoVectorEnv(n, action_type="discrete", device=dev, autoreset_mode="SameStep", env_index0=rank * n,
                                  reuse_buffers=True)
    else:
        env = cge.TrafficVectorEnv(n, device=dev, autoreset_mode="SameStep", env_index0=rank * n, reuse_buffers=True)
    env.reset(seed=0)
    # synthetic action stream, resident in HBM before timing
    ashape = (K + W, n, 9) if wl["env"] == "traffic" else (K + W, n)
    actions = torch.randint(0, N_ACTIONS[wl["env"]], ashape, dtype=torch.int32, device=dev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if args.path == "step":
        for t in range(W):
            env.step(actions[t])
        barrier()
        t0 = time.perf_counter()
        ev0.record()                      # same stream the kernels are launched on (torch current stream)
        for t in range(W, W + K):
            env.step(actions[t])
        ev1.record()
        barrier()
        wall = time.perf_counter() - t0
        launches = K
        kernel = KERNELS[(wl["env"], "step")]
    else:
        env.rollout(max(W, 1), action_seed=123, t0=0)
        barrier()
        t0 = time.perf_counter()
        ev0.record()
        env.rollout(K, action_seed=123, t0=W)
        ev1.record()
        barrier()
        wall = time.perf_counter() - t0
        launches = 1
        kernel = KERNELS[(wl["env"], "rollout")]
    gpu_ms = ev0.elapsed_time(ev1)
    if wl["env"] == "snake":
        assert env.invalid_action_count() == 0
    fused = None
    if args.path == "step":
        # reported beside the headline, outside its timed region: the same K steps fused in one launch
        # (state stays in registers; obs still written to HBM every step; device-side action hash)
        env.rollout(max(W, 1), action_seed=123, t0=0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.rollout(K, action_seed=123, t0=W)
        e1.record()
        torch.cuda.synchronize()
        f_ms = e0.elapsed_time(e1)
        f_ach = ALGO_BYTES[wl["env"]] * n * K / (f_ms * 1e-3) / 1e9
        fused = {"path": "rollout (one launch, K fused steps)", "kernel": KERNELS[(wl["env"], "rollout")],
                 "env_steps_per_s_per_gpu": n * K / (f_ms * 1e-3), "us_per_step": f_ms * 1e3 / K,
                 "roofline": {"bound": "hbm", "achieved": f_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": f_ach / HBM_PEAK_GBS}}

    wall_t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall_max = float(wall_t.item())

    if rank == 0:
        total_steps = n * world * K
        value = total_steps / wall_max
        algo = ALGO_BYTES[wl["env"]]
        launch_s = gpu_ms * 1e-3 / launches
        units_per_launch = n * (K if args.path == "rollout" else 1)
        achieved = algo * units_per_launch / launch_s / 1e9
        out = {
            "metric": "env steps/sec (whole node) at 1M parallel envs; achieved HBM GB/s vs peak",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall_max * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[wl["env"]], "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['desc']}", "envs_per_gpu": n, "path": args.path,
                       "autoreset": "SameStep", "parallelism": f"env-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(kernel),
                         "algorithmic_bytes_per_env_step": algo, "env_steps_per_launch": units_per_launch,
                         "avg_launch_us": launch_s * 1e6, "timing": "HIP events on the launch stream over the timed region"},
        }
        if fused is not None:
            out["fused_rollout"] = fused
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = (cpu_baseline_snake(wl["grid"]) if wl["env"] == "snake" else
                                   cpu_baseline_crypto() if wl["env"] == "crypto" else cpu_baseline_traffic())
        print(json.dumps(out), flush=True)
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
