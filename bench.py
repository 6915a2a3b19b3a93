#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched step() hot path on MI355X, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload snake_1m] [--path rollout|step]

A "step" is one pass of the hot path over one batch: one env transition (with fused auto-reset) of every
instance on the GPU, the full observation / reward / flags written to HBM.  Default workload = BASELINE.json
configs[1]: SnakeEnv 10x10, 1,048,576 parallel envs per GPU (weak scaling: per-GPU batch fixed, global env
indices sharded contiguously, no collective on the data path).  Two paths are measured in every run:

  rollout (headline `value`)  the K timed steps fused in ONE launch per GPU — the K-steps-per-launch entry point
                              SURVEY.md section 7 / BASELINE.md section 3 prescribe for the roofline target: env state
                              stays in registers, the observation is still written to HBM every step, actions come
                              from the device-side counter hash (cge_hash_action);
  step   (`api_step` block)   K separate C-ABI step() calls through the VectorEnv facade with HBM-resident
                              actions — what a gymnasium.vector consumer calls.

`--path step` makes the API path the headline instead.  Prints ONE JSON line on rank 0 carrying `roofline` for
the dominant kernel (algorithmic bytes of SURVEY 8d / HIP-event time on the launch stream; `traffic` = HBM bytes
per launch from the committed rocprofv3 PMC passes, profiles/traffic.json) and `cpu_baseline` (the oracle's C
port of the reference on this box's host cores: reported, not the target).
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
# algorithmic bytes per env-step (secondary envs: obs + actions + reward/flags + 2 x state + RNG)
ENVS = {
    "snake":   dict(algo=145,  n_act=4, act_shape=(),   dtype="i8",   step_kernel="cge::snake::step_kernel<10, 256, 1, 8>",
                    roll_kernel="cge::snake::rollout_kernel<10, 256, 1, 8>", ref_py="3.4e5-4.2e5 steps/s/process"),
    "crypto":  dict(algo=2346, n_act=5, act_shape=(),   dtype="f64",  step_kernel="cge::crypto::step_kernel<false>",
                    roll_kernel="cge::crypto::step_kernel<true>", ref_py="1.64e3-1.68e3 steps/s/process"),
    "traffic": dict(algo=1134, n_act=3, act_shape=(9,), dtype="int32", step_kernel="cge::traffic::step_kernel<false>",
                    roll_kernel="cge::traffic::step_kernel<true>", ref_py="1.75e3-1.90e3 steps/s/process"),
    "parking": dict(algo=662,  n_act=8, act_shape=(),   dtype="f64",  step_kernel="cge::parking::step_kernel<false>",
                    roll_kernel="cge::parking::step_kernel<true>", ref_py="2.66e4 steps/s/process"),
    "climate": dict(algo=218,  n_act=None, act_shape=None, dtype="f64", step_kernel="cge::climate::step_kernel<false>",
                    roll_kernel="cge::climate::step_kernel<true>", ref_py="1.27e4 steps/s/process"),
    "fleet":   dict(algo=642,  n_act=8, act_shape=(3,), dtype="f64",  step_kernel="cge::fleet::step_kernel + cge::fleet::dense_kernel",
                    roll_kernel="cge::fleet::step_kernel + cge::fleet::dense_kernel", launches_per_step=True, ref_py="2.01e4 steps/s/process"),
    # 292 obs + 2 x 336 state + action/reward/flags; plus 10 bytes (quality f64 + meta u16) per product in the system, which the
    # per-type np.mean of the observation has to read every step: added from the measured mean occupancy (algo_per_product)
    # 972 obs + 2 x 768 state + action/reward/flags + ~70 MT19937 words read and written per step (2 x 280)
    "hospital": dict(algo=3078, n_act=35, act_shape=(), dtype="f64", step_kernel="cge::hosp::step_kernel<false>",
                     roll_kernel="cge::hosp::step_kernel<true>", ref_py="not in BASELINE.md"),
    "manufacturing": dict(algo=974, algo_per_product=10, n_act=25, act_shape=(), dtype="f64", step_kernel="cge::mfg::step_kernel<false>",
                          roll_kernel="cge::mfg::step_kernel<true>", ref_py="not in BASELINE.md"),
}
WORKLOADS = {
    "snake_1m": dict(env="snake", n=1 << 20, desc="SnakeEnv 10x10, 1,048,576 parallel envs per GPU, random actions, fused auto-reset"),
    "snake_64k": dict(env="snake", n=1 << 16, desc="SnakeEnv 10x10, 65,536 envs per GPU (quick check)"),
    "crypto_1m": dict(env="crypto", n=1 << 20, desc="crypto_trading_env discrete, 1,048,576 parallel envs per GPU"),
    "traffic_262k": dict(env="traffic", n=1 << 18, desc="traffic_management_env (9 intersections), 262,144 parallel envs per GPU"),
    "parking_131k": dict(env="parking", n=1 << 17, desc="smart_parking_env, 131,072 parallel envs per GPU"),
    "climate_131k": dict(env="climate", n=1 << 17, desc="smartclimate, 131,072 parallel envs per GPU"),
    "fleet_131k": dict(env="fleet", n=1 << 17, desc="fleet_management_env, 131,072 parallel envs per GPU"),
    "manufacturing_131k": dict(env="manufacturing", n=1 << 17, desc="smart_manufacturing_env, 131,072 parallel envs per GPU"),
    "hospital_131k": dict(env="hospital", n=1 << 17, desc="hospital_management_env, 131,072 parallel envs per GPU"),
    "hetero_split_131k": dict(env="hetero_split", n=1 << 17,
                              desc="heterogeneous batch, placement A: every env type x 131,072, the types dealt round-robin over the GPUs "
                                   "(one type per GPU at N=8), results identical to placement B because seeds follow the global env index"),
    "hetero_131k": dict(env="hetero", n=1 << 17,
                        desc="heterogeneous batch: every implemented env type x 131,072, co-resident on each GPU, one HIP stream per type"),
}


def make_env(cge, name, n, dev, env0):
    kw = dict(device=dev, autoreset_mode="SameStep", env_index0=env0, reuse_buffers=True)
    if name == "snake":
        return cge.SnakeVectorEnv(n, grid_size=10, **kw)
    if name == "crypto":
        return cge.CryptoVectorEnv(n, action_type="discrete", **kw)
    return {"traffic": cge.TrafficVectorEnv, "parking": cge.ParkingVectorEnv, "climate": cge.ClimateVectorEnv,
            "fleet": cge.FleetVectorEnv, "manufacturing": cge.ManufacturingVectorEnv, "hospital": cge.HospitalVectorEnv}[name](n, **kw)


def make_actions(name, steps, n, dev):
    """Synthetic action stream resident in HBM before the timed region."""
    if name == "climate":
        return (torch.rand((steps, n, 1), device=dev) * 16 + 16, torch.randint(0, 2, (steps, n, 4), dtype=torch.int8, device=dev))
    spec = ENVS[name]
    return torch.randint(0, spec["n_act"], (steps, n) + spec["act_shape"], dtype=torch.int32, device=dev)


def act_at(name, actions, t):
    return (actions[0][t], actions[1][t]) if name == "climate" else actions[t]


def cpu_baseline(name, budget_s=12.0):
    """Times the oracle (C port of the reference, single-threaded per handle) on the box's CPU share: one handle per
    thread (ctypes releases the GIL), a bounded sample of the same workload (hash actions, auto-reset)."""
    import oracle
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))        # a 1-GPU box's CPU share is 16 cores
    n_each, k = {"snake": (8192, 250), "crypto": (256, 100), "traffic": (1024, 100), "parking": (1024, 100),
                 "climate": (2048, 100), "fleet": (1024, 100), "manufacturing": (512, 100), "hospital": (256, 100)}[name]
    ctor = {"snake": lambda: oracle.SnakeOracle(n_each, 10, oracle.SAME_STEP),
            "crypto": lambda: oracle.CryptoOracle(n_each, "discrete", oracle.SAME_STEP),
            "traffic": lambda: oracle.TrafficOracle(n_each, oracle.SAME_STEP), "parking": lambda: oracle.ParkingOracle(n_each, oracle.SAME_STEP),
            "climate": lambda: oracle.ClimateOracle(n_each, oracle.SAME_STEP), "fleet": lambda: oracle.FleetOracle(n_each, oracle.SAME_STEP),
            "manufacturing": lambda: oracle.ManufacturingOracle(n_each, oracle.SAME_STEP),
            "hospital": lambda: oracle.HospitalOracle(n_each, oracle.SAME_STEP)}[name]

    def new(c):
        h = ctor()
        h.seed(np.arange(c * n_each, (c + 1) * n_each, dtype=np.uint64))
        h.reset()
        return h

    o = new(0)
    t = time.perf_counter()
    o.rollout(k, 123, 0, 0)
    one = time.perf_counter() - t
    reps = max(1, min(int(budget_s / max(one, 1e-3)), 2000))
    handles = [new(c) for c in range(cores)]

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
                sample=f"{cores} threads x {n_each} envs x {reps * k} steps of the same workload, oracle/orc_{name}.c; "
                       f"single-core rate {n_each * k / one:.3e}",
                reference_python_note=f"the reference's own Python, measured in the build container (8-core Xeon 2.6 GHz): "
                                      f"{ENVS[name]['ref_py']} (BASELINE.md section 2)")


def pmc_traffic(kernel):
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(kernel)
    except (OSError, ValueError):
        return None


def roofline(name, kernel, gpu_ms, launches, steps_per_launch, n, occupancy=0.0):
    algo = ENVS[name]["algo"] + ENVS[name].get("algo_per_product", 0) * occupancy
    launch_s = gpu_ms * 1e-3 / launches
    achieved = algo * n * steps_per_launch / launch_s / 1e9
    return {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic(kernel) if steps_per_launch == 1 else None, "algorithmic_bytes_per_env_step": algo,
            "env_steps_per_launch": n * steps_per_launch, "avg_launch_us": launch_s * 1e6,
            "timing": "HIP events on the launch stream over the timed region"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="snake_1m", choices=sorted(WORKLOADS))
    ap.add_argument("--path", default="rollout", choices=["rollout", "step"])
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
    # CGE_BENCH_REHEARSAL=1: rehearse the N>1 control flow on a box with fewer GPUs than ranks (ranks share devices, gloo
    # instead of RCCL).  Only for checking the launch / barrier / reduction logic; its numbers mean nothing.
    rehearsal = os.environ.get("CGE_BENCH_REHEARSAL") == "1" and torch.cuda.device_count() < world
    if rehearsal:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import custom_gymnasium_environments_amd as cge
    wl = WORKLOADS[args.workload]
    n, K, W = wl["n"], args.steps, args.warmup
    dev = torch.device("cuda", local_rank)
    if wl["env"] == "hetero":
        names = sorted(ENVS)
    elif wl["env"] == "hetero_split":                               # placement A: type k lives on rank k % world, whole (131,072 envs)
        names = [nm for k, nm in enumerate(sorted(ENVS)) if k % world == rank]
    else:
        names = [wl["env"]]
    split = wl["env"] == "hetero_split"
    envs = {nm: make_env(cge, nm, n, dev, 0 if split else rank * n) for nm in names}
    if len(names) > 1:
        streams = {nm: torch.cuda.Stream(device=dev) for nm in names}
    else:
        streams = {names[0]: torch.cuda.current_stream(dev)}
    for e in envs.values():
        e.reset(seed=0)
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_rollout(k, t0):
        for nm in names:
            with torch.cuda.stream(streams[nm]):
                envs[nm].rollout(k, action_seed=123, t0=t0)

    def run_steps(actions, lo, hi):
        for t in range(lo, hi):
            for nm in names:
                with torch.cuda.stream(streams[nm]):
                    envs[nm].step(act_at(nm, actions[nm], t))

    occupancy = {}

    def mean_occupancy():
        return {nm: float(envs[nm].info("in_system").mean().item()) for nm in names if "algo_per_product" in ENVS[nm]}

    def timed(fn, tag=None):
        before = mean_occupancy()
        barrier()
        evs = {nm: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for nm in names}
        t0 = time.perf_counter()
        for nm in names:
            evs[nm][0].record(streams[nm])          # on the stream the kernels are launched on
        fn()
        for nm in names:
            evs[nm][1].record(streams[nm])
        barrier()
        wall = time.perf_counter() - t0
        after = mean_occupancy()
        occupancy[tag] = {nm: 0.5 * (before[nm] + after[nm]) for nm in before}
        return wall, {nm: evs[nm][0].elapsed_time(evs[nm][1]) for nm in names}

    results = {}
    run_rollout(max(W, 1), 0)                                        # fused rollout: K steps in one launch per env type
    results["rollout"] = timed(lambda: run_rollout(K, W), "rollout")
    actions = {nm: make_actions(nm, K + W, n, dev) for nm in names}  # API path: K step() calls, HBM-resident actions
    run_steps(actions, 0, W)
    results["step"] = timed(lambda: run_steps(actions, W, W + K), "step")
    if "snake" in envs:
        assert envs["snake"].invalid_action_count() == 0

    def reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearsal else dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    walls = {p: reduce_max(results[p][0]) for p in results}
    if rank == 0:
        head, other = args.path, ("step" if args.path == "rollout" else "rollout")
        total_envs = n * len(ENVS) if split else n * len(names) * world

        def block(path):
            wall, gpu_ms = walls[path], results[path][1]
            b = {"path": path, "value": total_envs * K / wall, "unit": "env-steps/s", "ms_per_step": wall * 1e3 / K}
            rl = {}
            for nm in names:
                kern = ENVS[nm]["roll_kernel" if path == "rollout" else "step_kernel"]
                # fleet's rollout is K (step, dense) launch pairs, not one fused launch: price it per pair
                fused = path == "rollout" and not ENVS[nm].get("launches_per_step")
                rl[nm] = roofline(nm, kern, gpu_ms[nm], 1 if fused else K, K if fused else 1, n, occupancy[path].get(nm, 0.0))
            b["roofline"] = rl[names[0]] if len(names) == 1 else rl
            return b

        hb = block(head)
        out = {
            "metric": "env steps/sec (whole node) at 1M parallel envs; achieved HBM GB/s vs peak",
            "value": hb["value"], "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": hb["ms_per_step"], "higher_is_better": True, "scaling": "strong" if split else "weak", "vs_baseline": None,
            "dtype": ENVS[names[0]]["dtype"] if len(names) == 1 else "mixed", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['desc']}", "envs_per_gpu": n * len(names), "env_types": names,
                       "path": (("rollout: K (step, dense-reset) launch pairs queued by one C-ABI call, obs written to HBM every step, device-side action hash"
                                 if all(ENVS[nm].get("launches_per_step") for nm in names) else
                                 "fused rollout: the K steps in one launch per GPU, obs written to HBM every step, device-side action hash")
                                if head == "rollout" else "K C-ABI step() calls through the VectorEnv facade, HBM-resident actions"),
                       "autoreset": "SameStep", "parallelism": f"env-sharded x{world}, no data-path collective"},
            "roofline": hb["roofline"] if len(names) == 1 else hb["roofline"].get("snake", hb["roofline"][names[0]]),
        }
        if len(names) > 1:
            out["roofline_per_env_type"] = hb["roofline"]
        out["api_step" if other == "step" else "fused_rollout"] = block(other)
        if not args.no_cpu_baseline and world == 1:            # the CPU port is timed at N=1 only
            out["cpu_baseline"] = cpu_baseline(names[0] if len(names) == 1 else "snake")
        print(json.dumps(out), flush=True)
    for e in envs.values():
        e.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
