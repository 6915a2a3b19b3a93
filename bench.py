#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched step() hot path on MI355X, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload snake_1m] [--path rollout|step]

A "step" is one pass of the hot path over one batch: one env transition (with fused auto-reset) of every
instance on the GPU, the full observation / reward / flags written to HBM.  Default workload = BASELINE.json
configs[1]: SnakeEnv 10x10, 1,048,576 parallel envs per GPU (weak scaling: per-GPU batch fixed, global env
indices sharded contiguously, no collective on the data path).  Two paths are measured in every run:

  rollout (headline `value`)  the K timed steps fused, kc steps per launch (kc = K unless the [kc, N, obs] trajectory would
                              exceed --traj-gib): env state stays in registers, and EVERY step's observation, reward and
                              flag is written to its own place in HBM — a [kc, N, *obs] trajectory plus [kc, N] reward /
                              flag arrays, tens of GB, so no byte is absorbed by rewriting a cache-resident buffer; actions
                              come from the device-side counter hash (cge_hash_action);
  step   (`api_step` block)   K separate C-ABI step() calls through the VectorEnv facade with HBM-resident
                              actions — what a gymnasium.vector consumer calls.

`--path step` makes the API path the headline instead.  `--gpus N` with N > 1 and no launcher environment starts the N ranks
itself (fresh child processes, before anything touches the GPU) and relays rank 0's line.  Prints ONE JSON line on rank 0
carrying `roofline` for the dominant kernel and `cpu_baseline` (the oracle's C port of the reference on this box's host
cores: reported, not the target).  roofline fields:
  achieved     bytes the timed kernel is OBLIGED to move per launch (`algorithmic_bytes_per_env_step` x env-steps per launch)
               / average launch time (HIP events on the launch stream); step(): SURVEY 8d's per-step figure; fused rollout:
               the same figure minus the state that legitimately stays in registers and the action read
  traffic      HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/traffic.json,
               2*FETCH_SIZE + WRITE_SIZE per env-step x env-steps per launch); frac_moved = traffic / time / peak
  peak         8 TB/s (spec); peak_measured = the larger of this box's device-to-device copy bandwidth (bytes read + written) and its
               pure fill_ (store) rate, both measured in this run and both reported — a write-dominated kernel is held against the fill rate
The K-step timed region (barrier + synchronize on both sides, max over ranks) is run `--repeats` times (default 9; episodes restarted
before each one where the workload fixes the episode phase): `value` / `ms_per_step` are the MEDIAN region, `spread` holds min, max and
every region; the roofline's launch time is the tighter of two HIP-event brackets of the same K steps — the median region's own events and a
second pass queued behind a primer copy (`first_pass_launch_us`, `second_pass_launch_us`, `roofline.timing`).  Every leg starts after half a second of idle.
snake_1m's regions are not preceded by a reset, and W warm-up steps after the batch was reset together it is still leaving that state (every digit
ring full; `spread.all_ms_per_step` shows the regions getting slower): `steady_state` times the same K steps again `--steady-steps` (2,000) steps later.
Single-type workloads time the rollout once more with the
terminal-observation side output registered (`rollout_with_final_obs`; the headline leg runs without it, like rounds 1-3).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
# Bytes per env-step.  algo = SURVEY.md 8d's figure for one step() call (secondary envs: obs + actions + reward /
# flags + 2 x state + RNG).  The fused rollout is priced on what IT must move: obs + reward (4) + flag (1) + stream (generator
# words and tables that live in HBM even inside a fused launch) + 2 x the state that is NOT register-resident, plus the resident
# record once per launch (2 x resident / kc); it reads no actions (device-side hash).
ENVS = {
    # the 16-byte hot record is all a launch reads and writes of the state (snakes of <= 25 cells); the digit ring adds < 1 B / env-step
    # ep_typ: a typical episode length under random actions — sizes the terminal-row side output of the rollout leg (collect_final_obs)
    "snake":   dict(algo=145,  obs=100, state=16, resident=16, stream=0, ep_typ=12, n_act=4, act_shape=(),   dtype="i8",
                    ref_py="3.4e5-4.2e5 steps/s/process"),
    # rollout: the 50-candle window (1,200 B) is resident in LDS for a launch (read once: resident_ro); per step only the new candle
    # (24 B, written through) and the generator words (144 B) stream
    "crypto":  dict(algo=2346, obs=1044, state=64, resident=64, resident_ro=1200, stream=168, n_act=5, act_shape=(),   dtype="f64", ref_py="1.64e3-1.68e3 steps/s/process"),
    # round 4: array-of-structs record of 62 dwords padded to 256 B, register-resident over a fused launch
    "traffic": dict(algo=1134, obs=520, state=256, resident=256, stream=60, n_act=3, act_shape=(9,), dtype="int32", ref_py="1.75e3-1.90e3 steps/s/process"),
    "parking": dict(algo=662,  obs=52, state=288, resident=288, stream=24, n_act=8, act_shape=(),   dtype="f64", ref_py="2.66e4 steps/s/process"),
    "climate": dict(algo=218,  obs=36, state=80, resident=80, stream=0, n_act=None, act_shape=None, dtype="f64", ref_py="1.27e4 steps/s/process"),
    # fleet's rollout is K (step, dense) launch pairs: the record goes through HBM every step
    "fleet":   dict(algo=642,  obs=304, state=160, resident=0, stream=0, ep_typ=30, n_act=8, act_shape=(3,), dtype="f64", launches_per_step=True, ref_py="2.01e4 steps/s/process"),
    # step(): SURVEY 8d's figure (972 obs + 2 x 768 state + action/reward/flags + ~70 MT19937 words read and written per step, 2 x 280).
    # Fused rollout, round 4: the whole 848-byte record stays in registers for the launch (rounds 1-3 re-loaded 672 of 768 bytes per step
    # and the rollout was credited 2 x 672 B per env-step for it): obs + reward + flag + the generator words
    "hospital": dict(algo=3078, obs=972, state=848, resident=848, stream=560, n_act=35, act_shape=(), dtype="f64", ref_py="not in BASELINE.md"),
    # 292 obs + 2 x 336 state + action/reward/flags; plus 10 bytes (quality f64 + meta u16) per product in the system, which the
    # per-type np.mean of the observation has to read every step: added from the measured mean occupancy (algo_per_product)
    "manufacturing": dict(algo=1028, algo_per_product=8, obs=292, state=368, resident=368, stream=0, n_act=25, act_shape=(), dtype="f64", ref_py="not in BASELINE.md"),
}
WORKLOADS = {
    "snake_1m": dict(env="snake", n=1 << 20, desc="SnakeEnv 10x10, 1,048,576 parallel envs per GPU, random actions, fused auto-reset"),
    "snake_64k": dict(env="snake", n=1 << 16, desc="SnakeEnv 10x10, 65,536 envs per GPU (quick check)"),
    "crypto_1m": dict(env="crypto", n=1 << 20, episode_start=True, episode=1000, traj_gib=128.0,
                      desc="crypto_trading_env discrete, 1,048,576 parallel envs per GPU, timed from the start of an episode"),
    # episode_start: every env is re-seeded and reset right before each timed region, so the K timed steps are steps 1..K of an
    # episode — the spawn phase (the RNG-heavy one: the reference stops spawning at 50 vehicles, ~step 200) is inside the window,
    # and --steps 1000 times exactly one whole episode
    "traffic_262k": dict(env="traffic", n=1 << 18, episode_start=True, episode=1000,
                         desc="traffic_management_env (9 intersections), 262,144 parallel envs per GPU, timed from the start of an episode"),
    "parking_131k": dict(env="parking", n=1 << 17, episode_start=True, episode=1440,
                         desc="smart_parking_env, 131,072 parallel envs per GPU, timed from the start of an episode"),
    "climate_131k": dict(env="climate", n=1 << 17, episode_start=True, episode=1440,
                         desc="smartclimate, 131,072 parallel envs per GPU, timed from the start of an episode"),
    "fleet_131k": dict(env="fleet", n=1 << 17, episode_start=True, episode=800,
                       desc="fleet_management_env, 131,072 parallel envs per GPU, timed from the start of an episode (episodes end early and restart in place)"),
    "manufacturing_131k": dict(env="manufacturing", n=1 << 17, episode_start=True, episode=1500,
                               desc="smart_manufacturing_env, 131,072 parallel envs per GPU, timed from the start of an episode"),
    "hospital_131k": dict(env="hospital", n=1 << 17, episode_start=True, episode=1440,
                          desc="hospital_management_env, 131,072 parallel envs per GPU, timed from the start of an episode"),
    "hetero_split_131k": dict(env="hetero_split", n=1 << 17,
                              desc="heterogeneous batch, placement A: every env type x 131,072, the types dealt round-robin over the GPUs "
                                   "(one type per GPU at N=8), results identical to placement B because seeds follow the global env index"),
    "hetero_131k": dict(env="hetero", n=1 << 17,
                        desc="heterogeneous batch: every implemented env type x 131,072, co-resident on each GPU, one HIP stream per type"),
}


def roll_algo(name, kc, occupancy=0.0, final_rate=0.0):
    """final_rate: terminal rows delivered per env-step (measured: the rollout leg registers the side output of SAME_STEP rollouts, so
    that a fused launch hands over everything the k step() calls it replaces return): obs bytes + the 8-byte index entry each"""
    s = ENVS[name]
    return (s["obs"] + 4 + 1 + s["stream"] + 2 * (s["state"] - s["resident"]) + (2.0 * s["resident"] + s.get("resident_ro", 0)) / max(kc, 1)
            + s.get("algo_per_product", 0) * occupancy + (s["obs"] + 8) * final_rate)


def step_algo(name, occupancy=0.0):
    return ENVS[name]["algo"] + ENVS[name].get("algo_per_product", 0) * occupancy


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
    if actions is None:                                  # dry run
        return None
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
    """bytes per env-step of `kernel` from the committed PMC passes (tools/profile.sh -> profiles/traffic.json), or None"""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            rec = json.load(f).get(kernel)
        return float(rec["bytes_per_env_step"]) if isinstance(rec, dict) else None
    except (OSError, ValueError, KeyError, TypeError):
        return None


def measure_copy_bandwidth(dev):
    """The second roofline denominator SURVEY 8d asks for: this box's device-to-device copy bandwidth (bytes read + bytes
    written per second) and its pure-store rate, measured here with torch on 2-GiB buffers."""
    x = torch.empty(1 << 29, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)

    def t(fn, reps=8):
        fn()
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b) / reps * 1e-3
    gb = x.numel() * 4 / 1e9
    out = {"copy_GBs": 2 * gb / t(lambda: y.copy_(x)), "fill_GBs": gb / t(lambda: x.fill_(1.0))}
    del x, y
    torch.cuda.empty_cache()
    return out


def roofline(name, path, kernel, gpu_ms, launches, steps_per_launch, n, occupancy, measured, final_rate=0.0):
    algo = roll_algo(name, steps_per_launch, occupancy, final_rate) if path == "rollout" and not ENVS[name].get("launches_per_step") else \
        (roll_algo(name, 1, occupancy, final_rate) if path == "rollout" else step_algo(name, occupancy))
    launch_s = gpu_ms * 1e-3 / launches
    env_steps = n * steps_per_launch
    achieved = algo * env_steps / launch_s / 1e9
    per_step = pmc_traffic(kernel)
    r = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": per_step * env_steps if per_step is not None else None,
         "traffic_bytes_per_env_step": per_step,
         "frac_moved": (per_step * env_steps / launch_s / 1e9 / HBM_PEAK_GBS) if per_step is not None else None,
         "traffic_over_algorithmic": (per_step / algo) if per_step is not None else None,
         "algorithmic_bytes_per_env_step": algo, "env_steps_per_launch": env_steps, "avg_launch_us": launch_s * 1e6,
         "timing": "HIP events on the launch stream over the timed region",
         "algorithmic_note": ("fused rollout: obs + reward + flag + generator/table stream + non-resident state, resident record once per launch"
                              if path == "rollout" else "SURVEY 8d per-step figure")}
    if per_step is not None:
        r["traffic_note"] = ("2 x FETCH_SIZE + WRITE_SIZE: the gfx950 half-count of wide coalesced reads (MI355X_MICROARCH.md, HBM) applied to ALL "
                             "reads of the kernel — for scattered 32..128-byte generator reads that may over-count the read side (conservative)")
    if measured:
        r["peak_copy_measured"] = measured["copy_GBs"]
        r["peak_fill_measured"] = measured["fill_GBs"]
        r["peak_measured"] = max(measured["copy_GBs"], measured["fill_GBs"])
        r["frac_of_copy"] = achieved / measured["copy_GBs"]
        r["frac_of_fill"] = achieved / measured["fill_GBs"]
        r["frac_of_measured"] = achieved / r["peak_measured"]
        r["peak_measured_note"] = ("this box, this run, 2-GiB torch buffers: copy_ = bytes read + written per second, fill_ = bytes written per second; "
                                   "peak_measured is the larger: the yardstick for a write-dominated kernel is the fill rate")
    return r


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: the parent starts the ranks itself, as fresh processes, before it makes any GPU call
def self_launch(n_ranks):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    out0 = procs[0].communicate()[0]
    rcs = [procs[0].returncode]
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:           # rank 0 is done and a sibling is not: end exactly that child
            p.kill()
            rcs.append(p.wait())
    lines = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    for ln in (out0 or "").splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if any(rcs) or not lines:
        sys.stderr.write(f"bench.py: rank exit codes {rcs}\n")
        sys.exit(max([abs(rc) for rc in rcs if rc] + [1]))
    print(lines[-1], flush=True)
    sys.exit(0)


class _DryEnv:
    """CGE_BENCH_DRYRUN=1 stand-in for an env handle: no device, no work.  It exists so the N > 1 control flow (self-launch,
    rendezvous, barriers, max-over-ranks, rank-0 JSON) can be exercised by a CPU test; a dry run's numbers mean nothing and
    its line says so."""

    def __init__(self, n):
        self.n = n

    def reset(self, seed=None):
        return None

    def rollout(self, k, **kw):
        time.sleep(1e-4 * k)

    def step(self, a):
        time.sleep(1e-4)

    def info(self, f):
        return torch.zeros(1)

    def invalid_action_count(self):
        return 0

    def last_kernel(self):
        return "dry-run"

    def close(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="snake_1m", choices=sorted(WORKLOADS))
    ap.add_argument("--path", default="rollout", choices=["rollout", "step"])
    ap.add_argument("--steady-steps", type=int, default=2000,
                    help="workloads timed without a reset before every region (snake_1m): after the legs, this many more untimed rollout steps, "
                         "then the same K steps timed again -> the `steady_state` block (0: skip)")
    ap.add_argument("--traj-gib", type=float, default=None,
                    help="cap on the per-env-type trajectory buffer of the rollout leg (default 24; crypto_1m 128: its rows are 1,044 B, and a launch "
                         "should be long enough to amortise the 1,200-byte window load — 125 steps per launch on the 288-GB part)")
    ap.add_argument("--manifest", default=None, help="write {kernel: env-steps launched} here (tools/profile.sh uses it to turn PMC bytes into bytes per env-step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=9, help="the K-step timed region is run this many times; value / ms_per_step = the median region")
    ap.add_argument("--graph", action="store_true", help="third leg: the step() calls captured in ONE HIP graph (torch.cuda.CUDAGraph) and replayed — "
                                                         "what the host-bound small batches gain when the per-call launch path is taken off the clock")
    ap.add_argument("--no-final-obs-leg", action="store_true", help="skip the second rollout leg (the same K steps with the terminal-observation side "
                                                                    "output registered): tools/profile.sh, so that a kernel's PMC bytes are the headline leg's")
    ap.add_argument("--episode", action="store_true", help="--steps := the workload's episode length (one whole episode per timed region)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)                       # never returns

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    dry = os.environ.get("CGE_BENCH_DRYRUN") == "1"
    # CGE_BENCH_REHEARSAL=1: rehearse the N>1 control flow on a box with fewer GPUs than ranks (ranks share devices, gloo
    # instead of RCCL).  Only for checking the launch / barrier / reduction logic; its numbers mean nothing.
    rehearsal = dry or (os.environ.get("CGE_BENCH_REHEARSAL") == "1" and torch.cuda.device_count() < world)
    if not dry:
        if torch.cuda.device_count() == 0:
            sys.exit("bench.py needs an MI355X (no HIP device visible); there is no CPU path")
        if rehearsal:
            local_rank = local_rank % torch.cuda.device_count()
        elif local_rank >= torch.cuda.device_count():
            sys.exit(f"rank {rank}: --gpus {world} but only {torch.cuda.device_count()} device(s) visible")
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    wl = WORKLOADS[args.workload]
    if args.episode and wl.get("episode"):
        args.steps = wl["episode"]
    n, K, W, R = wl["n"], args.steps, args.warmup, max(1, args.repeats)
    dev = torch.device("cpu") if dry else torch.device("cuda", local_rank)
    if wl["env"] == "hetero":
        names = sorted(ENVS)
    elif wl["env"] == "hetero_split":                               # placement A: type k lives on rank k % world, whole (131,072 envs)
        names = [nm for k, nm in enumerate(sorted(ENVS)) if k % world == rank]
    else:
        names = [wl["env"]]
    split = wl["env"] == "hetero_split"
    if dry:
        envs = {nm: _DryEnv(n) for nm in names}
        streams = {nm: None for nm in names}
    else:
        import custom_gymnasium_environments_amd as cge
        envs = {nm: make_env(cge, nm, n, dev, 0 if split else rank * n) for nm in names}
        if len(names) > 1:
            streams = {nm: torch.cuda.Stream(device=dev) for nm in names}
        else:
            streams = {names[0]: torch.cuda.current_stream(dev)}
    for e in envs.values():
        e.reset(seed=0)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    class on_stream:                                     # `with on_stream(nm):` = torch.cuda.stream(...) or nothing in a dry run
        def __init__(self, nm):
            self.cm = None if dry or len(names) == 1 else torch.cuda.stream(streams[nm])   # one env type: its stream IS the current one

        def __enter__(self):
            return self.cm.__enter__() if self.cm else None

        def __exit__(self, *a):
            return self.cm.__exit__(*a) if self.cm else False

    # rollout leg: kc steps per launch, every step's obs / reward / flag to its own place in a [kc, N, ...] trajectory
    traj_gib = args.traj_gib if args.traj_gib is not None else wl.get("traj_gib", 24.0)
    budget = traj_gib * (1 << 30) / (1 if len(names) == 1 else 4)
    kc = {nm: max(1, min(K, int(budget // (n * ENVS[nm]["obs"])))) for nm in names}
    # SAME_STEP rollouts can also deliver the terminal observation of every episode that ends (cge_<env>_rollout_final_obs: what
    # infos["final_obs"] carries on the step() leg) to a side output the caller registers.  The headline leg runs without it, as in rounds 1-3;
    # single-type workloads then time the same K steps once more WITH it (rollout_with_final_obs: rows delivered / dropped, what it costs).
    # Sized for ~3x the episode ends a launch is expected to see.
    fin_rows = {nm: min(kc[nm], 3 * -(-kc[nm] // ENVS[nm].get("ep_typ", 1000)) + 2) for nm in names}
    final_stats = {}
    launched = {}                                        # kernel name -> env-steps launched (for --manifest)
    ran = {}                                             # (env type, path) -> the kernel the library says it launched last

    def count(nm, path, steps):
        # the library reports the kernel it dispatched (cge_<env>_last_kernel): nothing here guesses a template instance
        kern = envs[nm].last_kernel()
        ran[(nm, path)] = kern
        launched[kern] = launched.get(kern, 0) + n * steps

    last_launch = {}

    def run_rollout(k_total, t0, marks=None):
        for nm in names:
            with on_stream(nm):
                done = 0
                while done < k_total:
                    k = min(kc[nm], k_total - done)
                    envs[nm].rollout(k, action_seed=123, t0=t0 + done, trajectory=True, per_step=True)
                    last_launch[nm] = k
                    count(nm, "rollout", k)
                    done += k
                    if marks is not None and not dry:                # one event per launch: the cost of a step early / late in the region
                        ev = torch.cuda.Event(enable_timing=True)
                        ev.record(streams[nm])
                        marks[nm].append((ev, done))

    def run_steps(actions, lo, hi, marks=None):
        seg = max(1, (hi - lo) // 10)
        for t in range(lo, hi):
            for nm in names:
                with on_stream(nm):
                    envs[nm].step(act_at(nm, actions[nm], t))
                    if marks is not None and not dry and ((t + 1 - lo) % seg == 0 or t + 1 == hi):
                        ev = torch.cuda.Event(enable_timing=True)
                        ev.record(streams[nm])
                        marks[nm].append((ev, t + 1 - lo))
        for nm in names:
            count(nm, "step", hi - lo)

    occupancy = {}
    host_latency_ms = {}
    second_ms = {}
    first_ms = {}
    spread = {}
    segments = {}
    primer = [None]

    def mean_occupancy():
        return {nm: float(envs[nm].info("in_system").mean().item()) for nm in names if "algo_per_product" in ENVS[nm]}

    def restart_episodes():
        if wl.get("episode_start"):
            for e in envs.values():
                e.reset(seed=0)

    def median(xs):
        xs = sorted(xs)
        return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])

    def cool():
        # every leg starts after half a second of idle, so that no leg is measured right behind another one's last launch
        sync()
        if not dry:
            time.sleep(0.5)

    def timed(fn, tag, second_pass):
        """R repeats of one timed region = fn(rep, marks): K steps between two barriers (the contract's clock).  Returns (median wall
        seconds, per-env GPU milliseconds of the median region's launches).  The wall clock is `value` / `ms_per_step`.  The GPU time
        feeds the roofline and must be the KERNELS' time: HIP events recorded on an idle stream also count the host's way to the first
        launch (stream switch, argument checks, the launch itself: 15-100 us, up to a fifth of a 20-step launch), so with `second_pass` the
        same launches run again queued behind a primer that keeps the card busy while the host enqueues event, launches and event —
        those events see the kernels back to back.  Without it the first passes' events are used."""
        walls, firsts, occ = [], [], []
        cool()
        for rep in range(R):
            restart_episodes()
            before = mean_occupancy()
            barrier()
            evs = {}
            if not dry:
                evs = {nm: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for nm in names}
            marks = {nm: [] for nm in names}
            t0 = time.perf_counter()
            for nm in evs:
                evs[nm][0].record(streams[nm])          # on the stream the kernels are launched on
            fn(rep, marks)
            for nm in evs:
                evs[nm][1].record(streams[nm])
            barrier()
            walls.append(time.perf_counter() - t0)
            after = mean_occupancy()
            occ.append({nm: 0.5 * (before[nm] + after[nm]) for nm in before})
            firsts.append({nm: (evs[nm][0].elapsed_time(evs[nm][1]) if evs else walls[-1] * 1e3) for nm in names})
            if evs:                                     # us per step of each segment of the region (a launch of the rollout leg, a tenth of the step() leg)
                seg = {}
                for nm in names:
                    prev_ev, prev_done, out = evs[nm][0], 0, []
                    for ev, done in marks[nm]:
                        out.append({"steps": [prev_done + 1, done], "us_per_step": prev_ev.elapsed_time(ev) * 1e3 / max(done - prev_done, 1)})
                        prev_ev, prev_done = ev, done
                    seg[nm] = out
                segments[tag] = seg
        occupancy[tag] = {nm: sum(o[nm] for o in occ) / len(occ) for nm in occ[0]} if occ and occ[0] else {}
        if occupancy[tag] and wl.get("episode_start") and not dry:
            # regions that start at a reset (and may end at the next one) see an empty system at both ends: the mean over the region
            # comes from one more, untimed, pass of the same K steps sampled at ten points
            restart_episodes()
            acc = {nm: 0.0 for nm in occupancy[tag]}
            pts, done_k = 10, 0
            for c in range(pts):
                k = (K * (c + 1)) // pts - done_k
                if k > 0:
                    for nm in acc:
                        envs[nm].rollout(k, action_seed=123, t0=t_roll + done_k, trajectory=False)
                    done_k += k
                cur = mean_occupancy()
                for nm in acc:
                    acc[nm] += cur[nm] / pts
            occupancy[tag] = acc
            restart_episodes()
        wall = median(walls)
        spread[tag] = {"repeats": R, "min_ms_per_step": min(walls) * 1e3 / K, "max_ms_per_step": max(walls) * 1e3 / K,
                       "all_ms_per_step": [w * 1e3 / K for w in walls]}
        first = {nm: median([f[nm] for f in firsts]) for nm in names}
        host_latency_ms[tag] = None
        if second_pass and not dry:
            seconds = []
            for rep in range(min(R, 3)):
                ev2 = {nm: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for nm in names}
                restart_episodes()
                if primer[0] is None:
                    primer[0] = (torch.empty(1 << 28, dtype=torch.float32, device=dev), torch.empty(1 << 28, dtype=torch.float32, device=dev))
                sync()
                for nm in names:
                    with on_stream(nm):
                        primer[0][0].copy_(primer[0][1])  # a 1-GiB copy (~0.45 ms at full HBM rate: a spinning wave would let the clocks drop) on this
                    ev2[nm][0].record(streams[nm])        # stream: the queue behind it fills before it ends
                fn(R + rep, None)
                for nm in names:
                    ev2[nm][1].record(streams[nm])
                barrier()
                seconds.append({nm: ev2[nm][0].elapsed_time(ev2[nm][1]) for nm in names})
            second = {nm: median([x[nm] for x in seconds]) for nm in names}
            host_latency_ms[tag] = {nm: first[nm] - second[nm] for nm in names}
            second_ms[tag] = second
            first_ms[tag] = first
            # Both brackets can only OVER-estimate the kernels' time for the K steps: the median region's own events also hold the host's way
            # to the first launch and, on the step() leg, whatever gaps the host leaves between calls; the second pass runs after R
            # regions, and where the regions are not preceded by a reset (snake_1m) the batch is R x K steps further from the reset it
            # started with: rocprofv3's per-dispatch times of the 1M-env snake launch go from 23 us per step in the first regions to
            # 28-29 in the last (profiles/r04_snake_summary.txt) - read as a clock drift at first, it is the batch leaving the state
            # it was reset into (the steady_state leg below and profiles/r04_snake_warmup_sweep.txt).  The roofline takes the tighter
            # of the two and reports both.
            return wall, {nm: min(first[nm], second[nm]) for nm in names}
        return wall, first

    sync()
    results = {}
    # buffer priming, untimed and not counted as warm-up steps: one launch of the timed shape, so that the [kc, N, ...] trajectory
    # (tens of GB) is allocated and touched before the timed region — on a fresh box the first hipMalloc of that size took ~0.5 s
    for nm in names:
        with on_stream(nm):
            envs[nm].rollout(min(kc[nm], K), action_seed=123, t0=0, trajectory=True, per_step=True)
        count(nm, "rollout", min(kc[nm], K))
    sync()
    run_rollout(max(W, 1), K)                                        # W untimed warm-up steps
    t_roll = 0 if wl.get("episode_start") else K + max(W, 1)
    # the events' second pass only where it repeats the first pass's work: snake (episodes of tens of steps: a steady state) and the
    # workloads that restart their episodes before every region
    same_work = len(names) == 1 and (names[0] == "snake" or bool(wl.get("episode_start")))
    results["rollout"] = timed(lambda rep, marks: run_rollout(K, t_roll if wl.get("episode_start") else t_roll + rep * K, marks), "rollout", same_work)
    withfin_raw = None
    if not dry and len(names) == 1 and not args.no_final_obs_leg:
        nm = names[0]
        envs[nm].collect_final_obs(rows_per_env=fin_rows[nm])
        R_keep, R = R, min(R, 3)
        kern_main, launched_main = ran[(nm, "rollout")], dict(launched)   # the with-rows leg may be another kernel instance (snake) and is not in the manifest
        res = timed(lambda rep, marks: run_rollout(K, t_roll if wl.get("episode_start") else t_roll + rep * K, None), "rollout_fin", same_work)
        R = R_keep
        _, _, cnt, cap = envs[nm]._fin                   # terminal rows of the LAST launch: per env-step (-> obliged bytes), and whether the side output held them all
        tot = int(cnt.sum().item())
        final_stats[nm] = {"rows_per_env": fin_rows[nm], "delivered_last_launch": tot, "dropped_last_launch": envs[nm].final_obs_dropped(),
                           "per_env_step": tot / float(n * max(last_launch.get(nm, 1), 1))}
        withfin_raw = (res, ran[(nm, "rollout")])
        ran[(nm, "rollout")] = kern_main
        launched.clear(); launched.update(launched_main)
        envs[nm].collect_final_obs(0)
    # Steady state.  A batch that was reset together is not in its steady state for hundreds of steps: every env starts with a full digit
    # ring, no ring needs topping up for the first ~100 steps, and the share of waves in which SOME lane refills grows until the envs are
    # out of phase (snake_1m, K = 20: 20.6 us per step 25-45 steps after the reset, 23.2 after 500, 25.7 after 2,000 and flat from there:
    # profiles/r04_snake_warmup_sweep.txt).  The contract's W warm-up steps decide where `value` is measured; this leg says what a consumer
    # that never resets the batch sees: --steady-steps more untimed steps, then the same K steps, 3 regions.
    steady_raw = None
    if not dry and len(names) == 1 and not wl.get("episode_start") and args.steady_steps > 0:
        t_ss = t_roll + (R + 8) * K
        run_rollout(args.steady_steps, t_ss)
        R_keep, R = R, min(R, 3)
        steady_raw = timed(lambda rep, marks: run_rollout(K, t_ss + args.steady_steps + rep * K, None), "rollout_steady", False)
        R = R_keep
    if not dry:
        for e in envs.values():                                      # the trajectory buffers are not needed by the API leg
            e._bufs.pop("traj", None)
        torch.cuda.empty_cache()
    actions = {nm: (None if dry else make_actions(nm, K + W, n, dev)) for nm in names}  # API path: K step() calls, HBM-resident actions
    run_steps(actions, 0, W)
    results["step"] = timed(lambda rep, marks: run_steps(actions, W, W + K, marks), "step", same_work)
    graph_leg = None
    if args.graph and len(names) == 1 and not dry:
        nm, Kg = names[0], min(K, 256)
        g = torch.cuda.CUDAGraph()
        restart_episodes()
        sync()
        with torch.cuda.graph(g):                                  # the facade's buffers exist (reuse_buffers=True, the step() leg ran): nothing allocates
            for t in range(W, W + Kg):
                envs[nm].step(act_at(nm, actions[nm], t))
        gw = []
        for rep in range(R):
            restart_episodes()
            barrier()
            t0 = time.perf_counter()
            g.replay()
            barrier()
            gw.append(time.perf_counter() - t0)
        gm = median(gw)
        graph_leg = {"path": "graph", "steps_per_replay": Kg, "_wall": gm, "unit": "env-steps/s", "spread": {"repeats": R, "min_ms_per_step": min(gw) * 1e3 / Kg, "max_ms_per_step": max(gw) * 1e3 / Kg},
                     "note": f"{Kg} step() calls captured once in a torch.cuda.CUDAGraph and replayed; wall clock between two barriers, max over ranks"}
        del g
    if "snake" in envs:
        assert envs["snake"].invalid_action_count() == 0
    measured = None
    if not dry and rank == 0:
        del actions
        torch.cuda.empty_cache()
        measured = measure_copy_bandwidth(dev)

    def reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearsal else dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    walls = {p: reduce_max(results[p][0]) for p in results}
    withfin_wall = reduce_max(withfin_raw[0][0]) if withfin_raw else None      # (every rank: a collective)
    steady_wall = reduce_max(steady_raw[0]) if steady_raw else None
    if graph_leg:
        gwall = reduce_max(graph_leg.pop("_wall"))
        graph_leg["value"] = n * world * graph_leg["steps_per_replay"] / gwall
        graph_leg["ms_per_step"] = gwall * 1e3 / graph_leg["steps_per_replay"]
    if rank == 0:
        head, other = args.path, ("step" if args.path == "rollout" else "rollout")
        total_envs = n * len(ENVS) if split else n * len(names) * world

        def block(path):
            wall, gpu_ms = walls[path], results[path][1]
            b = {"path": path, "value": total_envs * K / wall, "unit": "env-steps/s", "ms_per_step": wall * 1e3 / K, "spread": spread[path]}
            if segments.get(path) and len(names) == 1:
                sg = segments[path][names[0]]
                if sg:
                    b["phases"] = {"early_us_per_step": sg[0]["us_per_step"], "late_us_per_step": sg[-1]["us_per_step"], "segments": sg,
                                   "note": "HIP events between the launches (rollout leg) / every tenth of the region (step() leg) of the last repeat"}
            rl = {}
            for nm in names:
                kern = ran[(nm, path)]                   # set by the timed region's own launches (the last thing each leg ran)
                # fleet's rollout is K (step, dense) launch pairs, not one fused launch: price it per pair
                fused = path == "rollout" and not ENVS[nm].get("launches_per_step")
                launches = -(-K // kc[nm]) if fused else K
                rl[nm] = roofline(nm, path, kern, gpu_ms[nm], launches, K / launches, n, occupancy[path].get(nm, 0.0), measured)
                hl = host_latency_ms.get(path)
                rl[nm]["timing"] = "HIP events on the launch stream around the K steps of the median timed region"
                if hl is not None:
                    rl[nm]["first_pass_launch_us"] = first_ms[path][nm] * 1e3 / launches
                    rl[nm]["second_pass_launch_us"] = second_ms[path][nm] * 1e3 / launches
                    rl[nm]["timing"] = ("HIP events on the launch stream, the TIGHTER of two brackets of the same K steps: first_pass_launch_us = the median timed "
                                        "region's own events (they also hold the host's way to the first launch and any gap the host leaves between step() calls); "
                                        "second_pass_launch_us = the same K steps once more after the R regions, queued behind a 1-GiB device copy (no host "
                                        "latency inside; for snake_1m the batch is R x K steps further from its reset by then: see steady_state)")
            b["roofline"] = rl[names[0]] if len(names) == 1 else rl
            if len(names) > 1:
                # co-resident types: their kernels share the card (and its hardware queues), so one type's launch-stream events
                # do not isolate its kernel; the batch as a whole is priced instead: every type's algorithmic bytes over the wall time
                tot = sum(r["algorithmic_bytes_per_env_step"] for r in rl.values()) * n * K
                moved = [r["traffic_bytes_per_env_step"] for r in rl.values()]
                agg = {"bound": "hbm", "kernel": "all of: " + "; ".join(r["kernel"] for r in rl.values()), "achieved": tot / wall / 1e9,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": tot / wall / 1e9 / HBM_PEAK_GBS,
                       "traffic": (sum(moved) * n if all(m is not None for m in moved) else None),
                       "timing": "wall clock of the timed region (barrier + synchronize on both sides), all env types' streams together",
                       "algorithmic_note": "sum over the env types of each type's own figure (roofline_per_env_type)"}
                if measured:
                    agg["peak_measured"] = max(measured["copy_GBs"], measured["fill_GBs"])
                    agg["frac_of_measured"] = agg["achieved"] / agg["peak_measured"]
                b["roofline_aggregate"] = agg
            return b

        hb = block(head)
        out = {
            "metric": "env steps/sec (whole node) at 1M parallel envs; achieved HBM GB/s vs peak",
            "value": hb["value"], "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": hb["ms_per_step"], "spread": hb["spread"], "higher_is_better": True, "scaling": "strong" if split else "weak", "vs_baseline": None,
            "dtype": ENVS[names[0]]["dtype"] if len(names) == 1 else "mixed", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['desc']}", "envs_per_gpu": n * len(names), "env_types": names,
                       "path": (("rollout: K (step, dense-reset) launch pairs queued by one C-ABI call, every step's obs / reward / flag written to "
                                 "its own slot of a [K, N, ...] trajectory in HBM, device-side action hash"
                                 if all(ENVS[nm].get("launches_per_step") for nm in names) else
                                 f"fused rollout: {kc[names[0]]} steps per launch, every step's obs / reward / flag written to its own slot of a "
                                 f"[{kc[names[0]]}, N, ...] trajectory in HBM, device-side action hash")
                                if head == "rollout" else "K C-ABI step() calls through the VectorEnv facade, HBM-resident actions"),
                       "autoreset": "SameStep", "parallelism": f"env-sharded x{world}, no data-path collective",
                       "episode_phase": (f"steps 1..{K} of an episode (envs reset right before every timed region"
                                         + (f"; a whole episode is {wl['episode']} steps: --episode" if wl.get("episode") and K != wl["episode"] else "") + ")"
                                         if wl.get("episode_start")
                                         else f"the timed regions start {K + max(W, 1) + min(kc[names[0]], K)} steps after the batch was reset together (W warm-up steps + one priming launch); "
                                              "the batch's steady state is the steady_state block")},
            "roofline": hb["roofline"] if len(names) == 1 else hb["roofline_aggregate"],
        }
        if dry:
            out["data"] = "DRY RUN (CGE_BENCH_DRYRUN=1): no device work, numbers are meaningless"
            out["roofline"] = None
        if hb.get("phases"):
            out["phases"] = hb["phases"]
        if withfin_raw:
            nm = names[0]
            launches = K if ENVS[nm].get("launches_per_step") else -(-K // kc[nm])
            rr = roofline(nm, "rollout", withfin_raw[1], withfin_raw[0][1][nm], launches, K / launches, n, occupancy["rollout"].get(nm, 0.0), measured,
                          final_stats[nm]["per_env_step"])
            out["rollout_with_final_obs"] = {
                "us_per_step": rr["avg_launch_us"] / (K / launches), "frac": rr["frac"], "algorithmic_bytes_per_env_step": rr["algorithmic_bytes_per_env_step"],
                "ms_per_step_wall": withfin_wall * 1e3 / K, **final_stats[nm],
                "note": "the same K steps with the terminal-observation side output registered (cge_<env>_rollout_final_obs): trajectory + side output = "
                        "everything the K step() calls return; its bytes (obs + 8 per episode end) are part of this block's obliged bytes"}
        if steady_raw:
            nm = names[0]
            launches = -(-K // kc[nm])
            rr = roofline(nm, "rollout", ran[(nm, "rollout")], steady_raw[1][nm], launches, K / launches, n, occupancy["rollout"].get(nm, 0.0), measured)
            out["steady_state"] = {
                "value": total_envs * K / steady_wall, "unit": "env-steps/s", "ms_per_step": steady_wall * 1e3 / K, "spread": spread["rollout_steady"],
                "us_per_step": rr["avg_launch_us"] / (K / launches), "frac": rr["frac"], "kernel": rr["kernel"],
                "after_steps": args.steady_steps,
                "note": f"the same fused rollout of K = {K} steps, timed again after {args.steady_steps} more untimed steps (3 regions, median; us_per_step / frac from "
                        "HIP events on the launch stream): the batch's steady state - envs out of phase, digit rings topped up at their steady rate. "
                        "`value` above is measured W warm-up steps after the reset, as the contract says, where every ring is still full"}
        if graph_leg:
            out["graph_step"] = graph_leg
        if len(names) > 1:
            out["roofline_per_env_type"] = hb["roofline"]
        out["api_step" if other == "step" else "fused_rollout"] = block(other)
        if not args.no_cpu_baseline and world == 1 and not dry:            # the CPU port is timed at N=1 only
            out["cpu_baseline"] = cpu_baseline(names[0] if len(names) == 1 else "snake")
        print(json.dumps(out), flush=True)
        if args.manifest:
            with open(args.manifest, "w") as f:
                json.dump(launched, f, indent=1)
    for e in envs.values():
        e.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
