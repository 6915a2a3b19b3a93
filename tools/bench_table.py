"""profiles/r0N_bench_all.jsonl -> the markdown table of DESIGN.md section 6"""
import json
import sys
rows = []
fin = []
for l in open(sys.argv[1] if len(sys.argv) > 1 else "profiles/r04_bench_all.jsonl"):
    d = json.loads(l)
    if d.get("failed"):
        rows.append(f"| {d['workload']} | — | FAILED | | | | | | |")
        continue
    w = d["config"]["workload"].split(":")[0]
    n = d["config"]["envs_per_gpu"]
    k = d["steps"]
    if w.startswith("hetero"):
        rows.append(f"| {w}: 8 types × 131,072 | both | — | {k} | {d['ms_per_step'] * 1e3:.0f} / {d['api_step']['ms_per_step'] * 1e3:.0f} per step of all eight (rollout / `step()`) | {d['value']:.2e} / {d['api_step']['value']:.2e} | — | — | — | — |")
        continue
    wf = d.get("rollout_with_final_obs")
    if wf:
        r0 = d["roofline"]
        us0 = r0["avg_launch_us"] / (r0["env_steps_per_launch"] / n)
        fin.append(f"| {w} | {us0:.1f} | {wf['us_per_step']:.1f} | {wf['per_env_step'] * 100:.2f} % | {wf['delivered_last_launch']:,} / {wf['dropped_last_launch']} | {wf['algorithmic_bytes_per_env_step']:.0f} | {wf['frac']:.2f} |")
    for path, blk in (("fused rollout", d), ("`step()`", d["api_step"])):
        r = blk["roofline"]
        steps_per_launch = r["env_steps_per_launch"] / n
        us = r["avg_launch_us"] / steps_per_launch
        tr = r.get("traffic_bytes_per_env_step")
        fr = f"{r['frac']:.2f} / {r.get('frac_of_copy', 0):.2f} / {r.get('frac_of_fill', 0):.2f}"
        rows.append(f"| {w} ({n:,}) | {path} | `{r['kernel'].replace('cge::', '')}` | {k} | {us:.1f} | {blk['value']:.2e} | {r['algorithmic_bytes_per_env_step']:.0f} | {r['achieved']:.0f} | {fr} | {'%.0f (%.2f×)' % (tr, r['traffic_over_algorithmic']) if tr else '—'} |")
print("| workload (envs per GPU) | path | kernel | K (steps timed) | µs per step (HIP events) | env-steps/s | obliged B / env-step | achieved GB/s | frac of 8 TB/s / of the box's copy / of its fill | PMC B / env-step (÷ obliged) |")
print("|---|---|---|---|---|---|---|---|---|---|")
print("\n".join(rows))
if fin:
    print()
    print("| workload | rollout µs per step | with the terminal-row side output | episode ends per env-step | rows delivered / dropped (last launch) | obliged B / env-step with the rows | frac of 8 TB/s |")
    print("|---|---|---|---|---|---|---|")
    print("\n".join(fin))
