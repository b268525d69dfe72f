"""Least-squares fit of  time per sparse-attention launch = jobs x c + executed tiles x tau  over the density sweep of r06_sparse_job_cost.sh."""
import json
import sys

import numpy as np

rows = {}
for line in open(sys.argv[1]):
    try:
        d = json.loads(line)
    except Exception:
        continue
    r = d["roofline"]
    o = dict(x.split("=") for x in d["config"].get("engine_options", []))
    g = int(o.get("nabla_group_rows", 0))
    N, H = d["config"]["tokens"], 28
    nb = N // 64
    dens, ue = r["kept_block_density"], r["union_efficiency"]
    tiles = dens * nb * nb * H / ue / g          # 64-key tiles walked by all jobs of one launch (a job = g query blocks sharing a list)
    jobs = H * nb / g
    rows.setdefault(g, []).append((dens, ue, jobs, tiles, r["avg_launch_ms"] * 1e3))
for g, rr in sorted(rows.items()):
    rr.sort()
    A = np.array([[x[2], x[3]] for x in rr])
    t = np.array([x[4] for x in rr])
    (c_tot, tau), res, *_ = np.linalg.lstsq(A, t, rcond=None)
    # c_tot, tau: us of the whole device per job / per tile (all jobs share the 256 CUs)
    print(f"--- {64 * g}-query workgroups: {int(rr[0][2])} jobs per launch")
    print("density  union_eff  tiles/job  launch us   fit us   fixed share")
    for dens, ue, jobs, tiles, us in rr:
        fit = jobs * c_tot + tiles * tau
        print(f"{dens:7.4f}  {ue:8.3f}  {tiles / jobs:9.1f}  {us:9.1f}  {fit:7.1f}   {jobs * c_tot / fit:6.3f}")
    print(f"fit: {c_tot * 1e3:.2f} ns of device time per job + {tau * 1e3:.3f} ns per tile  (one job's fixed cost = {c_tot / tau:.1f} tiles' worth)")
