#!/usr/bin/env python3
"""CPU baseline of SURVEY.md 8d: the oracle's core substep (advect_vector + divergence + 50 RB-SOR iterations + gradient +
advect_scalars, S=1) on the CPUs this process may use (affinity and cgroup quota: 16 on the GPU boxes) and on one, at 64^3 and 128^3 (256^3 on
that CPU budget is what bench.py reports).
Pressure iterations are timed on a few sweeps and scaled to 50 (every sweep does the same work)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hnanosolver_amd import fields
from oracle_lib import OracleGrid, cpu_budget, oracle

L = oracle()
out = []
for cfg in ("64", "128"):
    origins, R = fields.config_leaves(cfg)
    f = fields.synthetic_fields(origins, R)
    G = OracleGrid(origins)
    vs, dt, inv_dx = 1.0 / R, 1.0 / 24.0, float(R)
    omega = float(L.orc_omega_compute(vs))
    for threads in (cpu_budget(), 1):
        L.orc_set_threads(threads)
        used = int(L.orc_get_threads())
        its = 10 if threads != 1 else (4 if cfg == "64" else 2)
        t0 = time.perf_counter()
        adv = G.advect_vector(f["vel"], dt, inv_dx)
        div = G.divergence(adv, inv_dx)
        t1 = time.perf_counter()
        p = np.zeros(G.N, dtype=np.float32)
        for _ in range(its):
            G.rbgs(div, p, vs, 0, omega); G.rbgs(div, p, vs, 1, omega)
        t2 = time.perf_counter()
        u = G.subtract_pressure_gradient(adv, p, inv_dx)
        G.advect_scalars(u, [f["density"]], dt, inv_dx)
        t3 = time.perf_counter()
        sub = (t1 - t0) + (t3 - t2) + (t2 - t1) / its * 50
        out.append({"config": cfg, "threads": used, "s_per_substep": round(sub, 4), "substeps_per_s": round(1.0 / sub, 3), "sweep_ms": round(1e3 * (t2 - t1) / its, 3)})
        print(out[-1], flush=True)
    L.orc_set_threads(cpu_budget())
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cpu_baseline.json"), "w"), indent=1)
