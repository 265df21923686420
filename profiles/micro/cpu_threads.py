#!/usr/bin/env python3
"""How many OpenMP threads should the CPU baseline use on this host? Times one RB-SOR sweep of the oracle per thread count."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hnanosolver_amd import fields
from oracle_lib import OracleGrid, oracle
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f): print(f, open(f).read().strip())
os.system("nproc; lscpu | grep -E 'Model name|Socket|Core|Thread' | head -5")
L = oracle()
cfg = sys.argv[1] if len(sys.argv) > 1 else "128"
origins, R = fields.config_leaves(cfg)
G = OracleGrid(origins)
div = np.random.default_rng(0).standard_normal(G.N).astype(np.float32)
for th in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    L.orc_set_threads(th)
    p = np.zeros(G.N, dtype=np.float32)
    G.rbgs(div, p, 1.0 / R, 0, 1.9)
    t0 = time.perf_counter()
    for _ in range(2):
        G.rbgs(div, p, 1.0 / R, 0, 1.9); G.rbgs(div, p, 1.0 / R, 1, 1.9)
    print(cfg, "threads", th, "sweep_ms", round(1e3 * (time.perf_counter() - t0) / 2, 2), flush=True)
