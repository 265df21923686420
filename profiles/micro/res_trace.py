#!/usr/bin/env python3
"""Per-phase wall-clock stamps (100 MHz) of the resident SOR kernel (library built with -DHNS_RES_TRACE, profiles/micro/exp/sorresident_trace.patch):
thread 0 of the first 64 workgroups, rounds 2 and 3 of a 24-round solve. argv: config"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hnanosolver_amd as H
from hnanosolver_amd import api, device as D, fields, _lib

cfg = sys.argv[1]
origins, R = fields.config_leaves(cfg)
grid = api.create_grid_from_leaves(origins, 1.0 / R)
N = len(origins) * 512
div = torch.randn(N, device="cuda"); p_a = torch.zeros(N, device="cuda"); p_b = torch.zeros(N, device="cuda")
for _ in range(3):
    D.rbgs_iterate(grid, div, p_a, p_b, 1.0 / R, 1.9, 48)
torch.cuda.synchronize()
print(D.rbgs_plan(grid, 48)[0])
buf = (C.c_ulonglong * (64 * 32))()
lib = _lib.load_library()
lib.hns_res_trace_read.argtypes = [C.c_void_p]
assert lib.hns_res_trace_read(buf) == 0
t = np.array(buf[:], dtype=np.int64).reshape(64, 2, 16)[:, :, :9]
names = ["own stores at memory (vmcnt 0)", "barrier", "flag up + neighbours' flags seen", "barrier", "halo loads back", "staged + barrier", "four sweeps + barrier", "stores issued"]
d = np.diff(t, axis=2) * 10.0  # ns
for rnd in range(2):
    print(f"{cfg} round {rnd + 2}: median ns per phase over 64 workgroups")
    for j, nm in enumerate(names):
        print(f"  {nm:36s} {np.median(d[:, rnd, j]):8.0f}  (min {d[:, rnd, j].min():7.0f} max {d[:, rnd, j].max():7.0f})")
    print(f"  round (top to stores issued)         {np.median(t[:, rnd, 8] - t[:, rnd, 0]) * 10.0:8.0f}")
print(f"  top of round 2 -> top of round 3     {np.median(t[:, 1, 0] - t[:, 0, 0]) * 10.0:8.0f}")
