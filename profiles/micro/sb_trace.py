#!/usr/bin/env python3
"""Per-stage s_memtime stamps of the temporally blocked SOR kernel (library built with -DHNS_SB_TRACE, see profiles/micro/exp/build.sh):
thread 0 of the first 64 workgroups. argv: config lb k"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hnanosolver_amd as H
from hnanosolver_amd import api, device as D, fields, _lib

cfg, lb, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
origins, R = fields.config_leaves(cfg)
grid = api.create_grid_from_leaves(origins, 1.0 / R)
N = len(origins) * 512
div = torch.randn(N, device="cuda"); p_a = torch.zeros(N, device="cuda"); p_b = torch.zeros(N, device="cuda")
H.set_option("rbgs", "block"); H.set_option("sor_block_lb", lb); H.set_option("sor_block_k", k)
for _ in range(3):
    D.rbgs_iterate(grid, div, p_a, p_b, 1.0 / R, 1.9, 4 * k)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (64 * 16))()
lib = _lib.load_library()
lib.hns_sb_trace_read.argtypes = [C.c_void_p]
assert lib.hns_sb_trace_read(buf) == 0
t = np.array(buf[:], dtype=np.int64).reshape(64, 16)
n = 7 + 2 * k
names = ["start", "rec loaded", "loads issued", "data split", "staged+barrier"] + [f"sweep {s}" for s in range(1, 2 * k + 1)] + ["(pre-store)", "stored"]
d = np.diff(t[:, :n], axis=1)
print(f"{cfg} lb={lb} k={k}: median cycles per stage over 64 workgroups (100 MHz ticks if s_memtime is the constant clock)")
for j in range(n - 1):
    print(f"  {names[j + 1]:16s} {np.median(d[:, j]):9.0f}  (min {d[:, j].min():7d} max {d[:, j].max():7d})")
print(f"  total            {np.median(t[:, n - 1] - t[:, 0]):9.0f}")
