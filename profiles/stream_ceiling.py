#!/usr/bin/env python3
"""Practical ceiling for the SOR kernel's traffic pattern on this GPU: a plain elementwise q = p + d over 256^3 floats
(read 8 B/voxel, write 4 B/voxel = the 12 algorithmic B/voxel of one RB-SOR iteration), timed like the kernel itself.
Not product code: a calibration point for DESIGN.md section 7."""
import json
import sys

import torch

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256  # 256: the working set (201 MB) fits the Infinity Cache; 512: it does not
n = R ** 3
p = torch.rand(n, device="cuda")
d = torch.rand(n, device="cuda")
q = torch.empty(n, device="cuda")
for _ in range(5):
    torch.add(p, d, out=q)
torch.cuda.synchronize()
res = {"R": R}
for reps in (50, 200):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        if i & 1:
            torch.add(q, d, out=p)
        else:
            torch.add(p, d, out=q)
    b.record()
    torch.cuda.synchronize()
    us = 1e3 * a.elapsed_time(b) / reps
    res[f"us_per_pass_{reps}"] = us
    res[f"GBps_{reps}"] = 12 * n / (us * 1e-6) / 1e9
print(json.dumps(res))
