#!/usr/bin/env python3
"""One rank of BASELINE config 5 alone on the GPU, EXCHANGED pressure loop (option dist_mirror = 0: what RCCL ranks run), loopback transport: for
rocprofv3 --kernel-trace [--memory-copy-trace] of its substeps. argv: sweeps_per_exchange [--rccl] [--rank=N] [option=value ...]"""
import os
import runpy
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, root)
import hnanosolver_amd as H  # noqa: E402

H.set_option("dist_mirror", "0")
rest = []
for a in sys.argv[2:]:
    if "=" in a and not a.startswith("--"):
        H.set_option(*a.split("=", 1))
    else:
        rest.append(a)
if not any(a.startswith("--rank=") for a in rest):
    rest.append("--rank=4")
sys.argv = ["dist_overhead.py", "plume1024", "8", sys.argv[1], "--partition", "--lone-only", "--no-plain"] + rest
runpy.run_path(os.path.join(root, "profiles", "micro", "dist_overhead.py"), run_name="__main__")
