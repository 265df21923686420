#!/usr/bin/env python3
"""advect kernel times of several builds of the library (HNS_LIBRARY paths; "-" = the product build), each in its own
subprocess, two rounds. argv: config lib [lib ...]"""
import os, subprocess, sys
cfg, libs = sys.argv[1], sys.argv[2:]
here = os.path.dirname(os.path.abspath(__file__))
for rep in range(2):
    for lib in libs:
        env = dict(os.environ)
        if lib != "-":
            env["HNS_LIBRARY"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, os.path.join(here, "advect_ab.py"), "rev", "1", "1", cfg], env=env, capture_output=True, text=True)
        print(os.path.basename(lib), (out.stdout.strip().splitlines() or ["?" + out.stderr[-300:]])[-1], flush=True)
