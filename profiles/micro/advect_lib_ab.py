#!/usr/bin/env python3
"""advect kernel times of two builds of the library (HNS_LIBRARY paths), each in its own subprocess, alternating.
argv: lib_a lib_b [config=256]"""
import os, subprocess, sys
a, b = sys.argv[1], sys.argv[2]
cfg = sys.argv[3] if len(sys.argv) > 3 else "256"
here = os.path.dirname(os.path.abspath(__file__))
for rep in range(2):
    for lib in (a, b):
        env = dict(os.environ, HNS_LIBRARY=lib)
        out = subprocess.run([sys.executable, os.path.join(here, "advect_ab.py"), "rev", "1", "1", cfg], env=env, capture_output=True, text=True).stdout.strip().splitlines()
        print(os.path.basename(lib), out[-1] if out else "?")
