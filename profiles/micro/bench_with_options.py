#!/usr/bin/env python3
"""bench.py with library options set first: python profiles/micro/bench_with_options.py opt=val[,opt=val...] <bench.py arguments>; prints substeps/s and the kernel times of the line"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hnanosolver_amd as H
import bench

opts = [o for o in sys.argv[1].split(",") if o and o != "-"]
for o in opts:
    k, v = o.split("=")
    H.set_option(k, v)
sys.argv = ["bench.py"] + sys.argv[2:]
buf = io.StringIO()
with redirect_stdout(buf):
    bench.main()
j = json.loads(buf.getvalue().strip().splitlines()[-1])
r = j["roofline"]
print(",".join(opts) or "default", " ".join(sys.argv[1:]), "| substeps/s", round(j["value"], 1), "|",
      {k.split("<")[0]: round(1e3 * v["ms_per_launch"], 2) for k, v in r["kernels"].items()}, flush=True)
