#!/usr/bin/env python3
"""Stress: several host threads cooking on ONE grid handle at once; every result must equal the serial answer."""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from hnanosolver_amd import api, fields
from test_operators_gpu import build_data, snapshot

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
origins, R = fields.dense_leaves(32), 32
vs = 1.0 / R
p = api.CombustionParams()
h = api.IndexGridHandle()
api.CreateIndexGrid(build_data(origins, R), h, vs)
amps = [60.0, 96.0, 140.0, 200.0]
serial = []
for a in amps:
    d = build_data(origins, R, amplitude=a)
    api.Compute_Sim(d, h, 10, 1.0 / 24.0, vs, p, False)
    serial.append(snapshot(d))
bad = 0
for r in range(rounds):
    results, errors = [None] * len(amps), []
    def work(i):
        try:
            for _ in range(3):
                d = build_data(origins, R, amplitude=amps[i])
                api.Compute_Sim(d, h, 10, 1.0 / 24.0, vs, p, False)
                results[i] = snapshot(d)
        except Exception as e:
            errors.append(repr(e))
    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(amps))]
    [t.start() for t in ts]; [t.join() for t in ts]
    for e in errors: print("round", r, "error", e[:300]); bad += 1
    for i in range(len(amps)):
        if results[i] is None: continue
        for n in serial[i]:
            if not np.array_equal(serial[i][n], results[i][n]):
                diff = np.nonzero(np.asarray(serial[i][n]).reshape(len(results[i][n]), -1) != np.asarray(results[i][n]).reshape(len(results[i][n]), -1))[0]
                print("round", r, "thread", i, "field", n, "mismatching voxels", len(np.unique(diff)), "first", diff[:3], "leaf", diff[:3] // 512)
                bad += 1
print("rounds", rounds, "bad", bad, "env", {k: v for k, v in os.environ.items() if k.startswith("HNS_")})
