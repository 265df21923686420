#!/usr/bin/env python3
"""Writes tests/golden/nanovdb_export_v1.json: size and sha256 of hns_grid_export_nanovdb's buffer for the leaf sets of
tests/test_nanovdb_export.py. Run only after that file's reference-backed tests pass (they are what certifies the
bytes; this fixture lets a machine without the reference checkout notice a change)."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

from test_nanovdb_export import CASES, export  # noqa: E402

out = {}
for name, (mk, vs) in CASES.items():
    g, buf = export(mk(), vs)
    out[name] = {"bytes": int(len(buf)), "sha256": hashlib.sha256(buf.tobytes()).hexdigest()}
    g.reset()
json.dump(out, open(os.path.join(HERE, "nanovdb_export_v1.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
