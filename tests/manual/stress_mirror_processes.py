"""Stress of the mirroring pressure loop between PROCESSES sharing one GPU at cache-resident sizes (everything fits an XCD's
L2, where a stale line would be hit if a ghost row were ever read before its refresh): N repetitions of 2- and 3-process
runs, 6 substeps x 50 iterations, compared bit for bit with the single grid. Not part of the pytest suite (minutes)."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_dist_gpu import _run_processes, single_grid  # noqa: E402
from dist_process_worker import case_leaves  # noqa: E402
from hnanosolver_amd import dist as HD  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
bad = 0
for case, world in (("dense32", 2), ("dense64", 2), ("plume", 3), ("dense64", 4)):
    origins, R = case_leaves(case)
    names, iters, substeps = ["density", "temperature"], 50, 6
    _, want = single_grid(origins, R, names, iters, substeps)
    ids = [HD.owned_ids_of(origins, world, r) for r in range(world)]  # (slabs along the cheapest axis since round 5: the plume's ranks are not contiguous leaf ranges)
    for rep in range(reps):
        with tempfile.TemporaryDirectory() as tmp:
            got = _run_processes(world, case, 1, iters, substeps, tmp)
        ok = all(np.array_equal(g["vel"], HD.take_leaves(want["vel"], ids[r])) and all(np.array_equal(g[n], HD.take_leaves(want[n], ids[r])) for n in names)
                 for r, g in enumerate(got))
        bad += 0 if ok else 1
        print(case, world, "rep", rep, "ok" if ok else "DIFFERENT", flush=True)
print("different runs:", bad)
sys.exit(1 if bad else 0)
