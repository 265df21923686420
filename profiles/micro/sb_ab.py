import os, sys, torch
sys.path.insert(0, "/root/repo")
import hnanosolver_amd as H
from hnanosolver_amd import api, device as D, fields
tag = os.path.basename(os.environ.get("HNS_LIBRARY", "libhns.so"))
opts = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
for k, v in opts.items():
    H.set_option(k, v)
for c in [a for a in sys.argv[1:] if "=" not in a] or ["128", "256", "plume1024"]:
    origins, R = (fields.dense_leaves(int(c[1:])), int(c[1:])) if c.startswith("d") else fields.config_leaves(c)
    grid = api.create_grid_from_leaves(origins, 1.0 / R)
    N = len(origins) * 512
    div = torch.randn(N, device="cuda"); p_a = torch.zeros(N, device="cuda"); p_b = torch.zeros(N, device="cuda")
    ms = sorted(D.time_rbgs(grid, div, p_a, p_b, 1.0 / R, 1.97, 48, 3) for _ in range(7))
    print(tag, opts, c, "min %.2f median %.2f us/iter" % (1e3 * ms[0], 1e3 * ms[3]), flush=True)
    del grid, div, p_a, p_b
    torch.cuda.empty_cache()
