#!/usr/bin/env python3
"""Floor for the gradient kernel at 256^3: read 201 MB (+ 67 MB), write 201 MB, with the inputs NOT in the Infinity Cache
(a 400 MB scratch is streamed in between, as the pressure solve does to the velocity)."""
import torch
n = 256 ** 3
a = torch.rand(n, 3, device="cuda"); b = torch.empty_like(a); p = torch.rand(n, device="cuda")
junk = torch.rand(100 * 1024 * 1024, device="cuda"); junk2 = torch.empty_like(junk)
def timed(fn, reps=10):
    ts = []
    for _ in range(reps):
        junk2.copy_(junk)  # evict
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return min(ts), sum(ts) / len(ts)
print("copy 201 MB -> 201 MB, cold: min/avg us", timed(lambda: b.copy_(a)))
print("b = a - p[:,None] (201 + 67 MB in, 201 MB out), cold:", timed(lambda: torch.sub(a, p[:, None], out=b)))
