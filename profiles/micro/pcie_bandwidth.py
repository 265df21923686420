import torch, time
n = 201326592 // 4 * 4  # 201 MB worth of floats*... use 268 MB
n = 64 * 1024 * 1024
h = torch.empty(n, dtype=torch.float32).fill_(1.0)
hp = torch.empty(n, dtype=torch.float32).pin_memory()
d = torch.empty(n, dtype=torch.float32, device="cuda")
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
gb = n * 4 / 1e9
print("pageable H2D GB/s", gb / t(lambda: d.copy_(h)))
print("pageable D2H GB/s", gb / t(lambda: h.copy_(d)))
print("pinned   H2D GB/s", gb / t(lambda: d.copy_(hp, non_blocking=True)))
print("pinned   D2H GB/s", gb / t(lambda: hp.copy_(d, non_blocking=True)))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
d2 = torch.empty_like(d); hp2 = torch.empty(n, dtype=torch.float32).pin_memory()
def both():
    with torch.cuda.stream(s1): d.copy_(hp, non_blocking=True)
    with torch.cuda.stream(s2): hp2.copy_(d2, non_blocking=True)
print("pinned duplex GB/s (sum)", 2 * gb / t(both))
t0 = time.perf_counter(); r = torch.cuda.cudart().cudaHostRegister(h.data_ptr(), n * 4, 0); t1 = time.perf_counter()
print("hostRegister 268MB ms", (t1 - t0) * 1e3, r)
print("registered H2D GB/s", gb / t(lambda: d.copy_(h, non_blocking=True)))
t0 = time.perf_counter(); torch.cuda.cudart().cudaHostUnregister(h.data_ptr()); print("unregister ms", (time.perf_counter() - t0) * 1e3)
import os; print("cpus", os.cpu_count())
