import ctypes as C, os, sys, torch
torch.zeros(1, device="cuda")
lib = C.CDLL(os.environ["HNS_LIBRARY"])
for w in range(3): print("occupancy blocks/CU kernel", w, lib.hns_sb_occupancy(w))
