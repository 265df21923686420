#!/usr/bin/env python3
"""Generates tests/golden/kernel_goldens_v1.npz: the frozen outputs of the golden set G1-G4 for every
kernel alone, the projection driver with 1/2/50 iterations and the Compute driver (combustion zero / non-zero, collision).

Engine (round 4): THE REFERENCE'S OWN KERNELS -- src/Cuda/Kernel.cu compiled where it lies into
oracle/_ref/libhns_refk.so (oracle/Makefile, oracle/ref_kernels.cpp: g++, the image's CUDA runtime headers, strict IEEE:
-ffp-contract=off), launched in the reference's order (tests/oracle_lib.py: RefKernelGrid). The oracle must agree bit for
bit before anything is written. See tests/golden_cases.py for what these vectors are and are not. Small grids (G1, G2) keep whole arrays; G3 and G4 keep every 37th element plus SHA-256 / L2 / L-inf of the full
array (tests/golden/kernel_goldens_v1.json)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_cases as gc  # noqa: E402
from hnanosolver_amd import api  # noqa: E402
from oracle_lib import OracleGrid, RefKernelGrid  # noqa: E402


def main():
    arrays, meta = {}, {}
    for name in ("G1", "G2", "G3", "G4"):
        origins, R = gc.grid_leaves(name)
        out = gc.run_all(RefKernelGrid(origins), name, api.CombustionParams)
        orc = gc.run_all(OracleGrid(origins), name, api.CombustionParams)
        assert set(orc) == set(out) and all(np.array_equal(orc[k], out[k]) for k in out), f"{name}: oracle and reference disagree"
        meta[name] = {"leaves": int(len(origins)), "R": R, "inputs": {k: gc.digest(v) for k, v in gc.inputs(origins, R).items()}, "outputs": {}}
        for k, v in out.items():
            meta[name]["outputs"][k] = gc.digest(v)
            flat = np.ascontiguousarray(v, dtype=np.float32).reshape(len(origins) * 512, -1)
            arrays[f"{name}/{k}"] = flat if name in ("G1", "G2") else flat[:: gc.STRIDE]
    meta["_engine"] = {"outputs_of": "reference src/Cuda/Kernel.cu, compiled from /root/reference by oracle/Makefile (oracle/_ref/libhns_refk.so)",
                       "launch_sequences": "tests/oracle_lib.py: RefKernelGrid (HNanoSolver.cu:150-356, PressureProjection.cu:43-66)",
                       "floating_point": "strict IEEE (-ffp-contract=off), fused only where the reference calls __fmaf_rn / fmaf",
                       "cross_checked": "oracle/hns_oracle.c bit-identical on every array"}
    np.savez_compressed(os.path.join(HERE, "kernel_goldens_v1.npz"), **arrays)
    json.dump(meta, open(os.path.join(HERE, "kernel_goldens_v1.json"), "w"), indent=1, sort_keys=True)
    print(len(arrays), "arrays;", os.path.getsize(os.path.join(HERE, "kernel_goldens_v1.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
