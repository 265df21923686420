#!/usr/bin/env python3
"""Turns a checkout of the reference (ZephirFXEC/HNanoSolver) into the libhns build: the handful of token substitutions a maintainer
makes in src/SOP/** and src/Utils/Memory.hpp so that the five SOPs call libhns.so through integration/hns_shim.cpp instead of the CUDA
`Kernels` library (the list in integration/hns_shim.cpp's header, executable).

    python integration/port_reference.py /path/to/HNanoSolver [--check]      # --check: report only, write nothing

Every edit is anchored on (file, line, the token expected there): a checkout that has moved on fails loudly instead of being edited
blindly. The script carries identifiers only -- no reference source text -- and nothing in this repo's product path depends on it.
What is NOT automated: src/SOP/CMakeLists.txt's target list keeps its shape, but HIP and libhns have to be found on the build machine
(the two cmake lines below are a starting point), and the operators need the Houdini HDK to compile at all.
"""
from __future__ import annotations

import argparse
import os
import sys

HANDLE = "nanovdb::GridHandle<nanovdb::cuda::DeviceBuffer>"
# (file, line, expected token, replacement)
EDITS = [
    # hnanosolver: the all-in-one substep (SOP_HNanoSolver.hpp:82-85, SOP_HNanoSolver.cpp:226-256)
    ("src/SOP/HNanoSolver/SOP_HNanoSolver.hpp", 18, '"nanovdb/cuda/DeviceBuffer.h"', '"hns_shim.hpp"'),
    ("src/SOP/HNanoSolver/SOP_HNanoSolver.hpp", 82, HANDLE, "hns_shim::GridHandle"),
    ("src/SOP/HNanoSolver/SOP_HNanoSolver.hpp", 84, HANDLE, "hns_shim::GridHandle"),
    ("src/SOP/HNanoSolver/SOP_HNanoSolver.hpp", 85, "cudaStream_t", "hipStream_t"),
    ("src/SOP/HNanoSolver/SOP_HNanoSolver.cpp", 6, "<nanovdb/cuda/DeviceBuffer.h>", "<hip/hip_runtime_api.h>"),
    ("src/SOP/HNanoSolver/SOP_HNanoSolver.cpp", 227, HANDLE, "hns_shim::GridHandle"),
    ("src/SOP/HNanoSolver/SOP_HNanoSolver.cpp", 238, "cudaStream_t", "hipStream_t"),
    ("src/SOP/HNanoSolver/SOP_HNanoSolver.cpp", 239, "cudaStreamCreate", "hipStreamCreate"),
    ("src/SOP/HNanoSolver/SOP_HNanoSolver.cpp", 255, "cudaStreamSynchronize", "hipStreamSynchronize"),
    ("src/SOP/HNanoSolver/SOP_HNanoSolver.cpp", 256, "cudaStreamDestroy", "hipStreamDestroy"),
    # hnanoadvect (SOP_VDBAdvect.hpp:66, SOP_VDBAdvect.cpp:99-155)
    ("src/SOP/Advection/SOP_VDBAdvect.hpp", 10, '"nanovdb/cuda/DeviceBuffer.h"', '"hns_shim.hpp"'),
    ("src/SOP/Advection/SOP_VDBAdvect.hpp", 47, HANDLE, "hns_shim::GridHandle"),
    ("src/SOP/Advection/SOP_VDBAdvect.hpp", 66, "cudaStream_t", "hipStream_t"),
    ("src/SOP/Advection/SOP_VDBAdvect.cpp", 99, "cudaStream_t", "hipStream_t"),
    ("src/SOP/Advection/SOP_VDBAdvect.cpp", 100, "cudaStreamCreate", "hipStreamCreate"),
    ("src/SOP/Advection/SOP_VDBAdvect.cpp", 155, "cudaStreamDestroy", "hipStreamDestroy"),
    # hnanoadvectvelocity (SOP_VDBAdvectVelocity.hpp:60, .cpp:79-110)
    ("src/SOP/VelocityAdvection/SOP_VDBAdvectVelocity.hpp", 9, '"nanovdb/cuda/DeviceBuffer.h"', '"hns_shim.hpp"'),
    ("src/SOP/VelocityAdvection/SOP_VDBAdvectVelocity.hpp", 60, "cudaStream_t", "hipStream_t"),
    ("src/SOP/VelocityAdvection/SOP_VDBAdvectVelocity.cpp", 79, "cudaStream_t", "hipStream_t"),
    ("src/SOP/VelocityAdvection/SOP_VDBAdvectVelocity.cpp", 80, "cudaStreamCreate", "hipStreamCreate"),
    ("src/SOP/VelocityAdvection/SOP_VDBAdvectVelocity.cpp", 110, "cudaStreamDestroy", "hipStreamDestroy"),
    # hnanoprojectnondivergent (SOP_VDBProjectNonDivergent.hpp:69-70, .cpp:90-143; the unused BufferT alias goes to the host buffer)
    ("src/SOP/ProjectNonDivergent/SOP_VDBProjectNonDivergent.hpp", 13, '"nanovdb/cuda/DeviceBuffer.h"', '"hns_shim.hpp"'),
    ("src/SOP/ProjectNonDivergent/SOP_VDBProjectNonDivergent.hpp", 69, "cudaStream_t", "hipStream_t"),
    ("src/SOP/ProjectNonDivergent/SOP_VDBProjectNonDivergent.hpp", 70, "cudaStream_t", "hipStream_t"),
    ("src/SOP/ProjectNonDivergent/SOP_VDBProjectNonDivergent.cpp", 90, "cudaStream_t", "hipStream_t"),
    ("src/SOP/ProjectNonDivergent/SOP_VDBProjectNonDivergent.cpp", 91, "cudaStreamCreate", "hipStreamCreate"),
    ("src/SOP/ProjectNonDivergent/SOP_VDBProjectNonDivergent.cpp", 95, "nanovdb::cuda::DeviceBuffer", "nanovdb::HostBuffer"),
    ("src/SOP/ProjectNonDivergent/SOP_VDBProjectNonDivergent.cpp", 143, "cudaStreamDestroy", "hipStreamDestroy"),
    # hnanofromgrid: round-trip only, its kernel call is commented out (SOP_VDBFromGrid.cpp:120); only the alias names CUDA
    ("src/SOP/ReadWrite/SOP_VDBFromGrid.cpp", 11, "<cuda_runtime_api.h>", "<hip/hip_runtime_api.h>"),
    ("src/SOP/ReadWrite/SOP_VDBFromGrid.cpp", 16, '"nanovdb/cuda/DeviceBuffer.h"', '"nanovdb/HostBuffer.h"'),
    ("src/SOP/ReadWrite/SOP_VDBFromGrid.cpp", 92, "nanovdb::cuda::DeviceBuffer", "nanovdb::HostBuffer"),
    ("src/SOP/ReadWrite/SOP_VDBFromGrid.cpp", 94, "cudaStream_t", "hipStream_t"),
    ("src/SOP/ReadWrite/SOP_VDBFromGrid.cpp", 95, "cudaStreamCreate", "hipStreamCreate"),
    # pinned host blocks of the field container (Memory.hpp:31-32, :84)
    ("src/Utils/Memory.hpp", 3, "<cuda_runtime.h>", "<hip/hip_runtime_api.h>"),
    ("src/Utils/Memory.hpp", 31, "cudaError_t", "hipError_t"),
    ("src/Utils/Memory.hpp", 31, "cudaMallocHost", "hipHostMalloc"),
    ("src/Utils/Memory.hpp", 31, "cudaSuccess", "hipSuccess"),
    ("src/Utils/Memory.hpp", 32, "cudaGetErrorString", "hipGetErrorString"),
    ("src/Utils/Memory.hpp", 84, "cudaFreeHost", "hipHostFree"),
    # build glue (src/SOP/CMakeLists.txt:2-5, :41, :67): no CUDA language, no Kernels subdirectory, libhns + HIP runtime instead
    ("src/SOP/CMakeLists.txt", 2, "LANGUAGES CUDA CXX", "LANGUAGES CXX"),
    ("src/SOP/CMakeLists.txt", 5, "find_package(CUDAToolkit REQUIRED)", "find_package(hip REQUIRED)  # and: find_library(HNS_LIB hns) ; include_directories(<hnanosolver_amd>/include <hnanosolver_amd>/integration)"),
    ("src/SOP/CMakeLists.txt", 41, "add_subdirectory(../Cuda Kernels)", "add_library(Kernels SHARED <hnanosolver_amd>/integration/hns_shim.cpp) ; target_link_libraries(Kernels PRIVATE ${HNS_LIB} hip::host)"),
]


def apply(root: str, write: bool = True):
    """Returns (matched, problems). With write=False nothing is modified. With write=True NOTHING is written unless every anchor
    of every file matched: a checkout that has moved on is reported, never left half edited."""
    if write:
        matched, problems = apply(root, write=False)
        if problems:
            return matched, problems
    by_file = {}
    for e in EDITS:
        by_file.setdefault(e[0], []).append(e)
    matched, problems = 0, []
    for rel, edits in by_file.items():
        path = os.path.join(root, rel)
        if not os.path.exists(path):
            problems.append(f"{rel}: file not found")
            continue
        with open(path, encoding="utf-8", errors="surrogateescape") as f:
            lines = f.read().split("\n")
        for _, ln, want, repl in edits:
            if ln > len(lines) or want not in lines[ln - 1]:
                problems.append(f"{rel}:{ln}: expected token {want!r} is not on this line")
                continue
            lines[ln - 1] = lines[ln - 1].replace(want, repl)
            matched += 1
        if write:
            with open(path, "w", encoding="utf-8", errors="surrogateescape") as f:
                f.write("\n".join(lines))
    return matched, problems


def leftovers(root: str):
    """CUDA / NanoVDB-CUDA identifiers that remain in the files the SOPs are built from (should be none after apply())."""
    out = []
    for rel in sorted({e[0] for e in EDITS if not e[0].endswith("CMakeLists.txt")}):
        path = os.path.join(root, rel)
        if not os.path.exists(path):
            continue
        with open(path, encoding="utf-8", errors="surrogateescape") as f:
            for i, line in enumerate(f.read().split("\n"), 1):
                code = line.split("//")[0]
                if "cuda" in code:
                    out.append(f"{rel}:{i}")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("checkout")
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    n, bad = apply(a.checkout, write=not a.check)
    print(f"{n} of {len(EDITS)} edits {'match' if a.check else 'applied'}")
    for b in bad:
        print("  !!", b)
    if not a.check:
        for l in leftovers(a.checkout):
            print("  still names CUDA:", l)
    sys.exit(1 if bad else 0)
