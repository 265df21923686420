"""The frozen golden set G1-G4 of SURVEY.md 8c: grids, closed-form inputs and the list of kernel / driver calls whose
outputs are recorded. TEST INFRASTRUCTURE shared by the generator (tests/golden/make_kernel_goldens.py, oracle engine),
the CPU test (oracle reproduces the fixture) and the -m gpu test (HIP kernels reproduce the fixture).

Status of these vectors (round 4): they are outputs of THE REFERENCE'S OWN KERNELS -- src/Cuda/Kernel.cu compiled where it
lies for the host (oracle/Makefile -> oracle/_ref/libhns_refk.so, strict IEEE; oracle/ref_kernels.cpp is the launch) and
launched in the reference's order (tests/oracle_lib.py: RefKernelGrid), written by tests/golden/make_kernel_goldens.py
only after oracle/hns_oracle.c agreed bit for bit. tests/test_ref_kernels.py re-derives them from the reference where it
can be built. (Rounds 1-3: the same bits, then produced by the oracle alone.) Not covered: nvcc's own FMA contraction
(DESIGN.md 2). Every input below is closed-form and uses only IEEE +, -, *, comparisons (no libm), so another machine
regenerates bit-identical inputs.
"""
from __future__ import annotations

import hashlib
from typing import Dict

import numpy as np

from hnanosolver_amd import fields

STRIDE = 37  # G4: every 37th voxel is recorded; G3 uses the same


def grid_leaves(name: str):
    """(leaf origins in NanoVDB order, extent R)"""
    if name == "G1":  # 16^3 dense
        return fields.dense_leaves(16), 16
    if name == "G2":  # sparse 20-leaf set straddling the origin, ragged z-runs and lone leaves
        lat = np.array([[-1, -1, -1], [-1, -1, 0], [-1, 0, -1], [-1, 0, 0], [0, -1, -1], [0, -1, 0], [0, 0, -1], [0, 0, 0], [1, 0, 0], [2, 0, 0],
                        [0, 1, 0], [0, 2, 1], [0, 0, 1], [0, 0, 2], [-2, 0, 0], [-2, -1, 0], [1, 1, 1], [1, -1, 1], [-1, 1, 1], [2, 1, -1]], dtype=np.int32) * 8
        return np.ascontiguousarray(lat[fields.nanovdb_order(lat)]), 32
    if name == "G3":
        return fields.dense_leaves(32), 32
    if name == "G4":
        return fields.dense_leaves(64), 64
    raise KeyError(name)


def inputs(origins: np.ndarray, R: int) -> Dict[str, np.ndarray]:
    """Smooth closed-form fields from IEEE basic operations only (deterministic on every machine): parabola 'sines'
    s(t) = 4 t' (1 - t') on the fractional part, a compactly supported quartic bump for the plume."""
    c = fields.leaves_to_coords(origins).astype(np.float64)
    q = (c + 0.5) * (1.0 / R)
    t = q - np.floor(q)

    def s(x):  # one positive arch per unit interval, then mirrored: a C0 stand-in for sin(2 pi x)
        h = x * 2.0 - np.floor(x * 2.0)
        arch = 4.0 * h * (1.0 - h)
        return np.where(x - np.floor(x) < 0.5, arch, -arch)

    def co(x):
        return s(x + 0.25)

    qx, qy, qz = t[:, 0], t[:, 1], t[:, 2]
    r2 = (qx - 0.5) * (qx - 0.5) + (qy - 0.25) * (qy - 0.25) + (qz - 0.5) * (qz - 0.5)
    w = 1.0 - r2 * 9.0
    blob = np.where(w > 0.0, w * w, 0.0)
    A = 5.0 * 24.0 / R  # |u| dt/dx peaks near 5 voxels at dt = 1/24, like SURVEY 8d
    out = {
        "vel": np.stack([A * 0.5 * s(qy) * co(qz), A * (blob + 0.25 * s(qx)), A * 0.5 * co(qx) * s(qy)], axis=-1).astype(np.float32),
        "density": blob.astype(np.float32),
        "temperature": (23.0 + 50.0 * blob).astype(np.float32),
        "fuel": (0.2 * blob + 0.0005).astype(np.float32),  # + a sub-threshold floor: exercises fuel < 0.001 -> 0
        "waste": (0.9 * np.where(qy > 0.6, qy - 0.6, 0.0)).astype(np.float32),
        "flame": (0.05 * qz).astype(np.float32),
        "pressure0": (0.01 * s(qx) * s(qz) + 0.02 * co(qy)).astype(np.float32),
    }
    d = (r2 - 0.04) * R * 2.0  # sign-correct pseudo distance to a sphere, in voxels near its surface
    out["collision_sdf"] = d.astype(np.float32)
    out["collision_sdf"][::11] = np.float32(0.04)  # some voxels inside the blend margin [0, 0.1)
    return out


def run_all(K, name: str, params_cls) -> Dict[str, np.ndarray]:
    """Every recorded output of grid `name` through engine K (OracleGrid or HipKernels signatures)."""
    origins, R = grid_leaves(name)
    f = inputs(origins, R)
    vs, dt = 1.0 / R, 1.0 / 24.0
    inv_dx = float(np.float32(1.0) / np.float32(vs))
    vs32 = float(np.float32(vs))
    out: Dict[str, np.ndarray] = {}
    vel, sdf = f["vel"], f["collision_sdf"]
    # --- every kernel alone ---
    out["advect_vector"] = K.advect_vector(vel, dt, inv_dx)
    out["advect_vector_coll"] = K.advect_vector(vel, dt, inv_dx, sdf, True)
    out["advect_scalar"] = K.advect_scalar(vel, f["density"], dt, inv_dx)
    out["advect_scalar_coll"] = K.advect_scalar(vel, f["density"], dt, inv_dx, sdf, True)
    d0 = f["density"].copy()
    d0[0] = 100.0  # element 0 is what out-of-domain taps of advect_scalars read (Kernel.cu:133,192,225)
    out["advect_scalars_s1"] = K.advect_scalars(vel, [d0], dt, inv_dx)[0]
    five = [d0, f["temperature"], f["fuel"], f["waste"], f["flame"]]
    for k, a in enumerate(K.advect_scalars(vel, five, dt, inv_dx)):
        out[f"advect_scalars_s5_{k}"] = a
    for k, a in enumerate(K.advect_scalars(vel, five[:2], dt, inv_dx, sdf, True)):
        out[f"advect_scalars_coll_{k}"] = a
    div = K.divergence(vel, inv_dx)
    out["divergence"] = div
    omega = 1.7
    p = K.rbgs(div, f["pressure0"].copy(), vs32, 0, omega)
    out["rbgs_red"] = p.copy()
    out["rbgs_black"] = K.rbgs(div, p.copy(), vs32, 1, omega)
    out["rbgs_3_iterations"] = K.rbgs_iterations(div, vs32, omega, 3, f["pressure0"])
    out["gradient"] = K.subtract_pressure_gradient(vel, f["pressure0"], inv_dx)
    out["gradient_coll"] = K.subtract_pressure_gradient(vel, f["pressure0"], inv_dx, sdf, True)
    out["vorticity_fs1"] = K.vorticity_confinement(vel, dt, inv_dx, 0.3, 1.0)
    out["vorticity_fs2"] = K.vorticity_confinement(vel, dt, inv_dx, 1.0, 2.0)
    cf, cw, ct, cl, cd = K.combustion_oxygen(f["fuel"], f["waste"], f["temperature"], div.copy(), f["flame"], 0.5, 0.1)
    out.update(combustion_fuel=cf, combustion_waste=cw, combustion_temperature=ct, combustion_flame=cl, combustion_divergence=cd)
    out["buoyancy"] = K.temperature_buoyancy(vel, f["temperature"], dt, 23.0, 1.0)
    out["enforce_collision"] = K.enforce_collision_boundaries(vel.copy(), sdf, vs32)
    # --- pressure_projection_idx order (PressureProjection.cu:43-66) with 1, 2, 50 iterations ---
    for it in (1, 2, 50):
        u = vel.copy()
        assert K.project_non_divergent(u, it, vs) == 0
        out[f"project_{it}"] = u
    # --- Compute order (HNanoSolver.cu:150-356): combustion fields zero / non-zero, collision off / on ---
    names = ["density", "temperature", "fuel", "waste", "flame"]
    for tag, burning, coll, fs in (("zero", False, False, 0.5), ("burn", True, False, 1.0), ("burn_coll", True, True, 1.0)):
        cur = {n: (f[n].copy() if (burning or n in ("density", "temperature")) else np.zeros_like(f[n])) for n in names}
        if coll:
            cur["collision_sdf"] = sdf.copy()
        u = vel.copy()
        assert K.compute_sim(u, cur, 6, dt, vs, params_cls(factorScale=fs, vorticityScale=0.3), coll) == 0
        out[f"compute_{tag}_vel"] = u
        for n in names:
            out[f"compute_{tag}_{n}"] = cur[n]
    return out


def digest(a: np.ndarray) -> dict:
    """What the fixture keeps of one output array: SHA-256 of the raw float32 bytes, L2 and L-inf norms (float64), and for
    the small grids the array itself / for the large ones every STRIDE-th element."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    x = a.astype(np.float64)
    return {"sha256": hashlib.sha256(a.tobytes()).hexdigest(), "l2": float(np.sqrt((x * x).sum())), "linf": float(np.abs(x).max()) if x.size else 0.0}
