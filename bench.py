#!/usr/bin/env python3
"""bench.py -- solver substeps/s of the HNanoSolver hot path on MI355X, with the pressure stencil's HBM roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 256|128|64|plume] [--iterations 50]
    (N > 1 without a launcher: bench.py starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` itself,
    as a child process, before it touches the GPU; under a launcher -- WORLD_SIZE set -- it is a rank, as before)

A "step" is one core substep on device-resident fields: advect_vector -> divergence -> 50 red-black
SOR iterations -> pressure-gradient subtraction -> advect_scalars (S=1), i.e. 688 algorithmic bytes per voxel. Inputs
are the closed-form synthetic fields of hnanosolver_amd.fields, already resident in HBM when the timed region starts.

N = 1: the workload is BASELINE.json's roofline configuration, the 256^3 dense-active grid (16,777,216 voxels).
N > 1: weak scaling -- every rank owns one such x-slab of a (256*N) x 256 x 256 domain (--partition: ONE --config domain,
e.g. plume1024 = BASELINE.json's 1024^3-extent sparse grid, split across the ranks instead); ranks exchange the halo
voxels of u / div / p / phi over RCCL where the single-GPU code has a kernel boundary that a stencil crosses, under the
interior kernels (csrc/hns_dist_*.hip; hnanosolver_amd/dist.py is the host mirror). `value` = slab-substeps/s summed over ranks = N x (global substeps/s).

Rank 0 prints ONE JSON line. `roofline`: the SOR kernel (`kernel` names the form the library picked for this grid size).
Algorithmic bytes are 12 B/voxel per red+black iteration (read p, read div, write p once each); a launch of the temporally
blocked form holds `iterations_per_launch` iterations, so per KERNEL LAUNCH -- the unit of `algorithmic_bytes_per_launch`,
`ms_per_launch` and `traffic`, and what rocprofv3 --stats lists -- they are 12 B/voxel x iterations per launch; `achieved` is
their ratio either way. Times come from hipEvents recorded on the launch stream around the pressure loop of every timed step. `roofline.kernels` carries the same figures for all five kernels of the substep (each bracketed by hipEvents
on the launch stream, in a short pass of its own after the timed region) and `roofline.substep` the whole substep
against 688 B/voxel. `traffic` is NOT measured by this
run (bench.py cannot run rocprofv3 on itself): it is the PMC-derived HBM bytes per launch from the builder's committed
rocprofv3 passes of this same command (profiles/pmc_latest.json: per configuration and kernel), reported only while they
were taken from the kernel source this library was built from, and labelled with `traffic_source`. Three fractions of the
8 TB/s peak: `frac` (contractual: 12 B/voxel per ITERATION), `frac_compulsory` (12 B/voxel per LAUNCH: what the kernel must
move) and `frac_moved` (what it did move, PMC); `roofline.kernels` carries `traffic` / `frac_moved` for all five kernels. `cpu_baseline`: the oracle (C restatement
of the reference kernels, OpenMP over leaves) on the host cores, rank 0 at N=1 only, on a bounded sample.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
BYTES_PER_VOXEL_ITER = 12  # RB-SOR per iteration reads p, reads div, writes p once each
BYTES_PER_VOXEL_SUBSTEP = 688  # advect_vector 24 + divergence 16 + 50*12 + gradient 28 + advect_scalars(S=1) 20
# SURVEY.md 8d, algorithmic bytes per voxel of each stage of the core substep (pressure: per iteration)
STAGE_BYTES = {"advect_vector": 24, "divergence": 16, "pressure": BYTES_PER_VOXEL_ITER, "gradient": 28, "advect_scalars": 20}
# (the divergence kernel is picked by size: z-paired workgroups from 16,384 leaves, one leaf per workgroup below; the pressure kernel's name comes from hns_grid_rbgs_plan)
STAGE_KERNEL = {"advect_vector": ("k_advect_vector_n",), "divergence": ("k_divergence_zpair", "k_divergence_row"), "pressure": (), "gradient": ("k_subtract_gradient_s",),
                "advect_scalars": ("k_advect_scalars_n",)}


def kernel_source_sha16():
    """Identifies the kernel source a PMC profile belongs to (profiles/pmc_latest.json carries the same stamp)."""
    h = hashlib.sha256()
    for f in ("hns_pressure.hip", "hns_sorblock.hip", "hns_advect.hip", "hns_device.hpp", "hns_internal.hpp", "hns_flags.hpp"):
        h.update(open(os.path.join(ROOT, "hnanosolver_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="256", help="256 (default, roofline config) | 128 | 64 | plume")
    ap.add_argument("--iterations", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--partition", action="store_true", help="N > 1: split ONE --config domain across the ranks (strong scaling) instead of one slab per rank")
    ap.add_argument("--sweeps-per-exchange", type=int, default=0, help="N > 1: fused SOR sweeps between two halo refreshes of p (1..4, 0 = library default)")
    ap.add_argument("--transport", choices=["auto", "rccl", "ipc"], default="rccl",
                    help="N > 1 halo transport: rccl (default) = RCCL send/recv groups on a communication stream under the interior kernels; ipc = one-sided "
                         "stores into hipIpc-mapped peer memory, every kernel delivering its own halo (has only run between processes sharing ONE GPU so "
                         "far: opt-in); auto = ipc if it connects and reproduces three RCCL substeps bit for bit on this machine, else rccl. Whatever "
                         "runs is first checked against the single-GPU result of the whole domain (config.verified)")
    ap.add_argument("--no-verify", action="store_true", help="N > 1: skip the start-up comparison of the partitioned run with the single-GPU run of the whole domain")
    ap.add_argument("--share-one-gpu", action="store_true",
                    help="builder's check on a 1-GPU box: all N ranks on cuda:0, host rendezvous over gloo, --transport ipc (RCCL refuses two ranks on one device)")
    ap.add_argument("--cook", action="store_true", help="time the cook-equivalent hns_compute_sim call (upload + grid build + substep + download) instead")
    ap.add_argument("--full", action="store_true", help="time the FULL Compute_Sim substep instead (HNanoSolver.cu:150-356 on device-resident fields: five advected scalars, "
                                                        "combustion, buoyancy; SURVEY 8d's 812 B/voxel row) and print its own JSON line; single GPU")
    ap.add_argument("--strong-timeout", type=int, default=600, help="seconds EACH of the strong_scaling records may take before a watchdog thread abandons it and prints the line without it (0 = no limit)")
    ap.add_argument("--no-one-sided", action="store_true", help="N > 1, default workload: skip the third record `strong_scaling_one_sided` (config 5 again over the one-sided transport "
                                                                 "where it connects and reproduces RCCL bit for bit)")
    ap.add_argument("--no-strong", action="store_true", help="default workload only: skip the second record `strong_scaling` (BASELINE config 5, the 1024^3-extent plume "
                                                              "as ONE domain over the N ranks)")
    return ap.parse_args()


def cook_equivalent(args):
    """What one reference SOP cook costs (HNanoSolver.cu:87-133,361-371 + :375-384): CreateIndexGrid + Compute_Sim with all
    fields crossing PCIe both ways. Reported on its own line; never the headline value."""
    from hnanosolver_amd import api, fields

    origins, R = fields.config_leaves(args.config)
    f = fields.synthetic_fields(origins, R)
    d = api.GridIndexedData()
    coords = fields.leaves_to_coords(origins)
    d.allocateCoords(len(coords))
    d.pCoords()[:] = coords
    for n in ("density", "temperature", "fuel", "waste", "flame"):
        d.addValueBlock(n, d.FLOAT)
        d.pValues(n)[:] = f[n]
    d.addValueBlock("vel", d.VEC3F)
    d.pValues("vel")[:] = f["vel"]
    p = api.CombustionParams(vorticityScale=0.0)
    k = args.warmup
    out = {"metric": "cook-equivalent Compute_Sim (host arrays in, host arrays out; PCIe inclusive)", "config": args.config,
           "active_voxels": len(coords), "bytes_over_pcie_per_cook": int(len(coords) * (12 + 5 * 4) * 2)}
    # cold: what the reference does every cook -- release the previous grid and device buffers, build new ones. warm: the same handle again; CreateIndexGrid finds the topology unchanged and the grid
    # keeps the device buffers of the previous cook.
    # feedback: warm, and the caller vouches that its arrays still hold what the previous cook handed back -- what the reference's SOP does
    # with its feedback input (SOP_HNanoSolver.cpp:106): nothing is uploaded, the cook is the substep plus the downloads (hns_compute_sim_resident)
    # feedback_checked: the same with the promise CHECKED (a digest of every element at both ends of the cook) instead of vouched
    for mode in ("cold", "warm", "feedback", "feedback_checked"):
        times = {"release_ms": [], "create_index_grid_ms": [], "compute_sim_ms": [], "total_ms": []}
        h = api.IndexGridHandle()
        for it in range(args.warmup + args.steps):
            ta = time.perf_counter()
            if mode == "cold":
                h.reset()
            t0 = time.perf_counter()
            api.CreateIndexGrid(d, h, 1.0 / R)
            t1 = time.perf_counter()
            api.Compute_Sim(d, h, args.iterations, 1.0 / 24.0, 1.0 / R, p, False, feedback=(mode.startswith("feedback") and it > 0) or None, checked=mode == "feedback_checked")
            t2 = time.perf_counter()
            times["release_ms"].append(1e3 * (t0 - ta))
            times["create_index_grid_ms"].append(1e3 * (t1 - t0))
            times["compute_sim_ms"].append(1e3 * (t2 - t1))
            times["total_ms"].append(1e3 * (t2 - ta))
        h.reset()
        out[mode] = {n: float(np.median(v[k:])) for n, v in times.items()}
    print(json.dumps(out))


def cpu_baseline(origins, R, iterations, budget_s=25.0):
    """Time the oracle (test infrastructure, here only as the reported CPU baseline) on a bounded sample: the SAME
    workload with `it_s` pressure iterations instead of `iterations`, extrapolated linearly in the iteration count
    (every iteration does identical work)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hnanosolver_amd import fields
    from oracle_lib import OracleGrid, oracle

    L = oracle()  # sets the OpenMP thread count to the CPUs this process may use (affinity and cgroup quota)
    cores = int(L.orc_get_threads())
    G = OracleGrid(origins)
    f = fields.synthetic_fields(origins, R)
    vs = 1.0 / R
    inv_dx, dt = float(R), 1.0 / 24.0
    omega = float(L.orc_omega_compute(vs))
    t0 = time.perf_counter()
    adv = G.advect_vector(f["vel"], dt, inv_dx)
    div = G.divergence(adv, inv_dx)
    t_pre = time.perf_counter() - t0
    p = np.zeros(G.N, dtype=np.float32)
    it_s, t_it = 0, 0.0
    while it_s < iterations and (it_s < 2 or t_it + t_pre * 2 < budget_s * 0.6):
        t1 = time.perf_counter()
        G.rbgs(div, p, vs, 0, omega)
        G.rbgs(div, p, vs, 1, omega)
        t_it += time.perf_counter() - t1
        it_s += 1
    t2 = time.perf_counter()
    u = G.subtract_pressure_gradient(adv, p, inv_dx)
    G.advect_scalars(u, [f["density"]], dt, inv_dx)
    t_post = time.perf_counter() - t2
    per_substep = t_pre + t_post + (t_it / it_s) * iterations
    # ONE core on BASELINE.json's CPU-runnable configuration (64^3, the whole substep, nothing extrapolated): SURVEY 8d asks for it
    one = None
    try:
        o64 = fields.dense_leaves(64)
        G1, f1 = OracleGrid(o64), fields.synthetic_fields(o64, 64)
        L.orc_set_threads(1)
        t3 = time.perf_counter()
        a1 = G1.advect_vector(f1["vel"], dt, 64.0)
        d1 = G1.divergence(a1, 64.0)
        p1 = G1.rbgs_iterations(d1, 1.0 / 64, float(L.orc_omega_compute(1.0 / 64)), iterations)
        u1 = G1.subtract_pressure_gradient(a1, p1, 64.0)
        G1.advect_scalars(u1, [f1["density"]], dt, 64.0)
        one = {"value": 1.0 / (time.perf_counter() - t3), "unit": "substeps/s", "cores": 1, "workload": "64^3 dense-active grid, the whole core substep"}
    finally:
        L.orc_set_threads(cores)
    # the REFERENCE ITSELF (src/Cuda/Kernel.cu built for the host into oracle/_ref/libhns_refk.so; one thread, serial launch emulation)
    # on BASELINE.json's configs[0], the 64^3 smoke plume: the same core substep, when that prebuilt library travelled with the repo
    ref64 = None
    try:
        from oracle_lib import RefKernelGrid, reference_kernels, reference_samplers

        if reference_kernels() is not None and reference_samplers() is not None:
            o64 = fields.dense_leaves(64)
            K, f1 = RefKernelGrid(o64), fields.synthetic_fields(o64, 64)
            t4 = time.perf_counter()
            a1 = K.advect_vector(f1["vel"], dt, 64.0)
            d1 = K.divergence(a1, 64.0)
            p1 = K.rbgs_iterations(d1, 1.0 / 64, float(L.orc_omega_compute(1.0 / 64)), iterations)
            u1 = K.subtract_pressure_gradient(a1, p1, 64.0)
            K.advect_scalars(u1, [f1["density"]], dt, 64.0)
            ref64 = {"value": 1.0 / (time.perf_counter() - t4), "unit": "substeps/s", "cores": 1, "kind": "reference",
                     "workload": "64^3 dense-active grid, the whole core substep, the reference's own kernels compiled for the host"}
    except Exception as e:  # noqa: BLE001 -- a reported extra, never the reason a bench line is missing
        ref64 = {"error": f"{type(e).__name__}: {e}"[:200]}
    return {
        "reference_64": ref64,
        "one_core_64": one,
        "value": 1.0 / per_substep,
        "unit": "substeps/s",
        "cores": cores,
        "kind": "port",
        # no parentheses in this string: the driver's record parser cuts it at the first one
        "sample": f"oracle core substep on the same {G.N}-voxel grid: advect_vector + divergence + gradient + advect_scalars S=1 timed once in "
                  f"{t_pre + t_post:.2f} s, then {it_s} of {iterations} RB-SOR iterations timed in {t_it:.2f} s and scaled linearly",
    }


class Watchdog:
    """Cuts a hang the main thread cannot see: a daemon timer THREAD that, after `seconds`, runs `last_words()` and leaves the process with os._exit(0). A Python signal
    handler (signal.alarm) only runs when the main thread returns to the bytecode loop, which a collective that never returns does not do; a thread runs while the main
    thread is blocked in C with the GIL released (dist.barrier, torch.cuda.synchronize, ctypes calls all release it). seconds <= 0: no watchdog."""

    def __init__(self, seconds, last_words):
        import threading

        self._fired = threading.Event()
        self._timer = None
        if seconds and seconds > 0:
            self._timer = threading.Timer(seconds, self._fire, args=(last_words,))
            self._timer.daemon = True
            self._timer.start()

    def _fire(self, last_words):
        self._fired.set()
        try:
            last_words()
        finally:
            os._exit(0)

    def cancel(self):
        if self._timer is not None:
            self._timer.cancel()
            if self._fired.is_set():  # (fired a moment ago: it is printing the line and taking the process down -- do not print a second one)
                time.sleep(3600)


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, as a CHILD process
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`), let rank 0's JSON line
    through on the inherited stdout and hand back the child's exit code. This parent never touches the GPU (no torch
    import, no hns_* call) and never replaces itself with another program."""
    import subprocess

    # --standalone: torchrun's own c10d rendezvous on a port IT binds (endpoint localhost:0) -- picking a "free" port here and passing its number on
    # leaves a window in which another process, or a second bench.py on the same box, takes it
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL and hipIpc need on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def timed_steps(step, steps, warmup, world, on_cpu, before_timed=None):
    """The contract's timed region: W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides; MAX over ranks."""
    import torch
    import torch.distributed as dist

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if before_timed:
        before_timed()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if on_cpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def ghost_note_of(runner):
    """after a timed region: do the ghost voxels the timed transport wrote hold their owners' bits? (collective; any failure is reported, not raised)"""
    try:
        n_pairs, bad = runner.rank_obj.ghost_check(None, stream=runner.stream)
        return (f"after the timed substeps the velocity and p ghost voxels of all {n_pairs} (owner, holder, field) pairs are bit-equal to their owners' values"
                if not bad else f"GHOST VOXELS DIFFER FROM THEIR OWNERS' VALUES after the timed substeps in {len(bad)} of {n_pairs} pairs, first {bad[0]}")
    except Exception as e:  # noqa: BLE001
        return f"ghost check did not complete: {type(e).__name__}: {e}"[:300]


def strong_scaling_record(args, world, rank, dt, transport=None):
    """BASELINE.json configs[4] beside the weak-scaling headline, in the same JSON line: the 1024^3-extent plume (65,944 leaves) as ONE domain split over the N
    ranks (N = 1: the whole domain on the one GPU -- the curve's first point). Same step, same timing rule, same checks (against the single-GPU run of the
    whole domain before, ghost voxels against their owners after). `value` = substeps/s of the one domain."""
    import torch

    from hnanosolver_amd import api, device as D, fields

    origins, R = fields.config_leaves("plume1024")
    vs = 1.0 / R
    rec = {"metric": "solver substeps/sec of ONE domain over N GPUs (strong scaling)", "unit": "substeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "scaling": "strong", "config": {"workload": "plume1024: BASELINE.json configs[4], 1024^3-extent sparse plume", "leaves": int(len(origins)),
                                           "active_voxels": int(len(origins)) * 512, "pressure_iterations": args.iterations}}
    if world == 1:
        f = fields.synthetic_fields(origins, R)
        grid = api.create_grid_from_leaves(origins, vs)
        sim = D.Sim(grid, ["density"])
        sim.upload({"vel": f["vel"], "density": f["density"]})
        stream = D.current_stream()
        elapsed = timed_steps(lambda: sim.core_substep(args.iterations, dt, vs, stream), args.steps, args.warmup, 1, False)
        rec["config"]["parallelism"] = "single GPU"
        sim.close()
    else:
        from hnanosolver_amd import dist as HD

        runner = HD.SlabBench(origins, R, rank, world, args.iterations, dt, partition=True, sweeps_per_exchange=args.sweeps_per_exchange, transport=transport or args.transport,
                              reference_transport="ipc" if args.share_one_gpu else "rccl")
        if not args.no_verify:
            try:
                runner.verify_against_single_gpu()
            except Exception as e:  # noqa: BLE001
                runner.verified_note = f"check against the single-GPU run did not complete: {type(e).__name__}: {e}"[:300]
                runner.rank_obj.upload(*runner._fields, stream=runner.stream)
        elapsed = timed_steps(runner.step, args.steps, args.warmup, world, args.share_one_gpu)
        info = runner.info()
        rec["config"].update({
            "leaves_on_rank_0": runner.n_owned, "verified": runner.verified_note, "ghosts": ghost_note_of(runner),
            "halo": {k: info[k] for k in ("boundary_leaves", "interior_leaves", "ghost_leaves", "peers", "halo_peers", "sweeps_per_exchange", "bytes_sent", "messages_sent", "exchanges")},
            "parallelism": f"one domain in {world} slabs of leaves along axis {'xyz'[runner.rank_obj.partition_axis] if runner.rank_obj.partition_axis >= 0 else 'x (contiguous leaf ranges)'}"
                           + ", halo transport: " + runner.transport_note + (" -- ALL RANKS SHARE ONE GPU (builder's check, not a scaling figure)" if args.share_one_gpu else "")})
        runner.rank_obj.close()
    torch.cuda.synchronize()
    rec["value"] = args.steps / elapsed
    rec["ms_per_step"] = 1e3 * elapsed / args.steps
    return rec


FULL_BYTES_PER_VOXEL = 812  # core 688 + vorticity 24 + combustion 40 + buoyancy 28 + four more advected scalars 4 x 8
# the five brackets of hns_sim_stage_timing over hns_sim_substep, with SURVEY 8d's algorithmic bytes per voxel (pressure: per iteration)
FULL_STAGE_BYTES = {"advect_vector": 24, "divergence": 16 + 40 + 28, "pressure": BYTES_PER_VOXEL_ITER, "gradient": 28, "advect_scalars": 12 + 8 * 5}
FULL_STAGE_NAME = {"advect_vector": "advect_vector", "divergence": "divergence_combustion_buoyancy", "pressure": "pressure", "gradient": "gradient", "advect_scalars": "advect_scalars_S5"}
FUSED_MIDDLE_BYTES = 60  # what the fused launch must move: u* in 12, div out 4, four scalars in 16 and out 16, buoyed u* out 12


def full_substep(args):
    """The whole Compute_Sim substep a SOP cook runs (reference HNanoSolver.cu:150-356; SURVEY 8d's last row): advect_vector, [vorticity confinement:
    an exact copy at the default factor_scale 0.5, skipped], divergence, combustion, buoyancy, 50 RB-SOR iterations, gradient subtraction, advect_scalars
    over the five float fields -- on device-resident fields, K timed substeps between synchronisations. Round 6: divergence + combustion + buoyancy are ONE
    launch and the four combustion fields are advected out of one 16-byte-per-voxel array (option "fuse"; bit-identical to the separate launches).
    `kernels`: the five stages bracketed by hipEvents on the launch stream INSIDE substeps of a short pass of its own after the timed region
    (hns_sim_stage_timing; the rocprofv3 kernel statistics of this command, profiles/r06_full256_kernel_stats.csv, must agree)."""
    import torch

    import hnanosolver_amd as H
    from hnanosolver_amd import api, device as D, fields

    origins, R = fields.config_leaves(args.config)
    vs, dt = 1.0 / R, 1.0 / 24.0
    n_vox = len(origins) * 512
    names = ["density", "temperature", "fuel", "waste", "flame"]
    f = fields.synthetic_fields(origins, R)
    grid = api.create_grid_from_leaves(origins, vs)
    sim = D.Sim(grid, names)
    sim.upload({"vel": f["vel"], **{n: f[n] for n in names}})
    prm = api.CombustionParams(vorticityScale=0.0)  # SOP defaults, vorticityScale = 0 as SURVEY 8d prescribes for metric runs (factor_scale 0.5: a no-op either way)
    stream = D.current_stream()

    def step():
        sim.substep(args.iterations, dt, vs, prm, False, stream)

    sim.timing(args.steps)
    elapsed = timed_steps(step, args.steps, args.warmup, 1, False)
    p_ms, p_iters = sim.pressure_time()
    ms = 1e3 * elapsed / args.steps
    sor_form, sor_launches, _ = D.rbgs_plan(grid, args.iterations)
    n_st = min(args.steps, 10)
    sim.timing(0)
    sim.stage_timing(n_st)
    for _ in range(n_st):
        step()
    torch.cuda.synchronize()
    stage_ms, n_sub = sim.stage_times()
    fused = H.get_option("fuse") == "1"
    kernels = {}
    for st, tot in stage_ms.items():
        t_ms = tot / max(1, n_sub)
        alg = FULL_STAGE_BYTES[st] * n_vox * (args.iterations if st == "pressure" else 1)
        gbs = alg / (t_ms * 1e-3) / 1e9 if t_ms > 0 else None
        ent = {"ms_per_substep": t_ms, "algorithmic_bytes": alg, "achieved": gbs, "frac": gbs / HBM_PEAK_GBS if gbs else None, "share_of_substep": t_ms / ms,
               "timed": "hipEvents on the launch stream inside the substeps of a pass after the timed region"}
        if st == "divergence":
            ent["launches"] = 1 if fused else 3
            if fused:
                ent["frac_compulsory"] = FUSED_MIDDLE_BYTES * n_vox / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if t_ms > 0 else None
                ent["note"] = ("one launch (k_divergence_combust_buoyancy): `frac` prices SURVEY 8d's 16 + 40 + 28 B/voxel of the three reference launches, "
                               f"`frac_compulsory` the {FUSED_MIDDLE_BYTES} B/voxel the fused launch must move")
        kernels[FULL_STAGE_NAME[st]] = ent
    sub_bytes = (FULL_BYTES_PER_VOXEL - 600 + 12 * args.iterations - 24) * n_vox  # (the vorticity pass, 24 B/voxel, is an exact copy at factor_scale < 1 and skipped)
    gbs = sub_bytes / (ms * 1e-3) / 1e9
    print(json.dumps({
        "metric": "full Compute_Sim substeps/sec (advect + vorticity + combustion + buoyancy + 50 red-black SOR iterations + project + advect of 5 scalars) at N active voxels",
        "value": args.steps / elapsed, "unit": "substeps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}^3 dense-active grid" if args.config.isdigit() else args.config, "active_voxels_per_gpu": n_vox, "pressure_iterations": args.iterations,
                   "substep": "Compute_Sim order (HNanoSolver.cu:150-356), S = 5 advected scalars, combustion, buoyancy; vorticity confinement at factor_scale 0.5 is an exact copy and skipped; "
                              + ("divergence + combustion + buoyancy fused into one launch, {fuel, waste, temperature, flame} advected out of one 16-byte-per-voxel array" if fused
                                 else "option fuse = 0: the reference's three launches over five float arrays"),
                   "algorithmic_bytes_per_voxel_substep": FULL_BYTES_PER_VOXEL - 600 + 12 * args.iterations - 24, "sor_form": sor_form},
        "roofline": {"kernel": "whole substep", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_substep": sub_bytes, "pressure_ms_per_iteration": p_ms / max(1, p_iters), "kernel_launches_per_solve": sor_launches, "kernels": kernels},
    }))
    sim.close()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} but the launcher started WORLD_SIZE = {world} ranks: pass the same N to both")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    if args.share_one_gpu:
        local_rank = 0
        import hnanosolver_amd as H

        H.set_option("dist_mirror", "guarded")  # processes sharing a GPU must not wait inside their sweeps (see hns_dist_*.hip)
        if args.transport == "rccl":
            args.transport = "ipc"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from hnanosolver_amd import api, device as D, fields

    if args.cook:
        cook_equivalent(args)
        return
    if args.full:
        if world != 1:
            raise SystemExit("bench.py --full is the single-GPU Compute_Sim substep")
        full_substep(args)
        return
    origins, R = fields.config_leaves(args.config)
    vs, dt = 1.0 / R, 1.0 / 24.0
    n_vox_rank = len(origins) * 512

    if world == 1:
        f = fields.synthetic_fields(origins, R)
        grid = api.create_grid_from_leaves(origins, vs)
        sim = D.Sim(grid, ["density"])
        sim.upload({"vel": f["vel"], "density": f["density"]})
        stream = D.current_stream()

        def step():
            sim.core_substep(args.iterations, dt, vs, stream)

        def timing_on():
            sim.timing(args.steps)

        def pressure_time():
            return sim.pressure_time()

        def stage_times():
            # per-kernel figures from a pass of its own, after the timed region: six more events per substep on the launch
            # stream would cost the timed loop ~1 % at 256^3 (and 10 % at 64^3)
            n = min(args.steps, 10)
            sim.stage_timing(n)
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            return sim.stage_times()
    else:
        from hnanosolver_amd import dist as HD

        runner = HD.SlabBench(origins, R, rank, world, args.iterations, dt, partition=args.partition, sweeps_per_exchange=args.sweeps_per_exchange,
                              transport=args.transport, reference_transport="ipc" if args.share_one_gpu else "rccl")
        n_vox_rank = runner.n_owned * 512
        step, pressure_time, stage_times = runner.step, runner.pressure_time, None
        if not args.no_verify:
            try:
                runner.verify_against_single_gpu()
            except Exception as e:  # noqa: BLE001 -- the check must not cost the measurement; it says that it did not run
                runner.verified_note = f"check against the single-GPU run did not complete: {type(e).__name__}: {e}"[:300]
                runner.rank_obj.upload(*runner._fields, stream=runner.stream)

        def timing_on():
            runner.timing_on(args.steps)

    elapsed = timed_steps(step, args.steps, args.warmup, world, args.share_one_gpu, before_timed=timing_on)
    ghosts_note = ghost_note_of(runner) if world > 1 else None
    p_ms, launches = pressure_time()  # (launches: red+black iterations inside the bracketed pressure loops)
    sor_form, sor_launches, sor_k = D.rbgs_plan(grid, args.iterations) if world == 1 else ("", args.iterations, 1)
    stages, n_sub = stage_times() if stage_times else ({}, 0)
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = (1 if args.partition and world > 1 else world) * args.steps / elapsed  # slab-substeps/s over all ranks (partitioned: substeps/s of the one domain)
        ms_iter = p_ms / max(1, launches)  # per red+black iteration
        # per KERNEL LAUNCH (what rocprofv3 --stats lists): a solve of `iterations` iterations is sor_launches launches
        iters_per_launch = args.iterations / max(1, sor_launches)
        ms_launch = ms_iter * iters_per_launch
        achieved = BYTES_PER_VOXEL_ITER * n_vox_rank / (ms_iter * 1e-3) / 1e9 if launches else None
        # HBM-side bytes per kernel launch from the builder's committed PMC passes (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction), per
        # configuration and kernel, quoted only while they were taken from the kernel sources this library was built from
        traffic, traffic_source, pmc_kernels = None, None, {}
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc) and world == 1:
            try:
                j = json.load(open(pmc))
                if j.get("kernel_source_sha16") == kernel_source_sha16():
                    pmc_kernels = j.get("configs", {}).get(args.config, {}).get("kernels", {})
                    traffic_source = ("profiles/pmc_latest.json: builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate) of `bench.py --config " + args.config +
                                      "` on this kernel source, 2 x FETCH_SIZE + WRITE_SIZE; not measured by this run")
            except Exception:
                pmc_kernels = {}

        def pmc_of(kernel_name):  # the profile keys kernels by their name without template arguments
            return (pmc_kernels.get(kernel_name.split("<")[0].split(":")[0]) or {}).get("hbm_bytes_per_kernel_launch")

        traffic = pmc_of(sor_form) if sor_form else None
        if traffic is None:
            traffic_source = None
        kernels = {}
        if n_sub:
            for st, ms in stages.items():
                n_l = sor_launches if st == "pressure" else 1
                per = ms / n_sub / n_l  # ms per kernel launch
                alg = STAGE_BYTES[st] * n_vox_rank * (args.iterations / n_l if st == "pressure" else 1)
                gbs = alg / (per * 1e-3) / 1e9 if per > 0 else None
                if st == "pressure":
                    kname = sor_form.split(":")[0]
                else:  # (whichever of the stage's kernels the committed PMC profile of this configuration saw; else the first)
                    kname = next((k for k in STAGE_KERNEL[st] if pmc_of(k) is not None), STAGE_KERNEL[st][0] if n_vox_rank >= 16384 * 512 or len(STAGE_KERNEL[st]) == 1 else STAGE_KERNEL[st][-1])
                moved = pmc_of(kname)
                ent = {"stage": st, "ms_per_launch": per, "launches_per_substep": n_l, "algorithmic_bytes_per_launch": alg, "achieved": gbs,
                       "frac": gbs / HBM_PEAK_GBS if gbs else None,
                       # what the kernel really moves per launch (PMC) and that rate against the peak
                       "traffic": moved, "frac_moved": (moved / (per * 1e-3) / 1e9 / HBM_PEAK_GBS) if (moved and per > 0) else None}
                if st == "pressure" and n_l != args.iterations:
                    ent["frac_compulsory"] = (BYTES_PER_VOXEL_ITER * n_vox_rank / (per * 1e-3) / 1e9 / HBM_PEAK_GBS) if per > 0 else None
                    ent["note"] = (f"one launch = {args.iterations / n_l:g} red+black iterations: `frac` prices SURVEY 8d's 12 B/voxel PER ITERATION and can exceed 1; "
                                   "`frac_compulsory` prices one pass over p, p, div per LAUNCH (what the kernel must move), `frac_moved` the PMC bytes")
                kernels[kname] = ent
        sub_bytes = (BYTES_PER_VOXEL_SUBSTEP - 600 + 12 * args.iterations) * n_vox_rank
        sub_gbs = sub_bytes / (ms_per_step * 1e-3) / 1e9
        out = {
            "metric": "solver substeps/sec (advect + 50 red-black SOR iterations + project) at N active voxels",
            "value": value,
            "unit": "substeps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if (args.partition and world > 1) else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}^3 dense-active grid" if args.config.isdigit() else args.config,
                "active_voxels_per_gpu": n_vox_rank,
                "leaves_per_gpu": len(origins) if not (args.partition and world > 1) else n_vox_rank // 512,
                "pressure_iterations": args.iterations,
                "substep": "advect_vector + divergence + RB-SOR + gradient subtraction + advect_scalars(S=1)",
                "algorithmic_bytes_per_voxel_substep": BYTES_PER_VOXEL_SUBSTEP - 600 + 12 * args.iterations,
                "halo": None if world == 1 else {k: runner.info()[k] for k in ("boundary_leaves", "interior_leaves", "ghost_leaves", "peers", "halo_peers", "sweeps_per_exchange",
                                                                                "bytes_sent", "messages_sent", "exchanges")},
                "verified": "single GPU path (tests/ tie it to the oracle)" if world == 1 else runner.verified_note,
                "ghosts": ghosts_note,
                "parallelism": "single GPU" if world == 1 else (
                    (f"one domain in {world} slabs of leaves (axis {runner.rank_obj.partition_axis}; -1 = contiguous ranges of the leaf order)" if args.partition else f"x-slab leaf partition over {world} ranks")
                    + ", halo transport: " + runner.transport_note
                    + (" -- ALL RANKS SHARE ONE GPU (builder's check, not a scaling figure)" if args.share_one_gpu else "")),
            },
            "roofline": {
                "kernel": sor_form if world == 1 else
                          "pressure loop of rank 0 INCLUDING its halo exchanges (comm-inclusive; the kernel alone is the N=1 figure)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "traffic_achieved": (traffic / (ms_launch * 1e-3) / 1e9) if (traffic and ms_launch) else None,  # GB/s of HBM-side bytes actually moved
                # the physical roofline next to the contractual one: bytes really moved (PMC) and bytes that MUST move per launch (one pass
                # over p in, p out and div = 12 B/voxel per LAUNCH, however many iterations it holds), both against the same 8 TB/s
                "frac_moved": (traffic / (ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and ms_launch) else None,
                "frac_compulsory": (BYTES_PER_VOXEL_ITER * n_vox_rank / (ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBS) if (launches and ms_launch) else None,
                "note": ("`achieved` / `frac` price the ALGORITHMIC bytes of SURVEY 8d -- one pass over p, p and div per red+black iteration, 12 B/voxel -- against the "
                         "launch time; the temporally blocked kernel does several iterations per pass (iterations_per_launch), so `frac` can exceed 1: it is an accounting "
                         "figure in the contract's unit, not HBM utilisation. `frac_compulsory` = the bytes one launch must move (12 B/voxel per LAUNCH) / time / peak; "
                         "`frac_moved` = the HBM-side bytes it really moved (PMC `traffic`) / time / peak") if iters_per_launch > 1 else None,
                # per KERNEL LAUNCH, as rocprofv3 --stats lists the kernel: a launch of the temporally blocked form holds several red+black
                # iterations (SURVEY 8d's unit of 12 B/voxel is the iteration), so its algorithmic bytes are 12 B/voxel x iterations per launch
                "algorithmic_bytes_per_launch": BYTES_PER_VOXEL_ITER * n_vox_rank * iters_per_launch,
                "ms_per_launch": ms_launch,
                "iterations_per_launch": iters_per_launch,
                "kernel_launches_per_solve": sor_launches,
                "ms_per_iteration": ms_iter,
                "algorithmic_bytes_per_iteration": BYTES_PER_VOXEL_ITER * n_vox_rank,
                "iterations_timed": launches,
                "kernels": kernels,
                "substep": {"algorithmic_bytes": sub_bytes, "ms": ms_per_step, "achieved": sub_gbs, "frac": sub_gbs / HBM_PEAK_GBS},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(origins, R, args.iterations)
    # One driver command, all curves: the default workload also runs BASELINE config 5 as one domain over the same ranks (collective: every rank takes part) -- `strong_scaling`
    # over the transport of the headline (RCCL by default) and, for N > 1, `strong_scaling_one_sided`: --transport auto semantics, i.e. the one-sided transport (every kernel
    # stores its halo into the peers' hipIpc-mapped ghost voxels, no exchanges) if it connects on this machine and reproduces three RCCL substeps bit for bit, else RCCL again.
    # The headline line must come out whatever happens to these records. An exception is reported in the record's place. A HANG (a collective that never returns on a
    # machine this path has not seen yet) blocks the main thread inside C (dist.barrier, torch.cuda.synchronize, a ctypes hns_dist_* call), where no Python signal handler
    # runs: a WATCHDOG THREAD cuts it -- those calls release the GIL. When it fires, rank 0 prints the line with everything measured so far and a note in place of
    # the record that hung, and every rank leaves the process at once without touching the GPU again (a hung collective cannot be unwound; nothing re-execs). All ranks exit 0:
    # the headline in the line is complete and valid, the abandoned record says so itself, and a non-zero exit would make the launcher discard the line.
    records = []
    if args.config == "256" and not args.partition and not args.no_strong:
        records.append(("strong_scaling", None))
        if world > 1 and not args.no_one_sided:
            records.append(("strong_scaling_one_sided", "auto"))
    if records and world > 1:
        runner.rank_obj.close()
    for key, transport in records:
        def last_words(key=key):
            if rank == 0:
                out[key] = {"error": f"the {key} record did not complete within {args.strong_timeout} s and was abandoned (watchdog thread); everything else in this line is complete"}
                sys.stdout.write(json.dumps(out) + "\n")
                sys.stdout.flush()

        dog = Watchdog(args.strong_timeout, last_words)
        try:
            rec = strong_scaling_record(args, world, rank, dt, transport)
        except Exception as e:  # noqa: BLE001
            rec = {"error": f"{type(e).__name__}: {e}"[:300]}
        dog.cancel()
        if rank == 0:
            out[key] = rec
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
