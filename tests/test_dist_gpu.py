"""GPU check of the multi-rank building blocks on ONE device: the HIP engine with ghost leaves (n_active, outside
element, pack kernel) driven in lockstep for 2 and 3 emulated ranks, exchanging through in-process tensor copies.
Owned results must be bit-identical to the single-grid device run (which the other GPU tests tie to the oracle)."""
import numpy as np
import pytest

from hnanosolver_amd import dist as HD
from hnanosolver_amd import fields

pytestmark = pytest.mark.gpu


def lockstep_exchange(solvers, field_lists):
    """What HaloExchanger.exchange does over the wire, done with device copies between the emulated ranks."""
    sends = [s.halo.pack_sends(fl) for s, fl in zip(solvers, field_lists)]
    recvs = [s.halo.recv_targets(fl) for s, fl in zip(solvers, field_lists)]
    for r, rv in enumerate(recvs):
        for q, dst in rv.items():
            dst.copy_(sends[q][r])
    for s, fl in zip(solvers, field_lists):
        s.halo.finish(fl)


@pytest.mark.parametrize("name,world", [("dense32", 2), ("plume", 3)])
def test_emulated_ranks_match_single_grid(name, world):
    import torch

    from hnanosolver_amd import api, device as D

    if name == "dense32":
        origins, R = fields.dense_leaves(32), 32
    else:
        origins, R = fields.plume_leaves(8, 1.5, 0.35), 64
    f = fields.synthetic_fields(origins, R)
    vs, dt, iters = 1.0 / R, 1.0 / 24.0, 7

    # single grid on the device
    grid = api.create_grid_from_leaves(origins, vs)
    sim = D.Sim(grid, ["density", "temperature"])
    arrays = {"vel": f["vel"].copy(), "density": f["density"].copy(), "temperature": f["temperature"].copy()}
    sim.upload(arrays)
    for _ in range(2):
        sim.core_substep(iters, dt, vs, D.current_stream())
    sim.download(arrays)

    solvers = []
    for r in range(world):
        plan = HD.make_plan(origins, world, r)
        eng = HD.HipEngine(plan.local_origins, plan.n_owned, vs)
        sol = HD.DistributedSolver(plan, eng, vs, n_scalars=2)
        loc = np.concatenate([plan.owned_global, plan.ghost_global])
        sel = (loc[:, None] * 512 + np.arange(512)[None, :]).reshape(-1)
        vel, den, tem = f["vel"][sel].copy(), f["density"][sel].copy(), f["temperature"][sel].copy()
        g0 = plan.n_owned * 512
        vel[g0:], den[g0:], tem[g0:] = 9.0, -1.0, 4.0  # stale ghosts: the first exchange must repair them
        sol.load_local(vel, [den, tem])
        solvers.append(sol)

    for _ in range(2):  # DistributedSolver.core_substep, stage by stage, all ranks in lockstep
        lockstep_exchange(solvers, [[s.u] + s.phi for s in solvers])
        for s in solvers:
            s.e.advect_vector(s.u, s.adv, dt, s.inv_dx)
        lockstep_exchange(solvers, [[s.adv] for s in solvers])
        for s in solvers:
            s.e.divergence(s.adv, s.div, s.inv_dx)
        lockstep_exchange(solvers, [[s.div] for s in solvers])
        for s in solvers:
            s.p_a.zero_()
            s.p_b.zero_()
            s._src, s._dst = s.p_a, s.p_b
        for _it in range(iters):
            exch = (_it + 1) % HD.DistributedSolver.SWEEPS_PER_EXCHANGE == 0 or _it + 1 == iters
            for s in solvers:
                s.e.rbgs_iteration(s.div, s._src, s._dst, s.vs, s.omega, include_ghosts=not exch)
            if exch:
                lockstep_exchange(solvers, [[s._dst] for s in solvers])
            for s in solvers:
                s._src, s._dst = s._dst, s._src
        for s in solvers:
            s.p = s._src
            s.e.subtract_pressure_gradient(s.adv, s.p, s.u, s.inv_dx)
        lockstep_exchange(solvers, [[s.u] for s in solvers])
        for s in solvers:
            s.e.advect_scalars(s.u, s.phi, s.phi_next, dt, s.inv_dx)
            s.phi, s.phi_next = s.phi_next, s.phi
    torch.cuda.synchronize()
    for s in solvers:
        own = s.plan.owned_global
        sel = (own[:, None] * 512 + np.arange(512)[None, :]).reshape(-1)
        u = s.owned(s.u).cpu().numpy()
        assert np.array_equal(u, arrays["vel"][sel]), f"rank {s.plan.rank} velocity"
        assert np.array_equal(s.owned(s.phi[0]).cpu().numpy(), arrays["density"][sel]), f"rank {s.plan.rank} density"
        assert np.array_equal(s.owned(s.phi[1]).cpu().numpy(), arrays["temperature"][sel]), f"rank {s.plan.rank} temperature"


def test_world_size_one_solver_equals_sim():
    import torch

    from hnanosolver_amd import api, device as D

    origins, R = fields.dense_leaves(32), 32
    f = fields.synthetic_fields(origins, R)
    vs, dt, iters = 1.0 / R, 1.0 / 24.0, 5
    plan = HD.make_plan(origins, 1, 0)
    sol = HD.DistributedSolver(plan, HD.HipEngine(plan.local_origins, plan.n_owned, vs), vs, n_scalars=1)
    sol.load_local(f["vel"], [f["density"]])
    sol.core_substep(iters, dt)
    grid = api.create_grid_from_leaves(origins, vs)
    sim = D.Sim(grid, ["density"])
    arrays = {"vel": f["vel"].copy(), "density": f["density"].copy()}
    sim.upload(arrays)
    sim.core_substep(iters, dt, vs, D.current_stream())
    sim.download(arrays)
    torch.cuda.synchronize()
    assert np.array_equal(sol.u.cpu().numpy(), arrays["vel"])
    assert np.array_equal(sol.phi[0].cpu().numpy(), arrays["density"])


def test_slab_bench_driver_single_rank():
    """bench.py's multi-GPU driver, exercised with world = 1 (its halo exchange is then a no-op)."""
    import torch

    b = HD.SlabBench(fields.dense_leaves(32), 32, 0, 1, 5, 1.0 / 24.0)
    b.step()
    b.timing_on()
    b.step()
    b.step()
    ms, launches = b.pressure_time()
    assert launches == 10 and ms > 0.0
    torch.cuda.synchronize()
    assert torch.isfinite(b.solver.u).all()


def test_partitioned_bench_driver_matches_the_slab_driver_at_world_one():
    """bench.py --partition splits ONE domain across the ranks (BASELINE.json's 1024^3-extent configuration); with one
    rank both drivers run the same substeps on the same leaves."""
    import torch

    o = fields.plume_leaves(8, 1.0, 0.3)
    a = HD.SlabBench(o, 64, 0, 1, 6, 1.0 / 24.0)
    b = HD.SlabBench(o, 64, 0, 1, 6, 1.0 / 24.0, partition=True)
    for _ in range(2):
        a.step()
        b.step()
    torch.cuda.synchronize()
    assert torch.equal(a.solver.u, b.solver.u) and torch.equal(a.solver.phi[0], b.solver.phi[0])
    assert b.plan.n_owned == len(o)
