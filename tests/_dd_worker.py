"""One rank of the GPU domain-decomposition test: `python _dd_worker.py RANK WORLD PORT GX GY GZ OUT`.
All ranks share GPU 0; the process group is gloo (device buffers are staged through the host, which
exercises everything except the RCCL transport itself).  Rank 0 writes the final global snapshot."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd"), os.path.join(ROOT, "scripts"), HERE]

import faulthandler
import signal

import torch
import torch.distributed as dist

faulthandler.register(signal.SIGUSR1, all_threads=True, chain=False)


def build_kg(hoomd, decomposition, n, steps, sorter):
    """Bead-spring melt: WCA pairs with bond exclusions + FENE bonds, chains of 8 along x."""
    import lj_liquid
    sim, cell, wca, fene, _ = lj_liquid.make_kg_simulation(hoomd, n, chain_length=8, seed=5, sorter_period=7, decomposition=decomposition)
    if not sorter:
        sim.operations.tuners.clear()
    thermo = hoomd.md.compute.ThermodynamicProperties(hoomd.filter.All())
    sim.operations.computes.append(thermo)
    return sim, cell, wca, thermo


def build_tail(hoomd, decomposition, n, steps, sorter):
    """The LJ liquid without an energy shift and with the long-range (tail) correction of energy and pressure."""
    return build(hoomd, decomposition, n, steps, sorter, tail=True)


def build(hoomd, decomposition, n, steps, sorter, tail=False):
    import lj_liquid
    snap = lj_liquid.make_snapshot(hoomd, n, kT=1.5, seed=11)
    rng = np.random.default_rng(2)
    snap.particles.position[:] += rng.normal(0, 0.05, snap.particles.position.shape)
    snap.wrap()
    sim = hoomd.Simulation(device=hoomd.device.GPU(), seed=3)
    if decomposition is None:
        sim.create_state_from_snapshot(snap)
    else:
        sim.create_state_from_snapshot(snap, domain_decomposition=decomposition)
    cell = hoomd.md.nlist.Cell(buffer=0.4)
    lj = hoomd.md.pair.LJ(nlist=cell, default_r_cut=2.5, mode="none", tail_correction=True) if tail else \
        hoomd.md.pair.LJ(nlist=cell, default_r_cut=2.5, mode="shift")
    lj.params[("A", "A")] = dict(epsilon=1.0, sigma=1.0)
    method = hoomd.md.methods.ConstantVolume(filter=hoomd.filter.All())
    sim.operations.integrator = hoomd.md.Integrator(dt=0.005, methods=[method], forces=[lj])
    if sorter:
        sim.operations.tuners[0].trigger = 7
    else:
        sim.operations.tuners.clear()
    thermo = hoomd.md.compute.ThermodynamicProperties(hoomd.filter.All())
    sim.operations.computes.append(thermo)
    return sim, cell, lj, thermo


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    grid = tuple(int(v) for v in sys.argv[4:7])
    out = sys.argv[7]
    n = tuple(int(v) for v in sys.argv[8:11])
    steps = int(sys.argv[11])
    sorter = bool(int(sys.argv[12]))
    workload = sys.argv[13] if len(sys.argv) > 13 else "lj"
    import hoomd_amd as hoomd
    dd = None
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from hoomd_amd import comm
        dd = comm.DomainDecomposition(grid=grid)
    torch.cuda.set_device(0)
    sim, cell, lj, thermo = {"kg": build_kg, "tail": build_tail}.get(workload, build)(hoomd, dd, n, steps, sorter)
    sim.run(0)
    if dd is not None:
        # a stuck run says which stream holds unfinished work before the launcher's limit ends it
        import threading
        done = threading.Event()

        def watchdog():
            if not done.wait(25.0):
                side = dd.communicator.transport.side
                print(f"[watchdog rank {rank}] step {sim.timestep}: default stream idle={torch.cuda.default_stream().query()}, "
                      f"exchange stream idle={side[1].query() if side else None}", flush=True)
        threading.Thread(target=watchdog, daemon=True).start()
    e0 = (thermo.kinetic_energy, thermo.potential_energy)
    sim.run(steps)
    e1 = (thermo.kinetic_energy, thermo.potential_energy)
    if dd is not None:
        done.set()
    s = sim.state.get_snapshot()
    dof = thermo.degrees_of_freedom  # collective under domain decomposition: every rank asks
    pair_energy = lj.energy          # likewise (summed over ranks, hoomd/ForceCompute.cc:196-226)
    pressure = thermo.pressure
    stats = dict(N_local=int(sim._sysdef.getParticleData().N), n_ghosts=int(sim._sysdef.getParticleData().n_ghosts), builds=int(cell.num_builds),
                 overlapped=int(cell._cpp_obj.num_overlapped), additional_energy=float(lj.additional_energy))
    if dd is not None:
        ex = dd.communicator
        stats.update(migrated_out=int(ex.n_migrated_out), migrated_in=int(ex.n_migrated_in), vote_exchanges=int(ex.n_vote_exchanges),
                     fixed_migrate=int(ex.fixed_migrate), slot_rows=int(ex.migrate_slot_rows), count_exchanges=int(ex.n_count_exchanges),
                     ghost_slot_overflows=int(ex.n_ghost_slot_overflows))
        allstats = [None] * world
        dist.all_gather_object(allstats, stats)
    else:
        allstats = [stats]
    if rank == 0:
        np.savez(out, position=s.particles.position, velocity=s.particles.velocity, image=s.particles.image, e0=e0, e1=e1,
                 dof=dof, pair_energy=pair_energy, pressure=pressure, additional_energy=[a["additional_energy"] for a in allstats], N_local=[a["N_local"] for a in allstats], n_ghosts=[a["n_ghosts"] for a in allstats],
                 builds=[a["builds"] for a in allstats], migrated=[a.get("migrated_out", 0) for a in allstats],
                 vote_exchanges=[a.get("vote_exchanges", 0) for a in allstats], overlapped=[a["overlapped"] for a in allstats],
                 fixed_migrate=[a.get("fixed_migrate", 0) for a in allstats], slot_rows=[a.get("slot_rows", 0) for a in allstats],
                 count_exchanges=[a.get("count_exchanges", 0) for a in allstats],
                 ghost_slot_overflows=[a.get("ghost_slot_overflows", 0) for a in allstats])
    if dd is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
