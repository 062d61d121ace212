"""One rank of the CPU baseline: the reference's CPU path (the oracle's restatement: binned half neighbor list with
ghost layer, third-law pair loop, velocity Verlet, distance check) on one sub-box of a domain decomposition, one
process per core, halos and migration over gloo -- the shape of the reference's MPI mode (hoomd/Communicator.cc:1510-1569:
migrate and exchange ghosts when any rank's distance check fires, update ghost positions otherwise).

    python cpu_decomposed.py RANK WORLD PORT STATE.npz STEPS OUT.json

STATE.npz holds the global state to start from (position (N,3) wrapped into the box, velocity (N,3), box lengths L);
rank 0 writes OUT.json with the wall time of STEPS steps (maximum over ranks, barriers on both sides).
Measurement infrastructure: the simulation product never imports this file."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd")]


def load_exchange_logic():
    """hoomd_amd/comm.py without the package around it: the migration / ghost-exchange logic on CPU tensors is plain
    torch code, while importing the package loads the GPU libraries (and opens the GPU on a machine that has one, in
    every one of these CPU processes).  The one name comm.py takes from the extension module at import time is the
    base class of its GPU communicator, which is not used here."""
    import importlib.util
    import types
    pkg_dir = os.path.join(ROOT, "hoomd-blue_amd", "hoomd_amd")
    pkg = types.ModuleType("hoomd_amd")
    pkg.__path__ = [pkg_dir]
    ext = types.ModuleType("hoomd_amd._hoomd_amd")
    ext.Communicator = object
    sys.modules.setdefault("hoomd_amd", pkg)
    sys.modules.setdefault("hoomd_amd._hoomd_amd", ext)
    spec = importlib.util.spec_from_file_location("hoomd_amd.comm", os.path.join(pkg_dir, "comm.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["hoomd_amd.comm"] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    state, steps, out = sys.argv[4], int(sys.argv[5]), sys.argv[6]
    r_cut, r_buff, dt = float(os.environ.get("CPU_DD_RCUT", "2.5")), float(os.environ.get("CPU_DD_RBUFF", "0.4")), 0.005
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = load_exchange_logic()
    from oracle import oracle as orc

    s = np.load(state)
    gpos, gvel, L = s["position"], s["velocity"], s["L"]
    N_global = len(gpos)
    dd = comm.DomainDecomposition(grid=None)
    dd._set_box(L, -L / 2)
    mine = np.nonzero(dd.owner_of(gpos) == rank)[0]
    pos4 = np.concatenate([gpos[mine], np.zeros((len(mine), 1))], 1)
    vel4 = np.concatenate([gvel[mine], np.ones((len(mine), 1))], 1)
    del gpos, gvel
    A = comm.HostArrays(pos4, vel4, mine, N_global)
    ex = comm.ParticleExchange(dd, A)
    r_ghost = r_cut + r_buff
    ghost_width = tuple(r_ghost if d else 0.0 for d in dd.decomposed)
    periodic = tuple(0 if d else 1 for d in dd.decomposed)
    box = orc.Box.from_lo_hi(list(dd.lo), list(dd.hi), periodic=periodic)
    params, rcutsq = [orc.lj_params(1.0, 1.0)], [[r_cut * r_cut]]
    rcut_max = np.array([r_cut])
    st = dict(nl=None, last=None, builds=0, migrated=0)

    def rebuild():
        ex.migrate()
        ex.exchange_ghosts(r_ghost)
        n, ng = A.N, A.n_ghosts
        pos = A.pos[:n + ng].numpy()
        st["nl"] = orc.nlist_binned(pos, n, box, [[r_cut]], r_buff, storage_full=False, n_ghost=ng, ghost_width=ghost_width)
        st["last"] = pos[:n].copy()
        st["builds"] += 1

    def forces():
        n, ng = A.N, A.n_ghosts
        nl = st["nl"]
        F, _ = orc.pair_forces(orc.EV_LJ, A.pos[:n + ng].numpy(), n, box, nl["n_neigh"], nl["nlist"], nl["head"], params, rcutsq,
                               third_law=True, compute_virial=False, n_total=n + ng)
        return F[:n]

    rebuild()
    F = forces()
    A.accel[:A.N] = torch.from_numpy(orc.compute_accel(A.vel[:A.N].numpy(), F))

    def step():
        n = A.N
        orc.step_one(A.pos[:n].numpy(), A.vel[:n].numpy(), A.accel[:n].numpy(), A.image[:n].numpy(), box, dt)
        stale = orc.distance_check(A.pos[:n].numpy(), st["last"], n, box, rcut_max, r_buff)
        if ex.any_rank(stale):
            rebuild()
        else:
            ex.update_ghosts()
        n = A.N
        F = forces()
        orc.step_two(A.vel[:n].numpy(), A.accel[:n].numpy(), F, dt)
        return F

    for _ in range(2):  # untimed: first-touch of the buffers
        step()
    b0 = st["builds"]
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        F = step()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # conserved quantities as a sanity check of the run that was timed
    ke, pe = orc.thermo(A.vel[:A.N].numpy(), F)
    e = torch.tensor([ke, pe, float(A.N), float(A.n_ghosts)], dtype=torch.float64)
    dist.all_reduce(e)
    if rank == 0:
        with open(out, "w") as f:
            json.dump(dict(seconds=float(t.item()), steps=steps, N=int(e[2].item()), ranks=world, grid=list(dd.grid), rebuilds=st["builds"] - b0,
                           kinetic_energy=float(e[0].item()), potential_energy=float(e[1].item()), mean_ghosts_per_rank=float(e[3].item()) / world,
                           migrated=int(ex.n_migrated_out)), f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
