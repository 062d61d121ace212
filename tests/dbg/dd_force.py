import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd"), os.path.join(ROOT, "scripts"), os.path.join(ROOT, "tests")]
import torch, torch.distributed as dist
import _dd_worker as W
from _comm_worker import lj_forces
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
dist.init_process_group("gloo", rank=rank, world_size=world)
import hoomd_amd as hoomd
from hoomd_amd import comm
dd = comm.DomainDecomposition(grid=(1, 1, 2))
sim, cell, lj, thermo = W.build(hoomd, dd, (16, 16, 16), 0, 0)
sim.run(0)
pd = sim._sysdef.getParticleData()
A = dd.communicator.exchange.arrays
N, ng = pd.N, pd.n_ghosts
pos = A.pos[:N + ng].cpu().numpy(); tag = A.tag[:N + ng].cpu().numpy()
F = lj._cpp_obj.getForcesAndEnergies()[:N]
L = dd.gL
# global positions via gather
snap = sim.state.get_snapshot()
gpos = snap.particles.position
Fref = lj_forces(gpos[tag[:N]], gpos, L, [True] * 3, skip_self=tag[:N])
Floc = lj_forces(pos[:N, :3], pos[:, :3], L, [True, True, False], skip_self=np.arange(N))
print(rank, "N", N, "ng", ng, "lo", dd.lo, "hi", dd.hi, "dim", cell.dimensions)
print(rank, "brute local vs global", np.abs(Floc - Fref).max())
err = np.abs(F[:, :3] - Fref).max(1)
bad = np.nonzero(err > 1e-6)[0]
print(rank, "gpu vs global: nbad", len(bad), "max", err.max())
if len(bad):
    print(rank, "z of bad", np.sort(pos[bad, 2])[:5], np.sort(pos[bad, 2])[-5:])
    nn = cell._cpp_obj.getNNeighArray(N)
    print(rank, "n_neigh mean", nn.mean(), "bad mean", nn[bad].mean())
    # which neighbours are missing for the first bad particle
    i = bad[0]
    head = cell._cpp_obj.getHeadListArray(N); nl = cell._cpp_obj.getNListArray()
    mine = set(nl[head[i]:head[i] + nn[i]].tolist())
    d = pos[:, :3] - pos[i, :3]; d[:, :2] -= np.round(d[:, :2] / L[:2]) * L[:2]
    want = set(np.nonzero(((d * d).sum(1) <= 2.9**2))[0].tolist()) - {i}
    miss = sorted(want - mine)
    print(rank, "particle", i, pos[i], "listed", len(mine), "want", len(want), "missing", miss[:10], "ghost?", [m >= N for m in miss[:10]], pos[miss[:5]])
dist.barrier()
