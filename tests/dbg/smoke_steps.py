"""The smoke configuration (hot, rough start) step by step against the oracle; with and without the tile path."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd")]
import hoomd_amd as hoomd
from oracle import oracle as orc
n, rho = 12, 0.84
a = (1.0 / rho) ** (1.0 / 3.0)
L = n * a
g = (np.arange(n) + 0.5) * a - L / 2
xyz = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
xyz += np.random.default_rng(1).normal(0, 0.05, xyz.shape)
xyz = (xyz + L / 2) % L - L / 2
N = len(xyz)
vel = np.random.default_rng(2).normal(0, 1.0, (N, 3))
for tiles in (True, False):
    snap = hoomd.Snapshot()
    snap.configuration.box = [L, L, L, 0, 0, 0]
    snap.particles.N = N
    snap.particles.types = ["A"]
    snap.particles.position[:] = xyz
    snap.particles.velocity[:] = vel
    sim = hoomd.Simulation(device=hoomd.device.GPU(), seed=1)
    sim.create_state_from_snapshot(snap)
    cell = hoomd.md.nlist.Cell(buffer=0.4)
    lj = hoomd.md.pair.LJ(nlist=cell, default_r_cut=2.5)
    lj.params[("A", "A")] = dict(epsilon=1.0, sigma=1.0)
    sim.operations.integrator = hoomd.md.Integrator(dt=0.005, methods=[hoomd.md.methods.ConstantVolume(hoomd.filter.All())], forces=[lj])
    sim.run(0)
    cell._cpp_obj.tiles = tiles
    ref = orc.Sim(orc.pack_pos(xyz, np.zeros(N, int)), np.concatenate([vel, np.ones((N, 1))], 1), np.array([L, L, L]), r_cut=2.5, r_buff=0.4, dt=0.005)
    ref.run(0)
    for step in range(1, 21):
        try:
            sim.run(1)
        except RuntimeError as e:
            print("tiles", tiles, "step", step, "FAILED", str(e).splitlines()[0]); break
        ref.run(1)
        p_ref, v_ref, _, f_ref = ref.get()
        s = sim.state.get_snapshot()
        dp = np.abs(s.particles.position - p_ref[:, :3]); dp = np.minimum(dp, L - dp).max()
        dv = np.abs(s.particles.velocity - v_ref[:, :3]).max()
        print("tiles", tiles, "step", step, "dpos %.2e dvel %.2e" % (dp, dv), "builds", cell.num_builds, ref.stats()["num_builds"], "vmax %.1f" % np.abs(v_ref[:, :3]).max(), flush=True)
