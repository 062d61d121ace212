"""Writes tests/golden/nve_drift_cpu_1728.npz: the CPU oracle's total energy over 1e5 NVE steps of a small LJ
liquid (12^3 sites, rho 0.84, kT 1.0, r_cut 2.5 shifted, buffer 0.4, dt 0.005), sampled every 1000 steps, together
with the initial state, so that the GPU path can be run from the same configuration over the same 1e5 steps and its
drift set against the CPU path's.  About ten minutes on one core:  python tests/golden/make_drift_fixture.py"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scripts")]
import lj_liquid  # noqa: E402
from oracle import oracle as orc  # noqa: E402

n = (12, 12, 12)
xyz, L = lj_liquid.lattice(n)
N = len(xyz)
rng = np.random.default_rng(3)
xyz = xyz + rng.normal(0, 0.05, xyz.shape)
Lv = np.array([L, L, L], dtype=float) if np.ndim(L) == 0 else np.asarray(L, dtype=float)
xyz = (xyz + Lv / 2) % Lv - Lv / 2
v = lj_liquid.velocities(N, kT=1.0, seed=3)
pos = orc.pack_pos(xyz, np.zeros(N, dtype=int))
vel = np.concatenate([v, np.ones((N, 1))], axis=1)
sim = orc.Sim(pos, vel, Lv, r_cut=2.5, r_buff=0.4, dt=0.005, shift_mode=orc.SHIFT_SHIFT)
sim.run(0)
every, total = 1000, 100000
trace = [sim.energies()]
t0 = time.time()
for k in range(total // every):
    sim.run(every)
    trace.append(sim.energies())
    if k % 10 == 9:
        print(k + 1, "k steps", round(time.time() - t0), "s", sum(trace[-1]) - sum(trace[0]), flush=True)
trace = np.array(trace)
st = sim.stats()
np.savez_compressed(os.path.join(HERE, "nve_drift_cpu_1728.npz"), position=xyz, velocity=v, L=Lv, every=every, energies=trace,
                    num_builds=st["num_builds"], r_cut=2.5, r_buff=0.4, dt=0.005, kT=1.0)
print("drift per particle", (trace[-1].sum() - trace[0].sum()) / N)
