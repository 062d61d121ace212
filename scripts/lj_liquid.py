"""LJ-liquid workload builder shared by bench.py and the tests: simple-cubic lattice at rho = 0.84,
Gaussian velocities at kT = 1.5 (NumPy default_rng(1234)), zero net momentum."""
import numpy as np


def lattice(n, rho=0.84):
    """n = (nx, ny, nz) lattice sites; returns positions (N,3) and box lengths."""
    n = (n, n, n) if np.isscalar(n) else tuple(n)
    a = (1.0 / rho) ** (1.0 / 3.0)
    L = np.array(n, dtype=np.float64) * a
    gx, gy, gz = [(np.arange(k) + 0.5) * a - l / 2 for k, l in zip(n, L)]
    # x fastest in memory: neighbouring tags are neighbouring sites
    z, y, x = np.meshgrid(gz, gy, gx, indexing="ij")
    return np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1), L


def velocities(N, kT=1.5, seed=1234):
    rng = np.random.default_rng(seed)
    v = rng.normal(0.0, np.sqrt(kT), (N, 3))
    v -= v.mean(axis=0)
    v *= np.sqrt(kT * 3 * (N - 1) / (v * v).sum())  # exactly kT on 3(N-1) degrees of freedom
    return v


def make_snapshot(hoomd, n, rho=0.84, kT=1.5, seed=1234):
    pos, L = lattice(n, rho)
    snap = hoomd.Snapshot()
    snap.configuration.box = [L[0], L[1], L[2], 0, 0, 0]
    snap.particles.N = len(pos)
    snap.particles.types = ["A"]
    snap.particles.position[:] = pos
    snap.particles.velocity[:] = velocities(len(pos), kT, seed)
    return snap


def make_simulation(hoomd, n, rho=0.84, kT=1.5, r_cut=2.5, buffer=0.4, dt=0.005, seed=1234, thermostat=None, sorter_period=200,
                    decomposition=None, mode="none"):
    dev = hoomd.device.GPU()
    sim = hoomd.Simulation(device=dev, seed=seed)
    sim.create_state_from_snapshot(make_snapshot(hoomd, n, rho, kT, seed), domain_decomposition=decomposition) if decomposition is not None \
        else sim.create_state_from_snapshot(make_snapshot(hoomd, n, rho, kT, seed))
    cell = hoomd.md.nlist.Cell(buffer=buffer)
    lj = hoomd.md.pair.LJ(nlist=cell, default_r_cut=r_cut, mode=mode)
    lj.params[("A", "A")] = dict(epsilon=1.0, sigma=1.0)
    method = hoomd.md.methods.ConstantVolume(filter=hoomd.filter.All(), thermostat=thermostat)
    sim.operations.integrator = hoomd.md.Integrator(dt=dt, methods=[method], forces=[lj])
    sim.operations.tuners[0].trigger = sorter_period
    return sim, cell, lj, method


def make_ka_simulation(hoomd, n, rho=1.2, kT=1.0, buffer=0.4, dt=0.005, seed=1234, thermostat=None, sorter_period=200, decomposition=None):
    """Kob-Andersen 80:20 binary LJ mixture: types by rng.random(N) < 0.8, eps AA/AB/BB = 1/1.5/0.5,
    sigma = 1/0.8/0.88, r_cut = 2.5 sigma_ij."""
    snap = make_snapshot(hoomd, n, rho, kT, seed)
    N = snap.particles.N
    snap.particles.types = ["A", "B"]
    snap.particles.typeid[:] = (np.random.default_rng(seed + 1).random(N) >= 0.8).astype(np.uint32)
    sim = hoomd.Simulation(device=hoomd.device.GPU(), seed=seed)
    if decomposition is not None:
        sim.create_state_from_snapshot(snap, domain_decomposition=decomposition)
    else:
        sim.create_state_from_snapshot(snap)
    cell = hoomd.md.nlist.Cell(buffer=buffer)
    lj = hoomd.md.pair.LJ(nlist=cell, default_r_cut=2.5)
    for pair, eps, sig in ((("A", "A"), 1.0, 1.0), (("A", "B"), 1.5, 0.8), (("B", "B"), 0.5, 0.88)):
        lj.params[pair] = dict(epsilon=eps, sigma=sig)
        lj.r_cut[pair] = 2.5 * sig
    method = hoomd.md.methods.ConstantVolume(filter=hoomd.filter.All(), thermostat=thermostat)
    sim.operations.integrator = hoomd.md.Integrator(dt=dt, methods=[method], forces=[lj])
    sim.operations.tuners[0].trigger = sorter_period
    return sim, cell, lj, method


def make_kg_snapshot(hoomd, n, chain_length=32, kT=1.0, seed=1234, jitter=0.02, across=1.05):
    """Kremer-Grest melt start: linear chains laid along x on an n[0] x n[1] x n[2] lattice (spacing 0.97 along the
    chains, `across` between them: 1.05 gives rho = 0.935, 1.1013 rho = 0.85), every chain_length consecutive sites of
    a row bonded (rows wrap around the periodic box, so n[0] must be a multiple of chain_length)."""
    if isinstance(n, int):
        n = (n, n, n)
    if n[0] % chain_length:
        raise ValueError("n[0] must be a multiple of chain_length")
    a = np.array([0.97, across, across])
    L = np.array(n) * a
    rng = np.random.default_rng(seed)
    g = [(np.arange(k) + 0.5) * s - l / 2 for k, s, l in zip(n, a, L)]
    z, y, x = np.meshgrid(g[2], g[1], g[0], indexing="ij")
    xyz = np.stack([x.ravel(), y.ravel(), z.ravel()], 1) + rng.normal(0, jitter, (x.size, 3))
    N = len(xyz)
    snap = hoomd.Snapshot()
    snap.configuration.box = [L[0], L[1], L[2], 0, 0, 0]
    snap.particles.N = N
    snap.particles.types = ["A"]
    snap.particles.position[:] = xyz
    v = rng.normal(0, np.sqrt(kT), (N, 3))
    snap.particles.velocity[:] = v - v.mean(0)
    idx = np.arange(N).reshape(-1, n[0])                      # rows along x
    left = idx.reshape(-1, chain_length)[:, :-1].ravel()       # every site but the last of its chain
    snap.bonds.N = len(left)
    snap.bonds.types = ["polymer"]
    snap.bonds.group[:] = np.stack([left, left + 1], 1)
    snap.wrap()
    return snap


def make_kg_simulation(hoomd, n, chain_length=32, kT=1.0, buffer=0.4, dt=0.005, seed=1234, thermostat=None, sorter_period=200, decomposition=None,
                       across=1.05):
    """WCA pair force (LJ cut at 2^(1/6), shifted) + FENE/WCA bonds (k = 30, r0 = 1.5), bonded pairs excluded from
    the pair force: the Kremer-Grest bead-spring melt."""
    snap = make_kg_snapshot(hoomd, n, chain_length, kT, seed, across=across)
    sim = hoomd.Simulation(device=hoomd.device.GPU(), seed=seed)
    if decomposition is not None:
        sim.create_state_from_snapshot(snap, domain_decomposition=decomposition)
    else:
        sim.create_state_from_snapshot(snap)
    cell = hoomd.md.nlist.Cell(buffer=buffer, exclusions=("bond",))
    wca = hoomd.md.pair.LJ(nlist=cell, default_r_cut=2.0 ** (1.0 / 6.0), mode="shift")
    wca.params[("A", "A")] = dict(epsilon=1.0, sigma=1.0)
    fene = hoomd.md.bond.FENEWCA()
    fene.params["polymer"] = dict(k=30.0, r0=1.5, epsilon=1.0, sigma=1.0, delta=0.0)
    method = hoomd.md.methods.ConstantVolume(filter=hoomd.filter.All(), thermostat=thermostat)
    sim.operations.integrator = hoomd.md.Integrator(dt=dt, methods=[method], forces=[wca, fene])
    sim.operations.tuners[0].trigger = sorter_period
    return sim, cell, wca, fene, method
