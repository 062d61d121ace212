"""Multi-rank tests of the domain decomposition host logic on CPU tensors (gloo, world size 2 and 4):
ownership, migration (including across the periodic boundary and to non-adjacent domains), the ghost
layer against a brute-force expectation, forces from owned + ghost particles against the global
minimum-image result, the per-step ghost update and the rebuild vote."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(script, world, extra, timeout=240, env=None):
    """One process per rank.  A rank that fails ends the run at once (the others would wait for it in a
    collective until the time limit) and its output is what the assertion shows."""
    import tempfile
    import time
    port = free_port()
    logs = [tempfile.TemporaryFile() for _ in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, script), str(r), str(world), str(port)] + [str(e) for e in extra],
                              stdout=logs[r], stderr=subprocess.STDOUT, env=env) for r in range(world)]
    deadline = time.monotonic() + timeout
    failed = None
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                failed = next(r for r, c in enumerate(codes) if c is None)
                # workers that registered faulthandler on SIGUSR1 print where they are stuck
                import signal
                for p in procs:
                    if p.poll() is None:
                        p.send_signal(signal.SIGUSR1)
                time.sleep(2.0)
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    outs = []
    for f in logs:
        f.seek(0)
        outs.append(f.read().decode(errors="replace"))
        f.close()
    if failed is not None:
        others = "".join(f"\n---- rank {r} ----\n{o[-1500:]}" for r, o in enumerate(outs) if r != failed)
        raise AssertionError(f"rank {failed} failed or timed out (exit code {procs[failed].returncode}):\n{outs[failed][-4000:]}{others}")
    return outs


@pytest.mark.parametrize("world,grid", [(2, (1, 1, 2)), (2, (2, 1, 1)), (4, (2, 2, 1)), (4, (0, 0, 0))])
def test_migration_and_halo_exchange(world, grid):
    launch("_comm_worker.py", world, grid)


def test_grid_choice_minimises_surface():
    """hoomd/DomainDecomposition.cc:327-408."""
    from hoomd_amd import comm
    assert comm.find_grid(8, [271.32, 271.32, 135.66]) == (2, 2, 2)
    assert comm.find_grid(8, [100.0, 100.0, 100.0]) == (2, 2, 2)
    assert comm.find_grid(2, [135.66, 135.66, 135.66 * 2]) == (1, 1, 2)
    assert comm.find_grid(4, [10.0, 10.0, 40.0]) == (1, 1, 4)
    assert comm.find_grid(4, [10.0, 10.0, 1.0], twod=True) == (2, 2, 1)
    assert comm.find_grid(6, [30.0, 20.0, 10.0]) == (3, 2, 1)
    assert comm.find_grid(1, [1.0, 1.0, 1.0]) == (1, 1, 1)
    for n in (2, 3, 4, 6, 8, 12):
        assert int(np.prod(comm.find_grid(n, [3.0, 2.0, 1.0]))) == n


@pytest.mark.parametrize("world", [2, 4])
def test_decomposed_cpu_baseline_matches_the_single_domain_cpu_path(tmp_path, world):
    """scripts/cpu_decomposed.py (the CPU baseline of the benchmark: the oracle's per-domain step driven by the gloo
    exchange) against the oracle's single-domain run from the same state: same energies after the same number of
    steps, with rebuilds and migration on the way."""
    import json
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "scripts"))
    import lj_liquid
    from oracle import oracle as orc
    n = (14, 14, 14)
    pos, L = lj_liquid.lattice(n)
    rng = np.random.default_rng(3)
    pos = pos + rng.normal(0, 0.08, pos.shape)
    pos -= np.floor((pos + L / 2) / L) * L
    N = len(pos)
    vel = lj_liquid.velocities(N, 1.5, 7)
    state = str(tmp_path / "state.npz")
    np.savez(state, position=pos, velocity=vel, L=L)
    steps = 40
    outs = []
    out = str(tmp_path / "out.json")
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(os.path.dirname(HERE), "scripts", "cpu_decomposed.py"), str(r), str(world), str(port), state,
                               str(steps), out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    for p in procs:
        o, _ = p.communicate(timeout=240)
        outs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-2000:] for o in outs)
    res = json.load(open(out))
    assert res["N"] == N and res["ranks"] == world and res["rebuilds"] >= 3 and res["mean_ghosts_per_rank"] > 0
    ref = orc.Sim(orc.pack_pos(pos, np.zeros(N, dtype=int)), np.concatenate([vel, np.ones((N, 1))], 1), L)
    ref.run(steps + 2)  # the worker runs two untimed steps first
    ke, pe = ref.energies()
    assert res["kinetic_energy"] == pytest.approx(ke, rel=1e-9) and res["potential_energy"] == pytest.approx(pe, rel=1e-9)
