"""Domain decomposition end to end on the GPU: 2, 4 and 8 ranks (all on GPU 0, gloo transport) against
the single-domain run of the same system.  Trajectories agree to rounding (force sums run in a
different neighbor order), energies and the rebuild cadence agree, particles migrate."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from test_comm_gloo import launch  # noqa: E402


@pytest.mark.parametrize("world,grid,sorter,n,overlap", [
    (2, (1, 1, 2), 0, (16, 16, 16), False), (4, (2, 2, 1), 1, (16, 16, 16), False), (8, (2, 2, 2), 1, (20, 20, 20), False),
    # sub-boxes long enough along the decomposed axes to have tile blocks that see no ghosts: their forces are
    # computed on a second stream while the halo exchange runs
    (2, (1, 1, 2), 1, (10, 10, 36), True), (4, (2, 2, 1), 0, (36, 36, 10), True), (8, (2, 2, 2), 1, (36, 36, 36), True)])
def test_decomposed_run_matches_single_domain(tmp_path, world, grid, sorter, n, overlap):
    steps = 60
    ref = str(tmp_path / "ref.npz")
    out = str(tmp_path / "dd.npz")
    launch("_dd_worker.py", 1, list((1, 1, 1)) + [ref] + list(n) + [steps, sorter], timeout=60)
    # one case starts with migration messages of a single row: the headers tell every rank that it was not
    # enough, and all of them send again with room for the largest message
    env = dict(os.environ, HOOMD_AMD_MIGRATE_SLOT_ROWS="1") if world == 4 and not overlap else None
    # Ranks that share a GPU keep the exchange stream at normal priority (chosen by the communicator: a high-priority
    # queue of one process that waits on a normal-priority queue can keep the queues of the other processes off the
    # hardware, two processes are enough for that).  The production priority runs in the one-rank test below.
    # Ghost messages are sized from the previous exchange of the same pair of ranks.  One case gives them no spare room
    # at all: whenever a message grows, the headers tell every rank, and all of them exchange counts first that time.
    tight = world == 8 and not overlap
    if tight:
        env = dict(os.environ, HOOMD_AMD_GHOST_SLOT_ROOM="0,0")
    launch("_dd_worker.py", world, list(grid) + [out] + list(n) + [steps, sorter], timeout=60, env=env)
    a, b = np.load(ref), np.load(out)
    assert sum(b["N_local"]) == n[0] * n[1] * n[2] and all(g > 0 for g in b["n_ghosts"])
    assert sum(b["migrated"]) > 0
    assert min(b["fixed_migrate"]) == 1
    # counts are exchanged for the first ghost layer of the run and after a message that was too small, never otherwise
    assert len(set(b["ghost_slot_overflows"])) == 1 and len(set(b["count_exchanges"])) == 1
    assert b["count_exchanges"][0] == 1 + b["ghost_slot_overflows"][0] and b["count_exchanges"][0] < min(b["builds"])
    assert (b["ghost_slot_overflows"][0] > 0) == tight
    if env is not None:
        assert min(b["slot_rows"]) > 1 and len(set(b["slot_rows"])) == 1
    # every rank is every other rank's neighbor in these grids: the rebuild vote rode in the halo exchange
    assert min(b["vote_exchanges"]) > 0
    if overlap:
        assert min(b["overlapped"]) > steps // 2
    else:
        assert max(b["overlapped"]) == 0
    assert np.allclose(a["e0"], b["e0"], rtol=1e-11)
    assert np.allclose(a["e1"], b["e1"], rtol=1e-7)
    assert a["dof"] == b["dof"]
    L = np.array(n) * (1 / 0.84) ** (1 / 3)
    ua = a["position"] + a["image"] * L
    ub = b["position"] + b["image"] * L
    assert np.abs(ua - ub).max() < 1e-7
    assert np.abs(a["velocity"] - b["velocity"]).max() < 1e-6
    if not sorter:
        assert len(set(b["builds"])) == 1 and b["builds"][0] == a["builds"][0]
    # ... and the CPU path itself (half list + third law on one domain) from the same start: the decomposed GPU
    # trajectory is the reference algorithm's trajectory, not merely the single-domain GPU run's
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "scripts"))
    import hoomd_amd
    import lj_liquid
    from oracle import oracle as orc
    snap = lj_liquid.make_snapshot(hoomd_amd, n, kT=1.5, seed=11)
    snap.particles.position[:] += np.random.default_rng(2).normal(0, 0.05, snap.particles.position.shape)
    snap.wrap()
    vel = np.concatenate([snap.particles.velocity, snap.particles.mass[:, None]], axis=1)
    o = orc.Sim(orc.pack_pos(snap.particles.position, snap.particles.typeid), vel, snap.configuration.box[:3], shift_mode=orc.SHIFT_SHIFT)
    o.run(0)
    assert np.allclose(o.energies(), b["e0"], rtol=1e-10)
    o.run(steps)
    op, ov, oi, _ = o.get()
    # the snapshot built here carries the images of its own wrap(); the workers start from the same wrapped state
    uo = op[:, :3] + (oi + snap.particles.image) * L
    assert np.abs(uo - ub).max() < 1e-7
    assert np.abs(ov[:, :3] - b["velocity"]).max() < 1e-6
    assert np.allclose(o.energies(), b["e1"], rtol=1e-7)
    if not sorter:
        assert o.stats()["num_builds"] == b["builds"][0]


@pytest.mark.parametrize("world,grid,sorter", [(2, (2, 1, 1), 0), (4, (2, 2, 1), 1), (8, (2, 2, 2), 1)])
def test_bonded_melt_under_decomposition_matches_single_domain(tmp_path, world, grid, sorter):
    """Bead-spring melt (WCA + FENE, bonded pairs excluded) on 2, 4 and 8 ranks against the single-domain run of the
    same system: bonds and exclusions that cross sub-box faces, chains that migrate.  (The single-domain run is the
    one tests/test_gpu_bonds_md.py checks against the CPU oracle.)"""
    n, steps = (16, 12, 12), 240  # long enough for beads to cross sub-box faces from the lattice start
    ref, out = str(tmp_path / "ref.npz"), str(tmp_path / "dd.npz")
    launch("_dd_worker.py", 1, [1, 1, 1, ref] + list(n) + [steps, sorter, "kg"], timeout=60)
    launch("_dd_worker.py", world, list(grid) + [out] + list(n) + [steps, sorter, "kg"], timeout=90)
    a, b = np.load(ref), np.load(out)
    assert sum(b["N_local"]) == n[0] * n[1] * n[2] and all(g > 0 for g in b["n_ghosts"])
    assert np.allclose(a["e0"], b["e0"], rtol=1e-11)
    assert np.allclose(a["e1"], b["e1"], rtol=1e-7)
    L = np.array(n) * np.array([0.97, 1.05, 1.05])
    ua, ub = a["position"] + a["image"] * L, b["position"] + b["image"] * L
    assert np.abs(ua - ub).max() < 1e-6
    assert np.abs(a["velocity"] - b["velocity"]).max() < 1e-5
    assert sum(b["migrated"]) > 0


def test_tail_correction_under_decomposition_uses_global_counts(tmp_path):
    """Long-range corrections (hoomd/md/PotentialPair.h:283-337): from the global particle counts, added once -- the
    decomposed run reports the single-domain energy and pressure, and only rank 0 carries the correction."""
    n, steps = (16, 16, 16), 20
    ref, out = str(tmp_path / "ref.npz"), str(tmp_path / "dd.npz")
    launch("_dd_worker.py", 1, [1, 1, 1, ref] + list(n) + [steps, 0, "tail"], timeout=60)
    launch("_dd_worker.py", 2, [1, 1, 2, out] + list(n) + [steps, 0, "tail"], timeout=60)
    a, b = np.load(ref), np.load(out)
    N = n[0] * n[1] * n[2]
    V = N / 0.84
    # closed form for one type (hoomd/md/EvaluatorPairLJ.h:213-248): 2 pi N rho * 4 eps (sigma^12 / (9 rc^9) - sigma^6 / (3 rc^3))
    expected = 2 * np.pi * N * (N / V) * 4 * (1 / (9 * 2.5 ** 9) - 1 / (3 * 2.5 ** 3))
    assert np.isclose(a["additional_energy"][0], expected, rtol=1e-12)
    assert np.isclose(b["additional_energy"][0], expected, rtol=1e-12) and b["additional_energy"][1] == 0.0
    assert np.isclose(a["pair_energy"], b["pair_energy"], rtol=1e-9)
    assert np.allclose(a["e0"], b["e0"], rtol=1e-11) and np.allclose(a["e1"], b["e1"], rtol=1e-7)
    assert np.isclose(a["pressure"], b["pressure"], rtol=1e-7)


def test_one_rank_decomposition_through_torch_distributed_when_asked():
    """HOOMD_AMD_TRANSPORT=torch: the collectives of the same run go through torch.distributed (the path every gloo
    test uses, here on the nccl backend)."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_PORT="29534", HOOMD_AMD_TRANSPORT="torch")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dd", "--lattice", "40x40x40", "--equil", "400",
                        "--warmup", "20", "--steps", "200", "--no-cpu-baseline"], capture_output=True, text=True, timeout=140, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])["config"]["decomposition"]
    assert d["backend"] == "nccl" and d["transport"] == "torch.distributed nccl"


@pytest.mark.parametrize("priority", ["normal", "high"])
def test_one_rank_decomposition_over_rccl(priority):
    """The production transport, as far as one GPU allows: one rank (nothing to exchange: the migration, ghost and vote
    collectives run empty), RCCL collectives on the exchange stream -- at normal priority (the default) and at high
    priority (HOOMD_AMD_COMM_PRIORITY=high)."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_PORT="29533")
    env.pop("HOOMD_AMD_COMM_PRIORITY", None)
    if priority == "high":
        env["HOOMD_AMD_COMM_PRIORITY"] = "high"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dd", "--lattice", "40x40x40", "--equil", "400",
                        "--warmup", "20", "--steps", "200", "--no-cpu-baseline"], capture_output=True, text=True, timeout=140, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    d = out["config"]["decomposition"]
    assert d["backend"] == "nccl" and d["exchange_stream_high_priority"] is (priority == "high")
    # RCCL is called from the C++ communicator, through the copy of the library the process already holds
    assert d["transport"].startswith("rccl (C++") and "already loaded" in d["transport"]
    # at normal priority a stream that runs beside the default stream was found by the probe
    assert (d["exchange_stream_kept_after_probe"] >= 0) == (priority == "normal")
    assert out["config"]["rebuilds_in_timed_region"] > 10 and 60 < out["config"]["mean_listed_neighbors"] < 110
    assert out["value"] > 0


@pytest.mark.parametrize("scaling,lattice,total", [("weak", "24x24x24", 2 * 24 ** 3), ("strong", "24x48x24", 24 * 48 * 24)])
def test_bench_starts_its_own_ranks(scaling, lattice, total):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the parent starts two fresh ranks (here sharing
    GPU 0 over gloo, the rehearsal form), relays rank 0's line; --scaling decides whether the lattice is per GPU or the
    whole system's."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--scaling", scaling, "--lattice", lattice,
                        "--equil", "300", "--warmup", "20", "--steps", "100", "--no-cpu-baseline", "--launch-timeout", "150"],
                       capture_output=True, text=True, timeout=170, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["config"]["N"] == total
    assert out["config"]["domain_grid"] == ([1, 1, 2] if scaling == "weak" else [1, 2, 1])
    d = out["config"]["decomposition"]
    assert d["backend"] == "gloo" and d["ghosts_rank0"] > 0 and d["exchange_stream_high_priority"] is False
    assert out["config"]["rebuilds_in_timed_region"] > 5 and out["value"] > 0


def test_rccl_transport_issues_send_and_receive_from_cpp():
    """The C++ transport's ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on a created stream, as far as one GPU
    allows: a one-rank communicator whose own rows are sent through RCCL to itself (in a run they are copied), with
    unequal row sizes; the device-to-device copy path; the reductions of a one-rank group."""
    import torch
    from hoomd_amd import _hoomd_amd as cpp
    t = cpp.RcclTransport(cpp.RcclTransport.unique_id(), 0, 1)
    assert "librccl" in cpp.RcclTransport.library_origin()
    s = torch.cuda.Stream()
    for row_bytes, rows in ((32, 1000), (144, 37), (4, 1)):
        src = torch.randint(0, 255, (rows * row_bytes,), dtype=torch.uint8, device="cuda")
        for through in (True, False):
            dst = torch.zeros_like(src)
            t.self_through_rccl = through
            torch.cuda.synchronize()
            t.allToAll(src.data_ptr(), [rows], dst.data_ptr(), [rows], row_bytes, s.cuda_stream)
            s.synchronize()
            assert torch.equal(src, dst), (row_bytes, rows, through)
    t.allToAll(0, [0], 0, [0], 32, s.cuda_stream)  # nothing to move: no call at all
    w = torch.tensor([7], dtype=torch.int32, device="cuda")
    t.allReduceMaxWord(w.data_ptr(), s.cuda_stream)
    s.synchronize()
    assert int(w.item()) == 7 and t.anyRank(True) is True and t.anyRank(False) is False
    assert t.allReduceSumHost([1.5, -2.0]) == [1.5, -2.0]
    with pytest.raises(ValueError):
        t.allToAll(src.data_ptr(), [1, 2], dst.data_ptr(), [1], 4, 0)
    with pytest.raises(ValueError):
        cpp.RcclTransport(b"short", 0, 1)


def test_transport_self_test_tells_a_transport_that_delivers_from_one_that_does_not():
    """Communicator._transport_self_test (run once when the RCCL transport has been set up, before the simulation depends
    on it): passes for the C++ transport, through RCCL and through its copy path, and names what went wrong for transports
    that deliver nothing, something else, or raise."""
    import types
    import torch
    from hoomd_amd import _hoomd_amd as cpp
    from hoomd_amd.comm import Communicator
    dd = types.SimpleNamespace(world=1, rank=0)
    dev = torch.device("cuda", torch.cuda.current_device())
    t = cpp.RcclTransport(cpp.RcclTransport.unique_id(), 0, 1)
    for through in (False, True):
        t.self_through_rccl = through
        assert Communicator._transport_self_test(t, dd, dev) is None

    class Silent:
        def allToAll(self, *a):
            pass

        def allReduceMaxWord(self, *a):
            pass

    class Garbled(Silent):
        def allToAll(self, d_send, send_rows, d_recv, recv_rows, row_bytes, stream):
            t.allToAll(d_send, send_rows, d_recv, recv_rows, row_bytes, stream)
            torch.cuda.synchronize()
            from hoomd_amd.comm import _DevMem
            torch.as_tensor(_DevMem(d_recv, (1,), "<i4"), device=dev).fill_(12345)

    class Raising(Silent):
        def allToAll(self, *a):
            raise RuntimeError("no route to peer")

    assert "arrived as [-1" in Communicator._transport_self_test(Silent(), dd, dev)
    assert "arrived as [12345" in Communicator._transport_self_test(Garbled(), dd, dev)
    assert "no route to peer" in Communicator._transport_self_test(Raising(), dd, dev)
