"""One rank of the CPU (gloo) test of the migration / halo-exchange logic.  Launched by
test_comm_gloo.py as `python _comm_worker.py RANK WORLD PORT GX GY GZ`; exits non-zero on failure."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd")]

import torch
import torch.distributed as dist


def lj_forces(xi, xall, L, periodic, r_cut=2.5, skip_self=None):
    """Brute-force LJ forces on the rows of xi from every row of xall (minimum image in `periodic` dims)."""
    d = xi[:, None, :] - xall[None, :, :]
    for a in range(3):
        if periodic[a]:
            d[:, :, a] -= np.round(d[:, :, a] / L[a]) * L[a]
    r2 = (d * d).sum(-1)
    if skip_self is not None:
        r2[np.arange(len(xi)), skip_self] = np.inf
    ok = r2 < r_cut * r_cut
    r2 = np.where(ok, r2, 1.0)
    inv6 = 1.0 / r2**3
    fdivr = np.where(ok, (48.0 * inv6 * inv6 - 24.0 * inv6) / r2, 0.0)
    return (d * fdivr[:, :, None]).sum(1)


def expected_ghosts(dd, pos, tags, owner, w):
    """(tag, image position) of every periodic image inside this rank's sub-box extended by w."""
    out = {}
    shifts = [[0.0] if not dd.decomposed[a] else [-dd.gL[a], 0.0, dd.gL[a]] for a in range(3)]
    for sx in shifts[0]:
        for sy in shifts[1]:
            for sz in shifts[2]:
                p = pos + np.array([sx, sy, sz])
                inside = np.ones(len(p), dtype=bool)
                for a in range(3):
                    if dd.decomposed[a]:
                        inside &= (p[:, a] >= dd.lo[a] - w) & (p[:, a] < dd.hi[a] + w)
                if (sx, sy, sz) == (0.0, 0.0, 0.0):
                    inside &= owner != dd.rank
                for k in np.nonzero(inside)[0]:
                    out.setdefault(int(tags[k]), []).append(p[k])
    return out


def check_state(dd, ex, A, gpos, gvel, gimg0, L, w, what):
    N, ng = A.N, A.n_ghosts
    tag = A.tag[:N + ng].numpy().astype(np.int64)
    pos = A.pos[:N + ng, :3].numpy()
    # owners: inside the sub-box, unwrapped position conserved, velocity travelled with the tag
    for a in range(3):
        if dd.decomposed[a]:
            assert (pos[:N, a] >= dd.lo[a] - 1e-12).all() and (pos[:N, a] < dd.hi[a] + 1e-12).all(), what
    unwrapped = pos[:N] + A.image[:N].numpy() * L
    assert np.allclose(unwrapped, gpos[tag[:N]] + gimg0[tag[:N]] * L, atol=1e-12), what
    assert np.array_equal(A.vel[:N].numpy(), gvel[tag[:N]]), what
    assert np.array_equal(A.pos[:N, 3].numpy().view(np.int64), tag[:N] % 3), what + ": type word must survive bit for bit"
    # every tag owned exactly once over all ranks
    owned = [None] * dd.world
    dist.all_gather_object(owned, tag[:N].tolist())
    allt = np.sort(np.concatenate([np.asarray(o, dtype=np.int64) for o in owned]))
    assert np.array_equal(allt, np.arange(len(gpos))), what
    # ghost layer equals the brute-force expectation
    wrapped = gpos - np.floor((gpos - dd.glo) / L) * L * np.array(dd.decomposed)
    owner = dd.owner_of(np.clip(wrapped, dd.glo, dd.glo + L - 1e-12))
    exp = expected_ghosts(dd, wrapped, np.arange(len(gpos)), owner, w)
    got = {}
    for t, p in zip(tag[N:], pos[N:]):
        got.setdefault(int(t), []).append(p)
    assert sorted(got) == sorted(exp), f"{what}: ghost tag sets differ ({len(got)} vs {len(exp)})"
    for t in exp:
        e = np.array(sorted(map(tuple, exp[t])))
        g = np.array(sorted(map(tuple, got[t])))
        assert e.shape == g.shape and np.allclose(e, g, atol=1e-12), f"{what}: images of tag {t}"
    # reverse lookup
    rt = A.rtag.numpy()
    assert (rt[tag[:N]] == np.arange(N)).all(), what
    assert (rt >= 0).sum() == len(set(tag.tolist())), what
    # forces on owned particles from owned + ghosts (no minimum image across decomposed directions)
    periodic = [not d for d in dd.decomposed]
    F = lj_forces(pos[:N], pos, L, periodic, skip_self=np.arange(N))
    F_ref = lj_forces(gpos[tag[:N]], gpos, L, [True] * 3, skip_self=tag[:N])
    assert np.abs(F - F_ref).max() <= 1e-9 * max(1.0, np.abs(F_ref).max()), what


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    grid = tuple(int(v) for v in sys.argv[4:7])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hoomd_amd import comm

    n, a = 14, 1.06
    L = np.array([n * a] * 3)
    g = (np.arange(n) + 0.5) * a - L[0] / 2
    z, y, x = np.meshgrid(g, g, g, indexing="ij")
    rng = np.random.default_rng(5)
    gpos = np.stack([x.ravel(), y.ravel(), z.ravel()], 1) + rng.normal(0, 0.05, (n**3, 3))
    gpos -= np.floor((gpos + L / 2) / L) * L
    N = len(gpos)
    gvel = np.concatenate([rng.normal(0, 1, (N, 3)), np.ones((N, 1))], 1)
    gimg0 = np.zeros((N, 3), dtype=np.int64)
    w = 2.9

    dd = comm.DomainDecomposition(grid=grid if any(grid) else None)
    dd._set_box(L, -L / 2)
    if not any(grid):
        assert dd.grid == comm.find_grid(world, L)
    mine = np.nonzero(dd.owner_of(gpos) == rank)[0]
    pos4 = np.concatenate([gpos[mine], np.zeros((len(mine), 1))], 1)
    pos4[:, 3] = (mine % 3).astype(np.int64).view(np.float64)  # type ids ride in the bit pattern of w
    A = comm.HostArrays(pos4, gvel[mine], mine, N)
    ex = comm.ParticleExchange(dd, A)

    assert ex.migrate() is False
    ex.exchange_ghosts(w)
    check_state(dd, ex, A, gpos, gvel, gimg0, L, w, "initial exchange")

    # large random moves: many particles change owner, some across the periodic boundary, some to
    # domains that are not direct neighbors of the sender
    for it in range(2):
        step = rng.uniform(-2.5, 2.5, (N, 3)) if it == 0 else rng.normal(0, 0.3, (N, 3))
        gpos = gpos + step
        tag = A.tag[:A.N].numpy().astype(np.int64)
        A.pos[:A.N, :3] += torch.from_numpy(step[tag])
        before = A.N
        moved = ex.migrate()
        ex.exchange_ghosts(w)
        check_state(dd, ex, A, gpos, gvel, gimg0, L, w, f"after move {it}")
        assert moved
    tot = torch.tensor([ex.n_migrated_out, ex.n_migrated_in], dtype=torch.int64)
    dist.all_reduce(tot)
    assert tot[0] == tot[1] and tot[0] > 0

    # per-step path: small displacement, positions of ghosts follow their owners
    step = rng.normal(0, 0.01, (N, 3))
    gpos = gpos + step
    tag = A.tag[:A.N].numpy().astype(np.int64)
    A.pos[:A.N, :3] += torch.from_numpy(step[tag])
    ex.update_ghosts()
    gt = A.tag[A.N:A.N + A.n_ghosts].numpy().astype(np.int64)
    gp = A.pos[A.N:A.N + A.n_ghosts, :3].numpy()
    d = gp - gpos[gt]
    d -= np.round(d / L) * L
    assert np.abs(d).max() < 1e-12
    assert np.array_equal(A.pos[A.N:A.N + A.n_ghosts, 3].numpy().view(np.int64), gt % 3)

    # a migration that involves two ranks only: the others still have to take part in every collective
    tag = A.tag[:A.N].numpy().astype(np.int64)
    x = A.pos[:A.N, :3].numpy()
    step = np.zeros((N, 3))
    if world >= 2 and dd.grid[0] >= 2:
        if dd.coords == (0, 0, 0):
            sel = tag[(x[:, 0] > dd.hi[0] - 0.3)]
            step[sel, 0] = 0.4
        allstep = [None] * world
        dist.all_gather_object(allstep, step)
        step = sum(allstep)
        gpos = gpos + step
        A.pos[:A.N, :3] += torch.from_numpy(step[tag])
        ex.migrate()
        ex.exchange_ghosts(w)
        check_state(dd, ex, A, gpos, gvel, gimg0, L, w, "two-rank migration")

    # rebuild vote
    assert ex.any_rank(rank == world - 1) is True
    assert ex.any_rank(False) is False
    s = dd._allreduce_sum_list([1.0, float(rank)])
    assert s == [float(world), float(sum(range(world)))]
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
