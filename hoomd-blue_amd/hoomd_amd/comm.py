"""Spatial domain decomposition and halo exchange: one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI on the GPUs, "gloo" on CPU tensors for host-side tests).

Role of hoomd/DomainDecomposition.cc (grid choice :327-408, sub-box bounds), hoomd/Communicator.cc
(communicate :1510-1569, migrateParticles, exchangeGhosts, beginUpdateGhosts) and
hoomd/CommunicatorGPU.cu (ghost plan :306-358, pack with periodic shift :789-875).

The exchange itself (migration, ghost layer, per-step ghost update, rebuild vote) is the C++ class
``_cpp.CommunicatorGPU`` (host/communicator.cc); this module sets the decomposition up (grid, sub-boxes,
scatter / gather of snapshots), provides the message-passing layer the C++ class calls (``TorchTransport``:
``all_to_all_single`` / ``all_reduce`` on the device buffers) and keeps the tensor-operation form of the same
protocol for CPU tensors (``ParticleExchange``: host-side tests and the CPU baseline).
"""
import contextlib
import itertools
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _hoomd_amd as _cpp

NOT_LOCAL = -1  # 0xffffffff seen through an int32 view
_ELEM = 18      # int64 words of one migrating particle: pos4 vel4 accel3 image3 tag body charge diameter


def find_grid(nranks, L, twod=False):
    """Decomposition with the least inter-domain surface (hoomd/DomainDecomposition.cc:327-408).
    Ties keep the first candidate found scanning nx, then ny, upward -- more domains along z."""
    best, best_area = None, None
    for nx in range(1, nranks + 1):
        if nranks % nx:
            continue
        for ny in range(1, nranks // nx + 1):
            if (nranks // nx) % ny:
                continue
            nz = nranks // nx // ny
            if twod and nz > 1:
                continue
            area = L[0] * L[1] * (nz - 1) + L[0] * L[2] * (ny - 1) + L[1] * L[2] * (nx - 1)
            if best is None or area < best_area:
                best, best_area = (nx, ny, nz), area
    return best


class HostArrays:
    """Per-particle arrays held in ordinary torch tensors (any device).  Used by the CPU tests of the
    exchange logic; the GPU path binds the same interface to the C++ ParticleData (``PDataArrays``)."""

    def __init__(self, pos4, vel4, tag, n_global, image=None, accel=None, device="cpu"):
        n = len(pos4)
        self.N, self.n_ghosts, self.n_global = n, 0, int(n_global)
        self.max_n = max(n, 1)
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=device).clone()
        self.pos = t(pos4, torch.float64).reshape(-1, 4)
        self.vel = t(vel4, torch.float64).reshape(-1, 4)
        self.accel = t(np.zeros((n, 3)) if accel is None else accel, torch.float64).reshape(-1, 3)
        self.image = t(np.zeros((n, 3), dtype=np.int32) if image is None else image, torch.int32).reshape(-1, 3)
        self.tag = t(tag, torch.int32)
        self.body = torch.full((n,), -1, dtype=torch.int32, device=device)
        self.charge = torch.zeros(n, dtype=torch.float64, device=device)
        self.diameter = torch.ones(n, dtype=torch.float64, device=device)
        self.rtag = torch.full((self.n_global,), NOT_LOCAL, dtype=torch.int32, device=device)
        self.generation = 0

    _PER_PARTICLE = ("pos", "vel", "accel", "image", "tag", "body", "charge", "diameter")

    def _grow(self, need):
        if need <= self.max_n:
            return
        new = int(need * 1.125) + 64
        for name in self._PER_PARTICLE:
            a = getattr(self, name)
            b = torch.zeros((new,) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
            b[: len(a)] = a
            setattr(self, name, b)
        self.max_n = new
        self.generation += 1

    def resize_owned(self, n):
        self._grow(n + self.n_ghosts)
        self.N = n

    def set_n_ghosts(self, n):
        self._grow(self.N + n)
        self.n_ghosts = n

    def notify_sort(self):
        pass


class _DevMem:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class PDataArrays:
    """Zero-copy torch views of the device arrays owned by the C++ ParticleData."""

    def __init__(self, pdata):
        self._p = pdata
        self._key = None
        self._bind()

    def _bind(self):
        p = self._p
        key = (p.max_n, p.pos_ptr, p.rtag_ptr, p.N_global)
        if key == self._key:
            return
        self._key = key
        m = max(p.max_n, 1)
        dev = torch.device("cuda", torch.cuda.current_device())
        v = lambda ptr, shape, ts: torch.as_tensor(_DevMem(ptr, shape, ts), device=dev)
        self.pos = v(p.pos_ptr, (m, 4), "<f8")
        self.vel = v(p.vel_ptr, (m, 4), "<f8")
        self.accel = v(p.accel_ptr, (m, 3), "<f8")
        self.image = v(p.image_ptr, (m, 3), "<i4")
        self.tag = v(p.tag_ptr, (m,), "<i4")
        self.body = v(p.body_ptr, (m,), "<i4")
        self.charge = v(p.charge_ptr, (m,), "<f8")
        self.diameter = v(p.diameter_ptr, (m,), "<f8")
        self.rtag = v(p.rtag_ptr, (max(p.N_global, 1),), "<i4")
        self.generation = getattr(self, "generation", 0) + 1

    N = property(lambda self: self._p.N)
    n_ghosts = property(lambda self: self._p.n_ghosts)
    n_global = property(lambda self: self._p.N_global)
    max_n = property(lambda self: self._p.max_n)
    ntypes = property(lambda self: len(self._p.types))

    def refresh(self):
        self._bind()

    def resize_owned(self, n):
        self._p.resizeOwned(int(n))
        self._bind()

    def set_n_ghosts(self, n):
        self._p.setNGhosts(int(n))
        self._bind()

    def notify_sort(self):
        self._p.notifyParticleSort()


class DomainDecomposition:
    """Regular grid of sub-boxes over the ranks of a process group (hoomd/DomainDecomposition.h).

    ``grid``: (nx, ny, nz) or None to pick the least-surface split once the box is known.
    Rank r sits at grid coordinates (i, j, k) with r = (k * ny + j) * nx + i.
    Every rank passes the same global snapshot to ``create_state_from_snapshot``; each keeps the
    particles inside its own sub-box.  Triclinic boxes are not decomposed (raises)."""

    def __init__(self, grid=None, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised before DomainDecomposition is created")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.grid = tuple(int(g) for g in grid) if grid is not None else None
        if self.grid is not None and int(np.prod(self.grid)) != self.world:
            raise ValueError(f"grid {self.grid} does not match {self.world} ranks")
        self.communicator = None
        self._sim = None
        self._shares_gpu = None

    def ranks_share_a_gpu(self):
        """True when two ranks of the group run on the same GPU (rehearsals and tests; production is one rank per
        GPU).  Every rank gets the same answer: (host name, PCI bus id of the current device) of all ranks are
        gathered once.  The neighbor list then keeps its exchange stream at normal priority: with more processes on
        a GPU than it has hardware queues, a high-priority queue waiting for an event of a normal-priority queue
        keeps the scheduler from ever running the latter."""
        if self._shares_gpu is None:
            import socket
            me = None
            if torch.cuda.is_available():
                p = torch.cuda.get_device_properties(torch.cuda.current_device())
                ident = getattr(p, "uuid", None) or getattr(p, "pci_bus_id", None)
                me = (socket.gethostname(), str(ident) if ident is not None else f"ordinal {torch.cuda.current_device()}")
            ids = [None] * self.world
            dist.all_gather_object(ids, me, group=self.group)
            ids = [i for i in ids if i is not None]
            self._shares_gpu = len(set(ids)) < len(ids)
        return self._shares_gpu

    # ---- geometry ----
    def _set_box(self, L, lo, twod=False):
        self.gL = np.asarray(L, dtype=np.float64)
        self.glo = np.asarray(lo, dtype=np.float64)
        if self.grid is None:
            self.grid = find_grid(self.world, self.gL, twod)
        nx, ny, nz = self.grid
        r = self.rank
        self.coords = (r % nx, (r // nx) % ny, r // (nx * ny))
        g = np.asarray(self.grid, dtype=np.float64)
        c = np.asarray(self.coords, dtype=np.float64)
        self.lo = self.glo + self.gL * (c / g)
        self.hi = self.glo + self.gL * ((c + 1) / g)
        for d in range(3):  # the faces of the global box are hit exactly
            if self.coords[d] == self.grid[d] - 1:
                self.hi[d] = self.glo[d] + self.gL[d]
        self.decomposed = tuple(n > 1 for n in self.grid)

    def rank_at(self, i, j, k):
        nx, ny, nz = self.grid
        return ((k % nz) * ny + (j % ny)) * nx + (i % nx)

    def owner_of(self, xyz):
        """Owner rank of each position (numpy, positions inside the global box)."""
        f = (np.asarray(xyz, dtype=np.float64) - self.glo) / self.gL
        g = np.asarray(self.grid)
        c = np.clip(np.floor(f * g).astype(np.int64), 0, g - 1)
        return (c[:, 2] * self.grid[1] + c[:, 1]) * self.grid[0] + c[:, 0]

    # ---- hooks used by Simulation / State ----
    def _make_sysdef(self, snapshot, cbox, ndim, seed):
        b = snapshot.configuration._box
        if b.xy != 0 or b.xz != 0 or b.yz != 0:
            raise NotImplementedError("domain decomposition of triclinic boxes is not supported")
        L = np.array([b.Lx, b.Ly, b.Lz if ndim == 3 else 1.0])
        self._set_box(L, -L / 2, twod=(ndim == 2))
        d = snapshot._to_dict()
        n_global = int(d["N"])
        mine = np.nonzero(self.owner_of(d["position"]) == self.rank)[0]
        local = {"N": len(mine), "types": d["types"], "tag": mine.astype(np.uint32)}
        for k, v in d.items():
            if k not in ("N", "types"):
                local[k] = np.ascontiguousarray(v[mine])
        sysdef = _cpp.SystemDefinition(local, cbox, ndim, seed, n_global)
        lbox = _cpp.BoxDim(b.Lx, b.Ly, b.Lz, 0.0, 0.0, 0.0)
        lbox.setLoHi(list(self.lo), list(self.hi))
        lbox.setPeriodic(*[not dec for dec in self.decomposed])
        sysdef.getParticleData().setBoxes(cbox, lbox)
        return sysdef

    def _attach(self, sim):
        self._sim = sim
        self.communicator = Communicator(self, sim._state._sysdef)
        sim._state._sysdef.setCommunicator(self.communicator.cpp)

    def _nlist_attached(self, nlist):
        self.communicator.add_nlist(nlist)
        nlist._cpp_obj.ranks_share_gpu = bool(self.ranks_share_a_gpu())
        if os.environ.get("HOOMD_AMD_COMM_PRIORITY") == "high" and not self.ranks_share_a_gpu():
            nlist._cpp_obj.comm_stream_high_priority = True

    def _allreduce_sum_list(self, values):
        tr = self.communicator.transport if self.communicator is not None else None
        dev = tr.device if tr is not None and not tr.stage else "cpu"
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.tolist()

    def _gather_snapshot(self, d):
        parts = [None] * self.world
        dist.all_gather_object(parts, {k: v for k, v in d.items() if k != "types"}, group=self.group)
        n = sum(int(p["N"]) for p in parts)
        out = {"N": n, "types": d["types"]}
        tags = np.concatenate([np.asarray(p["tag"], dtype=np.int64) for p in parts])
        for k in d:
            if k in ("N", "types", "tag"):
                continue
            cat = np.concatenate([p[k] for p in parts])
            full = np.empty_like(cat)
            full[tags] = cat
            out[k] = full
        out["tag"] = np.arange(n, dtype=np.uint32)
        return out

    def _scatter_snapshot(self, snapshot, pdata):
        d = snapshot._to_dict()
        mine = np.nonzero(self.owner_of(d["position"]) == self.rank)[0]
        local = {"N": len(mine), "types": d["types"], "tag": mine.astype(np.uint32)}
        for k, v in d.items():
            if k not in ("N", "types"):
                local[k] = np.ascontiguousarray(v[mine])
        pdata.initializeFromSnapshot(local)
        if self.communicator is not None:
            self.communicator.cpp.forceMigrate()


class ParticleExchange:
    """Migration and halo exchange written with tensor operations on a set of per-particle arrays (``HostArrays``).

    The same protocol as the C++ communicator of the GPU path (``_cpp.CommunicatorGPU``: one send list sorted by
    destination rank, one ``all_to_all_single`` per exchange), for CPU tensors over gloo: the host-side tests of the
    decomposition logic and the decomposed CPU baseline of the benchmark run on it.  ``arrays`` provides N, n_ghosts,
    the tensors pos/vel/accel/image/tag/body/charge/diameter/rtag and resize_owned / set_n_ghosts / notify_sort."""

    def __init__(self, decomposition, arrays):
        self.dd = decomposition
        self.arrays = arrays
        self.device = arrays.pos.device
        dd = decomposition
        # one entry per direction towards a neighbor domain, sorted by destination rank
        dirs = []
        for d in itertools.product((-1, 0, 1), repeat=3):
            if d == (0, 0, 0) or any(d[a] != 0 and not dd.decomposed[a] for a in range(3)):
                continue
            mask, shift = 0, [0.0, 0.0, 0.0]
            for a in range(3):
                if d[a] == 1:
                    mask |= 1 << (2 * a)
                    if dd.coords[a] + 1 >= dd.grid[a]:
                        shift[a] = -dd.gL[a]
                elif d[a] == -1:
                    mask |= 2 << (2 * a)
                    if dd.coords[a] - 1 < 0:
                        shift[a] = dd.gL[a]
            dest = dd.rank_at(dd.coords[0] + d[0], dd.coords[1] + d[1], dd.coords[2] + d[2])
            dirs.append((dest, mask, shift))
        dirs.sort(key=lambda e: (e[0], e[1]))
        self.dir_dest = [e[0] for e in dirs]
        self.dir_mask = torch.tensor([e[1] for e in dirs], dtype=torch.int32, device=self.device)
        self.shift_table = torch.tensor([e[2] for e in dirs], dtype=torch.float64, device=self.device).reshape(-1, 3)
        self.lo = torch.tensor(dd.lo, dtype=torch.float64, device=self.device)
        self.hi = torch.tensor(dd.hi, dtype=torch.float64, device=self.device)
        self.glo = torch.tensor(dd.glo, dtype=torch.float64, device=self.device)
        self.gL = torch.tensor(dd.gL, dtype=torch.float64, device=self.device)
        self.gridf = torch.tensor(dd.grid, dtype=torch.float64, device=self.device)
        self.gridi = torch.tensor(dd.grid, dtype=torch.int64, device=self.device)
        self.dec = torch.tensor(dd.decomposed, dtype=torch.bool, device=self.device)
        self.send_idx = torch.zeros(0, dtype=torch.int64, device=self.device)
        self.send_dir = torch.zeros(0, dtype=torch.uint8, device=self.device)
        self.send_shift = torch.zeros((0, 3), dtype=torch.float64, device=self.device)
        self.send_splits = [0] * dd.world
        self.recv_splits = [0] * dd.world
        self.n_migrated_out = 0
        self.n_migrated_in = 0
        self._vote = None

    # ---- collectives ----
    def _a2a(self, out, inp, out_splits, in_splits):
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.dd.group)

    def _exchange_counts(self, send_counts):
        s = torch.tensor(send_counts, dtype=torch.int64, device=self.device)
        r = torch.empty_like(s)
        self._a2a(r, s, None, None)
        return [int(x) for x in r.tolist()]

    # ---- migration (hoomd/Communicator.cc migrateParticles) ----
    def migrate(self):
        A, dd = self.arrays, self.dd
        N = A.N
        A.set_n_ghosts(0)  # ghosts are re-exchanged afterwards
        x = A.pos[:N, :3]
        f = (x - self.glo) / self.gL
        wrap = torch.where(self.dec, torch.floor(f), torch.zeros_like(f))
        c = torch.clamp(torch.floor((f - wrap) * self.gridf).to(torch.int64), min=torch.zeros_like(self.gridi), max=self.gridi - 1)
        dest = (c[:, 2] * dd.grid[1] + c[:, 1]) * dd.grid[0] + c[:, 0]
        leave = dest != dd.rank
        li = torch.nonzero(leave).flatten()
        n_leave = int(li.numel())
        send_counts = [0] * dd.world
        payload = torch.zeros((0, _ELEM), dtype=torch.int64, device=self.device)
        if n_leave:
            dl, order = torch.sort(dest[li].to(torch.int64), stable=True)
            li = li[order]
            send_counts = torch.bincount(dl, minlength=dd.world).tolist()
            w = wrap[li]
            p = A.pos[li].clone()
            p[:, :3] -= w * self.gL
            img = A.image[li].to(torch.int64) + w.to(torch.int64)
            payload = torch.cat([
                p.view(torch.int64), A.vel[li].view(torch.int64), A.accel[li].view(torch.int64), img,
                A.tag[li].to(torch.int64)[:, None], A.body[li].to(torch.int64)[:, None],
                A.charge[li].view(torch.int64)[:, None], A.diameter[li].view(torch.int64)[:, None]], dim=1).contiguous()
        recv_counts = self._exchange_counts(send_counts)
        n_recv = sum(recv_counts)
        self.n_migrated_out += n_leave
        self.n_migrated_in += n_recv
        # every rank takes part in the collective, also with nothing to send or receive
        incoming = torch.empty((n_recv, _ELEM), dtype=torch.int64, device=self.device)
        self._a2a(incoming, payload, recv_counts, send_counts)
        if n_leave == 0 and n_recv == 0:
            return False
        keep = torch.nonzero(~leave).flatten()
        n_keep = int(keep.numel())
        new_n = n_keep + n_recv
        kept = {name: getattr(A, name)[keep] for name in HostArrays._PER_PARTICLE} if n_leave else None
        A.resize_owned(new_n)
        if n_leave:
            for name, v in kept.items():
                getattr(A, name)[:n_keep] = v
        if n_recv:
            s = slice(n_keep, new_n)
            A.pos[s] = incoming[:, 0:4].contiguous().view(torch.float64)
            A.vel[s] = incoming[:, 4:8].contiguous().view(torch.float64)
            A.accel[s] = incoming[:, 8:11].contiguous().view(torch.float64)
            A.image[s] = incoming[:, 11:14].to(torch.int32)
            A.tag[s] = incoming[:, 14].to(torch.int32)
            A.body[s] = incoming[:, 15].to(torch.int32)
            A.charge[s] = incoming[:, 16].contiguous().view(torch.float64)
            A.diameter[s] = incoming[:, 17].contiguous().view(torch.float64)
        return True

    # ---- ghost exchange (hoomd/Communicator.cc exchangeGhosts; plan bits of CommunicatorGPU.cu:306-358) ----
    def exchange_ghosts(self, r_ghost):
        A, dd = self.arrays, self.dd
        N = A.N
        A.set_n_ghosts(0)
        width = dd.hi - dd.lo
        for a in range(3):
            if dd.decomposed[a] and width[a] <= 2.0 * r_ghost:
                raise RuntimeError("Simulation box too small for domain decomposition: a sub-box must be wider than twice the ghost layer")
        ndir = len(self.dir_dest)
        x = A.pos[:N, :3]
        f = (x - self.lo) / (self.hi - self.lo)
        g = r_ghost / (self.hi - self.lo)
        bits = torch.tensor([[1, 4, 16], [2, 8, 32]], dtype=torch.int32, device=self.device)
        decm = self.dec.to(torch.int32)
        plan = (((f >= (1.0 - g)).to(torch.int32) * bits[0] + (f < g).to(torch.int32) * bits[1]) * decm).sum(dim=1, dtype=torch.int32)
        b = torch.nonzero(plan).flatten()
        if ndir and b.numel():
            m = (plan[b][None, :] & self.dir_mask[:, None]) == self.dir_mask[:, None]   # [ndir, nb]
            nz = torch.nonzero(m)
            per_dir = m.sum(dim=1).tolist()
            self.send_idx = b[nz[:, 1]].contiguous()
            self.send_dir = nz[:, 0].to(torch.uint8).contiguous()
            self.send_shift = self.shift_table[nz[:, 0]].contiguous()
        else:
            per_dir = [0] * ndir
            self.send_idx = torch.zeros(0, dtype=torch.int64, device=self.device)
            self.send_dir = torch.zeros(0, dtype=torch.uint8, device=self.device)
            self.send_shift = torch.zeros((0, 3), dtype=torch.float64, device=self.device)
        send_counts = [0] * dd.world
        for k, n in enumerate(per_dir):
            send_counts[self.dir_dest[k]] += int(n)
        self.send_splits = send_counts
        self.recv_splits = self._exchange_counts(send_counts)
        n_ghost = sum(self.recv_splits)
        A.set_n_ghosts(n_ghost)
        s = slice(N, N + n_ghost)
        self._a2a(A.tag[s], A.tag[self.send_idx].contiguous(), self.recv_splits, self.send_splits)
        A.body[s] = -1
        A.charge[s] = 0.0
        A.diameter[s] = 1.0
        A.image[s] = 0
        self.update_ghosts()
        A.rtag.fill_(NOT_LOCAL)
        n_all = N + n_ghost
        A.rtag[A.tag[:n_all].to(torch.int64)] = torch.arange(n_all, dtype=torch.int32, device=self.device)

    def pack(self):
        A = self.arrays
        buf = A.pos[self.send_idx]
        buf[:, :3] += self.send_shift
        return buf

    # ---- per-step ghost position update (hoomd/Communicator.cc beginUpdateGhosts/finishUpdateGhosts) ----
    def update_ghosts(self):
        A = self.arrays
        N, ng = A.N, A.n_ghosts
        self._a2a(A.pos[N:N + ng], self.pack(), self.recv_splits, self.send_splits)

    def any_rank(self, local):
        if self._vote is None:
            self._vote = torch.zeros(1, dtype=torch.int32, device=self.device)
        t = self._vote
        t.fill_(1 if local else 0)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.dd.group)
        return bool(t.item())


class TorchTransport(getattr(_cpp, "ExchangeTransport", object)):
    """The message-passing layer under the C++ communicator: device buffers go to ``torch.distributed``
    (``all_to_all_single`` / ``all_reduce``; backend "nccl" = RCCL moves them GPU to GPU).  With the gloo backend (tests
    and rehearsals with several ranks on one GPU) buffers are staged through pinned host memory with asynchronous copies
    on the stream the exchange kernels run on."""

    def __init__(self, group, device):
        super().__init__()
        self.group = group
        self.device = device
        self.stage = dist.get_backend(group) == "gloo"
        self._views = {}
        self._streams = {}
        self._pinned = {}
        self._empty = torch.zeros(0, dtype=torch.uint8, device=device)
        self._vote = torch.zeros(1, dtype=torch.int32, device="cpu" if self.stage else device)
        self.side = None  # (stream handle, torch stream) of the last exchange queued on a stream of its own

    def _view(self, ptr, nbytes, dtype=torch.uint8):
        if nbytes == 0:
            return self._empty
        key = (ptr, nbytes, dtype)
        v = self._views.get(key)
        if v is None:
            if len(self._views) > 512:
                self._views.clear()
            n = nbytes // dtype.itemsize
            v = torch.as_tensor(_DevMem(ptr, (n,), {torch.uint8: "|u1", torch.int32: "<i4"}[dtype]), device=self.device)
            self._views[key] = v
        return v

    def _pinned_like(self, name, t):
        nbytes = t.numel() * t.element_size()
        buf = self._pinned.get(name)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes * 1.25), 4096), dtype=torch.uint8, pin_memory=True)
            self._pinned[name] = buf
        return buf[:nbytes].view(t.dtype).view(t.shape)

    def _on(self, stream):
        if not stream:
            return contextlib.nullcontext()
        s = self._streams.get(stream)
        if s is None:
            s = torch.cuda.ExternalStream(stream)
            self._streams[stream] = s
        self.side = (stream, s)
        return torch.cuda.stream(s)

    def allToAll(self, d_send, send_rows, d_recv, recv_rows, row_bytes, stream):
        in_splits = [r * row_bytes for r in send_rows]
        out_splits = [r * row_bytes for r in recv_rows]
        inp, out = self._view(d_send, sum(in_splits)), self._view(d_recv, sum(out_splits))
        with self._on(stream):
            if self.stage:
                cur = torch.cuda.current_stream(self.device)
                hin, hout = self._pinned_like("in", inp), self._pinned_like("out", out)
                hin.copy_(inp, non_blocking=True)
                cur.synchronize()
                dist.all_to_all_single(hout, hin, out_splits, in_splits, group=self.group)
                out.copy_(hout, non_blocking=True)
                cur.synchronize()  # the buffers are reused by the next call
            else:
                dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)

    def allReduceMaxWord(self, d_word, stream):
        word = self._view(d_word, 4, torch.int32)
        with self._on(stream):
            if self.stage:
                cur = torch.cuda.current_stream(self.device)
                v = self._pinned_like("vote", word)
                v.copy_(word, non_blocking=True)
                cur.synchronize()
                dist.all_reduce(v, op=dist.ReduceOp.MAX, group=self.group)
                word.copy_(v, non_blocking=True)
                cur.synchronize()
            else:
                dist.all_reduce(word, op=dist.ReduceOp.MAX, group=self.group)

    def allReduceSumHost(self, values):
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device="cpu" if self.stage else self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.tolist()

    def anyRank(self, local):
        self._vote.fill_(1 if local else 0)
        dist.all_reduce(self._vote, op=dist.ReduceOp.MAX, group=self.group)
        return bool(self._vote.item())


class Communicator:
    """Holder of the C++ communicator (``_cpp.CommunicatorGPU``, interface of hoomd/Communicator.h:166-232: migration,
    ghost exchange, per-step ghost update, rebuild vote) and of its transport."""

    def __init__(self, decomposition, sysdef):
        dd = decomposition
        self.dd = dd
        self.sysdef = sysdef
        device = torch.device("cuda", torch.cuda.current_device())
        self.transport = self._make_transport(dd, device)
        self.cpp = _cpp.CommunicatorGPU(sysdef, self.transport, dd.rank, dd.world, [int(g) for g in dd.grid], [float(v) for v in dd.glo],
                                        [float(v) for v in dd.gL])
        self.cpp.setUseVoteRows(os.environ.get("HOOMD_AMD_VOTE_ROWS", "1") != "0")
        self.cpp.setFixedMigrate(os.environ.get("HOOMD_AMD_FIXED_MIGRATE", "1") != "0")
        self.cpp.setMigrateSlotRows(int(os.environ.get("HOOMD_AMD_MIGRATE_SLOT_ROWS", "1024")))
        self.cpp.setUseGhostSlots(os.environ.get("HOOMD_AMD_GHOST_SLOTS", "1") != "0")
        self.cpp.setUseDeviceCounts(os.environ.get("HOOMD_AMD_DEVICE_COUNTS", "1") != "0")
        if "HOOMD_AMD_GHOST_SLOT_ROOM" in os.environ:  # "div,pad" (experiments, tests of the too-small case)
            div, pad = (int(v) for v in os.environ["HOOMD_AMD_GHOST_SLOT_ROOM"].split(","))
            self.cpp.setGhostSlotRoom(div, pad)
        self.nlists = []

    @staticmethod
    def _make_transport(dd, device):
        """RCCL called from C++ (``_cpp.RcclTransport``) when the process group's backend is nccl; ``TorchTransport``
        (collectives through torch.distributed, host-staged for gloo) otherwise, when HOOMD_AMD_TRANSPORT=torch asks
        for it, or when the RCCL communicator cannot be set up on every rank."""
        want = os.environ.get("HOOMD_AMD_TRANSPORT", "")
        if want not in ("", "rccl", "torch"):
            raise ValueError("HOOMD_AMD_TRANSPORT must be 'rccl' or 'torch'")
        if dist.get_backend(dd.group) != "nccl" or want == "torch" or not hasattr(_cpp, "RcclTransport"):
            if want == "rccl":
                raise RuntimeError("HOOMD_AMD_TRANSPORT=rccl needs a process group with the nccl backend")
            return TorchTransport(dd.group, device)
        # rank 0 makes the id, the existing group carries its 128 bytes to the others
        box = [_cpp.RcclTransport.unique_id() if dd.rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(dd.group, 0) if dd.group is not None else 0, group=dd.group)
        transport, err = None, None
        try:
            transport = _cpp.RcclTransport(box[0], dd.rank, dd.world)
        except RuntimeError as e:  # the same decision on every rank, below
            err = str(e)
        oks = [None] * dd.world
        dist.all_gather_object(oks, err, group=dd.group)
        bad = [(r, e) for r, e in enumerate(oks) if e is not None]
        if bad:
            if want == "rccl":
                raise RuntimeError(f"RCCL transport: rank {bad[0][0]}: {bad[0][1]}")
            import warnings
            warnings.warn(f"RCCL transport not available (rank {bad[0][0]}: {bad[0][1]}); collectives go through torch.distributed")
            return TorchTransport(dd.group, device)
        # One exchange with unequal messages and one reduction, checked on every rank, before the simulation depends on
        # the communicator: a transport that does not deliver what was sent is replaced by torch.distributed's, on all
        # ranks alike.
        err = Communicator._transport_self_test(transport, dd, device)
        dist.all_gather_object(oks, err, group=dd.group)
        bad = [(r, e) for r, e in enumerate(oks) if e is not None]
        if bad:
            if want == "rccl":
                raise RuntimeError(f"RCCL transport: rank {bad[0][0]}: {bad[0][1]}")
            import warnings
            warnings.warn(f"RCCL transport failed its self-test (rank {bad[0][0]}: {bad[0][1]}); collectives go through torch.distributed")
            del transport
            return TorchTransport(dd.group, device)
        transport.device = device
        transport.stage = False
        transport.side = None
        return transport

    @staticmethod
    def _transport_self_test(transport, dd, device):
        """Rank s sends (s + r) % 3 + 1 rows of 32 bytes holding 1000 s + r to rank r and expects the mirror image; then the
        maximum of the ranks' numbers.  Returns None, or what went wrong."""
        try:
            w, me = dd.world, dd.rank
            send_rows = [(me + r) % 3 + 1 for r in range(w)]
            out = torch.cat([torch.full((n * 8,), 1000 * me + r, dtype=torch.int32, device=device) for r, n in enumerate(send_rows)])
            inp = torch.full((sum(send_rows) * 8,), -1, dtype=torch.int32, device=device)
            torch.cuda.synchronize(device)
            transport.allToAll(out.data_ptr(), send_rows, inp.data_ptr(), send_rows, 32, 0)
            word = torch.tensor([me], dtype=torch.int32, device=device)
            torch.cuda.synchronize(device)
            transport.allReduceMaxWord(word.data_ptr(), 0)
            torch.cuda.synchronize(device)
            got, off = inp.cpu(), 0
            for r, n in enumerate(send_rows):
                if not bool((got[off:off + 8 * n] == 1000 * r + me).all()):
                    return f"the message of rank {r} arrived as {got[off:off + 8 * n][:4].tolist()}..., expected {1000 * r + me}"
                off += 8 * n
            if int(word.item()) != w - 1:
                return f"the maximum over ranks came out as {int(word.item())}, expected {w - 1}"
            return None
        except Exception as e:  # (whatever it is: every rank must reach the gather that follows)
            return f"{type(e).__name__}: {e}"

    def add_nlist(self, nlist):
        self.nlists.append(nlist)
        self.cpp.addNeighborList(nlist._cpp_obj)

    # statistics of the exchange (tests, bench)
    transport_name = property(lambda self: "rccl (C++: " + _cpp.RcclTransport.library_origin() + ")" if isinstance(self.transport, getattr(_cpp, "RcclTransport", ()))
                              else "torch.distributed " + dist.get_backend(self.dd.group))
    n_migrated_out = property(lambda self: self.cpp.n_migrated_out)
    n_migrated_in = property(lambda self: self.cpp.n_migrated_in)
    n_vote_exchanges = property(lambda self: self.cpp.n_vote_exchanges)
    n_count_exchanges = property(lambda self: self.cpp.n_count_exchanges)
    n_device_count_exchanges = property(lambda self: self.cpp.n_device_count_exchanges)
    n_ghost_slot_overflows = property(lambda self: self.cpp.n_ghost_slot_overflows)
    fixed_migrate = property(lambda self: self.cpp.fixed_migrate)
    migrate_slot_rows = property(lambda self: self.cpp.migrate_slot_rows)
