"""ctypes front end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Loads ``oracle/libmd_oracle.so`` (built by ``oracle/Makefile`` from ``md_oracle.c``) and wraps
each function with NumPy arrays.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the shipped GPU path never does.
"""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmd_oracle.so")

NO_BODY = 0xFFFFFFFF
EV_LJ, EV_EXAMPLE = 0, 1
SHIFT_NONE, SHIFT_SHIFT, SHIFT_XPLOR = 0, 1, 2


def build():
    """Compile the C restatement (gcc, a second or two)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    src = os.path.join(_HERE, "md_oracle.c")
    if not os.path.exists(_LIB_PATH) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)):
        build()
    return C.CDLL(_LIB_PATH)


_lib = _load()


class Box(C.Structure):
    """Mirror of ``orc_box_t`` (hoomd/BoxDim.h:633-641)."""

    _fields_ = [
        ("lo", C.c_double * 3),
        ("hi", C.c_double * 3),
        ("L", C.c_double * 3),
        ("Linv", C.c_double * 3),
        ("xy", C.c_double),
        ("xz", C.c_double),
        ("yz", C.c_double),
        ("periodic", C.c_ubyte * 3),
    ]

    @classmethod
    def from_L(cls, Lx, Ly, Lz, xy=0.0, xz=0.0, yz=0.0, periodic=(1, 1, 1)):
        b = cls()
        _lib.orc_box_init(C.byref(b), C.c_double(Lx), C.c_double(Ly), C.c_double(Lz), C.c_double(xy),
                          C.c_double(xz), C.c_double(yz))
        for d in range(3):
            b.periodic[d] = int(periodic[d])
        return b

    @classmethod
    def from_lo_hi(cls, lo, hi, periodic=(1, 1, 1)):
        b = cls()
        _lib.orc_box_set_lo_hi(C.byref(b), (C.c_double * 3)(*lo), (C.c_double * 3)(*hi))
        b.xy = b.xz = b.yz = 0.0
        for d in range(3):
            b.periodic[d] = int(periodic[d])
        return b


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def _u3(v):
    return (C.c_uint * 3)(*[int(x) for x in v])


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def pack_pos(xyz, typeid=None):
    """(N,3) positions + integer type ids -> (N,4) float64 with the id bit-stuffed in w
    (hoomd/HOOMDMath.h:166-217)."""
    xyz = np.asarray(xyz, dtype=np.float64)
    n = xyz.shape[0]
    out = np.zeros((n, 4), dtype=np.float64)
    out[:, :3] = xyz
    if typeid is not None:
        w = np.zeros(n, dtype=np.int64)
        w[:] = np.asarray(typeid, dtype=np.int64) & 0xFFFFFFFF
        out[:, 3] = w.view(np.float64)
    return out


def unpack_int_w(col):
    """Low 32 bits of each double of a w column."""
    return (np.ascontiguousarray(col, dtype=np.float64).view(np.int64) & 0xFFFFFFFF).astype(np.uint32)


# ---------------------------------------------------------------- box
def nearest_plane_distance(box):
    out = (C.c_double * 3)()
    _lib.orc_box_nearest_plane_distance(C.byref(box), out)
    return np.array(out[:])


def make_fraction(box, v, ghost_width=(0, 0, 0)):
    out = (C.c_double * 3)()
    _lib.orc_box_make_fraction(C.byref(box), _d3(v), _d3(ghost_width), out)
    return np.array(out[:])


def min_image(box, v, rint=False):
    w = _d3(v)
    (_lib.orc_box_min_image_rint if rint else _lib.orc_box_min_image)(C.byref(box), w)
    return np.array(w[:])


def wrap(box, v, img):
    w = _d3(v)
    im = (C.c_int * 3)(*[int(x) for x in img])
    _lib.orc_box_wrap(C.byref(box), w, im)
    return np.array(w[:]), np.array(im[:])


_lib.orc_index2d.restype = C.c_uint
_lib.orc_index3d.restype = C.c_uint


def index2d(w, i, j):
    return _lib.orc_index2d(C.c_uint(w), C.c_uint(i), C.c_uint(j))


def index3d(w, h, i, j, k):
    return _lib.orc_index3d(C.c_uint(w), C.c_uint(h), C.c_uint(i), C.c_uint(j), C.c_uint(k))


# ---------------------------------------------------------------- cell list
def cell_dimensions(box, nominal_width, multiple=1, ghost_width=(0, 0, 0), ndim=3):
    dim = (C.c_uint * 3)()
    _lib.orc_cell_dimensions(C.byref(box), C.c_double(nominal_width), C.c_uint(multiple), _d3(ghost_width),
                             C.c_uint(ndim), dim)
    return tuple(dim[:])


_lib.orc_cell_n_adj.restype = C.c_uint


def cell_adj(dim, radius=1, ndim=3):
    n_adj = _lib.orc_cell_n_adj(_u3(dim), C.c_uint(radius), C.c_uint(ndim))
    ncells = dim[0] * dim[1] * dim[2]
    adj = np.zeros((ncells, n_adj), dtype=np.uint32)
    _lib.orc_cell_adj(_u3(dim), C.c_uint(radius), C.c_uint(ndim), _p(adj, C.c_uint))
    return adj


def compute_cell_list(pos, N, n_ghost, box, dim, Nmax, ghost_width=(0, 0, 0), flag_mode=0, charge=None,
                      body=None, want_type_body=False, want_idx=False):
    """Returns dict(cell_size, xyzf, type_body, idx, conditions)."""
    pos = _c(pos, np.float64)
    ncells = dim[0] * dim[1] * dim[2]
    cell_size = np.zeros(ncells, dtype=np.uint32)
    xyzf = np.zeros((ncells, Nmax, 4), dtype=np.float64)
    tb = np.zeros((ncells, Nmax, 2), dtype=np.uint32) if want_type_body else None
    idx = np.zeros((ncells, Nmax), dtype=np.uint32) if want_idx else None
    cond = (C.c_uint * 3)()
    charge = _c(charge, np.float64) if charge is not None else None
    body = _c(body, np.uint32) if body is not None else None
    _lib.orc_compute_cell_list(_p(pos, C.c_double), _p(charge, C.c_double), _p(body, C.c_uint), C.c_uint(N),
                               C.c_uint(n_ghost), C.byref(box), _u3(dim), C.c_uint(Nmax), _d3(ghost_width),
                               C.c_int(flag_mode), _p(cell_size, C.c_uint), _p(xyzf, C.c_double),
                               _p(tb, C.c_uint), _p(idx, C.c_uint), cond)
    return dict(cell_size=cell_size, xyzf=xyzf, type_body=tb, idx=idx, conditions=tuple(cond[:]))


def cell_list_auto(pos, N, n_ghost, box, dim, ghost_width=(0, 0, 0), **kw):
    """hoomd/CellList.cc:184-199, :232-241: start from ceil(N/ncells) slots and grow on overflow."""
    ncells = dim[0] * dim[1] * dim[2]
    Nmax = max(1, int(np.ceil((N + n_ghost) / ncells)))
    while True:
        r = compute_cell_list(pos, N, n_ghost, box, dim, Nmax, ghost_width, **kw)
        if r["conditions"][0] > Nmax:
            Nmax = r["conditions"][0]
            continue
        r["Nmax"] = Nmax
        return r


# ---------------------------------------------------------------- neighbor list
def update_rlist(r_cut, r_buff):
    r_cut = _c(r_cut, np.float64)
    nt = r_cut.shape[0]
    r_listsq = np.zeros((nt, nt))
    rcut_max = np.zeros(nt)
    mm = C.c_double()
    mn = C.c_double()
    _lib.orc_update_rlist(_p(r_cut, C.c_double), C.c_uint(nt), C.c_double(r_buff), _p(r_listsq, C.c_double),
                          _p(rcut_max, C.c_double), C.byref(mm), C.byref(mn))
    return r_listsq, rcut_max, mm.value, mn.value


_lib.orc_build_head_list.restype = C.c_size_t


def build_head_list(pos, N, Nmax_t):
    pos = _c(pos, np.float64)
    Nmax_t = _c(Nmax_t, np.uint32)
    head = np.zeros(N, dtype=np.uint64)
    total = _lib.orc_build_head_list(_p(pos, C.c_double), C.c_uint(N), _p(Nmax_t, C.c_uint), _p(head, C.c_size_t))
    return head, int(total)


def nlist_binned(pos, N, box, r_cut, r_buff, storage_full=True, n_ghost=0, ghost_width=(0, 0, 0), body=None,
                 filter_body=False, ndim=3, cell=None):
    """Full reference protocol: cell list -> build -> grow Nmax on overflow -> rebuild
    (hoomd/md/NeighborList.cc:406-434).  Returns dict(n_neigh, nlist, head, Nmax, cell)."""
    pos = _c(pos, np.float64)
    r_cut = _c(r_cut, np.float64)
    nt = r_cut.shape[0]
    r_listsq, rcut_max, rmax, _ = update_rlist(r_cut, r_buff)
    if cell is None:
        dim = cell_dimensions(box, rmax + r_buff, 1, ghost_width, ndim)
        cell = cell_list_auto(pos, N, n_ghost, box, dim, ghost_width)
        cell["dim"] = dim
    dim = cell["dim"]
    adj = cell_adj(dim, 1, ndim)
    body_a = _c(body, np.uint32) if body is not None else None
    Nmax_t = np.full(nt, 4, dtype=np.uint32)
    while True:
        head, total = build_head_list(pos, N, Nmax_t)
        nlist = np.zeros(max(total, 1), dtype=np.uint32)
        n_neigh = np.zeros(N, dtype=np.uint32)
        cond = np.zeros(nt, dtype=np.uint32)
        _lib.orc_nlist_binned(_p(pos, C.c_double), _p(body_a, C.c_uint), C.c_uint(N), C.byref(box), _u3(dim),
                              _d3(ghost_width), _p(cell["cell_size"], C.c_uint), _p(cell["xyzf"], C.c_double),
                              _p(adj, C.c_uint), C.c_uint(adj.shape[1]), C.c_uint(cell["Nmax"]),
                              _p(r_cut, C.c_double), _p(r_listsq, C.c_double), C.c_uint(nt),
                              C.c_int(1 if storage_full else 0), C.c_int(1 if filter_body else 0),
                              _p(head, C.c_size_t), _p(Nmax_t, C.c_uint), _p(nlist, C.c_uint),
                              _p(n_neigh, C.c_uint), _p(cond, C.c_uint))
        if _lib.orc_nlist_check_conditions(_p(cond, C.c_uint), _p(Nmax_t, C.c_uint), C.c_uint(nt)):
            continue
        break
    return dict(n_neigh=n_neigh, nlist=nlist, head=head, Nmax=Nmax_t, cell=cell, adj=adj, r_listsq=r_listsq,
                rcut_max=rcut_max)


def neighbor_sets(n_neigh, nlist, head):
    """Order-independent view: list of sorted index arrays, one per particle."""
    return [np.sort(nlist[int(head[i]):int(head[i]) + int(n_neigh[i])]) for i in range(len(n_neigh))]


def distance_check(pos, last_pos, N, box, rcut_max, r_buff, lam=(1.0, 1.0, 1.0)):
    pos = _c(pos, np.float64)
    last_pos = _c(last_pos, np.float64)
    rcut_max = _c(rcut_max, np.float64)
    return bool(_lib.orc_nlist_distance_check(_p(pos, C.c_double), _p(last_pos, C.c_double), C.c_uint(N),
                                              C.byref(box), _p(rcut_max, C.c_double), C.c_double(r_buff),
                                              _d3(lam)))


def nlist_filter(N, head, n_ex_idx, ex_list_idx, n_neigh, nlist):
    """In-place exclusion filter. ex_list_idx has shape (n_ex_max, pitch)."""
    ex_list_idx = _c(ex_list_idx, np.uint32)
    n_ex_idx = _c(n_ex_idx, np.uint32)
    _lib.orc_nlist_filter(C.c_uint(N), _p(head, C.c_size_t), _p(n_ex_idx, C.c_uint), _p(ex_list_idx, C.c_uint),
                          C.c_uint(ex_list_idx.shape[1]), _p(n_neigh, C.c_uint), _p(nlist, C.c_uint))


class NlistState(C.Structure):
    _fields_ = [
        ("last_updated_tstep", C.c_uint64),
        ("last_checked_tstep", C.c_uint64),
        ("last_check_result", C.c_int),
        ("force_update", C.c_int),
        ("rebuild_check_delay", C.c_uint64),
        ("dist_check", C.c_int),
        ("r_buff", C.c_double),
        ("updates", C.c_uint64),
        ("forced_updates", C.c_uint64),
        ("dangerous_updates", C.c_uint64),
        ("update_periods", C.c_uint64 * 100),
    ]

    @classmethod
    def new(cls, r_buff):
        s = cls()
        _lib.orc_nlist_state_init(C.byref(s), C.c_double(r_buff))
        return s

    def needs_updating(self, timestep, distance_check_fn):
        res = C.c_int()
        dang = C.c_int()
        if _lib.orc_needs_updating_begin(C.byref(self), C.c_uint64(timestep), C.byref(res), C.byref(dang)):
            chk = 1 if distance_check_fn() else 0
            return bool(_lib.orc_needs_updating_finish(C.byref(self), C.c_uint64(timestep), C.c_int(chk), dang))
        return bool(res.value)

    @property
    def num_builds(self):
        return self.updates + self.forced_updates

    @property
    def shortest_rebuild(self):
        return _lib.orc_nlist_smallest_rebuild(C.byref(self))


# ---------------------------------------------------------------- pair potential
def eval_pair(ev, rsq, rcutsq, params2, energy_shift):
    f = C.c_double(0.0)
    e = C.c_double(0.0)
    p = (C.c_double * 2)(*params2)
    ok = _lib.orc_eval_pair(C.c_int(ev), C.c_double(rsq), C.c_double(rcutsq), p, C.c_int(int(energy_shift)),
                            C.byref(f), C.byref(e))
    return bool(ok), f.value, e.value


def eval_lrc(ev, rcutsq, params2):
    e = C.c_double()
    pr = C.c_double()
    _lib.orc_eval_lrc(C.c_int(ev), C.c_double(rcutsq), (C.c_double * 2)(*params2), C.byref(e), C.byref(pr))
    return e.value, pr.value


def fene_params(k, r0, epsilon, sigma, delta):
    """hoomd/md/EvaluatorBondFENE.h:47-56: (k, r_0, sigma_6, epsilon_x_4, delta)."""
    return (k, r0, sigma * sigma * sigma * sigma * sigma * sigma, 4.0 * epsilon, delta)


def eval_bond_fene(rsq, params5):
    f = C.c_double(0.0)
    e = C.c_double(0.0)
    ok = _lib.orc_eval_bond_fene(C.c_double(rsq), (C.c_double * 5)(*params5), C.byref(f), C.byref(e))
    return bool(ok), f.value, e.value


def bond_forces_fene(pos, rtag, N, box, bonds, bond_type, params, compute_virial=True, n_total=None):
    """Forces, half energies and virials of FENE + WCA bonds (hoomd/md/PotentialBond.h:165-336).  Raises
    RuntimeError where the reference throws "bond out of bounds"."""
    n_total = pos.shape[0] if n_total is None else n_total
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    rtag = np.ascontiguousarray(rtag, dtype=np.uint32)
    bonds = np.ascontiguousarray(bonds, dtype=np.uint32).reshape(-1, 2)
    bond_type = np.ascontiguousarray(bond_type, dtype=np.uint32)
    prm = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, 5)
    force = np.zeros((n_total, 4))
    virial = np.zeros((6, n_total))
    rc = _lib.orc_bond_forces_fene(_p(pos, C.c_double), _p(rtag, C.c_uint), C.c_uint(N), C.c_uint(n_total), C.byref(box),
                                   _p(bonds, C.c_uint), _p(bond_type, C.c_uint), C.c_uint(len(bonds)), _p(prm, C.c_double),
                                   C.c_int(int(compute_virial)), _p(force, C.c_double), _p(virial, C.c_double),
                                   C.c_size_t(n_total))
    if rc:
        raise RuntimeError("Error in bond calculation: bond %d out of bounds" % (rc - 1))
    return force, virial


def lj_params(sigma, epsilon):
    """hoomd/md/EvaluatorPairLJ.h:136-140."""
    return (sigma * sigma * sigma * sigma * sigma * sigma, 4.0 * epsilon)


def pair_forces(ev, pos, N, box, n_neigh, nlist, head, params, rcutsq, ronsq=None, shift_mode=SHIFT_NONE,
                third_law=False, compute_virial=True, n_total=None):
    pos = _c(pos, np.float64)
    n_total = pos.shape[0] if n_total is None else n_total
    params = _c(params, np.float64)
    rcutsq = _c(rcutsq, np.float64)
    nt = rcutsq.shape[0]
    ronsq = _c(ronsq, np.float64) if ronsq is not None else np.zeros((nt, nt))
    force = np.zeros((n_total, 4))
    virial = np.zeros((6, n_total)) if compute_virial else None
    n_neigh = _c(n_neigh, np.uint32)
    nlist = _c(nlist, np.uint32)
    head = _c(head, np.uint64)
    _lib.orc_pair_forces(C.c_int(ev), _p(pos, C.c_double), C.c_uint(N), C.c_uint(n_total), C.byref(box),
                         _p(n_neigh, C.c_uint), _p(nlist, C.c_uint), _p(head, C.c_size_t), _p(params, C.c_double),
                         _p(rcutsq, C.c_double), _p(ronsq, C.c_double), C.c_uint(nt), C.c_int(shift_mode),
                         C.c_int(int(third_law)), C.c_int(int(compute_virial)), _p(force, C.c_double),
                         _p(virial, C.c_double), C.c_size_t(n_total))
    return force, virial


# ---------------------------------------------------------------- integrator
def step_one(pos, vel, accel, image, box, dt, rescale=1.0, group=None):
    """In place on float64 (N,4)/(N,4)/(N,3) and int32 (N,3) arrays."""
    n = pos.shape[0] if group is None else len(group)
    g = _c(group, np.uint32) if group is not None else None
    _lib.orc_step_one(_p(pos, C.c_double), _p(vel, C.c_double), _p(accel, C.c_double), _p(image, C.c_int),
                      _p(g, C.c_uint), C.c_uint(n), C.byref(box), C.c_double(rescale), C.c_double(dt))


def step_two(vel, accel, net_force, dt, rescale=1.0, group=None):
    n = vel.shape[0] if group is None else len(group)
    g = _c(group, np.uint32) if group is not None else None
    _lib.orc_step_two(_p(vel, C.c_double), _p(accel, C.c_double), _p(g, C.c_uint), C.c_uint(n),
                      _p(net_force, C.c_double), C.c_double(rescale), C.c_double(dt))


def compute_accel(vel, net_force):
    accel = np.zeros((vel.shape[0], 3))
    _lib.orc_compute_accel(_p(vel, C.c_double), _p(net_force, C.c_double), _p(accel, C.c_double),
                           C.c_uint(vel.shape[0]))
    return accel


def sum_net_force(forces, virials=None):
    n = forces[0].shape[0]
    nf = np.zeros((n, 4))
    nv = np.zeros((6, n)) if virials is not None else None
    farr = (C.POINTER(C.c_double) * len(forces))(*[_p(_c(f, np.float64), C.c_double) for f in forces])
    keep = [_c(f, np.float64) for f in forces]
    farr = (C.POINTER(C.c_double) * len(forces))(*[_p(f, C.c_double) for f in keep])
    if virials is not None:
        vkeep = [_c(v, np.float64) for v in virials]
        varr = (C.POINTER(C.c_double) * len(forces))(*[_p(v, C.c_double) for v in vkeep])
        vps = (C.c_size_t * len(forces))(*[n] * len(forces))
    else:
        varr, vps = None, None
    _lib.orc_sum_net_force(_p(nf, C.c_double), _p(nv, C.c_double), C.c_size_t(n), farr, varr, vps,
                           C.c_uint(len(forces)), C.c_uint(n))
    return nf, nv


def thermo(vel, net_force):
    ke = C.c_double()
    pe = C.c_double()
    _lib.orc_thermo(_p(_c(vel, np.float64), C.c_double), _p(_c(net_force, np.float64), C.c_double),
                    C.c_uint(vel.shape[0]), C.byref(ke), C.byref(pe))
    return ke.value, pe.value


# ---------------------------------------------------------------- whole-path driver
_lib.orc_sim_create.restype = C.c_void_p


class Sim:
    """LJ NVE run on the reference's CPU path (half list + third law)."""

    def __init__(self, pos, vel, L, r_cut=2.5, r_buff=0.4, epsilon=1.0, sigma=1.0, dt=0.005,
                 shift_mode=SHIFT_NONE):
        pos = _c(pos, np.float64)
        vel = _c(vel, np.float64)
        self.N = pos.shape[0]
        self._h = C.c_void_p(
            _lib.orc_sim_create(C.c_uint(self.N), _p(pos, C.c_double), _p(vel, C.c_double), _d3(L),
                                C.c_double(r_cut), C.c_double(r_buff), C.c_double(epsilon), C.c_double(sigma),
                                C.c_double(dt), C.c_int(shift_mode)))

    def run(self, nsteps):
        _lib.orc_sim_run(self._h, C.c_uint64(nsteps))

    def get(self):
        pos = np.zeros((self.N, 4))
        vel = np.zeros((self.N, 4))
        img = np.zeros((self.N, 3), dtype=np.int32)
        nf = np.zeros((self.N, 4))
        _lib.orc_sim_get(self._h, _p(pos, C.c_double), _p(vel, C.c_double), _p(img, C.c_int), _p(nf, C.c_double))
        return pos, vel, img, nf

    def energies(self):
        ke = C.c_double()
        pe = C.c_double()
        _lib.orc_sim_energies(self._h, C.byref(ke), C.byref(pe))
        return ke.value, pe.value

    def stats(self):
        nb = C.c_uint64()
        sr = C.c_uint()
        ts = C.c_uint64()
        _lib.orc_sim_stats(self._h, C.byref(nb), C.byref(sr), C.byref(ts))
        return dict(num_builds=nb.value, shortest_rebuild=sr.value, timestep=ts.value)

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.orc_sim_destroy(self._h)
            self._h = None
