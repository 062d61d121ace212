"""FENE + WCA bond kernels through the C ABI against the CPU oracle (hoomd/md/PotentialBond.h,
hoomd/md/EvaluatorBondFENE.h) and the reference's own test table (hoomd/md/pytest/test_bond.py)."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd")]
from oracle import oracle as orc  # noqa: E402

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
G = os.path.join(HERE, "golden")
F_RTOL = 1e-6  # tolerance the project states for forces, energies and virials


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a MI355X: torch.cuda.is_available() is False")
    from hoomd_amd import kernels
    return kernels


@pytest.fixture(scope="module")
def A():
    from hoomd_amd import _abi
    return _abi


def abi_params(p5):
    """oracle order (k, r_0, sigma_6, epsilon_x_4, delta) -> hmd_fene_params_t order (k, r_0, epsilon_x_4, sigma_6, delta)"""
    return [p5[0], p5[1], p5[3], p5[2], p5[4]]


def tag_tables(n_tags, bonds, bond_type):
    """Per-tag bond table as BondData keeps it: bonds in definition order on both of their ends."""
    per = [[] for _ in range(n_tags)]
    for (a, b), t in zip(bonds, bond_type):
        per[a].append((b, t))
        per[b].append((a, t))
    max_b = max(1, max(len(p) for p in per))
    n = np.array([len(p) for p in per], dtype=np.int32)
    tab = np.zeros((max_b, n_tags, 2), dtype=np.int32)
    for tag, p in enumerate(per):
        for k, e in enumerate(p):
            tab[k, tag] = e
    return n, tab


def gpu_bond_forces(K, pos, tag, rtag, N, gbox, bonds, bond_type, params5, compute_virial=True):
    n_tag, tab = tag_tables(len(rtag), bonds, bond_type)
    dev = "cuda"
    n_bonds, blist, flag = K.bond_table(torch.as_tensor(tag.astype(np.int32), device=dev), torch.as_tensor(rtag.astype(np.int32), device=dev), N,
                                        torch.as_tensor(n_tag, device=dev), torch.as_tensor(tab, device=dev))
    assert flag == 0
    F, V, flag = K.bond_forces_fene(torch.as_tensor(pos, device=dev), N, gbox, n_bonds, blist, [abi_params(p) for p in params5], compute_virial)
    return F.cpu().numpy(), (V.cpu().numpy() if V is not None else None), flag


def test_reference_fenewca_table(K, A):
    with open(os.path.join(G, "bond_fenewca_cases.json")) as f:
        cases = json.load(f)
    gbox = A.Box.from_L(5, 5, 5)
    for c in cases:
        d = c["distance"]
        pos = orc.pack_pos([[-d / 2, 0, 0.1], [d / 2, 0, 0.1]], [0, 0])
        F, V, flag = gpu_bond_forces(K, pos, np.arange(2), np.arange(2), 2, gbox, [(0, 1)], [0], [orc.fene_params(**c["params"])])
        assert flag == 0
        assert F[:, 3].sum() == pytest.approx(c["energy"], rel=1e-2)
        np.testing.assert_allclose(F[0, :3], [c["force"], 0, 0], rtol=1e-2, atol=1e-5)
        np.testing.assert_allclose(F[1, :3], [-c["force"], 0, 0], rtol=1e-2, atol=1e-5)


def chains(n_chains, length, L, seed, tilt=(0.0, 0.0, 0.0)):
    """Random-walk chains with bond lengths near 0.97 wrapped into the box; particle order shuffled."""
    rng = np.random.default_rng(seed)
    obox = orc.Box.from_L(L, L, L, *tilt)
    xyz = []
    bonds = []
    for c in range(n_chains):
        p = rng.uniform(-L / 2, L / 2, 3)
        for k in range(length):
            if k:
                step = rng.normal(size=3)
                p = p + step / np.linalg.norm(step) * rng.uniform(0.88, 1.08)
                bonds.append((c * length + k - 1, c * length + k))
            xyz.append(p.copy())
    xyz = np.array(xyz)
    wrapped = np.array([orc.wrap(obox, r, (0, 0, 0))[0] for r in xyz])
    n = len(xyz)
    order = rng.permutation(n)          # index -> tag
    tag = order
    rtag = np.empty(n, dtype=np.int64)
    rtag[tag] = np.arange(n)
    pos = orc.pack_pos(wrapped[tag], np.zeros(n, dtype=int))
    return pos, tag, rtag, np.array(bonds), obox


@pytest.mark.parametrize("tilt", [(0.0, 0.0, 0.0), (0.2, -0.1, 0.15)])
def test_chain_forces_match_oracle(K, A, tilt):
    L = 14.0
    pos, tag, rtag, bonds, obox = chains(40, 24, L, seed=3, tilt=tilt)
    gbox = A.Box.from_L(L, L, L, *tilt)
    N = len(tag)
    rng = np.random.default_rng(1)
    bond_type = rng.integers(0, 2, len(bonds))
    params = [orc.fene_params(30.0, 1.5, 1.0, 1.0, 0.0), orc.fene_params(22.0, 1.7, 0.8, 1.05, -0.1)]
    F_ref, V_ref = orc.bond_forces_fene(pos, rtag, N, obox, bonds, bond_type, params)
    F, V, flag = gpu_bond_forces(K, pos, tag, rtag, N, gbox, bonds.tolist(), bond_type.tolist(), params)
    assert flag == 0
    scale = np.abs(F_ref[:, :3]).max()
    assert np.abs(F[:, :3] - F_ref[:, :3]).max() <= F_RTOL * scale
    assert np.allclose(F[:, 3], F_ref[:, 3], rtol=F_RTOL, atol=F_RTOL)
    assert np.abs(V - V_ref).max() <= F_RTOL * np.abs(V_ref).max()
    # without the virial the forces are the same and the virial array is not touched
    F2, V2, _ = gpu_bond_forces(K, pos, tag, rtag, N, gbox, bonds.tolist(), bond_type.tolist(), params, compute_virial=False)
    assert np.array_equal(F2, F) and V2 is None


def test_particles_without_bonds_get_zero_and_table_follows_the_order(K, A):
    gbox = A.Box.from_L(10, 10, 10)
    xyz = [[0, 0, 0], [0.95, 0, 0], [3, 3, 3], [0, 1.0, 0]]
    tag = np.array([2, 0, 3, 1])   # index -> tag
    rtag = np.empty(4, dtype=np.int64)
    rtag[tag] = np.arange(4)
    pos = orc.pack_pos(np.array(xyz)[tag], [0] * 4)
    bonds = [(0, 1), (0, 3)]
    params = [orc.fene_params(30.0, 1.5, 1.0, 1.0, 0.0)]
    F_ref, _ = orc.bond_forces_fene(pos, rtag, 4, orc.Box.from_L(10, 10, 10), bonds, [0, 0], params)
    F, V, flag = gpu_bond_forces(K, pos, tag, rtag, 4, gbox, bonds, [0, 0], params)
    assert flag == 0
    assert np.allclose(F, F_ref, rtol=F_RTOL, atol=1e-12)
    assert np.all(F[rtag[2]] == 0) and np.all(V[:, rtag[2]] == 0)


def test_overstretched_and_incomplete_bonds_raise_their_flags(K, A):
    gbox = A.Box.from_L(10, 10, 10)
    pos = orc.pack_pos([[0, 0, 0], [1.6, 0, 0]], [0, 0])
    _, _, flag = gpu_bond_forces(K, pos, np.arange(2), np.arange(2), 2, gbox, [(0, 1)], [0], [orc.fene_params(30.0, 1.5, 1.0, 1.0, 0.0)])
    assert flag == 1
    # a partner whose rtag points past the local arrays: the table kernel reports it
    n_tag, tab = tag_tables(2, [(0, 1)], [0])
    rtag = torch.tensor([0, 0x7FFFFFFF], dtype=torch.int32, device="cuda")
    tag = torch.tensor([0], dtype=torch.int32, device="cuda")
    _, _, tflag = K.bond_table(tag, rtag, 1, torch.as_tensor(n_tag, device="cuda"), torch.as_tensor(tab, device="cuda"))
    assert tflag == 1


def test_bond_entry_points_reject_bad_arguments(K, A):
    import ctypes as C
    a = A.BondArgs()
    assert A.lib.hmd_compute_bond_forces_fene(C.byref(a), None, None) == A.ERR_INVALID_ARGUMENT
    t = A.BondTableArgs()
    assert A.lib.hmd_update_bond_table(C.byref(t), None) == A.ERR_INVALID_ARGUMENT
    assert A.lib.hmd_update_bond_table(None, None) == A.ERR_INVALID_ARGUMENT
