"""Rows laid out by type, checked at the kernel level on a Kob-Andersen-like configuration: neighbor sets and forces against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd"), os.path.join(ROOT, "tests")]
import torch
from hoomd_amd import kernels as K, _abi as A
from oracle import oracle as orc
from test_gpu_tiles import tile_nlist, expanded_sets
from test_gpu_kernels import boxes
n = int(sys.argv[1]); jitter = float(sys.argv[2]); rho = float(sys.argv[3]) if len(sys.argv) > 3 else 1.2
a = (1.0 / rho) ** (1.0 / 3.0)
L = n * a
g = (np.arange(n) + 0.5) * a - L / 2
xyz = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
rng = np.random.default_rng(3)
xyz = xyz + rng.normal(0, jitter, xyz.shape) if jitter > 0 else xyz
xyz = (xyz + L / 2) % L - L / 2
typ = (rng.random(len(xyz)) < 0.2).astype(int)
N = len(xyz)
gbox, obox = boxes(A, L)
pos = orc.pack_pos(xyz, typ)
r_cut = [[2.5, 2.0], [2.0, 2.2]]
P = [orc.lj_params(1.0, 1.0), orc.lj_params(0.8, 1.5), orc.lj_params(0.8, 1.5), orc.lj_params(0.88, 0.5)]
pos_t = torch.from_numpy(pos).cuda()
tl, cl, dim = tile_nlist(K, A, pos_t, N, gbox, obox, r_cut, 0.4)
print("dim", dim, "cell Nmax", cl["Nmax"], "Nmax", tl["Nmax"].tolist(), "builds", tl["n_builds"], "rows_by_type", tl["geom"].rows_by_type, "row_pieces", tl["geom"].row_pieces)
ref = orc.nlist_binned(pos, N, obox, r_cut, 0.4, storage_full=True)
nn = tl["n_neigh"].cpu().numpy().view(np.uint32)[:N]
print("n_neigh equal", np.array_equal(nn, ref["n_neigh"]), "max", nn.max())
sets = expanded_sets(K, tl, N, gbox)
rsets = orc.neighbor_sets(ref["n_neigh"], ref["nlist"], ref["head"])
bad = [i for i in range(N) if not np.array_equal(sets[i], rsets[i])]
print("sets differ for", len(bad), "particles", bad[:5])
half = orc.nlist_binned(pos, N, obox, r_cut, 0.4, storage_full=False)
rcutsq = np.square(r_cut)
F_ref, _ = orc.pair_forces(orc.EV_LJ, pos, N, obox, half["n_neigh"], half["nlist"], half["head"], P, rcutsq, np.zeros_like(rcutsq), shift_mode=0, third_law=True, compute_virial=False)
for env in (None, "1"):
    if env:
        os.environ["HOOMD_AMD_PFT_NO_BY_TYPE"] = env
    F, _ = K.pair_forces_tiles("lj", pos_t, N, gbox, tl, P, rcutsq, np.zeros_like(rcutsq), shift_mode=0, compute_virial=False)
    F = F.cpu().numpy()
    err = np.abs(F - F_ref).max(1)
    print("per-pair selects" if env else "by type", "max err", err.max(), "scale", np.abs(F_ref).max(), "sum F", np.abs(F[:, :3].sum(0)).max(), "worst", np.argsort(err)[-3:], "types", typ[np.argsort(err)[-3:]], "nn", nn[np.argsort(err)[-3:]])
