"""Minimal driver for counter collection: a few launches of the row and the tile force kernels."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd"), os.path.join(ROOT, "tests")]
import torch
from hoomd_amd import kernels as K, _abi as A
from oracle import oracle as orc
n = (128, 128, 64)
a = (1 / 0.84) ** (1 / 3)
L = np.array(n) * a
g = [(np.arange(k) + 0.5) * a - l / 2 for k, l in zip(n, L)]
z, y, x = np.meshgrid(g[2], g[1], g[0], indexing="ij")
xyz = np.stack([x.ravel(), y.ravel(), z.ravel()], 1) + np.random.default_rng(0).normal(0, 0.12, (x.size, 3))
xyz = (xyz + L / 2) % L - L / 2
N = len(xyz)
gbox, obox = A.Box.from_L(*L), orc.Box.from_L(*L)
pos_t = torch.from_numpy(orc.pack_pos(xyz, np.zeros(N, int))).cuda()
dim = orc.cell_dimensions(obox, 2.9, 1, (0, 0, 0), 3)
cl = K.cell_list_auto(pos_t, N, 0, gbox, dim, want_idx=True)
adj = torch.as_tensor(orc.cell_adj(dim, 1, 3).view(np.int32), device="cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "both"
P = [orc.lj_params(1.0, 1.0)]
if which in ("both", "tile"):
    tl = K.nlist_tiles(pos_t, N, gbox, [[2.5]], 0.4, cl, adj)
    for _ in range(3):
        K.pair_forces_tiles("lj", pos_t, N, gbox, tl, P, [[6.25]], compute_virial=False, fast=True)
if which in ("both", "row"):
    nl = K.nlist_binned(pos_t, N, gbox, [[2.5]], 0.4, cl, adj)
    for _ in range(3):
        K.pair_forces("lj", pos_t, N, gbox, nl["n_neigh"], nl["nlist"], nl["head"], P, [[6.25]], compute_virial=False)
torch.cuda.synchronize()
