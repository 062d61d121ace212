import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd"), os.path.join(ROOT, "tests")]
import torch
from hoomd_amd import kernels as K, _abi as A
from oracle import oracle as orc
import test_gpu_prefilter as T
L = (21.0, 21.0, 21.0)
r_list = 2.9
n_free = int(0.6 * L[0] * L[1] * L[2])
xyz = T._system(1, L, n_free, 4000, r_list)
if len(sys.argv) > 1 and sys.argv[1] == "plain":
    xyz = xyz[:n_free]
N = len(xyz)
pos = orc.pack_pos(xyz, np.zeros(N, dtype=int))
obox, gbox = orc.Box.from_L(*L), A.Box.from_L(*L)
dim = orc.cell_dimensions(obox, r_list)
ref = orc.nlist_binned(pos, N, obox, [[2.5]], 0.4, storage_full=True)
want = orc.neighbor_sets(ref["n_neigh"], ref["nlist"], ref["head"])
pos_t = torch.from_numpy(pos).cuda()
cl = K.cell_list_auto(pos_t, N, 0, gbox, dim, want_idx=True)
print("dim", dim, "cell Nmax", cl["Nmax"], "max cell", int(cl["cell_size"].max()))
adj = torch.as_tensor(orc.cell_adj(dim).view(np.int32), device="cuda")
nl = K.nlist_binned(pos_t, N, gbox, [[2.5]], 0.4, cl, adj)
nn = nl["n_neigh"].cpu().numpy().view(np.uint32)[:N]
got = orc.neighbor_sets(nn, nl["nlist"].cpu().numpy().view(np.uint32), nl["head"].cpu().numpy().view(np.uint64)[:N])
bad = np.nonzero(nn != ref["n_neigh"])[0]
print("mismatching n_neigh:", len(bad), "of", N, "Nmax", nl["Nmax"].tolist(), ref["Nmax"])
Lb = np.asarray(L)
for i in bad[:5]:
    g, w = set(got[i].tolist()), set(want[i].tolist())
    extra, missing = sorted(g - w), sorted(w - g)
    print("particle", i, "n", nn[i], ref["n_neigh"][i], "len(set got)", len(g), "extra", extra[:8], "missing", missing[:8])
    for j in extra[:4]:
        dx = xyz[j] - xyz[i]; dx -= Lb * np.rint(dx / Lb)
        print("    extra", j, "r", np.sqrt((dx * dx).sum()))
    # duplicates in the row?
    row = nl["nlist"].cpu().numpy().view(np.uint32)[int(nl["head"][i]):int(nl["head"][i]) + int(nn[i])]
    u, c = np.unique(row, return_counts=True)
    print("    duplicates:", u[c > 1][:8], c[c > 1][:8])
