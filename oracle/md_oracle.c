/* md_oracle.c -- TEST INFRASTRUCTURE ONLY (see md_oracle.h).
 *
 * CPU restatement, in plain C and IEEE double arithmetic, of the reference's CPU code path.
 * Build with -ffp-contract=off so that no multiply-add is fused: the reference's CPU build
 * targets baseline x86-64 and therefore rounds every product and sum separately.
 */
#include "md_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* helpers                                                                              */
/* ------------------------------------------------------------------------------------ */

/* hoomd/HOOMDMath.h:196-217 : the integer lives in the low 4 bytes of the double */
static inline int d_as_int(double v)
    {
    int r;
    memcpy(&r, &v, sizeof(int));
    return r;
    }

static inline double int_as_d(int a)
    {
    double v = 0.0;
    memcpy(&v, &a, sizeof(int));
    return v;
    }

/* ------------------------------------------------------------------------------------ */
/* box: hoomd/BoxDim.h                                                                  */
/* ------------------------------------------------------------------------------------ */

/* hoomd/BoxDim.h:174-187 (setL) and :125-130 (tilted constructor) */
void orc_box_init(orc_box_t* b, double Lx, double Ly, double Lz, double xy, double xz, double yz)
    {
    const double L[3] = {Lx, Ly, Lz};
    for (int d = 0; d < 3; d++)
        {
        b->hi[d] = L[d] / 2.0;
        b->lo[d] = -b->hi[d];
        b->Linv[d] = 1.0 / L[d];
        b->L[d] = L[d];
        b->periodic[d] = 1;
        }
    if (Lz == 0.0)
        b->Linv[2] = 0.0;
    b->xy = xy;
    b->xz = xz;
    b->yz = yz;
    }

/* hoomd/BoxDim.h:209-221 (setLoHi) */
void orc_box_set_lo_hi(orc_box_t* b, const double lo[3], const double hi[3])
    {
    for (int d = 0; d < 3; d++)
        {
        b->hi[d] = hi[d];
        b->lo[d] = lo[d];
        b->Linv[d] = 1.0 / (hi[d] - lo[d]);
        b->L[d] = hi[d] - lo[d];
        }
    if (hi[2] == 0.0 && lo[2] == 0.0)
        b->Linv[2] = 0.0;
    }

/* hoomd/BoxDim.h:536-552 (fast::rsqrt on a double is 1.0/sqrt on the host,
   hoomd/HOOMDMath.h:528-535) */
void orc_box_nearest_plane_distance(const orc_box_t* b, double out[3])
    {
    const double t = b->xy * b->yz - b->xz;
    out[0] = b->L[0] * (1.0 / sqrt(1.0 + b->xy * b->xy + t * t));
    out[1] = b->L[1] * (1.0 / sqrt(1.0 + b->yz * b->yz));
    out[2] = b->L[2];
    if (b->L[2] == 0.0)
        out[2] = 1.0;
    }

/* hoomd/BoxDim.h:260-268 */
void orc_box_make_fraction(const orc_box_t* b, const double v[3], const double gw[3], double out[3])
    {
    double delta[3] = {v[0] - b->lo[0], v[1] - b->lo[1], v[2] - b->lo[2]};
    delta[0] -= (b->xz - b->yz * b->xy) * v[2] + b->xy * v[1];
    delta[1] -= b->yz * v[2];
    double npd[3];
    orc_box_nearest_plane_distance(b, npd);
    for (int d = 0; d < 3; d++)
        {
        const double gf = gw[d] / npd[d];
        out[d] = (delta[d] * b->Linv[d] + gf) / (1.0 + 2.0 * gf);
        }
    }

/* hoomd/BoxDim.h:326-373 (host branch of minImage) */
void orc_box_min_image(const orc_box_t* b, double w[3])
    {
    const double* L = b->L;
    if (b->periodic[2])
        {
        if (w[2] >= b->hi[2])
            {
            w[2] -= L[2];
            w[1] -= L[2] * b->yz;
            w[0] -= L[2] * b->xz;
            }
        else if (w[2] < b->lo[2])
            {
            w[2] += L[2];
            w[1] += L[2] * b->yz;
            w[0] += L[2] * b->xz;
            }
        }
    if (b->periodic[1])
        {
        if (w[1] >= b->hi[1])
            {
            int i = (int)(w[1] * b->Linv[1] + 0.5);
            w[1] -= (double)i * L[1];
            w[0] -= (double)i * L[1] * b->xy;
            }
        else if (w[1] < b->lo[1])
            {
            int i = (int)(-w[1] * b->Linv[1] + 0.5);
            w[1] += (double)i * L[1];
            w[0] += (double)i * L[1] * b->xy;
            }
        }
    if (b->periodic[0])
        {
        if (w[0] >= b->hi[0])
            {
            int i = (int)(w[0] * b->Linv[0] + 0.5);
            w[0] -= (double)i * L[0];
            }
        else if (w[0] < b->lo[0])
            {
            int i = (int)(-w[0] * b->Linv[0] + 0.5);
            w[0] += (double)i * L[0];
            }
        }
    }

/* hoomd/BoxDim.h:306-325 (device branch of minImage: round-to-nearest image count) */
void orc_box_min_image_rint(const orc_box_t* b, double w[3])
    {
    const double* L = b->L;
    if (b->periodic[2])
        {
        double img = rint(w[2] * b->Linv[2]);
        w[2] -= L[2] * img;
        w[1] -= L[2] * b->yz * img;
        w[0] -= L[2] * b->xz * img;
        }
    if (b->periodic[1])
        {
        double img = rint(w[1] * b->Linv[1]);
        w[1] -= L[1] * img;
        w[0] -= L[1] * b->xy * img;
        }
    if (b->periodic[0])
        w[0] -= L[0] * rint(w[0] * b->Linv[0]);
    }

/* hoomd/BoxDim.h:391-458 with flags = (0,0,0) */
void orc_box_wrap(const orc_box_t* b, double w[3], int img[3])
    {
    const double* L = b->L;
    double origin[3];
    for (int d = 0; d < 3; d++)
        origin[d] = (b->hi[d] + b->lo[d]) / 2.0;
    if (!b->periodic[1])
        origin[1] = 0.0;
    if (!b->periodic[2])
        origin[2] = 0.0;

    if (b->periodic[0])
        {
        double tilt_x = (b->xz - b->xy * b->yz) * (w[2] - origin[2]) + b->xy * (w[1] - origin[1]);
        if (w[0] >= b->hi[0] + tilt_x)
            {
            w[0] -= L[0];
            img[0]++;
            }
        else if (w[0] < b->lo[0] + tilt_x)
            {
            w[0] += L[0];
            img[0]--;
            }
        }
    if (b->periodic[1])
        {
        double tilt_y = b->yz * (w[2] - origin[2]);
        if (w[1] >= b->hi[1] + tilt_y)
            {
            w[1] -= L[1];
            w[0] -= L[1] * b->xy;
            img[1]++;
            }
        else if (w[1] < b->lo[1] + tilt_y)
            {
            w[1] += L[1];
            w[0] += L[1] * b->xy;
            img[1]--;
            }
        }
    if (b->periodic[2])
        {
        if (w[2] >= b->hi[2])
            {
            w[2] -= L[2];
            w[1] -= L[2] * b->yz;
            w[0] -= L[2] * b->xz;
            img[2]++;
            }
        else if (w[2] < b->lo[2])
            {
            w[2] += L[2];
            w[1] += L[2] * b->yz;
            w[0] += L[2] * b->xz;
            img[2]--;
            }
        }
    }

/* hoomd/Index1D.h:51-54 */
unsigned int orc_index2d(unsigned int w, unsigned int i, unsigned int j)
    {
    return j * w + i;
    }

/* hoomd/Index1D.h:119-122 */
unsigned int orc_index3d(unsigned int w, unsigned int h, unsigned int i, unsigned int j, unsigned int k)
    {
    return (k * h + j) * w + i;
    }

/* ------------------------------------------------------------------------------------ */
/* cell list: hoomd/CellList.cc                                                         */
/* ------------------------------------------------------------------------------------ */

/* hoomd/CellList.cc:74-79, :83-131 */
void orc_cell_dimensions(const orc_box_t* box,
                         double nominal_width,
                         unsigned int multiple,
                         const double gw[3],
                         unsigned int ndim,
                         unsigned int dim[3])
    {
    double L[3];
    orc_box_nearest_plane_distance(box, L);
    for (int d = 0; d < 3; d++)
        {
        unsigned int v = (unsigned int)((L[d] + 2.0 * gw[d]) / nominal_width);
        dim[d] = (v / multiple) * multiple;
        }
    if (ndim != 3)
        dim[2] = 1;
    for (int d = 0; d < 3; d++)
        if (dim[d] == 0)
            dim[d] = 1;
    }

/* hoomd/CellList.cc:257-273 */
unsigned int orc_cell_n_adj(const unsigned int dim[3], unsigned int radius, unsigned int ndim)
    {
    unsigned int u[3];
    for (int d = 0; d < 3; d++)
        u[d] = dim[d] > radius * 2 + 1 ? radius * 2 + 1 : dim[d];
    return ndim == 2 ? u[0] * u[1] : u[0] * u[1] * u[2];
    }

static int cmp_uint(const void* a, const void* b)
    {
    unsigned int x = *(const unsigned int*)a, y = *(const unsigned int*)b;
    return (x > y) - (x < y);
    }

/* hoomd/CellList.cc:346-402 : per cell, the sorted, de-duplicated, periodically wrapped
   list of the (2r+1)^d surrounding cells, written at adj[n_adj*cell + k] */
void orc_cell_adj(const unsigned int dim[3], unsigned int radius, unsigned int ndim, unsigned int* adj_out)
    {
    const unsigned int n_adj = orc_cell_n_adj(dim, radius, ndim);
    const int r = (int)radius;
    const int rk = (ndim == 2) ? 0 : r;
    const int mx = (int)dim[0], my = (int)dim[1], mz = (int)dim[2];
    const int cap = (2 * r + 1) * (2 * r + 1) * (2 * rk + 1);
    unsigned int* tmp = (unsigned int*)malloc(sizeof(unsigned int) * (size_t)cap);

    for (int k = 0; k < mz; k++)
        for (int j = 0; j < my; j++)
            for (int i = 0; i < mx; i++)
                {
                const unsigned int cur = orc_index3d(dim[0], dim[1], i, j, k);
                int n = 0;
                for (int nk = k - rk; nk <= k + rk; nk++)
                    for (int nj = j - r; nj <= j + r; nj++)
                        for (int ni = i - r; ni <= i + r; ni++)
                            {
                            int wi = ni % mx, wj = nj % my, wk = nk % mz;
                            if (wi < 0)
                                wi += mx;
                            if (wj < 0)
                                wj += my;
                            if (wk < 0)
                                wk += mz;
                            tmp[n++] = orc_index3d(dim[0], dim[1], wi, wj, wk);
                            }
                qsort(tmp, (size_t)n, sizeof(unsigned int), cmp_uint);
                int m = 0;
                for (int q = 0; q < n; q++)
                    if (q == 0 || tmp[q] != tmp[q - 1])
                        tmp[m++] = tmp[q];
                /* m == n_adj by construction (hoomd/CellList.cc:257-275) */
                for (int q = 0; q < m && q < (int)n_adj; q++)
                    adj_out[(size_t)n_adj * cur + q] = tmp[q];
                }
    free(tmp);
    }

/* hoomd/CellList.cc:404-547 */
void orc_compute_cell_list(const orc_d4* pos,
                           const double* charge,
                           const unsigned int* body,
                           unsigned int N,
                           unsigned int n_ghost,
                           const orc_box_t* box,
                           const unsigned int dim[3],
                           unsigned int Nmax,
                           const double gw[3],
                           int flag_mode,
                           unsigned int* cell_size,
                           orc_d4* xyzf,
                           orc_u2* type_body,
                           unsigned int* cell_idx,
                           unsigned int cond[3])
    {
    const unsigned int ncells = dim[0] * dim[1] * dim[2];
    cond[0] = cond[1] = cond[2] = 0;
    memset(cell_size, 0, sizeof(unsigned int) * ncells);

    const unsigned int ntot = N + n_ghost;
    for (unsigned int n = 0; n < ntot; n++)
        {
        const double p[3] = {pos[n].x, pos[n].y, pos[n].z};
        if (isnan(p[0]) || isnan(p[1]) || isnan(p[2]))
            {
            cond[1] = n + 1;
            continue;
            }
        double f[3];
        orc_box_make_fraction(box, p, gw, f);
        int ib = (int)(f[0] * dim[0]);
        int jb = (int)(f[1] * dim[1]);
        int kb = (int)(f[2] * dim[2]);

        if ((f[0] < -0.00001 || f[0] >= 1.00001) || (f[1] < -0.00001 || f[1] >= 1.00001)
            || (f[2] < -0.00001 || f[2] >= 1.00001))
            {
            if (n < N)
                cond[2] = n + 1;
            continue;
            }

        if (ib == (int)dim[0] && box->periodic[0])
            ib = 0;
        if (jb == (int)dim[1] && box->periodic[1])
            jb = 0;
        if (kb == (int)dim[2] && box->periodic[2])
            kb = 0;

        if (ib < 0 || ib >= (int)dim[0] || jb < 0 || jb >= (int)dim[1] || kb < 0 || kb >= (int)dim[2])
            {
            if (n < N)
                cond[2] = n + 1;
            continue;
            }
        const unsigned int bin = orc_index3d(dim[0], dim[1], (unsigned)ib, (unsigned)jb, (unsigned)kb);

        double flag;
        if (flag_mode == 1)
            flag = charge[n];
        else if (flag_mode == 2)
            flag = pos[n].w;
        else
            flag = int_as_d((int)n);

        const unsigned int offset = cell_size[bin];
        if (offset < Nmax)
            {
            const size_t slot = (size_t)orc_index2d(Nmax, offset, bin);
            if (xyzf)
                {
                xyzf[slot].x = p[0];
                xyzf[slot].y = p[1];
                xyzf[slot].z = p[2];
                xyzf[slot].w = flag;
                }
            if (type_body)
                {
                type_body[slot].x = (unsigned int)d_as_int(pos[n].w);
                type_body[slot].y = body ? body[n] : ORC_NO_BODY;
                }
            if (cell_idx)
                cell_idx[slot] = n;
            }
        else
            {
            if (offset + 1 > cond[0])
                cond[0] = offset + 1;
            }
        cell_size[bin]++;
        }
    }

/* ------------------------------------------------------------------------------------ */
/* neighbor list: hoomd/md/NeighborList.cc, hoomd/md/NeighborListBinned.cc              */
/* ------------------------------------------------------------------------------------ */

/* hoomd/md/NeighborList.cc:452-529 (single consumer matrix already folded into r_cut) */
void orc_update_rlist(const double* r_cut,
                      unsigned int ntypes,
                      double r_buff,
                      double* r_listsq,
                      double* rcut_max,
                      double* rcut_max_max,
                      double* rcut_min)
    {
    double gmax = 0.0;
    for (unsigned int i = 0; i < ntypes; i++)
        {
        double mi = 0.0;
        for (unsigned int j = 0; j < ntypes; j++)
            {
            const double rc = r_cut[orc_index2d(ntypes, i, j)];
            if (rc > mi)
                mi = rc;
            const double rl = (rc > 0.0) ? rc + r_buff : 0.0;
            r_listsq[orc_index2d(ntypes, i, j)] = rl * rl;
            }
        rcut_max[i] = mi;
        if (mi > gmax)
            gmax = mi;
        }
    *rcut_max_max = gmax;
    double gmin = gmax;
    for (unsigned int p = 0; p < ntypes * ntypes; p++)
        if (r_cut[p] > 0.0 && r_cut[p] < gmin)
            gmin = r_cut[p];
    *rcut_min = gmin;
    }

/* hoomd/md/NeighborList.cc:1544-1565 */
size_t orc_build_head_list(const orc_d4* pos, unsigned int N, const unsigned int* Nmax, size_t* head_list)
    {
    size_t head = 0;
    for (unsigned int i = 0; i < N; i++)
        {
        head_list[i] = head;
        head += Nmax[(unsigned int)d_as_int(pos[i].w)];
        }
    return head;
    }

/* hoomd/md/NeighborList.cc:1601-1617 */
int orc_nlist_check_conditions(const unsigned int* conditions, unsigned int* Nmax, unsigned int ntypes)
    {
    int overflow = 0;
    for (unsigned int t = 0; t < ntypes; t++)
        if (conditions[t] > Nmax[t])
            {
            Nmax[t] = (conditions[t] > 4) ? (conditions[t] + 3) & ~3u : 4;
            overflow = 1;
            }
    return overflow;
    }

/* hoomd/md/NeighborListBinned.cc:36-179 */
void orc_nlist_binned(const orc_d4* pos,
                      const unsigned int* body,
                      unsigned int N,
                      const orc_box_t* box,
                      const unsigned int dim[3],
                      const double gw[3],
                      const unsigned int* cell_size,
                      const orc_d4* cell_xyzf,
                      const unsigned int* cell_adj,
                      unsigned int n_adj,
                      unsigned int cell_Nmax,
                      const double* r_cut,
                      const double* r_listsq,
                      unsigned int ntypes,
                      int storage_full,
                      int filter_body,
                      const size_t* head_list,
                      const unsigned int* Nmax,
                      unsigned int* nlist,
                      unsigned int* n_neigh,
                      unsigned int* conditions)
    {
    for (unsigned int i = 0; i < N; i++)
        {
        unsigned int cnt = 0;
        const double my[3] = {pos[i].x, pos[i].y, pos[i].z};
        const unsigned int ti = (unsigned int)d_as_int(pos[i].w);
        const unsigned int body_i = body ? body[i] : ORC_NO_BODY;
        const unsigned int Nmax_i = Nmax[ti];
        const size_t head = head_list[i];

        double f[3];
        orc_box_make_fraction(box, my, gw, f);
        int ib = (int)(unsigned int)(f[0] * dim[0]);
        int jb = (int)(unsigned int)(f[1] * dim[1]);
        int kb = (int)(unsigned int)(f[2] * dim[2]);
        if (ib == (int)dim[0] && box->periodic[0])
            ib = 0;
        if (jb == (int)dim[1] && box->periodic[1])
            jb = 0;
        if (kb == (int)dim[2] && box->periodic[2])
            kb = 0;
        const unsigned int my_cell = orc_index3d(dim[0], dim[1], (unsigned)ib, (unsigned)jb, (unsigned)kb);

        for (unsigned int a = 0; a < n_adj; a++)
            {
            const unsigned int nc = cell_adj[(size_t)n_adj * my_cell + a];
            const unsigned int sz = cell_size[nc];
            for (unsigned int s = 0; s < sz; s++)
                {
                const orc_d4* c = &cell_xyzf[(size_t)orc_index2d(cell_Nmax, s, nc)];
                const unsigned int j = (unsigned int)d_as_int(c->w);
                const unsigned int tj = (unsigned int)d_as_int(pos[j].w);
                const double rc = r_cut[orc_index2d(ntypes, ti, tj)];

                int excluded = (i == j) || (rc <= 0.0);
                if (filter_body && body_i != ORC_NO_BODY)
                    excluded = excluded | (body_i == body[j]);
                if (excluded)
                    continue;

                double dx[3] = {my[0] - c->x, my[1] - c->y, my[2] - c->z};
                orc_box_min_image(box, dx);
                const double drsq = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];

                if (drsq <= r_listsq[orc_index2d(ntypes, ti, tj)])
                    {
                    if (storage_full || i < j)
                        {
                        if (cnt < Nmax_i)
                            nlist[head + cnt] = j;
                        else if (cnt + 1 > conditions[ti])
                            conditions[ti] = cnt + 1;
                        cnt++;
                        }
                    }
                }
            }
        n_neigh[i] = cnt;
        }
    }

/* hoomd/md/NeighborList.cc:1220-1289 (single rank) */
int orc_nlist_distance_check(const orc_d4* pos,
                             const orc_d4* last_pos,
                             unsigned int N,
                             const orc_box_t* box,
                             const double* rcut_max,
                             double r_buff,
                             const double lambda[3])
    {
    double lmin = (lambda[0] < lambda[1]) ? lambda[0] : lambda[1];
    lmin = (lmin < lambda[2]) ? lmin : lambda[2];
    for (unsigned int i = 0; i < N; i++)
        {
        const unsigned int ti = (unsigned int)d_as_int(pos[i].w);
        const double old_rmin = rcut_max[ti];
        const double rmax = old_rmin + r_buff;
        const double delta_max = (rmax * lmin - old_rmin) / 2.0;
        const double maxsq = (delta_max > 0) ? delta_max * delta_max : 0;
        double dx[3] = {pos[i].x - lambda[0] * last_pos[i].x,
                        pos[i].y - lambda[1] * last_pos[i].y,
                        pos[i].z - lambda[2] * last_pos[i].z};
        orc_box_min_image(box, dx);
        if (dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2] >= maxsq)
            return 1;
        }
    return 0;
    }

/* hoomd/md/NeighborList.cc:1293-1311 */
void orc_nlist_set_last_pos(const orc_d4* pos, unsigned int N, orc_d4* last_pos)
    {
    for (unsigned int i = 0; i < N; i++)
        {
        last_pos[i].x = pos[i].x;
        last_pos[i].y = pos[i].y;
        last_pos[i].z = pos[i].z;
        last_pos[i].w = 0.0;
        }
    }

/* hoomd/md/NeighborList.cc:1489-1536 ; ex_list_idx[ex_pitch*k + idx] is exclusion k of idx */
void orc_nlist_filter(unsigned int N,
                      const size_t* head_list,
                      const unsigned int* n_ex_idx,
                      const unsigned int* ex_list_idx,
                      unsigned int ex_pitch,
                      unsigned int* n_neigh,
                      unsigned int* nlist)
    {
    for (unsigned int idx = 0; idx < N; idx++)
        {
        const size_t head = head_list[idx];
        const unsigned int nn = n_neigh[idx];
        const unsigned int nex = n_ex_idx[idx];
        unsigned int kept = 0;
        for (unsigned int k = 0; k < nn; k++)
            {
            const unsigned int j = nlist[head + k];
            int excluded = 0;
            for (unsigned int e = 0; e < nex; e++)
                if (ex_list_idx[(size_t)ex_pitch * e + idx] == j)
                    {
                    excluded = 1;
                    break;
                    }
            if (!excluded)
                nlist[head + kept++] = j;
            }
        n_neigh[idx] = kept;
        }
    }

/* hoomd/md/NeighborList.cc:32-50 (constructor: last_updated = last_checked = 0, forced
   update pending, distance check on); rebuild_check_delay = 1 is what the Python wrapper
   sets (hoomd/md/nlist.py:378) */
void orc_nlist_state_init(orc_nlist_state_t* s, double r_buff)
    {
    memset(s, 0, sizeof(*s));
    s->rebuild_check_delay = 1;
    s->dist_check = 1;
    s->r_buff = r_buff;
    s->force_update = 1;
    }

/* hoomd/md/NeighborList.cc:1325-1387 : everything up to (not including) distanceCheck */
int orc_needs_updating_begin(orc_nlist_state_t* s, uint64_t t, int* result, int* dangerous)
    {
    *dangerous = 0;
    if (s->last_checked_tstep == t)
        {
        if (s->force_update)
            {
            s->force_update = 0;
            *result = 1;
            return 0;
            }
        *result = s->last_check_result;
        return 0;
        }
    s->last_checked_tstep = t;

    if (!s->force_update && t < s->last_updated_tstep + s->rebuild_check_delay)
        {
        s->last_check_result = 0;
        *result = 0;
        return 0;
        }

    if (s->dist_check && s->rebuild_check_delay > 1 && t == s->last_updated_tstep + s->rebuild_check_delay)
        *dangerous = 1;

    if (s->force_update)
        {
        s->force_update = 0;
        s->forced_updates += 1;
        s->last_updated_tstep = t;
        s->last_check_result = 1;
        *dangerous = 0;
        *result = 1;
        return 0;
        }

    if (s->r_buff < 1e-6
        || (!s->dist_check
            && (s->rebuild_check_delay == 0
                || (s->rebuild_check_delay > 1 && t == s->last_updated_tstep + s->rebuild_check_delay))))
        {
        *result = orc_needs_updating_finish(s, t, 1, *dangerous);
        return 0;
        }
    return 1;
    }

/* hoomd/md/NeighborList.cc:1389-1418 */
int orc_needs_updating_finish(orc_nlist_state_t* s, uint64_t t, int check_result, int dangerous)
    {
    if (check_result)
        {
        if (t > s->last_updated_tstep)
            {
            uint64_t period = t - s->last_updated_tstep;
            if (period >= 100)
                period = 99;
            s->update_periods[period]++;
            }
        s->last_updated_tstep = t;
        s->updates += 1;
        }
    if (check_result && dangerous)
        s->dangerous_updates += 1;
    s->last_check_result = check_result;
    return check_result;
    }

/* hoomd/md/NeighborList.cc:1428-1436 */
unsigned int orc_nlist_smallest_rebuild(const orc_nlist_state_t* s)
    {
    for (unsigned int i = 0; i < 100; i++)
        if (s->update_periods[i] != 0)
            return i;
    return 100;
    }

/* ------------------------------------------------------------------------------------ */
/* evaluators                                                                           */
/* ------------------------------------------------------------------------------------ */

/* hoomd/md/EvaluatorPairLJ.h:162-166,190-211 and
   example_plugins/pair_plugin/EvaluatorPairExample.h:83-86,110-133 */
int orc_eval_pair(int ev, double rsq, double rcutsq, const double* p, int energy_shift, double* fdr, double* eng)
    {
    if (ev == ORC_EV_LJ)
        {
        const double sigma_6 = p[0], eps_x_4 = p[1];
        const double lj1 = eps_x_4 * sigma_6 * sigma_6;
        const double lj2 = eps_x_4 * sigma_6;
        if (rsq < rcutsq && lj1 != 0)
            {
            const double r2inv = 1.0 / rsq;
            const double r6inv = r2inv * r2inv * r2inv;
            *fdr = r2inv * r6inv * (12.0 * lj1 * r6inv - 6.0 * lj2);
            *eng = r6inv * (lj1 * r6inv - lj2);
            if (energy_shift)
                {
                const double rc2inv = 1.0 / rcutsq;
                const double rc6inv = rc2inv * rc2inv * rc2inv;
                *eng -= rc6inv * (lj1 * rc6inv - lj2);
                }
            return 1;
            }
        return 0;
        }
    else
        {
        const double k = p[0], sigma = p[1];
        if (rsq < rcutsq)
            {
            const double r = sqrt(rsq);
            const double rinv = 1 / r;
            const double overlap = sigma - r;
            *fdr = k * overlap * rinv;
            *eng = 0.5 * k * overlap * overlap;
            if (energy_shift)
                {
                const double rcut = sqrt(rcutsq);
                const double co = sigma - rcut;
                *eng -= 0.5 * k * co * co;
                }
            return 1;
            }
        return 0;
        }
    }

/* hoomd/md/EvaluatorPairLJ.h:213-248 ; EvaluatorPairExample.h:136-145 */
void orc_eval_lrc(int ev, double rcutsq, const double* p, double* e_int, double* p_int)
    {
    *e_int = 0.0;
    *p_int = 0.0;
    if (ev != ORC_EV_LJ || rcutsq == 0)
        return;
    const double lj1 = p[1] * p[0] * p[0];
    const double lj2 = p[1] * p[0];
    const double rc3inv = 1.0 / pow(rcutsq, 1.5);
    const double rc9inv = rc3inv * rc3inv * rc3inv;
    *p_int = lj1 * 4.0 / 3.0 * rc9inv - lj2 * 2.0 * rc3inv;
    *e_int = lj1 / 9.0 * rc9inv - lj2 / 3.0 * rc3inv;
    }

/* hoomd/md/PotentialPair.h:587-791 (tail correction excluded) */
void orc_pair_forces(int ev,
                     const orc_d4* pos,
                     unsigned int N,
                     unsigned int n_total,
                     const orc_box_t* box,
                     const unsigned int* n_neigh,
                     const unsigned int* nlist,
                     const size_t* head_list,
                     const double* params,
                     const double* rcutsq_t,
                     const double* ronsq_t,
                     unsigned int ntypes,
                     int shift_mode,
                     int third_law,
                     int compute_virial,
                     orc_d4* force,
                     double* virial,
                     size_t vp)
    {
    memset(force, 0, sizeof(orc_d4) * n_total);
    if (virial)
        memset(virial, 0, sizeof(double) * 6 * vp);

    for (unsigned int i = 0; i < N; i++)
        {
        const double pi[3] = {pos[i].x, pos[i].y, pos[i].z};
        const unsigned int ti = (unsigned int)d_as_int(pos[i].w);
        double fi[3] = {0, 0, 0};
        double pei = 0.0;
        double vi[6] = {0, 0, 0, 0, 0, 0};

        const size_t head = head_list[i];
        const unsigned int sz = n_neigh[i];
        for (unsigned int k = 0; k < sz; k++)
            {
            const unsigned int j = nlist[head + k];
            double dx[3] = {pi[0] - pos[j].x, pi[1] - pos[j].y, pi[2] - pos[j].z};
            const unsigned int tj = (unsigned int)d_as_int(pos[j].w);
            orc_box_min_image(box, dx);
            const double rsq = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];

            const unsigned int tp = orc_index2d(ntypes, ti, tj);
            const double rcutsq = rcutsq_t[tp];
            double ronsq = 0.0;
            if (shift_mode == ORC_SHIFT_XPLOR)
                ronsq = ronsq_t[tp];
            int energy_shift = 0;
            if (shift_mode == ORC_SHIFT_SHIFT)
                energy_shift = 1;
            else if (shift_mode == ORC_SHIFT_XPLOR && ronsq > rcutsq)
                energy_shift = 1;

            double fdr = 0.0, eng = 0.0;
            if (!orc_eval_pair(ev, rsq, rcutsq, &params[2 * tp], energy_shift, &fdr, &eng))
                continue;

            if (shift_mode == ORC_SHIFT_XPLOR && rsq >= ronsq && rsq < rcutsq)
                {
                const double old_eng = eng, old_fdr = fdr;
                const double denom_inv = 1.0 / ((rcutsq - ronsq) * (rcutsq - ronsq) * (rcutsq - ronsq));
                const double d = rsq - rcutsq;
                const double s = d * d * (rcutsq + 2.0 * rsq - 3.0 * ronsq) * denom_inv;
                const double ds_dr_divr = 12.0 * (rsq - ronsq) * d * denom_inv;
                eng = old_eng * s;
                fdr = s * old_fdr - ds_dr_divr * old_eng;
                }

            const double f2 = fdr * 0.5;
            fi[0] += dx[0] * fdr;
            fi[1] += dx[1] * fdr;
            fi[2] += dx[2] * fdr;
            pei += eng * 0.5;
            if (compute_virial)
                {
                vi[0] += f2 * dx[0] * dx[0];
                vi[1] += f2 * dx[0] * dx[1];
                vi[2] += f2 * dx[0] * dx[2];
                vi[3] += f2 * dx[1] * dx[1];
                vi[4] += f2 * dx[1] * dx[2];
                vi[5] += f2 * dx[2] * dx[2];
                }
            if (third_law && j < N)
                {
                force[j].x -= dx[0] * fdr;
                force[j].y -= dx[1] * fdr;
                force[j].z -= dx[2] * fdr;
                force[j].w += eng * 0.5;
                if (compute_virial)
                    {
                    virial[0 * vp + j] += f2 * dx[0] * dx[0];
                    virial[1 * vp + j] += f2 * dx[0] * dx[1];
                    virial[2 * vp + j] += f2 * dx[0] * dx[2];
                    virial[3 * vp + j] += f2 * dx[1] * dx[1];
                    virial[4 * vp + j] += f2 * dx[1] * dx[2];
                    virial[5 * vp + j] += f2 * dx[2] * dx[2];
                    }
                }
            }
        force[i].x += fi[0];
        force[i].y += fi[1];
        force[i].z += fi[2];
        force[i].w += pei;
        if (compute_virial)
            for (int c = 0; c < 6; c++)
                virial[c * vp + i] += vi[c];
        }
    }

/* ------------------------------------------------------------------------------------ */
/* bonds                                                                                */
/* ------------------------------------------------------------------------------------ */

/* hoomd/md/EvaluatorBondFENE.h:90-94 (constructor) and :116-156 (evalForceAndEnergy) */
int orc_eval_bond_fene(double rsq, const double* p, double* force_divr, double* bond_eng)
    {
    const double K = p[0], r_0 = p[1], sigma_6 = p[2], epsilon_x_4 = p[3], delta = p[4];
    const double lj1 = epsilon_x_4 * sigma_6 * sigma_6;
    const double lj2 = epsilon_x_4 * sigma_6;
    double rmdoverr;
    const double rtemp = sqrt(rsq) - delta;
    rmdoverr = rtemp / sqrt(rsq);
    rsq = rtemp * rtemp;
    const double r2inv = 1.0 / rsq;
    const double r6inv = r2inv * r2inv * r2inv;
    double wca_divr = 0.0, pair_eng = 0.0;
    const double sigma6inv = lj2 / lj1;
    const double epsilon = lj2 * lj2 / 4.0 / lj1;
    if (lj1 != 0 && r6inv > sigma6inv / 2.0)
        {
        wca_divr = r2inv * r6inv * (12.0 * lj1 * r6inv - 6.0 * lj2);
        pair_eng = (r6inv * (lj1 * r6inv - lj2) + epsilon);
        }
    *force_divr = wca_divr * rmdoverr;
    *bond_eng = pair_eng;
    if (K != 0.0)
        {
        if (rsq >= r_0 * r_0)
            return 0;
        *force_divr += -K / (1.0 - rsq / (r_0 * r_0)) * rmdoverr;
        *bond_eng += -0.5 * K * (r_0 * r_0) * log(1.0 - rsq / (r_0 * r_0));
        }
    return 1;
    }

/* hoomd/md/PotentialBond.h:165-336: loop over bonds, both ends receive the force and half the energy */
int orc_bond_forces_fene(const orc_d4* pos, const unsigned int* rtag, unsigned int N, unsigned int n_total, const orc_box_t* box,
                         const unsigned int* bonds, const unsigned int* bond_type, unsigned int n_bonds, const double* params,
                         int compute_virial, orc_d4* force, double* virial, size_t vp)
    {
    memset(force, 0, sizeof(orc_d4) * n_total);
    if (virial)
        memset(virial, 0, sizeof(double) * 6 * vp);
    for (unsigned int i = 0; i < n_bonds; i++)
        {
        const unsigned int a = rtag[bonds[2 * i]], b = rtag[bonds[2 * i + 1]];
        if (a >= n_total || b >= n_total)
            return (int)(1 + i);
        double dx[3] = {pos[b].x - pos[a].x, pos[b].y - pos[a].y, pos[b].z - pos[a].z};
        orc_box_min_image(box, dx);
        const double rsq = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
        double force_divr = 0.0, bond_eng = 0.0;
        const int evaluated = orc_eval_bond_fene(rsq, params + 5 * bond_type[i], &force_divr, &bond_eng);
        bond_eng *= 0.5;
        if (!evaluated)
            return (int)(1 + i);
        double bv[6] = {0, 0, 0, 0, 0, 0};
        if (compute_virial)
            {
            const double f2 = 0.5 * force_divr;
            bv[0] = dx[0] * dx[0] * f2;
            bv[1] = dx[0] * dx[1] * f2;
            bv[2] = dx[0] * dx[2] * f2;
            bv[3] = dx[1] * dx[1] * f2;
            bv[4] = dx[1] * dx[2] * f2;
            bv[5] = dx[2] * dx[2] * f2;
            }
        if (b < N)
            {
            force[b].x += force_divr * dx[0];
            force[b].y += force_divr * dx[1];
            force[b].z += force_divr * dx[2];
            force[b].w += bond_eng;
            if (compute_virial)
                for (int c = 0; c < 6; c++)
                    virial[c * vp + b] += bv[c];
            }
        if (a < N)
            {
            force[a].x -= force_divr * dx[0];
            force[a].y -= force_divr * dx[1];
            force[a].z -= force_divr * dx[2];
            force[a].w += bond_eng;
            if (compute_virial)
                for (int c = 0; c < 6; c++)
                    virial[c * vp + a] += bv[c];
            }
        }
    return 0;
    }

/* ------------------------------------------------------------------------------------ */
/* integrator                                                                           */
/* ------------------------------------------------------------------------------------ */

/* hoomd/Integrator.cc:186-237 (force and virial only: point particles carry no torque) */
void orc_sum_net_force(orc_d4* net_force,
                       double* net_virial,
                       size_t nvp,
                       const orc_d4* const* forces,
                       const double* const* virials,
                       const size_t* vps,
                       unsigned int n_forces,
                       unsigned int n)
    {
    memset(net_force, 0, sizeof(orc_d4) * n);
    if (net_virial)
        memset(net_virial, 0, sizeof(double) * 6 * nvp);
    for (unsigned int f = 0; f < n_forces; f++)
        for (unsigned int j = 0; j < n; j++)
            {
            net_force[j].x += forces[f][j].x;
            net_force[j].y += forces[f][j].y;
            net_force[j].z += forces[f][j].z;
            net_force[j].w += forces[f][j].w;
            if (net_virial && virials && virials[f])
                for (int k = 0; k < 6; k++)
                    net_virial[k * nvp + j] += virials[f][k * vps[f] + j];
            }
    }

/* hoomd/Integrator.cc:106-127 */
void orc_compute_accel(const orc_d4* vel, const orc_d4* nf, orc_d3* accel, unsigned int N)
    {
    for (unsigned int j = 0; j < N; j++)
        {
        const double minv = 1.0 / vel[j].w;
        accel[j].x = nf[j].x * minv;
        accel[j].y = nf[j].y * minv;
        accel[j].z = nf[j].z * minv;
        }
    }

/* hoomd/md/TwoStepConstantVolume.cc:31-79 (no displacement limit) */
void orc_step_one(orc_d4* pos,
                  orc_d4* vel,
                  const orc_d3* accel,
                  orc_i3* image,
                  const unsigned int* group,
                  unsigned int gs,
                  const orc_box_t* box,
                  double rescale,
                  double dt)
    {
    for (unsigned int g = 0; g < gs; g++)
        {
        const unsigned int j = group ? group[g] : g;
        double v[3] = {vel[j].x, vel[j].y, vel[j].z};
        const double a[3] = {accel[j].x, accel[j].y, accel[j].z};
        for (int d = 0; d < 3; d++)
            {
            v[d] = v[d] + (1.0 / 2.0) * a[d] * dt;
            v[d] *= rescale;
            }
        pos[j].x += dt * v[0];
        pos[j].y += dt * v[1];
        pos[j].z += dt * v[2];
        vel[j].x = v[0];
        vel[j].y = v[1];
        vel[j].z = v[2];
        }
    for (unsigned int g = 0; g < gs; g++)
        {
        const unsigned int j = group ? group[g] : g;
        double p[3] = {pos[j].x, pos[j].y, pos[j].z};
        int im[3] = {image[j].x, image[j].y, image[j].z};
        orc_box_wrap(box, p, im);
        pos[j].x = p[0];
        pos[j].y = p[1];
        pos[j].z = p[2];
        image[j].x = im[0];
        image[j].y = im[1];
        image[j].z = im[2];
        }
    }

/* hoomd/md/TwoStepConstantVolume.cc:233-261 */
void orc_step_two(orc_d4* vel,
                  orc_d3* accel,
                  const unsigned int* group,
                  unsigned int gs,
                  const orc_d4* nf,
                  double rescale,
                  double dt)
    {
    for (unsigned int g = 0; g < gs; g++)
        {
        const unsigned int j = group ? group[g] : g;
        double v[3] = {vel[j].x, vel[j].y, vel[j].z};
        const double minv = 1.0 / vel[j].w;
        const double a[3] = {nf[j].x * minv, nf[j].y * minv, nf[j].z * minv};
        for (int d = 0; d < 3; d++)
            {
            v[d] *= rescale;
            v[d] += (1.0 / 2.0) * dt * a[d];
            }
        vel[j].x = v[0];
        vel[j].y = v[1];
        vel[j].z = v[2];
        accel[j].x = a[0];
        accel[j].y = a[1];
        accel[j].z = a[2];
        }
    }

/* hoomd/md/ComputeThermo.cc:99-150 */
void orc_thermo(const orc_d4* vel, const orc_d4* nf, unsigned int N, double* ke, double* pe)
    {
    double k = 0.0, p = 0.0;
    for (unsigned int i = 0; i < N; i++)
        {
        k += vel[i].w * (vel[i].x * vel[i].x + vel[i].y * vel[i].y + vel[i].z * vel[i].z);
        p += nf[i].w;
        }
    *ke = 0.5 * k;
    *pe = p;
    }

/* ------------------------------------------------------------------------------------ */
/* whole-path driver: System::run + IntegratorTwoStep + NeighborList::compute           */
/* ------------------------------------------------------------------------------------ */

struct orc_sim
    {
    unsigned int N;
    orc_box_t box;
    orc_d4 *pos, *vel, *net_force, *force, *last_pos;
    orc_d3* accel;
    orc_i3* image;
    double r_cut, r_buff, dt;
    double params[2];
    double rcutsq, ronsq, r_listsq, rcut_max;
    int shift_mode;
    uint64_t timestep;
    int prepared;
    /* cell list */
    unsigned int dim[3], cell_Nmax, n_adj;
    unsigned int *cell_size, *cell_adj;
    orc_d4* cell_xyzf;
    /* neighbor list (half storage, hoomd/md/pair/pair.py:135-142) */
    orc_nlist_state_t st;
    unsigned int Nmax_t[1], cond_t[1];
    size_t* head_list;
    unsigned int *n_neigh, *nlist;
    size_t nlist_cap;
    double last_L[3];
    };

orc_sim_t* orc_sim_create(unsigned int N,
                          const orc_d4* pos,
                          const orc_d4* vel,
                          const double Lbox[3],
                          double r_cut,
                          double r_buff,
                          double epsilon,
                          double sigma,
                          double dt,
                          int shift_mode)
    {
    orc_sim_t* s = (orc_sim_t*)calloc(1, sizeof(orc_sim_t));
    s->N = N;
    orc_box_init(&s->box, Lbox[0], Lbox[1], Lbox[2], 0, 0, 0);
    s->pos = (orc_d4*)malloc(sizeof(orc_d4) * N);
    s->vel = (orc_d4*)malloc(sizeof(orc_d4) * N);
    s->net_force = (orc_d4*)calloc(N, sizeof(orc_d4));
    s->force = (orc_d4*)calloc(N, sizeof(orc_d4));
    s->last_pos = (orc_d4*)calloc(N, sizeof(orc_d4));
    s->accel = (orc_d3*)calloc(N, sizeof(orc_d3));
    s->image = (orc_i3*)calloc(N, sizeof(orc_i3));
    memcpy(s->pos, pos, sizeof(orc_d4) * N);
    memcpy(s->vel, vel, sizeof(orc_d4) * N);
    s->r_cut = r_cut;
    s->r_buff = r_buff;
    s->dt = dt;
    /* hoomd/md/EvaluatorPairLJ.h:136-140 */
    s->params[0] = sigma * sigma * sigma * sigma * sigma * sigma;
    s->params[1] = 4.0 * epsilon;
    s->rcutsq = r_cut * r_cut;
    s->ronsq = 0.0;
    s->shift_mode = shift_mode;
    double mm, mn;
    orc_update_rlist(&s->r_cut, 1, r_buff, &s->r_listsq, &s->rcut_max, &mm, &mn);

    /* hoomd/md/NeighborListBinned.cc:39-45, hoomd/CellList.cc:208-344 */
    const double gw[3] = {0, 0, 0};
    orc_cell_dimensions(&s->box, r_cut + r_buff, 1, gw, 3, s->dim);
    const unsigned int ncells = s->dim[0] * s->dim[1] * s->dim[2];
    s->cell_Nmax = (unsigned int)ceil((double)N / (double)ncells);
    if (s->cell_Nmax == 0)
        s->cell_Nmax = 1;
    s->n_adj = orc_cell_n_adj(s->dim, 1, 3);
    s->cell_size = (unsigned int*)malloc(sizeof(unsigned int) * ncells);
    s->cell_adj = (unsigned int*)malloc(sizeof(unsigned int) * (size_t)ncells * s->n_adj);
    orc_cell_adj(s->dim, 1, 3, s->cell_adj);
    s->cell_xyzf = (orc_d4*)malloc(sizeof(orc_d4) * (size_t)ncells * s->cell_Nmax);

    orc_nlist_state_init(&s->st, r_buff);
    s->Nmax_t[0] = 4; /* hoomd/md/NeighborList.cc:145-152 */
    s->cond_t[0] = 0;
    s->head_list = (size_t*)malloc(sizeof(size_t) * N);
    s->n_neigh = (unsigned int*)calloc(N, sizeof(unsigned int));
    s->nlist_cap = 4 * (size_t)N;
    s->nlist = (unsigned int*)malloc(sizeof(unsigned int) * s->nlist_cap);
    orc_box_nearest_plane_distance(&s->box, s->last_L);
    return s;
    }

void orc_sim_destroy(orc_sim_t* s)
    {
    free(s->pos);
    free(s->vel);
    free(s->net_force);
    free(s->force);
    free(s->last_pos);
    free(s->accel);
    free(s->image);
    free(s->cell_size);
    free(s->cell_adj);
    free(s->cell_xyzf);
    free(s->head_list);
    free(s->n_neigh);
    free(s->nlist);
    free(s);
    }

/* hoomd/CellList.cc:184-199 */
static void sim_cell_list(orc_sim_t* s)
    {
    const double gw[3] = {0, 0, 0};
    for (;;)
        {
        unsigned int cond[3];
        orc_compute_cell_list(s->pos, NULL, NULL, s->N, 0, &s->box, s->dim, s->cell_Nmax, gw, 0,
                              s->cell_size, s->cell_xyzf, NULL, NULL, cond);
        if (cond[0] > s->cell_Nmax)
            {
            s->cell_Nmax = cond[0];
            const unsigned int ncells = s->dim[0] * s->dim[1] * s->dim[2];
            free(s->cell_xyzf);
            s->cell_xyzf = (orc_d4*)malloc(sizeof(orc_d4) * (size_t)ncells * s->cell_Nmax);
            continue;
            }
        break;
        }
    }

/* hoomd/md/NeighborList.cc:1544-1592 */
static void sim_head_list(orc_sim_t* s)
    {
    size_t total = orc_build_head_list(s->pos, s->N, s->Nmax_t, s->head_list);
    if (total > s->nlist_cap)
        {
        size_t cap = s->nlist_cap ? s->nlist_cap : 1;
        while (total > cap)
            cap = ((size_t)(((float)cap) * 1.125f)) + 1;
        cap = (cap > 4) ? (cap + 3) & ~(size_t)3 : 4;
        s->nlist = (unsigned int*)realloc(s->nlist, sizeof(unsigned int) * cap);
        s->nlist_cap = cap;
        }
    }

/* hoomd/md/NeighborList.cc:363-435 */
static void sim_nlist_compute(orc_sim_t* s, uint64_t t)
    {
    if (s->st.force_update)
        sim_head_list(s);

    int result, dangerous;
    if (orc_needs_updating_begin(&s->st, t, &result, &dangerous))
        {
        double npd[3], lambda[3];
        orc_box_nearest_plane_distance(&s->box, npd);
        for (int d = 0; d < 3; d++)
            lambda[d] = npd[d] / s->last_L[d];
        int chk = orc_nlist_distance_check(s->pos, s->last_pos, s->N, &s->box, &s->rcut_max, s->r_buff, lambda);
        result = orc_needs_updating_finish(&s->st, t, chk, dangerous);
        }
    if (!result)
        return;

    const double gw[3] = {0, 0, 0};
    for (;;)
        {
        sim_cell_list(s);
        orc_nlist_binned(s->pos, NULL, s->N, &s->box, s->dim, gw, s->cell_size, s->cell_xyzf, s->cell_adj,
                         s->n_adj, s->cell_Nmax, &s->r_cut, &s->r_listsq, 1, /*full*/ 0, 0, s->head_list,
                         s->Nmax_t, s->nlist, s->n_neigh, s->cond_t);
        if (orc_nlist_check_conditions(s->cond_t, s->Nmax_t, 1))
            {
            sim_head_list(s);
            s->cond_t[0] = 0;
            continue;
            }
        break;
        }
    orc_nlist_set_last_pos(s->pos, s->N, s->last_pos);
    orc_box_nearest_plane_distance(&s->box, s->last_L);
    }

/* hoomd/Integrator.cc:166-253 with one force */
static void sim_net_force(orc_sim_t* s, uint64_t t)
    {
    sim_nlist_compute(s, t);
    orc_pair_forces(ORC_EV_LJ, s->pos, s->N, s->N, &s->box, s->n_neigh, s->nlist, s->head_list, s->params,
                    &s->rcutsq, &s->ronsq, 1, s->shift_mode, /*third law*/ 1, 0, s->force, NULL, 0);
    const orc_d4* fl[1] = {s->force};
    orc_sum_net_force(s->net_force, NULL, 0, fl, NULL, NULL, 1, s->N);
    }

/* hoomd/md/IntegratorTwoStep.cc:233-288 then :51-130 per step */
void orc_sim_run(orc_sim_t* s, uint64_t nsteps)
    {
    if (!s->prepared)
        {
        sim_net_force(s, s->timestep);
        orc_compute_accel(s->vel, s->net_force, s->accel, s->N);
        s->prepared = 1;
        }
    for (uint64_t k = 0; k < nsteps; k++)
        {
        orc_step_one(s->pos, s->vel, s->accel, s->image, NULL, s->N, &s->box, 1.0, s->dt);
        sim_net_force(s, s->timestep + 1);
        orc_step_two(s->vel, s->accel, NULL, s->N, s->net_force, 1.0, s->dt);
        s->timestep++;
        }
    }

void orc_sim_get(orc_sim_t* s, orc_d4* pos, orc_d4* vel, orc_i3* image, orc_d4* net_force)
    {
    if (pos)
        memcpy(pos, s->pos, sizeof(orc_d4) * s->N);
    if (vel)
        memcpy(vel, s->vel, sizeof(orc_d4) * s->N);
    if (image)
        memcpy(image, s->image, sizeof(orc_i3) * s->N);
    if (net_force)
        memcpy(net_force, s->net_force, sizeof(orc_d4) * s->N);
    }

void orc_sim_energies(orc_sim_t* s, double* ke, double* pe)
    {
    orc_thermo(s->vel, s->net_force, s->N, ke, pe);
    }

void orc_sim_stats(orc_sim_t* s, uint64_t* num_builds, unsigned int* shortest_rebuild, uint64_t* timestep)
    {
    *num_builds = s->st.updates + s->st.forced_updates;
    *shortest_rebuild = orc_nlist_smallest_rebuild(&s->st);
    *timestep = s->timestep;
    }
