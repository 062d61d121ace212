// neighbor_list.hip -- binned neighbor-list build, rebuild check, head list and exclusion filter
// for gfx950.
//
// Contracts: hoomd/md/NeighborListGPUBinned.cu:52-298,756-823 (build),
// hoomd/md/NeighborListGPU.cu:42-128 (distance check), :355-456 (head list), :158-267 (filter),
// :270-340 (exclusion table); comparison arithmetic as hoomd/md/NeighborListBinned.cc:150-157 with
// every product and sum rounded separately, so the neighbor SET of each particle equals the CPU
// path's bit for bit.
//
// Build kernel design (wave64, cell-centric): one wave owns one cell.  The members of the 27
// adjacent cells (~580 candidates at liquid density) are loaded ONCE, coalesced, into registers
// -- candidate k lives in lane k%64, register slot k/64 -- and stay there while the wave walks the
// home cell's particles one at a time.  The home particle's position is wave-uniform (readlane into
// scalar registers), every lane tests its own candidates, and each 64-candidate slab is compacted
// with a ballot + lane-rank prefix so that hits are appended contiguously to the padded row
// nlist[head_i ...].  Candidate traffic from memory is therefore one read per (cell, candidate)
// instead of one per (particle, candidate); the row writes are dense 4-byte runs.
// Results of this unit are compared bit for bit with the CPU path: no product may be fused into a
// following sum (the CPU path rounds every product and sum separately).
#pragma clang fp contract(off)
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include "tile.cuh"

namespace
    {
constexpr int NL_MAXK = 12;        // candidate register slots per lane -> 768 candidates per pass
constexpr int NL_BLOCK = 256;      // 4 waves per workgroup
constexpr int NL_WAVES = NL_BLOCK / 64;

__device__ __forceinline__ unsigned int rl_u(unsigned int v, unsigned int lane)
    {
    return (unsigned int)__builtin_amdgcn_readlane((int)v, (int)lane);
    }
__device__ __forceinline__ double rl_d(double v, unsigned int lane)
    {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), (int)lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), (int)lane);
    return __hiloint2double(hi, lo);
    }

// bijective remap of the workgroup id so that each XCD (workgroups are dealt round-robin over 8)
// walks one contiguous range of cells and neighbouring cells share an L2
__device__ __forceinline__ unsigned int xcd_remap(unsigned int b, unsigned int nb)
    {
    const unsigned int q = nb / 8, r = nb % 8, x = b % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
    }

// exclusions by particle index, applied while rows are built (d_n_ex == nullptr: none)
struct nl_excl_t
    {
    const unsigned int* d_n_ex;
    const unsigned int* d_ex_list;
    hmd_index2d_t exli;
    };

// Half-width of the band of squared distances in which "p_i - (p_j + image)" and the CPU path's
// "(p_i - p_j) - image" may fall on different sides of r_list^2.  Both are roundings of the same real number: the
// shifted candidate carries an error of half an ulp of 1.5 |box|, the CPU path's difference one of half an ulp of
// |box|, so the components differ by less than 2^-51 |box| and the squared distances of a pair near r_list by less than
// 2 sqrt(3) r_list 2^-51 |box| < 2^-48 (r_list^2 + |box|^2) / 2.  |box| is the sum of the lengths of the box vectors'
// components.  (About one candidate in 10^9 falls into the band.)
__device__ __forceinline__ double nl_exact_band_box(const hmd_box_t& b)
    {
    const double l = b.L[0] + b.L[1] * (1.0 + fabs(b.xy)) + b.L[2] * (1.0 + fabs(b.xz) + fabs(b.yz));
    return 3.5527136788005009e-15 * l * l; // 2^-48
    }

// The CPU path's squared distance of home particle (pix, piy, piz) and a candidate at (cx, cy, cz) (coordinates as
// stored, not shifted) seen through the periodic image (ix, iy, iz) of its cell: the difference first, then
// BoxDim::minImage's subtractions in its order -- z image (with its tilt terms on y and x), y image (tilt on x), x
// image -- every product and difference rounded on its own (hoomd/md/NeighborListBinned.cc:150-157,
// hoomd/BoxDim.h:327-372).
__device__ __forceinline__ double nl_exact_rsq(const hmd_box_t& b, double pix, double piy, double piz, double cx, double cy, double cz, unsigned int code)
    {
    const double ix = (double)((int)(code & 3u) - 1), iy = (double)((int)((code >> 2) & 3u) - 1), iz = (double)((int)((code >> 4) & 3u) - 1);
    double dx = HMD_SUB(pix, cx), dy = HMD_SUB(piy, cy), dz = HMD_SUB(piz, cz);
    dz = HMD_SUB(dz, HMD_MUL(b.L[2], iz));
    dy = HMD_SUB(dy, HMD_MUL(HMD_MUL(b.L[2], b.yz), iz));
    dx = HMD_SUB(dx, HMD_MUL(HMD_MUL(b.L[2], b.xz), iz));
    dy = HMD_SUB(dy, HMD_MUL(b.L[1], iy));
    dx = HMD_SUB(dx, HMD_MUL(HMD_MUL(b.L[1], b.xy), iy));
    dx = HMD_SUB(dx, HMD_MUL(b.L[0], ix));
    return HMD_ADD(HMD_ADD(HMD_MUL(dx, dx), HMD_MUL(dy, dy)), HMD_MUL(dz, dz));
    }

// MODE 0: minimum image per pair (any box); 1: the same for orthorhombic boxes; 2: the periodic image
// of every adjacent cell is fixed up front (possible when each periodic direction has at least three
// cells: the 3x3x3 stencil then never holds two images of one cell), so candidates are shifted once
// when they are loaded and no minimum-image arithmetic (hoomd/BoxDim.h:306-325) runs per pair.
// p_i - (p_j + image) is not the CPU path's (p_i - p_j) - image in the last bit, so in MODE 2 a squared distance
// within a band around r_list^2 that is wider than the two can differ (nl_exact_band) is computed again from the
// candidate's stored coordinates with the CPU path's own sequence of operations (nl_exact_rsq): the neighbor SET is
// the CPU path's also for pairs within an ulp of r_list across a periodic face.
// TILE (needs MODE 2): rows receive 16-bit tile slots (tile_slot() of the candidate number in the 4x4x4-cell
// region around the home cell's 2x2x2 block, tile.cuh) instead of particle indices, padded to a multiple
// of four with the slot after the region's last candidate, where consumers keep a dummy far away.
template<bool SINGLE_TYPE, bool FILTER_BODY, int MODE, bool TILE>
__global__ void __launch_bounds__(NL_BLOCK) __attribute__((amdgpu_waves_per_eu(4))) nlist_binned_kernel(hmd_nlist_binned_args_t a, unsigned int ncells, unsigned int wpc,
                                                                unsigned short* d_tile, unsigned int tile_capacity, unsigned int* d_tile_overflow, hmd_uint4* d_slot_meta,
                                                                unsigned int* d_slot_of, hmd_scalar4* d_cpos, const unsigned int* d_block_group, unsigned int row_pieces,
                                                                nl_excl_t ex)
    {
    extern __shared__ __align__(16) unsigned char s_raw[];
    // per wave: 64 exclusive offsets + 64 adjacent cell ids; then the ntypes^2 r_list^2 table
    unsigned int* s_excl_all = reinterpret_cast<unsigned int*>(s_raw);
    unsigned int* s_cell_all = s_excl_all + 64 * NL_WAVES;
    double* s_shift_all = reinterpret_cast<double*>(s_cell_all + 64 * NL_WAVES);
    unsigned int* s_tileoff_all = reinterpret_cast<unsigned int*>(s_shift_all + 3 * 64 * NL_WAVES);
    unsigned int* s_img_all = s_tileoff_all + 64 * NL_WAVES;
    double* s_rl2 = reinterpret_cast<double*>(s_img_all + 64 * NL_WAVES);

    const unsigned int lane = threadIdx.x & 63;
    const unsigned int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned int* s_excl = s_excl_all + 64 * wave;
    unsigned int* s_cell = s_cell_all + 64 * wave;
    double* s_shift = s_shift_all + 3 * 64 * wave;
    unsigned int* s_tileoff = s_tileoff_all + 64 * wave;
    unsigned int* s_img = s_img_all + 64 * wave; // (ix + 1) | (iy + 1) << 2 | (iz + 1) << 4 of every adjacent cell
    const double band_box = MODE == 2 ? nl_exact_band_box(a.box) : 0.0;

    if (!SINGLE_TYPE)
        {
        const unsigned int np = a.ntypes * a.ntypes;
        for (unsigned int t = threadIdx.x; t < np; t += blockDim.x)
            {
            const double rc = a.d_r_cut[t];
            const double rl = HMD_ADD(rc, a.r_buff);
            s_rl2[t] = (rc > 0.0) ? HMD_MUL(rl, rl) : -1.0;
            }
        __syncthreads();
        }
    double rl2_single = -1.0;
    if (SINGLE_TYPE)
        {
        const double rc = a.d_r_cut[0];
        const double rl = HMD_ADD(rc, a.r_buff);
        rl2_single = (rc > 0.0) ? HMD_MUL(rl, rl) : -1.0;
        }

    const unsigned int cells_per_block = NL_WAVES / wpc;
    const unsigned int nblocks = gridDim.x;
    const unsigned int vb = xcd_remap(blockIdx.x, nblocks);
    const unsigned int cell = vb * cells_per_block + wave / wpc;
    const unsigned int sub = wave % wpc;
    if (cell >= ncells)
        return;

    const unsigned int n_home = __builtin_amdgcn_readfirstlane(min(a.d_cell_size[cell], a.cli.w));
    if (n_home == 0)
        return;

    // ---- adjacency table of this cell: sizes -> exclusive offsets (wave scan) ----
    const unsigned int n_adj = a.cadji.w;
    unsigned int adj_cell = 0, adj_size = 0;
    if (lane < n_adj)
        {
        adj_cell = a.d_cell_adj[hmd_idx2(a.cadji, lane, cell)];
        adj_size = min(a.d_cell_size[adj_cell], a.cli.w);
        }
    unsigned int incl = adj_size;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
        {
        const unsigned int v = __shfl_up(incl, off);
        if ((int)lane >= off)
            incl += v;
        }
    s_excl[lane] = incl - adj_size;
    s_cell[lane] = adj_cell;
    if (MODE == 2)
        {
        // image of the adjacent cell seen from the home cell: its grid coordinate differs by -1, 0 or +1
        // after unwrapping; r_image = r + ix*a1 + iy*a2 + iz*a3 (box vectors as in hoomd/BoxDim.h:391-458)
        double sx = 0.0, sy = 0.0, sz = 0.0;
        unsigned int code = 1u | (1u << 2) | (1u << 4);
        if (lane < n_adj)
            {
            const int dim[3] = {(int)a.ci.w, (int)a.ci.h, (int)a.ci.d};
            const int hc[3] = {(int)(cell % a.ci.w), (int)((cell / a.ci.w) % a.ci.h), (int)(cell / (a.ci.w * a.ci.h))};
            const int ac[3] = {(int)(adj_cell % a.ci.w), (int)((adj_cell / a.ci.w) % a.ci.h), (int)(adj_cell / (a.ci.w * a.ci.h))};
            double img[3];
            code = 0u;
#pragma unroll
            for (int d = 0; d < 3; d++)
                {
                const int diff = ac[d] - hc[d];
                img[d] = (a.box.periodic[d] && dim[d] >= 3) ? (diff > 1 ? -1.0 : (diff < -1 ? 1.0 : 0.0)) : 0.0;
                code |= (unsigned int)((int)img[d] + 1) << (2 * d);
                }
            sx = img[0] * a.box.L[0] + img[1] * (a.box.L[1] * a.box.xy) + img[2] * (a.box.L[2] * a.box.xz);
            sy = img[1] * a.box.L[1] + img[2] * (a.box.L[2] * a.box.yz);
            sz = img[2] * a.box.L[2];
            }
        s_shift[3 * lane + 0] = sx;
        s_shift[3 * lane + 1] = sy;
        s_shift[3 * lane + 2] = sz;
        s_img[lane] = code;
        }
    unsigned int tile_pad = 0; // what pads a tile row: the slot of the consumers' dummy candidate
    // tile rows: number of the home cell's first member among the home particles of its block, the block's
    // first group of rows, and the most entries a row has room for
    unsigned int tile_hoff = 0, tile_grp0 = 0;
    const unsigned int tile_row_cap = row_pieces * hmd::TL_PIECE;
    if (TILE)
        {
        const int dim[3] = {(int)a.ci.w, (int)a.ci.h, (int)a.ci.d};
        const int hc[3] = {(int)(cell % a.ci.w), (int)((cell / a.ci.w) % a.ci.h), (int)(cell / (a.ci.w * a.ci.h))};
        const int c0[3] = {hc[0] - hc[0] % hmd::TB_X, hc[1] - hc[1] % hmd::TB_Y, hc[2] - hc[2] % hmd::TB_Z};
        int rimg[3];
        const int rcell = (int)lane < hmd::TL_RC ? hmd::tile_region_cell(a.ci, a.box, c0[0], c0[1], c0[2], (int)lane, rimg) : -1;
        const unsigned int rsize = rcell >= 0 ? min(a.d_cell_size[rcell], a.cli.w) : 0u;
        const unsigned int rincl = hmd::tile_wave_scan(rsize, (int)lane);
        // d_tile_overflow[0]: a region outgrew the tile, or the rows need more groups than there is room for (set by
        // the kernel that numbers the groups); [1]: largest region population seen (what the consumers of these rows
        // need as LDS capacity)
        const unsigned int rtotal = rl_u(rincl, 63);
        if (rtotal > tile_capacity || __builtin_amdgcn_readfirstlane(d_tile_overflow[0]) != 0u)
            {
            if (lane == 0)
                d_tile_overflow[0] = 1u;
            return;
            }
            {
            const unsigned int nbx = (a.ci.w + hmd::TB_X - 1) / hmd::TB_X, nby = (a.ci.h + hmd::TB_Y - 1) / hmd::TB_Y;
            const unsigned int tb = ((unsigned int)(c0[2] / hmd::TB_Z) * nby + (unsigned int)(c0[1] / hmd::TB_Y)) * nbx + (unsigned int)(c0[0] / hmd::TB_X);
            tile_grp0 = __builtin_amdgcn_readfirstlane(d_block_group[tb]);
            const int my_hc = ((hc[2] - c0[2]) * hmd::TB_Y + (hc[1] - c0[1])) * hmd::TB_X + (hc[0] - c0[0]);
#pragma unroll
            for (int q = 0; q < hmd::TL_NH; q++)
                {
                const int qx = q % hmd::TB_X, qy = (q / hmd::TB_X) % hmd::TB_Y, qz = q / (hmd::TB_X * hmd::TB_Y);
                const bool home = c0[0] + qx < dim[0] && c0[1] + qy < dim[1] && c0[2] + qz < dim[2];
                const unsigned int sz = rl_u(rsize, (unsigned int)hmd::tile_home_rc(q));
                if (q < my_hc && home)
                    tile_hoff += sz;
                }
            }
        if (lane == 0 && sub == 0 && rtotal > d_tile_overflow[1])
            atomicMax(d_tile_overflow + 1, rtotal);
        tile_pad = hmd::tile_slot(rtotal);
        // region cell of each adjacent cell: home position in the block + unwrapped offset + 1
        unsigned int rc = 0;
        if (lane < n_adj)
            {
            const int ac[3] = {(int)(adj_cell % a.ci.w), (int)((adj_cell / a.ci.w) % a.ci.h), (int)(adj_cell / (a.ci.w * a.ci.h))};
            int r[3];
#pragma unroll
            for (int d = 0; d < 3; d++)
                {
                int diff = ac[d] - hc[d];
                if (a.box.periodic[d])
                    diff = diff > 1 ? diff - dim[d] : (diff < -1 ? diff + dim[d] : diff);
                r[d] = hc[d] - c0[d] + diff + 1;
                }
            rc = (unsigned int)hmd::tile_rc(min(max(r[0], 0), hmd::TR_X - 1), min(max(r[1], 0), hmd::TR_Y - 1), min(max(r[2], 0), hmd::TR_Z - 1));
            }
        s_tileoff[lane] = __shfl(rincl - rsize, (int)rc);
        }
    const unsigned int T = rl_u(incl, 63);
    // (single wave reads back its own LDS words: no workgroup barrier needed, only the wait
    //  for the ds_write the compiler inserts before the first dependent ds_read)

    // ---- walk the home cell in slabs of 64 particles (one slab in practice) ----
    for (unsigned int hbase = 0; hbase < n_home; hbase += 64)
        {
        const unsigned int nh = min(n_home - hbase, 64u);
        double hx = 0, hy = 0, hz = 0;
        unsigned int hidx = 0xffffffffu, htype = 0, hbody = HMD_NO_BODY, hnmax = 0, cnt = 0;
        unsigned long long hhead = 0;
        // (exclusions: their number and the first four per home particle, fetched here, one home particle per lane)
        unsigned int hnex = 0, hex[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        if (lane < nh)
            {
            const unsigned int w = hmd_idx2(a.cli, hbase + lane, cell);
            const double4 v = hmd_ld4(a.d_cell_xyzf + w);
            hx = v.x;
            hy = v.y;
            hz = v.z;
            hidx = hmd_double_as_uint(v.w);
            if (!SINGLE_TYPE || FILTER_BODY)
                {
                const uint2 tb = *reinterpret_cast<const uint2*>(a.d_cell_type_body + w);
                htype = tb.x;
                hbody = tb.y;
                }
            if (hidx < a.N)
                {
                hnmax = a.d_Nmax[SINGLE_TYPE ? 0 : htype];
                hhead = a.d_head_list[hidx];
                if (ex.d_n_ex)
                    {
                    hnex = ex.d_n_ex[hidx];
#pragma unroll
                    for (unsigned int o = 0; o < 4; o++)
                        if (o < hnex)
                            hex[o] = ex.d_ex_list[hmd_idx2(ex.exli, hidx, o)];
                    }
                }
            }

        // ---- candidate passes of up to NL_MAXK*64 ----
        for (unsigned int pbase = 0; pbase < T; pbase += NL_MAXK * 64)
            {
            const unsigned int Tp = __builtin_amdgcn_readfirstlane(min(T - pbase, (unsigned int)(NL_MAXK * 64)));
            double cx[NL_MAXK], cy[NL_MAXK], cz[NL_MAXK];
            unsigned int cj[NL_MAXK], ct[NL_MAXK], cb[NL_MAXK], cl[NL_MAXK];
#pragma unroll
            for (int c = 0; c < NL_MAXK; c++)
                {
                cx[c] = cy[c] = cz[c] = __longlong_as_double(0x7ff8000000000000ll); // NaN: never a hit
                cj[c] = 0xffffffffu;
                ct[c] = 0;
                cb[c] = HMD_NO_BODY;
                if ((unsigned int)(c * 64) < Tp)
                    {
                    const unsigned int k = pbase + c * 64 + lane;
                    if (k < T)
                        {
                        // largest table entry with excl <= k
                        unsigned int lo = 0;
#pragma unroll
                        for (unsigned int step = 32; step > 0; step >>= 1)
                            {
                            const unsigned int t = lo + step;
                            if (t < n_adj && s_excl[t] <= k)
                                lo = t;
                            }
                        const unsigned int w = hmd_idx2(a.cli, k - s_excl[lo], s_cell[lo]);
                        const double4 v = hmd_ld4(a.d_cell_xyzf + w);
                        cx[c] = v.x;
                        cy[c] = v.y;
                        cz[c] = v.z;
                        if (MODE == 2)
                            {
                            cx[c] = HMD_ADD(cx[c], s_shift[3 * lo + 0]);
                            cy[c] = HMD_ADD(cy[c], s_shift[3 * lo + 1]);
                            cz[c] = HMD_ADD(cz[c], s_shift[3 * lo + 2]);
                            }
                        cj[c] = hmd_double_as_uint(v.w);
                        if (TILE)
                            cl[c] = hmd::tile_slot(s_tileoff[lo] + (k - s_excl[lo]));
                        if (!SINGLE_TYPE || FILTER_BODY)
                            {
                            const uint2 tb = *reinterpret_cast<const uint2*>(a.d_cell_type_body + w);
                            ct[c] = tb.x;
                            cb[c] = tb.y;
                            }
                        if (TILE && !SINGLE_TYPE)
                            cl[c] |= ct[c] << hmd::TL_TYPE_SHIFT; // rows for several types carry the candidate's type
                        }
                    }
                }

            for (unsigned int i = sub; i < nh; i += wpc)
                {
                const unsigned int iu = __builtin_amdgcn_readfirstlane(i);
                const unsigned int i_idx = rl_u(hidx, iu);
                if (i_idx >= a.N)
                    continue; // ghosts get no row
                const double pix = rl_d(hx, iu), piy = rl_d(hy, iu), piz = rl_d(hz, iu);
                const unsigned int nmax_i = rl_u(hnmax, iu);
                const unsigned long long head_i = ((unsigned long long)rl_u((unsigned int)(hhead >> 32), iu) << 32) | rl_u((unsigned int)hhead, iu);
                unsigned int cnt_i = rl_u(cnt, iu);
                unsigned int ti = 0, body_i = HMD_NO_BODY;
                if (!SINGLE_TYPE)
                    ti = rl_u(htype, iu);
                if (FILTER_BODY)
                    body_i = rl_u(hbody, iu);
                unsigned int* row = a.d_nlist + head_i;
                unsigned short* trow = nullptr;
                unsigned int nmax_w = nmax_i;
                if (TILE)
                    {
                    const unsigned int h_i = tile_hoff + hbase + iu;
                    trow = d_tile + (size_t)(tile_grp0 + h_i / hmd::TL_GROUP) * hmd::tile_group_entries(row_pieces) + (h_i % hmd::TL_GROUP) * hmd::TL_PIECE;
                    nmax_w = min(nmax_i, tile_row_cap);
                    }
                // excluded partners of this particle: the first four in registers, longer lists read again per slab
                unsigned int nex_i = 0, ex_r[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
                if (ex.d_n_ex)
                    {
                    nex_i = rl_u(hnex, iu);
#pragma unroll
                    for (unsigned int o = 0; o < 4; o++)
                        ex_r[o] = rl_u(hex[o], iu);
                    }

#pragma unroll
                for (int c = 0; c < NL_MAXK; c++)
                    {
                    if ((unsigned int)(c * 64) < Tp)
                        {
                        double dx = HMD_SUB(pix, cx[c]);
                        double dy = HMD_SUB(piy, cy[c]);
                        double dz = HMD_SUB(piz, cz[c]);
                        if (MODE != 2)
                            hmd_min_image<MODE == 1>(a.box, dx, dy, dz);
                        double rsq = HMD_ADD(HMD_ADD(HMD_MUL(dx, dx), HMD_MUL(dy, dy)), HMD_MUL(dz, dz));
                        const double rl2 = SINGLE_TYPE ? rl2_single : s_rl2[hmd_idx2(hmd_index2d_t {a.ntypes, a.ntypes}, ti, ct[c])];
                        if (MODE == 2 && fabs(rsq - rl2) <= band_box + 3.5527136788005009e-15 * rl2)
                            {
                            // (rare) too close to r_list^2 for the shifted candidate: the CPU path's arithmetic on the
                            // coordinates as stored
                            const unsigned int k = pbase + c * 64 + lane;
                            unsigned int lo = 0;
#pragma unroll
                            for (unsigned int step = 32; step > 0; step >>= 1)
                                {
                                const unsigned int t = lo + step;
                                if (t < n_adj && s_excl[t] <= k)
                                    lo = t;
                                }
                            const double4 v = hmd_ld4(a.d_cell_xyzf + hmd_idx2(a.cli, k - s_excl[lo], s_cell[lo]));
                            rsq = nl_exact_rsq(a.box, pix, piy, piz, v.x, v.y, v.z, s_img[lo]);
                            }
                        bool hit = (rsq <= rl2) && (cj[c] != i_idx);
                        if (FILTER_BODY)
                            hit = hit && !(body_i != HMD_NO_BODY && body_i == cb[c]);
                        if (nex_i)
                            {
                            hit = hit && cj[c] != ex_r[0] && cj[c] != ex_r[1] && cj[c] != ex_r[2] && cj[c] != ex_r[3];
                            for (unsigned int o = 4; o < nex_i; o++)
                                hit = hit && cj[c] != ex.d_ex_list[hmd_idx2(ex.exli, i_idx, o)];
                            }
                        const unsigned long long mask = __ballot(hit);
                        if (mask)
                            {
                            const unsigned int pos = cnt_i + hmd_rank_in_mask(mask);
                            if (hit && pos < nmax_w)
                                {
                                if (TILE)
                                    trow[((pos & ~15u) << 4) | (pos & 15u)] = (unsigned short)cl[c];
                                else
                                    row[pos] = cj[c];
                                }
                            cnt_i += (unsigned int)__popcll(mask);
                            }
                        }
                    }
                if (lane == iu)
                    cnt = cnt_i;
                }
            }

        // ---- per-particle results ----
        if (TILE && d_slot_meta && lane < nh && hidx >= a.N && (lane % wpc) == sub)
            {
            hmd_uint4 m = {hidx, 0u, 0u, 0u}; // a ghost: no row
            const unsigned int w = hmd_idx2(a.cli, hbase + lane, cell);
            d_slot_meta[w] = m;
            if (d_slot_of)
                {
                d_slot_of[hidx] = w;
                hmd_st4(d_cpos + w, hx, hy, hz, a.d_pos[hidx].w);
                }
            }
        if (lane < nh && hidx < a.N && (lane % wpc) == sub)
            {
            if (TILE && d_slot_meta)
                {
                hmd_uint4 m = {hidx, cnt, 0u, 0u};
                const unsigned int w = hmd_idx2(a.cli, hbase + lane, cell);
                d_slot_meta[w] = m;
                if (d_slot_of)
                    {
                    // the cell list holds the same bits as pos[hidx]: the slot-ordered copy starts out current
                    d_slot_of[hidx] = w;
                    hmd_st4(d_cpos + w, hx, hy, hz, a.d_pos[hidx].w);
                    }
                }
            a.d_n_neigh[hidx] = cnt;
            if (TILE)
                {
                const unsigned int h_i = tile_hoff + hbase + lane;
                unsigned short* trow = d_tile + (size_t)(tile_grp0 + h_i / hmd::TL_GROUP) * hmd::tile_group_entries(row_pieces) + (h_i % hmd::TL_GROUP) * hmd::TL_PIECE;
                for (unsigned int k = cnt; k < ((cnt + 3u) & ~3u) && k < min(hnmax, tile_row_cap); k++)
                    trow[((k & ~15u) << 4) | (k & 15u)] = (unsigned short)tile_pad;
                }
            if (a.d_last_updated_pos)
                {
                // same bits as pos[hidx]: the cell list holds a copy of x,y,z and of the type id
                const unsigned int t = (SINGLE_TYPE && !FILTER_BODY) ? hmd_double_as_uint(a.d_pos[hidx].w) : htype;
                hmd_st4(a.d_last_updated_pos + hidx, hx, hy, hz, hmd_uint_as_double(t));
                }
            if (cnt >= hnmax)
                atomicMax(&a.d_conditions[SINGLE_TYPE ? 0 : htype], cnt);
            }
        }
    }

// ---- the same build with an fp32 pre-test (one particle type, no body filter, orthorhombic box, fixed images) ----
//
// The exact test of the CPU path -- rsq <= r_list^2 with rsq = dx*dx + dy*dy + dz*dz in fp64, every operation
// rounded on its own -- costs eight fp64 instructions per 64 candidates, and 85 % of the candidates fail it.
// Here every candidate and every home particle also exists as three fp32 coordinates relative to the first home
// particle of the cell, two slabs of candidates packed per register, so one packed-fp32 instruction serves 128
// tests.  The fp32 squared distance r32 differs from the true one by less than `delta` / 4 (bound computed by the
// launcher from the cell width), so r32 <= r_list^2 - delta is a certain hit, r32 > r_list^2 + delta a certain miss,
// and only a slab with a candidate in the band between them (about one per thousand home particles) is tested
// again with the exact fp64 arithmetic on reloaded coordinates -- the CPU path's own: the difference of the two
// positions, then the box length of the periodic image subtracted.  The set of neighbors is therefore, bit for bit,
// the one the CPU path finds, also for pairs within an ulp of r_list across the periodic boundary.  The home particle itself is removed from its slab's hit mask by position in the
// candidate list (scalar arithmetic), not by comparing indices.
typedef float nl_v2f __attribute__((ext_vector_type(2)));
constexpr int NL_PAIRS = NL_MAXK / 2;
constexpr int NL_ROWBUF = 128; // entries of a row staged in LDS per wave before they go to memory

// inclusive prefix sum over the 64 lanes of a wave on the vector ALU's data-parallel-primitive path: shifts by 1, 2,
// 4, 8 within rows of 16 lanes, then lane 15 of a row into the next row and lane 31 into the upper half
__device__ __forceinline__ unsigned int nl_wave_scan_dpp(unsigned int v)
    {
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); // row_shr:1
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true); // row_shr:2
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true); // row_shr:4
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true); // row_shr:8
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return v;
    }

struct nl_pref_t
    {
    float lo, hi; // r32 <= lo: hit; r32 > hi: miss
    };

// Bound on |r32 - true squared distance| for candidates near r_list: coordinates relative to the origin are below
// M = 2 * (widest cell) + r_list, each conversion to fp32 errs by at most u * M (u = 2^-24), the difference by 2 u M
// more, so a component of the separation by at most 4 u M and the sum of squares by 8 sqrt(3) u M r_list, plus three
// roundings of about u r_list^2 in the squares and sums.  delta is four times that, plus two fp32 ulps of r_list^2
// for the conversion of the thresholds themselves.
__device__ __forceinline__ nl_pref_t nl_pref_thresholds(double rl, double rl2, double wmax)
    {
    nl_pref_t t;
    if (!(rl2 > 0.0))
        {
        t.lo = t.hi = -1.0f; // the pair is switched off (r_cut <= 0): nothing passes
        return t;
        }
    const double u = 5.9604644775390625e-08;
    const double M = 2.0 * wmax + rl;
    const double delta = 4.0 * (13.8564064605510183 * u * M * rl + 3.0 * u * rl2) + rl2 * 2.384185791015625e-07;
    t.lo = (float)(rl2 - delta);
    t.hi = (float)(rl2 + delta);
    return t;
    }

// bytes of the cell-start flags of one wave: one per candidate of 27 full cells, rounded up to whole slabs
__host__ __device__ constexpr unsigned int nl_pref_flag_bytes(unsigned int cell_Nmax)
    {
    return (27u * cell_Nmax + 63u) / 64u * 64u + 64u;
    }

// (two types, tile rows: held to the registers of three waves per SIMD -- left alone the 14-slab version takes 180 and runs
// two)
template<bool TILE, bool EXCL, bool TWO, int MAXK>
__global__ void __launch_bounds__(NL_BLOCK) __attribute__((amdgpu_waves_per_eu((TILE && TWO && !EXCL) ? 3 : 1))) nlist_pref_kernel(hmd_nlist_binned_args_t a, unsigned int ncells, unsigned int wpc, double wmax, unsigned short* d_tile,
                                                              unsigned int tile_capacity, unsigned int* d_tile_overflow, hmd_uint4* d_slot_meta,
                                                              unsigned int* d_slot_of, hmd_scalar4* d_cpos, const unsigned int* d_block_group,
                                                              unsigned int row_pieces, nl_excl_t ex)
    {
    extern __shared__ __align__(16) unsigned char s_raw[];
    const unsigned int nwaves = blockDim.x >> 6; // waves per workgroup: four, or one (a workgroup per cell)
    unsigned int* s_excl_all = reinterpret_cast<unsigned int*>(s_raw);
    unsigned int* s_cell_all = s_excl_all + 64 * nwaves;
    double* s_shift_all = reinterpret_cast<double*>(s_cell_all + 64 * nwaves);
    unsigned int* s_tileoff_all = reinterpret_cast<unsigned int*>(s_shift_all + 3 * 64 * nwaves);
    unsigned int* s_rowbuf = s_tileoff_all + 64 * nwaves;
    // which adjacent cell a candidate belongs to, without a search: a byte per candidate number that is set where a
    // (non-empty) cell starts, and the non-empty cells in order
    unsigned int* s_map_all = s_rowbuf + NL_ROWBUF * nwaves;
    const unsigned int flag_bytes = nl_pref_flag_bytes(a.cli.w);
    unsigned char* s_flag_all = reinterpret_cast<unsigned char*>(s_map_all + 32 * nwaves);

    const unsigned int lane = threadIdx.x & 63;
    const unsigned int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned int* s_map = s_map_all + 32 * wave;
    unsigned char* s_flag = s_flag_all + flag_bytes * wave;
    unsigned int* s_excl = s_excl_all + 64 * wave;
    unsigned int* s_cell = s_cell_all + 64 * wave;
    double* s_shift = s_shift_all + 3 * 64 * wave;
    unsigned int* s_tileoff = s_tileoff_all + 64 * wave;

    // One or two particle types (TWO): thresholds per type pair, [tj * 2 + ti].  As integers: lo1 = bits(lo) + 1 (0 for a
    // pair that is switched off: no distance is a hit then), and one band for all pairs, the widest.
    constexpr int NTP = TWO ? 4 : 1;
    double rl2_tp[NTP];
    unsigned int lo1_tp[NTP], band_all = 0;
#pragma unroll
    for (int tp = 0; tp < NTP; tp++)
        {
        const double rc = a.d_r_cut[tp];
        const double rl = HMD_ADD(rc, a.r_buff);
        rl2_tp[tp] = (rc > 0.0) ? HMD_MUL(rl, rl) : -1.0;
        const nl_pref_t pf = nl_pref_thresholds(rl, rl2_tp[tp], wmax);
        lo1_tp[tp] = pf.hi > 0.0f ? __float_as_uint(pf.lo) + 1u : 0u;
        if (pf.hi > 0.0f)
            band_all = max(band_all, __float_as_uint(pf.hi) - __float_as_uint(pf.lo));
        }

    // One type: the same two decisions from a float difference, two slabs per instruction.  d = r32 - lo', lo' the float
    // after lo: the difference of two floats has the sign of the exact difference (it is never rounded to zero), so the
    // sign bit of d is set exactly for r32 <= lo; and 0 <= d <= hi - lo' (as bit patterns: below bits(band_f), band_f a
    // little above hi - lo) covers every r32 in (lo, hi].
    const nl_v2f lo_next2 = {__uint_as_float(lo1_tp[0]), __uint_as_float(lo1_tp[0])};
    unsigned int band_f_bits = 0;
        {
        const nl_pref_t pf0 = nl_pref_thresholds(HMD_ADD(a.d_r_cut[0], a.r_buff), rl2_tp[0], wmax);
        if (pf0.hi > 0.0f)
            band_f_bits = __float_as_uint((pf0.hi - pf0.lo) * 1.001f + 1.0e-30f);
        }

    const unsigned int cells_per_block = nwaves / wpc;
    const unsigned int vb = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned int cell = vb * cells_per_block + wave / wpc;
    const unsigned int sub = wave % wpc;
    if (cell >= ncells)
        return;
    const unsigned int n_home = __builtin_amdgcn_readfirstlane(min(a.d_cell_size[cell], a.cli.w));
    if (n_home == 0)
        return;

    // ---- the 27 adjacent cells, straight from the grid coordinates (lane t: offset (t % 3 - 1, t / 3 % 3 - 1, t / 9 - 1);
    // a cell beyond a non-periodic face does not exist, one beyond a periodic face is the cell on the far side seen
    // one box length away): sizes -> exclusive offsets (wave scan), image shifts.  No adjacency table is read, so the
    // sizes arrive in the same round trip as the home cell's. ----
    const unsigned int n_adj = 27;
    const int dim[3] = {(int)a.ci.w, (int)a.ci.h, (int)a.ci.d};
    const int hc[3] = {(int)(cell % a.ci.w), (int)((cell / a.ci.w) % a.ci.h), (int)(cell / (a.ci.w * a.ci.h))};
    const int aoff[3] = {(int)(lane % 3u) - 1, (int)((lane / 3u) % 3u) - 1, (int)(lane / 9u) - 1};
    unsigned int adj_cell = 0xffffffffu, adj_size = 0;
        {
        int ac[3];
        double sh[3];
        bool exists = lane < n_adj;
#pragma unroll
        for (int d = 0; d < 3; d++)
            {
            ac[d] = hc[d] + aoff[d];
            sh[d] = 0.0;
            if (ac[d] < 0)
                {
                if (a.box.periodic[d])
                    {
                    ac[d] += dim[d];
                    sh[d] = -a.box.L[d];
                    }
                else
                    exists = false;
                }
            else if (ac[d] >= dim[d])
                {
                if (a.box.periodic[d])
                    {
                    ac[d] -= dim[d];
                    sh[d] = a.box.L[d];
                    }
                else
                    exists = false;
                }
            }
        if (exists)
            {
            adj_cell = (unsigned int)((ac[2] * dim[1] + ac[1]) * dim[0] + ac[0]);
            adj_size = min(a.d_cell_size[adj_cell], a.cli.w);
            }
        s_shift[3 * lane + 0] = sh[0];
        s_shift[3 * lane + 1] = sh[1];
        s_shift[3 * lane + 2] = sh[2];
        }
    const unsigned int incl = hmd::tile_wave_scan(adj_size, (int)lane);
    s_excl[lane] = incl - adj_size;
    s_cell[lane] = adj_cell;
    // where the home cell's own members start in the candidate list (for the self exclusion): offset (0, 0, 0) is lane 13
    const unsigned int self_first = rl_u(incl - adj_size, 13u);
        {
        for (unsigned int i = 4u * lane; i < flag_bytes; i += 256u)
            *reinterpret_cast<unsigned int*>(s_flag + i) = 0u;
        const bool nonempty = adj_size != 0u; // (lanes 27 and up have no cell: size 0)
        const unsigned long long ne = __ballot(nonempty);
        if (nonempty)
            {
            s_map[hmd_rank_in_mask(ne)] = lane;
            s_flag[incl - adj_size] = 1;
            }
        }

    unsigned int tile_pad = 0, tile_hoff = 0, tile_grp0 = 0;
    const unsigned int tile_row_cap = row_pieces * hmd::TL_PIECE;
    if (TILE)
        {
        const int c0[3] = {hc[0] - hc[0] % hmd::TB_X, hc[1] - hc[1] % hmd::TB_Y, hc[2] - hc[2] % hmd::TB_Z};
        int rimg[3];
        const int rcell = (int)lane < hmd::TL_RC ? hmd::tile_region_cell(a.ci, a.box, c0[0], c0[1], c0[2], (int)lane, rimg) : -1;
        const unsigned int rsize = rcell >= 0 ? min(a.d_cell_size[rcell], a.cli.w) : 0u;
        const unsigned int rincl = hmd::tile_wave_scan(rsize, (int)lane);
        const unsigned int rtotal = rl_u(rincl, 63);
        if (rtotal > tile_capacity || __builtin_amdgcn_readfirstlane(d_tile_overflow[0]) != 0u)
            {
            if (lane == 0)
                d_tile_overflow[0] = 1u;
            return;
            }
        if (lane == 0 && sub == 0 && rtotal > d_tile_overflow[1])
            atomicMax(d_tile_overflow + 1, rtotal);
        tile_pad = hmd::tile_slot(rtotal);
        const unsigned int nbx = (a.ci.w + hmd::TB_X - 1) / hmd::TB_X, nby = (a.ci.h + hmd::TB_Y - 1) / hmd::TB_Y;
        const unsigned int tb = ((unsigned int)(c0[2] / hmd::TB_Z) * nby + (unsigned int)(c0[1] / hmd::TB_Y)) * nbx + (unsigned int)(c0[0] / hmd::TB_X);
        tile_grp0 = __builtin_amdgcn_readfirstlane(d_block_group[tb]);
        const int my_hc = ((hc[2] - c0[2]) * hmd::TB_Y + (hc[1] - c0[1])) * hmd::TB_X + (hc[0] - c0[0]);
#pragma unroll
        for (int q = 0; q < hmd::TL_NH; q++)
            {
            const int qx = q % hmd::TB_X, qy = (q / hmd::TB_X) % hmd::TB_Y, qz = q / (hmd::TB_X * hmd::TB_Y);
            const bool home = c0[0] + qx < dim[0] && c0[1] + qy < dim[1] && c0[2] + qz < dim[2];
            const unsigned int sz = rl_u(rsize, (unsigned int)hmd::tile_home_rc(q));
            if (q < my_hc && home)
                tile_hoff += sz;
            }
        // region cell of each adjacent cell: home position in the block + offset + 1
        unsigned int rc = 0;
        if (lane < n_adj)
            rc = (unsigned int)hmd::tile_rc(hc[0] - c0[0] + aoff[0] + 1, hc[1] - c0[1] + aoff[1] + 1, hc[2] - c0[2] + aoff[2] + 1);
        s_tileoff[lane] = __shfl(rincl - rsize, (int)rc);
        }
    const unsigned int T = rl_u(incl, 63);
    unsigned short* const trow_block = TILE ? d_tile + (size_t)tile_grp0 * hmd::tile_group_entries(row_pieces) : nullptr;

    // candidate k of the list: its cell-list slot, the shift of its cell's image, what a row stores for it
    auto cand_lookup = [&](unsigned int k, unsigned int& lo) -> unsigned int
    {
        lo = 0;
#pragma unroll
        for (unsigned int step = 32; step > 0; step >>= 1)
            {
            const unsigned int t = lo + step;
            if (t < n_adj && s_excl[t] <= k)
                lo = t;
            }
        return hmd_idx2(a.cli, k - s_excl[lo], s_cell[lo]);
    };

    for (unsigned int hbase = 0; hbase < n_home; hbase += 64)
        {
        const unsigned int nh = min(n_home - hbase, 64u);
        double hx = 0, hy = 0, hz = 0;
        unsigned int hidx = 0xffffffffu, hnmax = 0, cnt = 0, htype = 0;
        // Tile rows for two types are laid out by type: the neighbors of type 0 (cnt of them), the dummy up to a multiple of
        // four entries, then the neighbors of type 1 (cntB) -- every chunk of four entries that a lane of the force kernel
        // takes then has ONE type, and the kernel picks its parameters per chunk instead of per pair.
        constexpr bool BY_TYPE = TWO && TILE;
        unsigned int cntB = 0;
        // (the start of a home particle's row, the number of its exclusions and the first four of them are fetched here,
        // one home particle per lane, and handed out by lane reads in the loop below: fetched where they are used, each
        // was a round trip to memory that the whole wave waited for)
        constexpr unsigned int NEXR = 4;
        unsigned int hnex = 0, hex[EXCL ? NEXR : 1];
        unsigned int hhead_lo = 0, hhead_hi = 0;
#pragma unroll
        for (unsigned int o = 0; o < (EXCL ? NEXR : 1u); o++)
            hex[o] = 0xffffffffu;
        if (lane < nh)
            {
            const unsigned int w = hmd_idx2(a.cli, hbase + lane, cell);
            const double4 v = hmd_ld4(a.d_cell_xyzf + w);
            hx = v.x;
            hy = v.y;
            hz = v.z;
            hidx = hmd_double_as_uint(v.w);
            if (TWO)
                htype = a.d_cell_type_body[w].x & 1u;
            if (hidx < a.N)
                {
                hnmax = a.d_Nmax[htype];
                if (EXCL)
                    {
                    hnex = ex.d_n_ex[hidx];
#pragma unroll
                    for (unsigned int o = 0; o < (EXCL ? NEXR : 0u); o++)
                        if (o < hnex)
                            hex[o] = ex.d_ex_list[hmd_idx2(ex.exli, hidx, o)];
                    }
                if (!TILE)
                    {
                    const size_t head = a.d_head_list[hidx];
                    hhead_lo = (unsigned int)head;
                    hhead_hi = (unsigned int)(head >> 32);
                    }
                }
            }
        // origin of the fp32 coordinates: the slab's first home particle
        const double Ox = rl_d(hx, 0), Oy = rl_d(hy, 0), Oz = rl_d(hz, 0);
        const float hxf = (float)(hx - Ox), hyf = (float)(hy - Oy), hzf = (float)(hz - Oz);

        unsigned int cells_before = 0; // non-empty cells that start before the slab being looked up (slabs are taken in order)
        // The candidates are taken in passes of MAXK slabs (one pass is the rule).  Rows laid out by type need all
        // neighbors of type 0 before the first of type 1: a cell with more candidates than one pass holds is swept twice,
        // first for the hits of type 0 (mode 1), then for those of type 1 (mode 2); mode 0 places both at once.
        const unsigned int npass = (T + (unsigned int)(MAXK * 64) - 1u) / (unsigned int)(MAXK * 64);
        const unsigned int nsweep = (BY_TYPE && npass > 1u) ? 2u : 1u;
        for (unsigned int it = 0; it < npass * nsweep; it++)
            {
            const unsigned int pbase = (it % npass) * (unsigned int)(MAXK * 64);
            const unsigned int mode = nsweep == 2u ? 1u + it / npass : 0u;
            if (it % npass == 0u)
                cells_before = 0;
            const unsigned int Tp = __builtin_amdgcn_readfirstlane(min(T - pbase, (unsigned int)(MAXK * 64)));
            nl_v2f fx[(MAXK / 2)], fy[(MAXK / 2)], fz[(MAXK / 2)];
            unsigned int co[MAXK]; // what a row stores for the candidate: tile slot or particle index
            unsigned int cidx[(EXCL && TILE) ? MAXK : 1]; // tile rows with exclusions: the candidate's particle index as well
            unsigned int ctbits = 0;                          // two types: bit c = type of this lane's candidate in slab c
            // Loads in two batches of six slabs: first every address of a batch (the searches run on LDS), then every
            // load, then the conversions -- two memory round trips per pass instead of one per slab.
            constexpr int NL_BATCH = MAXK / 2;
#pragma unroll
            for (int b0 = 0; b0 < MAXK; b0 += NL_BATCH)
                {
                if ((unsigned int)(b0 * 64) >= Tp)
                    {
#pragma unroll
                    for (int c = b0; c < b0 + NL_BATCH; c++)
                        {
                        co[c] = 0xffffffffu;
                        if (EXCL && TILE)
                            cidx[c] = 0xffffffffu;
                        if (c & 1)
                            fx[c / 2].y = fy[c / 2].y = fz[c / 2].y = 1.0e18f;
                        else
                            fx[c / 2].x = fy[c / 2].x = fz[c / 2].x = 1.0e18f;
                        }
                    continue;
                    }
                unsigned int w_[NL_BATCH], lo_[NL_BATCH];
#pragma unroll
                for (int q = 0; q < NL_BATCH; q++)
                    {
                    const unsigned int k = pbase + (b0 + q) * 64 + lane;
                    lo_[q] = 0;
                    w_[q] = 0xffffffffu;
                    if ((unsigned int)((b0 + q) * 64) < Tp)
                        {
                        // the cell of candidate k: as many non-empty cells start at or before k as flags are set up to k
                        const unsigned long long starts = __ballot(s_flag[k] != 0);
                        const unsigned int below = hmd_rank_in_mask(starts) + (unsigned int)((starts >> lane) & 1ull);
                        if (k < T)
                            {
                            lo_[q] = s_map[cells_before + below - 1u];
                            w_[q] = hmd_idx2(a.cli, k - s_excl[lo_[q]], s_cell[lo_[q]]);
                            }
                        cells_before += (unsigned int)__popcll(starts);
                        }
                    }
                double2 vxy[NL_BATCH], vzw[NL_BATCH];
                unsigned int vty[TWO ? NL_BATCH : 1];
#pragma unroll
                for (int q = 0; q < NL_BATCH; q++)
                    {
                    // (empty slots read slot 0: a valid address, the result is not used)
                    const double2* src = reinterpret_cast<const double2*>(a.d_cell_xyzf + (w_[q] == 0xffffffffu ? 0u : w_[q]));
                    vxy[q] = src[0];
                    vzw[q] = src[1];
                    if (TWO)
                        vty[q] = a.d_cell_type_body[w_[q] == 0xffffffffu ? 0u : w_[q]].x;
                    }
#pragma unroll
                for (int q = 0; q < NL_BATCH; q++)
                    {
                    const int c = b0 + q;
                    // empty slots: far away (finite, so that the squared distance keeps a positive sign bit: the bit tests
                    // below would take a NaN with its sign set for a hit)
                    float x32 = 1.0e18f, y32 = x32, z32 = x32;
                    co[c] = 0xffffffffu;
                    if (EXCL && TILE)
                        cidx[c] = w_[q] != 0xffffffffu ? hmd_double_as_uint(vzw[q].y) : 0xffffffffu;
                    if (TWO && w_[q] != 0xffffffffu)
                        ctbits |= (vty[q] & 1u) << c;
                    if (w_[q] != 0xffffffffu)
                        {
                        const unsigned int lo = lo_[q];
                        x32 = (float)(HMD_ADD(vxy[q].x, s_shift[3 * lo + 0]) - Ox);
                        y32 = (float)(HMD_ADD(vxy[q].y, s_shift[3 * lo + 1]) - Oy);
                        z32 = (float)(HMD_ADD(vzw[q].x, s_shift[3 * lo + 2]) - Oz);
                        const unsigned int k = pbase + c * 64 + lane;
                        co[c] = TILE ? hmd::tile_slot(s_tileoff[lo] + (k - s_excl[lo])) : hmd_double_as_uint(vzw[q].y);
                        if (TILE && TWO)
                            co[c] |= (vty[q] & 1u) << hmd::TL_TYPE_SHIFT; // rows for several types carry the candidate's type
                        }
                    if (c & 1)
                        {
                        fx[c / 2].y = x32;
                        fy[c / 2].y = y32;
                        fz[c / 2].y = z32;
                        }
                    else
                        {
                        fx[c / 2].x = x32;
                        fy[c / 2].x = y32;
                        fz[c / 2].x = z32;
                        }
                    }
                }

            // 16-bit slabs count of this pass and the fp32 thresholds as integers: for non-negative floats the bit
            // patterns order like the values, so t = bits(r32) - (bits(lo) + 1) is negative (top bit set) exactly for
            // r32 <= lo, and below `band` exactly for lo < r32 <= hi
            const unsigned int nslab_dyn = (Tp + 63u) / 64u;
            const unsigned int band = band_all;
            // the candidates' type bits in the order of the hit words (slab c at bit nslab - 1 - c)
            const unsigned int tbr = BY_TYPE ? (__brev(ctbits) >> (32u - nslab_dyn)) : 0u;
            // The walk over the home particles, written once and compiled for the slab counts cells of a liquid have
            // (NS = 8 ... MAXK: every 'is slab c in use' below is then decided by the compiler and the loops over the slabs
            // are straight-line code) and once more with the count read at run time (NS = 0).  The kernel spends its time
            // waiting on its own instruction stream, not on the vector units: some sixty scalar compare-and-branch pairs
            // per home particle, half of them taken, were most of that stream.
            auto walk = [&](auto ns_tag)
            {
            constexpr unsigned int NS = decltype(ns_tag)::value;
            const unsigned int nslab = NS ? NS : nslab_dyn;
            auto live = [&](int c) -> bool { return NS ? (unsigned int)c < NS : (unsigned int)(c * 64) < Tp; };
            for (unsigned int i = sub; i < nh; i += wpc)
                {
                const unsigned int iu = __builtin_amdgcn_readfirstlane(i);
                const unsigned int i_idx = rl_u(hidx, iu);
                if (i_idx >= a.N)
                    continue; // ghosts get no row
                const float pxs = __uint_as_float(rl_u(__float_as_uint(hxf), iu)), pys = __uint_as_float(rl_u(__float_as_uint(hyf), iu)),
                            pzs = __uint_as_float(rl_u(__float_as_uint(hzf), iu));
                const nl_v2f px = {pxs, pxs}, py = {pys, pys}, pz = {pzs, pzs};
                const unsigned int nmax_i = rl_u(hnmax, iu);
                const unsigned int cnt_old = rl_u(cnt, iu);
                // (a row laid out by type holds up to three entries of the dummy between its parts: room for them beyond
                // the capacity the overflow protocol counts in, which stays the reference's)
                const unsigned int nmax_w = TILE ? min(nmax_i + (BY_TYPE ? 3u : 0u), tile_row_cap) : nmax_i;
                const size_t row_head = TILE ? 0 : (((size_t)rl_u(hhead_hi, iu) << 32) | (size_t)rl_u(hhead_lo, iu));
                // thresholds of this particle's type against candidates of type 0 and 1
                const unsigned int ti = TWO ? rl_u(htype, iu) : 0u;
                const unsigned int lo1 = TWO ? (ti ? lo1_tp[1] : lo1_tp[0]) : lo1_tp[0];
                const unsigned int lo1_d = TWO ? ((ti ? lo1_tp[3] : lo1_tp[2]) - lo1) : 0u; // added for candidates of type 1
                const double rl2 = TWO ? (ti ? rl2_tp[1] : rl2_tp[0]) : rl2_tp[0];
                const double rl2_b = TWO ? (ti ? rl2_tp[3] : rl2_tp[2]) : rl2;

                // ---- test phase: nothing but vector arithmetic.  Every lane collects one bit per slab (slab c ends up
                // at bit nslab - 1 - c) and the smallest distance of any of its candidates from the undecided band. ----
                unsigned int bits = 0, nearest = 0xffffffffu;
                if (band != 0u)
                    {
#pragma unroll
                    for (int s = 0; s < (MAXK / 2); s++)
                        {
                        if (live(2 * s))
                            {
                            const nl_v2f dx = px - fx[s], dy = py - fy[s], dz = pz - fz[s];
                            if constexpr (!TWO)
                                {
                                // (the threshold folded into the chain, fma(dx, dx, -lo') first, saves an instruction per two
                                // slabs and 3 % of the kernel run by itself -- and cost 6 % of the time step in the
                                // simulation, measured back to back on one box; left out)
                                nl_v2f r = dx * dx;
                                r = __builtin_elementwise_fma(dy, dy, r);
                                r = __builtin_elementwise_fma(dz, dz, r);
                                const nl_v2f d = r - lo_next2;
                                const unsigned int d0 = __float_as_uint(d.x), d1 = __float_as_uint(d.y);
                                bits = __builtin_amdgcn_alignbit(bits, d0, 31);
                                if (live(2 * s + 1))
                                    {
                                    bits = __builtin_amdgcn_alignbit(bits, d1, 31);
                                    // (one three-operand minimum; left to itself the compiler pairs up the slabs' minima
                                    // first, which costs half as many instructions again)
                                    asm("v_min3_u32 %0, %0, %1, %2" : "+v"(nearest) : "v"(d0), "v"(d1));
                                    }
                                else
                                    nearest = min(nearest, d0);
                                continue;
                                }
                            nl_v2f r = dx * dx;
                            r = __builtin_elementwise_fma(dy, dy, r);
                            r = __builtin_elementwise_fma(dz, dz, r);
                            unsigned int t0 = __float_as_uint(r.x) - lo1;
                            if (TWO)
                                t0 -= (unsigned int)__builtin_amdgcn_sbfe((int)ctbits, 2 * s, 1) & lo1_d;
                            bits = __builtin_amdgcn_alignbit(bits, t0, 31);
                            nearest = min(nearest, t0);
                            if (live(2 * s + 1))
                                {
                                unsigned int t1 = __float_as_uint(r.y) - lo1;
                                if (TWO)
                                    t1 -= (unsigned int)__builtin_amdgcn_sbfe((int)ctbits, 2 * s + 1, 1) & lo1_d;
                                bits = __builtin_amdgcn_alignbit(bits, t1, 31);
                                nearest = min(nearest, t1);
                                }
                            }
                        }
                    }
                if (__ballot(nearest < (TWO ? band : band_f_bits)))
                    {
                    // a candidate of this particle sits in the band around r_list^2: every slab again with the CPU
                    // path's own arithmetic -- the difference first, then the box length of the image
                    // (hoomd/BoxDim.h:326-372 with one image; L * 1 is exact)
                    const double pix = rl_d(hx, iu), piy = rl_d(hy, iu), piz = rl_d(hz, iu);
                    bits = 0;
                    for (unsigned int c = 0; c < nslab; c++)
                        {
                        const unsigned int k = pbase + c * 64 + lane;
                        bool hit = false;
                        if (k < T)
                            {
                            unsigned int lo;
                            const unsigned int w = cand_lookup(k, lo);
                            const double4 v = hmd_ld4(a.d_cell_xyzf + w);
                            const double dx64 = HMD_SUB(HMD_SUB(pix, v.x), s_shift[3 * lo + 0]);
                            const double dy64 = HMD_SUB(HMD_SUB(piy, v.y), s_shift[3 * lo + 1]);
                            const double dz64 = HMD_SUB(HMD_SUB(piz, v.z), s_shift[3 * lo + 2]);
                            const double rsq = HMD_ADD(HMD_ADD(HMD_MUL(dx64, dx64), HMD_MUL(dy64, dy64)), HMD_MUL(dz64, dz64));
                            hit = rsq <= ((TWO && ((ctbits >> c) & 1u)) ? rl2_b : rl2);
                            }
                        bits = (bits << 1) | (hit ? 1u : 0u);
                        }
                    }
                // the particle itself, by its place in the candidate list
                const unsigned int k_self = self_first + hbase + iu - pbase; // (wraps below pbase: then >= MAXK * 64)
                if ((k_self >> 6) < nslab && lane == (k_self & 63u))
                    bits &= ~(1u << (nslab - 1u - (k_self >> 6)));
                // excluded partners, by particle index
                if (EXCL)
                    {
                    const unsigned int nex_i = rl_u(hnex, iu);
                    auto leave_out = [&](unsigned int e)
                    {
#pragma unroll
                        for (int c = 0; c < MAXK; c++)
                            if (live(c) && (TILE ? cidx[c] : co[c]) == e)
                                bits &= ~(1u << (nslab - 1u - (unsigned int)c));
                    };
#pragma unroll
                    for (unsigned int o = 0; o < (EXCL ? NEXR : 0u); o++)
                        if (o < nex_i)
                            leave_out(rl_u(hex[o], iu));
                    for (unsigned int o = NEXR; o < nex_i; o++)
                        leave_out(__builtin_amdgcn_readfirstlane(ex.d_ex_list[hmd_idx2(ex.exli, i_idx, o)]));
                    }

                if constexpr (BY_TYPE)
                    {
                    // ---- compaction by type: one scan carries both counts (low and high half of a word) ----
                    const unsigned int bA = mode == 2u ? 0u : bits & ~tbr, bB = mode == 1u ? 0u : bits & tbr;
                    const unsigned int mA = (unsigned int)__popc(bA), mB = (unsigned int)__popc(bB);
                    const unsigned int inc = nl_wave_scan_dpp(mA | (mB << 16));
                    const unsigned int tot = rl_u(inc, 63);
                    const unsigned int hA = tot & 0xffffu, hB = tot >> 16;
                    const unsigned int cB_old = rl_u(cntB, iu);
                    const unsigned int cA_new = cnt_old + hA;
                    const unsigned int nA4 = (cA_new + 3u) & ~3u; // (final in modes 0 and 2)
                    const unsigned int baseA = cnt_old + (inc & 0xffffu) - mA;
                    const unsigned int baseB = nA4 + cB_old + (inc >> 16) - mB;
                    const unsigned int lay_new = mode == 1u ? cA_new : nA4 + cB_old + hB; // entries of the row laid out so far
                    const unsigned int lay_lo = mode == 2u ? nA4 + cB_old : cnt_old;      // the first one this pass writes
                    unsigned int* wrow = s_rowbuf + NL_ROWBUF * wave;
                    const unsigned int h_i = tile_hoff + hbase + iu;
                    unsigned short* trow16 = trow_block + ((h_i / hmd::TL_GROUP) * (unsigned int)hmd::tile_group_entries(row_pieces) + (h_i % hmd::TL_GROUP) * hmd::TL_PIECE);
                    char* trow = reinterpret_cast<char*>(trow16);
                    const bool fits = mode == 0u && lay_new <= min(nmax_w, (unsigned int)NL_ROWBUF);
                    if (fits)
                        {
                        // (as below, once per type; cnt_old and cB_old are zero in mode 0)
                        unsigned long long saved_exec;
                        unsigned int b = bA << (32u - nslab);
                        unsigned int addr = (unsigned int)(size_t)(__attribute__((address_space(3))) void*)wrow + 4u * baseA - 4u;
#pragma unroll
                        for (int c = 0; c < MAXK; c++)
                            {
                            if (live(c))
                                asm volatile("v_add_co_u32_e32 %0, vcc, %0, %0\n\t"
                                             "s_and_saveexec_b64 %2, vcc\n\t"
                                             "v_add_u32_e32 %1, 4, %1\n\t"
                                             "ds_write_b32 %1, %3\n\t"
                                             "s_or_b64 exec, exec, %2"
                                             : "+v"(b), "+v"(addr), "=&s"(saved_exec)
                                             : "v"(co[c])
                                             : "vcc", "memory");
                            }
                        b = bB << (32u - nslab);
                        addr = (unsigned int)(size_t)(__attribute__((address_space(3))) void*)wrow + 4u * baseB - 4u;
#pragma unroll
                        for (int c = 0; c < MAXK; c++)
                            {
                            if (live(c))
                                asm volatile("v_add_co_u32_e32 %0, vcc, %0, %0\n\t"
                                             "s_and_saveexec_b64 %2, vcc\n\t"
                                             "v_add_u32_e32 %1, 4, %1\n\t"
                                             "ds_write_b32 %1, %3\n\t"
                                             "s_or_b64 exec, exec, %2"
                                             : "+v"(b), "+v"(addr), "=&s"(saved_exec)
                                             : "v"(co[c])
                                             : "vcc", "memory");
                            }
                        if (lane < nA4 - cA_new)
                            wrow[cA_new + lane] = tile_pad; // the dummy between the two parts
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        const unsigned int padded = (lay_new + 3u) & ~3u;
                        const unsigned int e = 2u * lane;
                        if (e < padded)
                            {
                            const unsigned long long w = *reinterpret_cast<const volatile unsigned long long*>(wrow + e);
                            const unsigned int w0 = (unsigned int)w, w1 = (unsigned int)(w >> 32);
                            const unsigned int v0 = e < lay_new ? w0 : tile_pad, v1 = e + 1u < lay_new ? w1 : tile_pad;
                            *reinterpret_cast<unsigned int*>(trow + (((lane >> 3) << 9) | ((lane & 7u) << 2))) = (v0 & 0xffffu) | (v1 << 16);
                            }
                        }
                    else
                        {
                        // long rows and cells swept in several passes: every hit finds its place by counting
                        auto put = [&](unsigned int pos, unsigned int what)
                        {
                            if (pos < nmax_w)
                                {
                                if (pos < (unsigned int)NL_ROWBUF)
                                    wrow[pos] = what;
                                else
                                    trow16[((pos & ~15u) << 4) | (pos & 15u)] = (unsigned short)what;
                                }
                        };
#pragma unroll
                        for (int c = 0; c < MAXK; c++)
                            {
                            if (live(c))
                                {
                                const unsigned int bp = nslab - 1u - (unsigned int)c;
                                if ((bA >> bp) & 1u)
                                    put(baseA + (unsigned int)__popc(bA >> 1 >> bp), co[c]);
                                if ((bB >> bp) & 1u)
                                    put(baseB + (unsigned int)__popc(bB >> 1 >> bp), co[c]);
                                }
                            }
                        // (the dummy between the parts: here when both parts are placed at once, after the sweeps otherwise)
                        if (mode == 0u && lane < nA4 - cA_new)
                            put(cA_new + lane, tile_pad);
                        const unsigned int stop = min(min(lay_new, nmax_w), (unsigned int)NL_ROWBUF);
#pragma unroll
                        for (unsigned int e0 = 0; e0 < (unsigned int)NL_ROWBUF; e0 += 64)
                            {
                            const unsigned int e = e0 + lane;
                            if (e >= lay_lo && e < stop)
                                *reinterpret_cast<unsigned short*>(trow + (((e & ~15u) << 5) | ((e & 15u) << 1))) = (unsigned short)wrow[e];
                            }
                        }
                    if (lane == iu)
                        {
                        cnt = cA_new;
                        cntB = cB_old + hB;
                        }
                    continue;
                    }
                // ---- compaction: hits ordered by lane, then slab.  One scan over the wave places every lane's hits. ----
                const unsigned int mine = (unsigned int)__popc(bits);
                const unsigned int incl = nl_wave_scan_dpp(mine);
                const unsigned int n_hits = rl_u(incl, 63);
                const unsigned int base = cnt_old + incl - mine;
                unsigned int* wrow = s_rowbuf + NL_ROWBUF * wave;
                const unsigned int cnt_new = cnt_old + n_hits;
                // (decided once per particle: the whole row fits its capacity and the staging buffer -- the rule)
                const bool fits = cnt_new <= min(nmax_w, (unsigned int)NL_ROWBUF);
                if (fits)
                    {
                    // the slab bits leave through the top of the register one by one; the running position steps
                    // forward on every hit
                    // Two vector instructions per slab: b + b shifts the next bit out through the carry, and under the
                    // lanes the carry selects the row address steps forward by one entry and the entry is stored
                    // (in assembly: the compiler has no way to say "branch on the carry of this addition").
                    unsigned int b = bits << (32u - nslab);
                    unsigned int addr = (unsigned int)(size_t)(__attribute__((address_space(3))) void*)wrow + 4u * base - 4u;
                    unsigned long long saved_exec;
#pragma unroll
                    for (int c = 0; c < MAXK; c++)
                        {
                        if (live(c))
                            asm volatile("v_add_co_u32_e32 %0, vcc, %0, %0\n\t"
                                         "s_and_saveexec_b64 %2, vcc\n\t"
                                         "v_add_u32_e32 %1, 4, %1\n\t"
                                         "ds_write_b32 %1, %3\n\t"
                                         "s_or_b64 exec, exec, %2"
                                         : "+v"(b), "+v"(addr), "=&s"(saved_exec)
                                         : "v"(co[c])
                                         : "vcc", "memory");
                        }
                    // (the stores above are invisible to the compiler's own count of outstanding LDS operations)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                else
                    {
#pragma unroll
                    for (int c = 0; c < MAXK; c++)
                        {
                        if (live(c))
                            {
                            const unsigned int bp = nslab - 1u - (unsigned int)c;
                            const unsigned int pos = base + (unsigned int)__popc(bits >> 1 >> bp); // this lane's hits in earlier slabs sit above bit bp
                            if (((bits >> bp) & 1u) && pos < nmax_w)
                                {
                                if (pos < (unsigned int)NL_ROWBUF)
                                    wrow[pos] = co[c];
                                else if (TILE) // rows longer than the staging buffer: straight to memory
                                    {
                                    const unsigned int h_i = tile_hoff + hbase + iu;
                                    unsigned short* trow = d_tile + (size_t)(tile_grp0 + h_i / hmd::TL_GROUP) * hmd::tile_group_entries(row_pieces) + (h_i % hmd::TL_GROUP) * hmd::TL_PIECE;
                                    trow[((pos & ~15u) << 4) | (pos & 15u)] = (unsigned short)co[c];
                                    }
                                else
                                    a.d_nlist[row_head + pos] = co[c];
                                }
                            }
                        }
                    }
                // ---- the staged part of the row goes to memory ----
                const unsigned int stop = min(min(cnt_new, nmax_w), (unsigned int)NL_ROWBUF);
                if (TILE)
                    {
                    // (the block's rows start at trow_block; within a block the offset of a row fits 32 bits)
                    const unsigned int h_i = tile_hoff + hbase + iu;
                    char* trow = reinterpret_cast<char*>(trow_block + ((h_i / hmd::TL_GROUP) * (unsigned int)hmd::tile_group_entries(row_pieces) + (h_i % hmd::TL_GROUP) * hmd::TL_PIECE));
                    if (fits && cnt_old == 0)
                        {
                        // one store instruction: every lane packs two entries, the padding to a multiple of four included
                        const unsigned int padded = (cnt_new + 3u) & ~3u;
                        const unsigned int e = 2u * lane;
                        if (e < padded)
                            {
                            // (one 8-byte read whatever the row's length: asked for entry by entry the compiler reads each
                            // under its own condition, one round trip to LDS after the other)
                            const unsigned long long w = *reinterpret_cast<const volatile unsigned long long*>(wrow + e);
                            const uint2 w2 = make_uint2((unsigned int)w, (unsigned int)(w >> 32));
                            const unsigned int v0 = e < cnt_new ? w2.x : tile_pad, v1 = e + 1u < cnt_new ? w2.y : tile_pad;
                            *reinterpret_cast<unsigned int*>(trow + (((lane >> 3) << 9) | ((lane & 7u) << 2))) = (v0 & 0xffffu) | (v1 << 16);
                            }
                        }
                    else
                        {
#pragma unroll
                        for (unsigned int e0 = 0; e0 < (unsigned int)NL_ROWBUF; e0 += 64)
                            {
                            const unsigned int e = e0 + lane;
                            if (e >= cnt_old && e < stop)
                                *reinterpret_cast<unsigned short*>(trow + (((e & ~15u) << 5) | ((e & 15u) << 1))) = (unsigned short)wrow[e];
                            }
                        }
                    }
                else
                    {
                    unsigned int* row = a.d_nlist + row_head;
#pragma unroll
                    for (unsigned int e0 = 0; e0 < (unsigned int)NL_ROWBUF; e0 += 64)
                        {
                        const unsigned int e = e0 + lane;
                        if (e >= cnt_old && e < stop)
                            row[e] = wrow[e];
                        }
                    }
                if (lane == iu)
                    cnt = cnt_new;
                }
            };
            switch (nslab_dyn)
                {
            case 8:
                walk(std::integral_constant<unsigned int, 8> {});
                break;
            case 9:
                walk(std::integral_constant<unsigned int, 9> {});
                break;
            case 10:
                walk(std::integral_constant<unsigned int, 10> {});
                break;
            case 11:
                walk(std::integral_constant<unsigned int, 11> {});
                break;
            case 12:
                walk(std::integral_constant<unsigned int, 12> {});
                break;
            case 13:
                if constexpr (MAXK >= 13)
                    walk(std::integral_constant<unsigned int, 13> {});
                else
                    walk(std::integral_constant<unsigned int, 0> {});
                break;
            case 14:
                if constexpr (MAXK >= 14)
                    walk(std::integral_constant<unsigned int, 14> {});
                else
                    walk(std::integral_constant<unsigned int, 0> {});
                break;
            default:
                walk(std::integral_constant<unsigned int, 0> {});
                }
            }


        // ---- per-particle results ----
        if (TILE && d_slot_meta && lane < nh && hidx >= a.N && (lane % wpc) == sub)
            {
            hmd_uint4 m = {hidx, 0u, 0u, 0u}; // a ghost: no row
            const unsigned int w = hmd_idx2(a.cli, hbase + lane, cell);
            d_slot_meta[w] = m;
            if (d_slot_of)
                {
                d_slot_of[hidx] = w;
                hmd_st4(d_cpos + w, hx, hy, hz, a.d_pos[hidx].w);
                }
            }
        if (lane < nh && hidx < a.N && (lane % wpc) == sub)
            {
            const double wtype = a.d_pos[hidx].w;
            // rows laid out by type: what the row holds (the dummy entries between the parts included) and what it lists
            const unsigned int n_first = cnt, n_first4 = (cnt + 3u) & ~3u;
            const unsigned int n_listed = BY_TYPE ? cnt + cntB : cnt;
            if (BY_TYPE)
                cnt = n_first4 + cntB;
            if (TILE && d_slot_meta)
                {
                hmd_uint4 m = {hidx, cnt, 0u, 0u};
                const unsigned int w = hmd_idx2(a.cli, hbase + lane, cell);
                d_slot_meta[w] = m;
                if (d_slot_of)
                    {
                    d_slot_of[hidx] = w;
                    hmd_st4(d_cpos + w, hx, hy, hz, wtype);
                    }
                }
            a.d_n_neigh[hidx] = n_listed;
            if (TILE)
                {
                const unsigned int h_i = tile_hoff + hbase + lane;
                unsigned short* trow = d_tile + (size_t)(tile_grp0 + h_i / hmd::TL_GROUP) * hmd::tile_group_entries(row_pieces) + (h_i % hmd::TL_GROUP) * hmd::TL_PIECE;
                // (whole chunks of four: the force kernel evaluates the last chunk of a row to its end)
                const unsigned int room = min((hnmax + (BY_TYPE ? 3u : 0u) + 3u) & ~3u, tile_row_cap);
                for (unsigned int k = cnt; k < ((cnt + 3u) & ~3u) && k < room; k++)
                    trow[((k & ~15u) << 4) | (k & 15u)] = (unsigned short)tile_pad;
                // (cells swept once per type: the dummy between the parts was not placed with the hits)
                if (BY_TYPE && nsweep == 2u)
                    for (unsigned int k = n_first; k < n_first4 && k < room; k++)
                        trow[((k & ~15u) << 4) | (k & 15u)] = (unsigned short)tile_pad;
                // rows with too few pieces for the dummy entries (row_pieces sized without them): reported as an overflow
                if (BY_TYPE && n_listed < hnmax && cnt > min(hnmax + 3u, tile_row_cap))
                    atomicMax(&a.d_conditions[htype], hnmax + 1u);
                }
            if (a.d_last_updated_pos)
                hmd_st4(a.d_last_updated_pos + hidx, hx, hy, hz, hmd_uint_as_double(hmd_double_as_uint(wtype)));
            if (n_listed >= hnmax)
                atomicMax(&a.d_conditions[htype], n_listed);
            }
        }
    }

// ---- the build for sparsely filled cells: one wave per block of 2x2x2 cells ----
//
// With a short cutoff a cell holds a handful of particles (3.3 in the bead-spring melt, r_list = 1.52), and a wave per
// cell spends most of its instructions on set-up that serves three home particles.  Here a wave takes a whole block of
// eight cells: the candidates are the members of the 4x4x4 cells around it (the tile path's region, about 210
// particles: four slabs of 64 in registers), every home particle of the block tests all of them with the same fp32
// pre-test and exact re-test as above, and a per-lane mask removes the candidates of region cells that are not
// adjacent to the home particle's own cell, so that the pairs considered are exactly those of the 27-cell walk of
// hoomd/md/NeighborListBinned.cc:85-176.  2.4 times the distance tests, a tenth of the set-up.  One type, orthorhombic box,
// at least three cells along every periodic direction, at most MAXK members per cell (so one pass holds the region).
template<bool TILE, bool EXCL, int MAXK>
__global__ void __launch_bounds__(NL_BLOCK) nlist_block_kernel(hmd_nlist_binned_args_t a, unsigned int nblocks, double wmax, unsigned short* d_tile,
                                                               unsigned int tile_capacity, unsigned int* d_tile_overflow, hmd_uint4* d_slot_meta,
                                                               unsigned int* d_slot_of, hmd_scalar4* d_cpos, const unsigned int* d_block_group,
                                                               unsigned int row_pieces, nl_excl_t ex)
    {
    extern __shared__ __align__(16) unsigned char s_raw[];
    // per wave: 64 exclusive offsets, 64 region cells, 64 image shifts, 8 x 64 adjacency words, the row buffer
    constexpr unsigned int PER_WAVE = 64 + 64 + 6 * 64 + 8 * 64 + NL_ROWBUF;
    const unsigned int lane = threadIdx.x & 63;
    const unsigned int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned int* s_excl = reinterpret_cast<unsigned int*>(s_raw) + PER_WAVE * wave;
    unsigned int* s_cell = s_excl + 64;
    double* s_shift = reinterpret_cast<double*>(s_cell + 64);
    unsigned int* s_adj = s_cell + 64 + 6 * 64;
    unsigned int* wrow = s_adj + 8 * 64;

    const double rc0 = a.d_r_cut[0];
    const double rl0 = HMD_ADD(rc0, a.r_buff);
    const double rl2 = (rc0 > 0.0) ? HMD_MUL(rl0, rl0) : -1.0;
    const nl_pref_t pf = nl_pref_thresholds(rl0, rl2, wmax);
    const unsigned int lo1 = pf.hi > 0.0f ? __float_as_uint(pf.lo) + 1u : 0u;
    const unsigned int band = pf.hi > 0.0f ? __float_as_uint(pf.hi) - __float_as_uint(pf.lo) : 0u;

    const unsigned int tb = xcd_remap(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + wave; // (one wave per workgroup, or four)
    if (tb >= nblocks)
        return;
    const int dim[3] = {(int)a.ci.w, (int)a.ci.h, (int)a.ci.d};
    const unsigned int nbx = (a.ci.w + hmd::TB_X - 1) / hmd::TB_X, nby = (a.ci.h + hmd::TB_Y - 1) / hmd::TB_Y;
    const int c0[3] = {hmd::TB_X * (int)(tb % nbx), hmd::TB_Y * (int)((tb / nbx) % nby), hmd::TB_Z * (int)(tb / (nbx * nby))};

    // ---- the region: cells, sizes -> exclusive offsets, image shifts ----
    int rimg[3];
    const int rcell = (int)lane < hmd::TL_RC ? hmd::tile_region_cell(a.ci, a.box, c0[0], c0[1], c0[2], (int)lane, rimg) : -1;
    const unsigned int rsize = rcell >= 0 ? min(a.d_cell_size[rcell], a.cli.w) : 0u;
    const unsigned int rincl = hmd::tile_wave_scan(rsize, (int)lane);
    const unsigned int T = rl_u(rincl, 63);
    s_excl[lane] = rincl - rsize;
    s_cell[lane] = (unsigned int)rcell;
    s_shift[3 * lane + 0] = rcell >= 0 ? (double)rimg[0] * a.box.L[0] : 0.0;
    s_shift[3 * lane + 1] = rcell >= 0 ? (double)rimg[1] * a.box.L[1] : 0.0;
    s_shift[3 * lane + 2] = rcell >= 0 ? (double)rimg[2] * a.box.L[2] : 0.0;

    // home cells of the block (those inside the grid): sizes and their running sum
    unsigned int hcum[hmd::TL_NH + 1];
    hcum[0] = 0;
#pragma unroll
    for (int q = 0; q < hmd::TL_NH; q++)
        {
        const int qx = q % hmd::TB_X, qy = (q / hmd::TB_X) % hmd::TB_Y, qz = q / (hmd::TB_X * hmd::TB_Y);
        const bool home = c0[0] + qx < dim[0] && c0[1] + qy < dim[1] && c0[2] + qz < dim[2];
        hcum[q + 1] = hcum[q] + (home ? rl_u(rsize, (unsigned int)hmd::tile_home_rc(q)) : 0u);
        }
    const unsigned int H = hcum[hmd::TL_NH];

    unsigned int tile_pad = 0, tile_grp0 = 0;
    const unsigned int tile_row_cap = row_pieces * hmd::TL_PIECE;
    if (TILE)
        {
        if (T > tile_capacity || __builtin_amdgcn_readfirstlane(d_tile_overflow[0]) != 0u)
            {
            if (lane == 0)
                d_tile_overflow[0] = 1u;
            return;
            }
        if (lane == 0 && T > d_tile_overflow[1])
            atomicMax(d_tile_overflow + 1, T);
        tile_pad = hmd::tile_slot(T);
        tile_grp0 = __builtin_amdgcn_readfirstlane(d_block_group[tb]);
        }
    if (H == 0)
        return;
    if (T > (unsigned int)(MAXK * 64))
        __builtin_trap(); // the launcher admits at most MAXK members per cell

    // candidate k of the region: its region cell (largest entry of the offsets that does not exceed k)
    auto cand_cell = [&](unsigned int k) -> unsigned int
    {
        unsigned int lo = 0;
#pragma unroll
        for (unsigned int step = 32; step > 0; step >>= 1)
            if (s_excl[lo + step] <= k) // (lo + step <= 63)
                lo += step;
        return lo;
    };
    const unsigned int nslab = (T + 63u) / 64u;

    // ---- candidates into registers: fp32 coordinates relative to the block's first home particle, what a row stores
    // for them, and for every home cell q one word per lane with a bit per slab: is this candidate's cell adjacent to q ----
    // The rest of the kernel exists twice: for regions of up to six slabs of candidates (384: the rule where cells hold
    // three particles) and for up to MAXK.  Every loop over the slabs is unrolled over the register arrays that hold
    // them, so the short version runs half the loop bodies and their guards.
    auto run = [&](auto kc)
    {
    constexpr int K = decltype(kc)::value;
    nl_v2f fx[K / 2], fy[K / 2], fz[K / 2];
    unsigned int co[K];
    unsigned int cidx[(EXCL && TILE) ? K : 1];
    unsigned int adjw[hmd::TL_NH];
#pragma unroll
    for (int q = 0; q < hmd::TL_NH; q++)
        adjw[q] = 0;
    // origin of the fp32 coordinates: the first candidate of the region (any fixed point near the block serves)
    double Ox = 0, Oy = 0, Oz = 0;
    constexpr int BATCH = K / 2;
#pragma unroll
    for (int b0 = 0; b0 < K; b0 += BATCH)
        {
        unsigned int w_[BATCH], lo_[BATCH];
#pragma unroll
        for (int q = 0; q < BATCH; q++)
            {
            const unsigned int k = (b0 + q) * 64 + lane;
            lo_[q] = 0;
            w_[q] = 0xffffffffu;
            if ((unsigned int)((b0 + q) * 64) < T && k < T)
                {
                lo_[q] = cand_cell(k);
                w_[q] = hmd_idx2(a.cli, k - s_excl[lo_[q]], s_cell[lo_[q]]);
                }
            }
        double2 vxy[BATCH], vzw[BATCH];
#pragma unroll
        for (int q = 0; q < BATCH; q++)
            {
            const double2* src = reinterpret_cast<const double2*>(a.d_cell_xyzf + (w_[q] == 0xffffffffu ? 0u : w_[q]));
            vxy[q] = src[0];
            vzw[q] = src[1];
            }
        if (b0 == 0)
            {
            Ox = rl_d(HMD_ADD(vxy[0].x, s_shift[3 * lo_[0] + 0]), 0);
            Oy = rl_d(HMD_ADD(vxy[0].y, s_shift[3 * lo_[0] + 1]), 0);
            Oz = rl_d(HMD_ADD(vzw[0].x, s_shift[3 * lo_[0] + 2]), 0);
            }
#pragma unroll
        for (int q = 0; q < BATCH; q++)
            {
            const int c = b0 + q;
            float x32 = 1.0e18f, y32 = x32, z32 = x32;
            co[c] = 0xffffffffu;
            if (EXCL && TILE)
                cidx[c] = 0xffffffffu;
            if (w_[q] != 0xffffffffu)
                {
                const unsigned int lo = lo_[q];
                x32 = (float)(HMD_ADD(vxy[q].x, s_shift[3 * lo + 0]) - Ox);
                y32 = (float)(HMD_ADD(vxy[q].y, s_shift[3 * lo + 1]) - Oy);
                z32 = (float)(HMD_ADD(vzw[q].x, s_shift[3 * lo + 2]) - Oz);
                const unsigned int k = c * 64 + lane;
                co[c] = TILE ? hmd::tile_slot(k) : hmd_double_as_uint(vzw[q].y);
                if (EXCL && TILE)
                    cidx[c] = hmd_double_as_uint(vzw[q].y);
                // adjacency of the candidate's region cell (coordinates 0..3 per axis) to the home cells (1..2 per axis)
                const unsigned int rx = lo % hmd::TR_X, ry = (lo / hmd::TR_X) % hmd::TR_Y, rz = lo / (hmd::TR_X * hmd::TR_Y);
#pragma unroll
                for (int hq = 0; hq < hmd::TL_NH; hq++)
                    {
                    const int hx = hq % hmd::TB_X + 1, hy = (hq / hmd::TB_X) % hmd::TB_Y + 1, hz = hq / (hmd::TB_X * hmd::TB_Y) + 1;
                    const bool adj = (int)rx >= hx - 1 && (int)rx <= hx + 1 && (int)ry >= hy - 1 && (int)ry <= hy + 1 && (int)rz >= hz - 1 && (int)rz <= hz + 1;
                    if (adj)
                        adjw[hq] |= 1u << (nslab - 1u - (unsigned int)c); // slab c at bit nslab - 1 - c, like the hit words
                    }
                }
            if (c & 1)
                {
                fx[c / 2].y = x32;
                fy[c / 2].y = y32;
                fz[c / 2].y = z32;
                }
            else
                {
                fx[c / 2].x = x32;
                fy[c / 2].x = y32;
                fz[c / 2].x = z32;
                }
            }
        }
#pragma unroll
    for (int q = 0; q < hmd::TL_NH; q++)
        s_adj[q * 64 + lane] = adjw[q];

    // ---- the home particles of the block, 64 at a time ----
    for (unsigned int hbase = 0; hbase < H; hbase += 64)
        {
        const unsigned int nh = min(H - hbase, 64u);
        double hx = 0, hy = 0, hz = 0;
        unsigned int hidx = 0xffffffffu, hnmax = 0, cnt = 0, hq = 0, hself = 0, hslot = 0;
        // what the loop over the home particles needs from memory per particle -- the start of its row, the number of its
        // exclusions and the first four of them -- is fetched here, one home particle per lane, and handed out by lane
        // reads: fetched where it is used, every one of these was a round trip to memory that the whole wave waited for
        // (three or more per home particle, most of the kernel's time where cells hold three particles).
        constexpr unsigned int NEXR = 4;
        unsigned int hnex = 0, hex[NEXR] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        unsigned int hhead_lo = 0, hhead_hi = 0;
        if (lane < nh)
            {
            const unsigned int hn = hbase + lane;
#pragma unroll
            for (int q = 1; q < hmd::TL_NH; q++)
                hq += hn >= hcum[q] ? 1u : 0u;
            unsigned int first = 0, rc = 0;
#pragma unroll
            for (int q = 0; q < hmd::TL_NH; q++)
                if ((unsigned int)q == hq)
                    {
                    first = hcum[q];
                    rc = (unsigned int)hmd::tile_home_rc(q);
                    }
            const unsigned int m = hn - first;
            hslot = hmd_idx2(a.cli, m, s_cell[rc]);
            hself = s_excl[rc] + m;
            const double4 v = hmd_ld4(a.d_cell_xyzf + hslot);
            hx = v.x;
            hy = v.y;
            hz = v.z;
            hidx = hmd_double_as_uint(v.w);
            if (hidx < a.N)
                {
                hnmax = a.d_Nmax[0];
                if (EXCL)
                    {
                    hnex = ex.d_n_ex[hidx];
#pragma unroll
                    for (unsigned int o = 0; o < NEXR; o++)
                        if (o < hnex)
                            hex[o] = ex.d_ex_list[hmd_idx2(ex.exli, hidx, o)];
                    }
                if (!TILE)
                    {
                    const size_t head = a.d_head_list[hidx];
                    hhead_lo = (unsigned int)head;
                    hhead_hi = (unsigned int)(head >> 32);
                    }
                }
            }
        const float hxf = (float)(hx - Ox), hyf = (float)(hy - Oy), hzf = (float)(hz - Oz);

        for (unsigned int i = 0; i < nh; i++)
            {
            const unsigned int iu = __builtin_amdgcn_readfirstlane(i);
            const unsigned int i_idx = rl_u(hidx, iu);
            if (i_idx >= a.N)
                continue; // ghosts get no row
            const float pxs = __uint_as_float(rl_u(__float_as_uint(hxf), iu)), pys = __uint_as_float(rl_u(__float_as_uint(hyf), iu)),
                        pzs = __uint_as_float(rl_u(__float_as_uint(hzf), iu));
            const nl_v2f px = {pxs, pxs}, py = {pys, pys}, pz = {pzs, pzs};
            const unsigned int nmax_i = rl_u(hnmax, iu);
            const unsigned int nmax_w = TILE ? min(nmax_i, tile_row_cap) : nmax_i;
            const unsigned int q_i = rl_u(hq, iu);

            unsigned int bits = 0, nearest = 0xffffffffu;
            if (band != 0u)
                {
#pragma unroll
                for (int s = 0; s < K / 2; s++)
                    {
                    if ((unsigned int)(2 * s * 64) < T)
                        {
                        const nl_v2f dx = px - fx[s], dy = py - fy[s], dz = pz - fz[s];
                        nl_v2f r = dx * dx;
                        r = __builtin_elementwise_fma(dy, dy, r);
                        r = __builtin_elementwise_fma(dz, dz, r);
                        const unsigned int t0 = __float_as_uint(r.x) - lo1;
                        bits = __builtin_amdgcn_alignbit(bits, t0, 31);
                        nearest = min(nearest, t0);
                        if ((unsigned int)((2 * s + 1) * 64) < T)
                            {
                            const unsigned int t1 = __float_as_uint(r.y) - lo1;
                            bits = __builtin_amdgcn_alignbit(bits, t1, 31);
                            nearest = min(nearest, t1);
                            }
                        }
                    }
                }
            if (__ballot(nearest < band))
                {
                // a candidate sits in the band around r_list^2: every slab again with the CPU path's own arithmetic
                const double pix = rl_d(hx, iu), piy = rl_d(hy, iu), piz = rl_d(hz, iu);
                bits = 0;
                for (unsigned int c = 0; c < nslab; c++)
                    {
                    const unsigned int k = c * 64 + lane;
                    bool hit = false;
                    if (k < T)
                        {
                        const unsigned int lo = cand_cell(k);
                        const double4 v = hmd_ld4(a.d_cell_xyzf + hmd_idx2(a.cli, k - s_excl[lo], s_cell[lo]));
                        const double dx64 = HMD_SUB(HMD_SUB(pix, v.x), s_shift[3 * lo + 0]);
                        const double dy64 = HMD_SUB(HMD_SUB(piy, v.y), s_shift[3 * lo + 1]);
                        const double dz64 = HMD_SUB(HMD_SUB(piz, v.z), s_shift[3 * lo + 2]);
                        const double rsq = HMD_ADD(HMD_ADD(HMD_MUL(dx64, dx64), HMD_MUL(dy64, dy64)), HMD_MUL(dz64, dz64));
                        hit = rsq <= rl2;
                        }
                    bits = (bits << 1) | (hit ? 1u : 0u);
                    }
                }
            // only candidates of the 27 cells around the particle's own cell; not the particle itself
            bits &= s_adj[q_i * 64 + lane];
            const unsigned int k_self = rl_u(hself, iu);
            if (lane == (k_self & 63u))
                bits &= ~(1u << (nslab - 1u - (k_self >> 6)));
            if (EXCL)
                {
                const unsigned int nex_i = rl_u(hnex, iu);
                auto leave_out = [&](unsigned int e)
                {
#pragma unroll
                    for (int c = 0; c < K; c++)
                        if ((unsigned int)(c * 64) < T && (TILE ? cidx[c] : co[c]) == e)
                            bits &= ~(1u << (nslab - 1u - (unsigned int)c));
                };
#pragma unroll
                for (unsigned int o = 0; o < NEXR; o++)
                    if (o < nex_i)
                        leave_out(rl_u(hex[o], iu));
                for (unsigned int o = NEXR; o < nex_i; o++)
                    leave_out(__builtin_amdgcn_readfirstlane(ex.d_ex_list[hmd_idx2(ex.exli, i_idx, o)]));
                }

            // ---- compaction: hits ordered by lane, then slab ----
            const unsigned int mine = (unsigned int)__popc(bits);
            const unsigned int incl = nl_wave_scan_dpp(mine);
            const unsigned int cnt_new = rl_u(incl, 63);
            const unsigned int base = incl - mine;
            const unsigned int h_i = hbase + iu;
            unsigned short* trow16 = TILE ? d_tile + (size_t)(tile_grp0 + h_i / hmd::TL_GROUP) * hmd::tile_group_entries(row_pieces) + (h_i % hmd::TL_GROUP) * hmd::TL_PIECE
                                          : nullptr;
            unsigned int* irow = TILE ? nullptr : a.d_nlist + (((size_t)rl_u(hhead_hi, iu) << 32) | (size_t)rl_u(hhead_lo, iu));
            const bool fits = cnt_new <= min(nmax_w, (unsigned int)NL_ROWBUF);
            if (fits)
                {
                unsigned int b = bits << (32u - nslab);
                unsigned int pos = base - 1u;
#pragma unroll
                for (int c = 0; c < K; c++)
                    {
                    if ((unsigned int)(c * 64) < T)
                        {
                        const bool hit = (int)b < 0;
                        b <<= 1;
                        pos += hit ? 1u : 0u;
                        if (hit)
                            wrow[pos] = co[c];
                        }
                    }
                if (TILE)
                    {
                    const unsigned int padded = (cnt_new + 3u) & ~3u;
                    const unsigned int e = 2u * lane;
                    if (e < padded)
                        {
                        const uint2 w2 = *reinterpret_cast<const uint2*>(wrow + e);
                        const unsigned int v0 = e < cnt_new ? w2.x : tile_pad, v1 = e + 1u < cnt_new ? w2.y : tile_pad;
                        *reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(trow16) + (((lane >> 3) << 9) | ((lane & 7u) << 2))) = (v0 & 0xffffu) | (v1 << 16);
                        }
                    }
                else
                    {
#pragma unroll
                    for (unsigned int e0 = 0; e0 < (unsigned int)NL_ROWBUF; e0 += 64)
                        if (e0 + lane < cnt_new)
                            irow[e0 + lane] = wrow[e0 + lane];
                    }
                }
            else
                {
                // the row is longer than its capacity (the caller grows Nmax and builds again) or than the staging
                // buffer: entries that have room go straight to memory
#pragma unroll
                for (int c = 0; c < K; c++)
                    {
                    if ((unsigned int)(c * 64) < T)
                        {
                        const unsigned int bp = nslab - 1u - (unsigned int)c;
                        const unsigned int pos = base + (unsigned int)__popc(bits >> 1 >> bp);
                        if (((bits >> bp) & 1u) && pos < nmax_w)
                            {
                            if (TILE)
                                trow16[((pos & ~15u) << 4) | (pos & 15u)] = (unsigned short)co[c];
                            else
                                irow[pos] = co[c];
                            }
                        }
                    }
                }
            if (lane == iu)
                cnt = cnt_new;
            }

        // ---- per-particle results ----
        if (TILE && d_slot_meta && lane < nh && hidx >= a.N)
            {
            hmd_uint4 m = {hidx, 0u, 0u, 0u}; // a ghost: no row
            d_slot_meta[hslot] = m;
            if (d_slot_of)
                {
                d_slot_of[hidx] = hslot;
                hmd_st4(d_cpos + hslot, hx, hy, hz, a.d_pos[hidx].w);
                }
            }
        if (lane < nh && hidx < a.N)
            {
            const double wtype = a.d_pos[hidx].w;
            if (TILE && d_slot_meta)
                {
                hmd_uint4 m = {hidx, cnt, 0u, 0u};
                d_slot_meta[hslot] = m;
                if (d_slot_of)
                    {
                    d_slot_of[hidx] = hslot;
                    hmd_st4(d_cpos + hslot, hx, hy, hz, wtype);
                    }
                }
            a.d_n_neigh[hidx] = cnt;
            if (TILE)
                {
                const unsigned int h_i = hbase + lane;
                unsigned short* trow = d_tile + (size_t)(tile_grp0 + h_i / hmd::TL_GROUP) * hmd::tile_group_entries(row_pieces) + (h_i % hmd::TL_GROUP) * hmd::TL_PIECE;
                for (unsigned int k = cnt; k < ((cnt + 3u) & ~3u) && k < min(hnmax, tile_row_cap); k++)
                    trow[((k & ~15u) << 4) | (k & 15u)] = (unsigned short)tile_pad;
                }
            if (a.d_last_updated_pos)
                hmd_st4(a.d_last_updated_pos + hidx, hx, hy, hz, hmd_uint_as_double(hmd_double_as_uint(wtype)));
            if (cnt >= hnmax)
                atomicMax(&a.d_conditions[0], cnt);
            }
        }
    };
    if (MAXK > 6 && T <= 6u * 64u)
        run(std::integral_constant<int, (MAXK > 6 ? 6 : MAXK)> {});
    else
        run(std::integral_constant<int, MAXK> {});
    }

template<bool ORTHO>
__global__ void __launch_bounds__(256) nlist_check_kernel(hmd_nlist_check_args_t a)
    {
    const unsigned int idx = blockIdx.x * blockDim.x + threadIdx.x;
    bool moved = false;
    if (idx < a.N)
        {
        const double4 cur = hmd_ld4(a.d_pos + idx);
        const double4 last = hmd_ld4(a.d_last_pos + idx);
        double dx = HMD_SUB(cur.x, HMD_MUL(a.lambda.x, last.x));
        double dy = HMD_SUB(cur.y, HMD_MUL(a.lambda.y, last.y));
        double dz = HMD_SUB(cur.z, HMD_MUL(a.lambda.z, last.z));
        hmd_min_image<ORTHO>(a.box, dx, dy, dz);
        const double rmin = a.d_rcut_max[hmd_double_as_uint(cur.w)];
        const double rmax = HMD_ADD(rmin, a.r_buff);
        const double delta_max = HMD_SUB(HMD_MUL(rmax, a.lambda_min), rmin) / 2.0;
        const double maxsq = (delta_max > 0) ? HMD_MUL(delta_max, delta_max) : 0.0;
        moved = HMD_ADD(HMD_ADD(HMD_MUL(dx, dx), HMD_MUL(dy, dy)), HMD_MUL(dz, dz)) >= maxsq;
        }
    // one atomic per wave that saw a mover
    if (__ballot(moved) && (threadIdx.x & 63) == 0)
        atomicMax(a.d_result, a.checkn);
    }

// Queued behind the check kernel: hands the flag to the host through pinned memory.  One 8-byte
// system-scope store carries the flag and the caller's stamp together, so the polling host never
// sees one without the other.
__global__ void nlist_publish_kernel(const unsigned int* d_result, unsigned long long* d_published, unsigned int stamp)
    {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        {
        const unsigned long long v = ((unsigned long long)stamp << 32) | (unsigned long long)(*d_result);
        __hip_atomic_store(d_published, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }

// ---- head list: exclusive scan of Nmax[type_i] ----
constexpr unsigned int HL_ITEMS = 8;
constexpr unsigned int HL_BLOCK = 256;
constexpr unsigned int HL_TILE = HL_ITEMS * HL_BLOCK;

__device__ __forceinline__ unsigned long long block_exclusive_scan(unsigned long long v, unsigned long long* s_wave, unsigned long long& total)
    {
    const unsigned int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
        {
        const unsigned long long t = __shfl_up(incl, off);
        if ((int)lane >= off)
            incl += t;
        }
    if (lane == 63)
        s_wave[wave] = incl;
    __syncthreads();
    unsigned long long base = 0;
    total = 0;
    for (unsigned int w = 0; w < HL_BLOCK / 64; w++)
        {
        if (w < wave)
            base += s_wave[w];
        total += s_wave[w];
        }
    __syncthreads();
    return base + incl - v;
    }

__global__ void __launch_bounds__(HL_BLOCK) head_uniform_kernel(size_t* d_head, size_t* d_req, const unsigned int* d_Nmax, unsigned int N)
    {
    const unsigned int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long nm = d_Nmax[0];
    if (idx < N)
        d_head[idx] = (size_t)idx * nm;
    if (idx == 0)
        *d_req = (size_t)N * nm;
    }

__global__ void __launch_bounds__(HL_BLOCK) head_tile_sums_kernel(unsigned long long* d_tile_sum, const unsigned int* d_Nmax, const hmd_scalar4* d_pos, unsigned int N)
    {
    __shared__ unsigned long long s_wave[HL_BLOCK / 64];
    unsigned long long v = 0;
    const unsigned int base = blockIdx.x * HL_TILE + threadIdx.x * HL_ITEMS;
    for (unsigned int k = 0; k < HL_ITEMS; k++)
        if (base + k < N)
            v += d_Nmax[hmd_double_as_uint(d_pos[base + k].w)];
    unsigned long long total;
    block_exclusive_scan(v, s_wave, total);
    if (threadIdx.x == 0)
        d_tile_sum[blockIdx.x] = total;
    }

__global__ void __launch_bounds__(HL_BLOCK) head_scan_tiles_kernel(unsigned long long* d_tile_sum, unsigned int ntiles, size_t* d_req)
    {
    __shared__ unsigned long long s_wave[HL_BLOCK / 64];
    unsigned long long carry = 0;
    for (unsigned int start = 0; start < ntiles; start += HL_BLOCK)
        {
        const unsigned int t = start + threadIdx.x;
        const unsigned long long v = (t < ntiles) ? d_tile_sum[t] : 0ull;
        unsigned long long total;
        const unsigned long long ex = block_exclusive_scan(v, s_wave, total);
        if (t < ntiles)
            d_tile_sum[t] = carry + ex;
        carry += total;
        }
    if (threadIdx.x == 0)
        *d_req = (size_t)carry;
    }

__global__ void __launch_bounds__(HL_BLOCK) head_write_kernel(size_t* d_head, const unsigned long long* d_tile_sum, const unsigned int* d_Nmax, const hmd_scalar4* d_pos, unsigned int N)
    {
    __shared__ unsigned long long s_wave[HL_BLOCK / 64];
    unsigned int nm[HL_ITEMS];
    unsigned long long v = 0;
    const unsigned int base = blockIdx.x * HL_TILE + threadIdx.x * HL_ITEMS;
#pragma unroll
    for (unsigned int k = 0; k < HL_ITEMS; k++)
        {
        nm[k] = (base + k < N) ? d_Nmax[hmd_double_as_uint(d_pos[base + k].w)] : 0u;
        v += nm[k];
        }
    unsigned long long total;
    unsigned long long run = d_tile_sum[blockIdx.x] + block_exclusive_scan(v, s_wave, total);
#pragma unroll
    for (unsigned int k = 0; k < HL_ITEMS; k++)
        {
        if (base + k < N)
            d_head[base + k] = (size_t)run;
        run += nm[k];
        }
    }

__global__ void __launch_bounds__(256) nlist_filter_kernel(hmd_nlist_filter_args_t a)
    {
    const unsigned int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.N)
        return;
    const unsigned int n_ex = a.d_n_ex[idx];
    if (n_ex == 0)
        return;
    const unsigned int nn = a.d_n_neigh[idx];
    unsigned int* row = a.d_nlist + a.d_head_list[idx];
    unsigned int kept = 0;
    for (unsigned int k = 0; k < nn; k++)
        {
        const unsigned int j = row[k];
        bool excluded = false;
        for (unsigned int e = 0; e < n_ex; e++)
            excluded = excluded || (a.d_ex_list[hmd_idx2(a.exli, idx, e)] == j);
        if (!excluded)
            {
            if (kept != k)
                row[kept] = j;
            kept++;
            }
        }
    a.d_n_neigh[idx] = kept;
    }

__global__ void __launch_bounds__(256) update_exclusion_kernel(hmd_update_exclusion_args_t a)
    {
    const unsigned int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.N)
        return;
    const unsigned int tag = a.d_tag[idx];
    const unsigned int n = a.d_n_ex_tag[tag];
    a.d_n_ex_idx[idx] = n;
    for (unsigned int o = 0; o < n; o++)
        a.d_ex_list_idx[hmd_idx2(a.ex_list_indexer, idx, o)] = a.d_rtag[a.d_ex_list_tag[hmd_idx2(a.ex_list_tag_indexer, tag, o)]];
    }

// Numbers the groups of tile rows: block b (2x2x2 cells) gets ceil(members of its home cells / 16) groups starting at
// d_block_group[b]; d_block_group[number of blocks] = all groups.  One workgroup: every thread sums a run of
// consecutive blocks, the runs are scanned, every thread writes its run.
constexpr unsigned int BG_THREADS = 1024;
// Grids with more blocks than the numbering workgroup can keep in its LDS: the group counts of the blocks are worked out
// by a launch over the whole chip first (one thread per block, its eight cell sizes in flight together) and left in
// d_block_group, where the numbering kernel turns them into starts (cached == 2).
__global__ void __launch_bounds__(256) tile_block_counts_kernel(const unsigned int* d_cell_size, hmd_index3d_t ci, unsigned int cell_Nmax, unsigned int* d_block_group)
    {
    const unsigned int nbx = (ci.w + hmd::TB_X - 1) / hmd::TB_X, nby = (ci.h + hmd::TB_Y - 1) / hmd::TB_Y, nbz = (ci.d + hmd::TB_Z - 1) / hmd::TB_Z;
    const unsigned int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbx * nby * nbz)
        return;
    const unsigned int cx0 = hmd::TB_X * (b % nbx), cy0 = hmd::TB_Y * ((b / nbx) % nby), cz0 = hmd::TB_Z * (b / (nbx * nby));
    unsigned int v[8];
#pragma unroll
    for (int q = 0; q < 8; q++)
        {
        const unsigned int x = cx0 + q % hmd::TB_X, y = cy0 + (q / hmd::TB_X) % hmd::TB_Y, z = cz0 + q / (hmd::TB_X * hmd::TB_Y);
        v[q] = d_cell_size[(min(z, ci.d - 1u) * ci.h + min(y, ci.h - 1u)) * ci.w + min(x, ci.w - 1u)];
        }
    unsigned int members = 0;
#pragma unroll
    for (int q = 0; q < 8; q++)
        {
        const unsigned int x = cx0 + q % hmd::TB_X, y = cy0 + (q / hmd::TB_X) % hmd::TB_Y, z = cz0 + q / (hmd::TB_X * hmd::TB_Y);
        members += (x < ci.w && y < ci.h && z < ci.d) ? min(v[q], cell_Nmax) : 0u;
        }
    d_block_group[b] = (members + hmd::TL_GROUP - 1) / hmd::TL_GROUP;
    }

__global__ void __launch_bounds__(BG_THREADS) tile_block_groups_kernel(const unsigned int* d_cell_size, hmd_index3d_t ci, unsigned int cell_Nmax,
                                                                       unsigned int* d_block_group, unsigned int tile_groups, unsigned int* d_tile_overflow, unsigned int cached)
    {
    __shared__ unsigned int s_wave[BG_THREADS / 64];
    const unsigned int nbx = (ci.w + hmd::TB_X - 1) / hmd::TB_X, nby = (ci.h + hmd::TB_Y - 1) / hmd::TB_Y, nbz = (ci.d + hmd::TB_Z - 1) / hmd::TB_Z;
    const unsigned int nb = nbx * nby * nbz;
    const unsigned int per = (nb + BG_THREADS - 1) / BG_THREADS;
    const unsigned int first = min(threadIdx.x * per, nb), last = min(first + per, nb);
    // Cells past the edge of the grid: the load goes to a cell at the edge instead, without a branch, and the values pass
    // through an empty asm statement before they are used -- otherwise the compiler moves every load behind the test of
    // its cell and waits for each in turn (eight round trips to memory per block).
    static_assert(hmd::TL_NH == 8, "eight home cells per block");
    auto load_sizes = [&](unsigned int b, unsigned int (&v)[8])
    {
        const unsigned int cx0 = hmd::TB_X * (b % nbx), cy0 = hmd::TB_Y * ((b / nbx) % nby), cz0 = hmd::TB_Z * (b / (nbx * nby));
#pragma unroll
        for (int q = 0; q < 8; q++)
            {
            const unsigned int x = cx0 + q % hmd::TB_X, y = cy0 + (q / hmd::TB_X) % hmd::TB_Y, z = cz0 + q / (hmd::TB_X * hmd::TB_Y);
            v[q] = d_cell_size[(min(z, ci.d - 1u) * ci.h + min(y, ci.h - 1u)) * ci.w + min(x, ci.w - 1u)];
            }
    };
    auto arrived = [](unsigned int (&v)[8], unsigned int (&w)[8])
    { asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7])); };
    auto count_groups = [&](unsigned int b, const unsigned int (&v)[8]) -> unsigned int
    {
        const unsigned int cx0 = hmd::TB_X * (b % nbx), cy0 = hmd::TB_Y * ((b / nbx) % nby), cz0 = hmd::TB_Z * (b / (nbx * nby));
        unsigned int members = 0;
#pragma unroll
        for (int q = 0; q < 8; q++)
            {
            const unsigned int x = cx0 + q % hmd::TB_X, y = cy0 + (q / hmd::TB_X) % hmd::TB_Y, z = cz0 + q / (hmd::TB_X * hmd::TB_Y);
            members += (x < ci.w && y < ci.h && z < ci.d) ? min(v[q], cell_Nmax) : 0u;
            }
        return (members + hmd::TL_GROUP - 1) / hmd::TL_GROUP;
    };
    auto groups_of = [&](unsigned int b) -> unsigned int
    {
        unsigned int v[8], w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        load_sizes(b, v);
        arrived(v, w);
        return count_groups(b, v);
    };
    // counts first, with neighboring threads on neighboring blocks (when they fit the LDS the launcher provided):
    // every thread then sums and later numbers its run of consecutive blocks from LDS
    extern __shared__ unsigned int s_groups[];
    if (cached == 1u)
        {
        // four blocks per trip: 32 loads in flight per thread
        for (unsigned int b = threadIdx.x; b < nb; b += 4u * BG_THREADS)
            {
            const unsigned int b1 = min(b + BG_THREADS, nb - 1u), b2 = min(b + 2u * BG_THREADS, nb - 1u), b3 = min(b + 3u * BG_THREADS, nb - 1u);
            unsigned int v0[8], v1[8], v2[8], v3[8];
            load_sizes(b, v0);
            load_sizes(b1, v1);
            load_sizes(b2, v2);
            load_sizes(b3, v3);
            arrived(v0, v1);
            arrived(v2, v3);
            s_groups[b] = count_groups(b, v0);
            if (b + BG_THREADS < nb)
                s_groups[b1] = count_groups(b1, v1);
            if (b + 2u * BG_THREADS < nb)
                s_groups[b2] = count_groups(b2, v2);
            if (b + 3u * BG_THREADS < nb)
                s_groups[b3] = count_groups(b3, v3);
            }
        __syncthreads();
        }
    unsigned int mine = 0;
    for (unsigned int b = first; b < last; b++)
        mine += cached == 1u ? s_groups[b] : (cached == 2u ? d_block_group[b] : groups_of(b));
    // exclusive scan over the threads
    const unsigned int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned int incl = hmd::tile_wave_scan(mine, (int)lane);
    if (lane == 63)
        s_wave[wave] = incl;
    __syncthreads();
    unsigned int base = 0, total = 0;
    for (unsigned int w = 0; w < BG_THREADS / 64; w++)
        {
        if (w < wave)
            base += s_wave[w];
        total += s_wave[w];
        }
    unsigned int run = base + incl - mine;
    for (unsigned int b = first; b < last; b++)
        {
        const unsigned int c = cached == 1u ? s_groups[b] : (cached == 2u ? d_block_group[b] : groups_of(b)); // (read before it is overwritten)
        d_block_group[b] = run;
        run += c;
        }
    if (threadIdx.x == 0)
        {
        d_block_group[nb] = total;
        // the two words of the build that follows start from zero here (a launch less than a memset of their own)
        d_tile_overflow[0] = total > tile_groups ? 1u : 0u;
        d_tile_overflow[1] = 0u;
        }
    }

// threads per workgroup of the build kernels whose waves work on their own: 64, or HOOMD_AMD_NL_BLOCK (64, 128 or 256)
static unsigned int nl_wave_block()
    {
    if (const char* e = std::getenv("HOOMD_AMD_NL_BLOCK"))
        {
        const int v = std::atoi(e);
        if (v == 64 || v == 128 || v == 256)
            return (unsigned int)v;
        }
    return 64u;
    }

template<bool ST, bool FB>
int launch_nlist(const hmd_nlist_binned_args_t& a, unsigned int ncells, unsigned int wpc, size_t lds, hipStream_t s, unsigned short* d_tile = nullptr,
                 unsigned int tile_capacity = 0, unsigned int* d_tile_overflow = nullptr, hmd_uint4* d_slot_meta = nullptr,
                 unsigned int* d_slot_of = nullptr, hmd_scalar4* d_cpos = nullptr, const unsigned int* d_block_group = nullptr, unsigned int row_pieces = 0,
                 nl_excl_t ex = nl_excl_t {nullptr, nullptr, hmd_index2d_t {0, 0}})
    {
    const nl_excl_t no_ex = {nullptr, nullptr, hmd_index2d_t {0, 0}};
    const unsigned int cells_per_block = NL_WAVES / wpc;
    const dim3 grid(hmd_div_up(ncells, cells_per_block));
    // fixed images need a radius-1 stencil in 3D whose cells are distinct along every periodic direction
    const unsigned int dim[3] = {a.ci.w, a.ci.h, a.ci.d};
    bool fixed_images = a.cadji.w == 27;
    for (int d = 0; d < 3; d++)
        if (a.box.periodic[d] && dim[d] < 3)
            fixed_images = false;
    // sparsely filled cells (at most 12 members in any cell; one type): a wave per block of eight cells
    if (ST && !FB && fixed_images && hmd_box_is_ortho(a.box) && a.cli.w <= 12 && !std::getenv("HOOMD_AMD_NLIST_EXACT_ONLY")
        && !std::getenv("HOOMD_AMD_NLIST_NO_BLOCK_BUILD"))
        {
        double wmax = 0.0;
        const double gw[3] = {a.ghost_width.x, a.ghost_width.y, a.ghost_width.z};
        for (int d = 0; d < 3; d++)
            wmax = std::max(wmax, (a.box.L[d] + 2.0 * gw[d]) / (double)dim[d]);
        // (fp32 coordinates are relative to a point of the 4x4x4-cell region: up to four cell widths away, not two)
        wmax *= 2.0;
        const unsigned int nblk = hmd::tile_num_blocks(a.ci);
        // One wave per workgroup: the waves share nothing, and the wave slots and the LDS of a finished block are free at
        // once instead of when the slowest of four is done (Kremer-Grest step 0.378 -> 0.371 ms).  HOOMD_AMD_NL_BLOCK=256
        // brings the four-wave workgroups back (A/B).
        const unsigned int blk_block = nl_wave_block();
        const dim3 bgrid(hmd_div_up(nblk, blk_block / 64u));
        const size_t lds_blk = sizeof(unsigned int) * (64 + 64 + 6 * 64 + 8 * 64 + NL_ROWBUF) * (blk_block / 64u);
#define HMD_BLOCK_LAUNCH(TILE_, EXCL_, ...) \
    hipLaunchKernelGGL((nlist_block_kernel<TILE_, EXCL_, 12>), bgrid, dim3(blk_block), lds_blk, s, a, nblk, wmax, __VA_ARGS__)
        if (d_tile && ex.d_n_ex)
            HMD_BLOCK_LAUNCH(true, true, d_tile, tile_capacity, d_tile_overflow, d_slot_meta, d_slot_of, d_cpos, d_block_group, row_pieces, ex);
        else if (d_tile)
            HMD_BLOCK_LAUNCH(true, false, d_tile, tile_capacity, d_tile_overflow, d_slot_meta, d_slot_of, d_cpos, d_block_group, row_pieces, no_ex);
        else if (ex.d_n_ex)
            HMD_BLOCK_LAUNCH(false, true, nullptr, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, ex);
        else
            HMD_BLOCK_LAUNCH(false, false, nullptr, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, no_ex);
#undef HMD_BLOCK_LAUNCH
        return hmd_launch_status();
        }
    // one or two types, no body filter, fixed images, orthorhombic: the build with the fp32 pre-test
    if ((ST || a.ntypes == 2) && !FB && fixed_images && hmd_box_is_ortho(a.box) && !std::getenv("HOOMD_AMD_NLIST_EXACT_ONLY"))
        {
        double wmax = 0.0;
        const double gw[3] = {a.ghost_width.x, a.ghost_width.y, a.ghost_width.z};
        for (int d = 0; d < 3; d++)
            wmax = std::max(wmax, (a.box.L[d] + 2.0 * gw[d]) / (double)dim[d]);
        // A workgroup per cell (one wave) where a cell has one wave: the waves share nothing, and the wave slots and the LDS
        // of a finished cell are free at once, not when the slowest of four cells is done (build 423 -> 408 us on the LJ
        // liquid, Kob-Andersen step 1.365 -> 1.337 ms)
        const unsigned int pref_block = wpc == 1 ? nl_wave_block() : NL_BLOCK;
        const dim3 grid(hmd_div_up(ncells, (pref_block / 64u) / wpc));
        const size_t lds_pref = (sizeof(unsigned int) * (192 + NL_ROWBUF + 32) + sizeof(double) * 3 * 64 + nl_pref_flag_bytes(a.cli.w)) * (pref_block / 64u);
// candidates in registers: 12 slabs of 64 cover the 27 cells of liquids up to 28 particles per cell in one pass (the
// LJ liquid has 21.5); denser systems (the binary mixture at density 1.2: 29) get 14 slabs, at three waves per SIMD
// instead of four, rather than a second pass over a handful of candidates.  Decided by the mean occupancy.
        const bool dense = 27.0 * (double)a.N / (double)ncells > 0.92 * 12.0 * 64.0;
#define HMD_PREF_LAUNCH(TILE_, EXCL_, TWO_, ...)                                                                                                   \
    do                                                                                                                                             \
        {                                                                                                                                          \
        if (dense)                                                                                                                                 \
            hipLaunchKernelGGL((nlist_pref_kernel<TILE_, EXCL_, TWO_, 14>), grid, dim3(pref_block), lds_pref, s, a, ncells, wpc, wmax, __VA_ARGS__); \
        else                                                                                                                                       \
            hipLaunchKernelGGL((nlist_pref_kernel<TILE_, EXCL_, TWO_, 12>), grid, dim3(pref_block), lds_pref, s, a, ncells, wpc, wmax, __VA_ARGS__); \
        } while (0)
        const bool two = !ST;
        if (d_tile && ex.d_n_ex)
            {
            if (two)
                HMD_PREF_LAUNCH(true, true, true, d_tile, tile_capacity, d_tile_overflow, d_slot_meta, d_slot_of, d_cpos, d_block_group, row_pieces, ex);
            else
                HMD_PREF_LAUNCH(true, true, false, d_tile, tile_capacity, d_tile_overflow, d_slot_meta, d_slot_of, d_cpos, d_block_group, row_pieces, ex);
            }
        else if (d_tile)
            {
            if (two)
                HMD_PREF_LAUNCH(true, false, true, d_tile, tile_capacity, d_tile_overflow, d_slot_meta, d_slot_of, d_cpos, d_block_group, row_pieces, no_ex);
            else
                HMD_PREF_LAUNCH(true, false, false, d_tile, tile_capacity, d_tile_overflow, d_slot_meta, d_slot_of, d_cpos, d_block_group, row_pieces, no_ex);
            }
        else if (ex.d_n_ex)
            {
            if (two)
                HMD_PREF_LAUNCH(false, true, true, nullptr, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, ex);
            else
                HMD_PREF_LAUNCH(false, true, false, nullptr, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, ex);
            }
        else
            {
            if (two)
                HMD_PREF_LAUNCH(false, false, true, nullptr, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, no_ex);
            else
                HMD_PREF_LAUNCH(false, false, false, nullptr, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, no_ex);
            }
#undef HMD_PREF_LAUNCH
        return hmd_launch_status();
        }
    if (d_tile)
        {
        HMD_REQUIRE(fixed_images, HMD_ERR_UNSUPPORTED, "tile rows need a 27-cell stencil with at least 3 cells along every periodic direction");
        hipLaunchKernelGGL((nlist_binned_kernel<ST, FB, 2, true>), grid, dim3(NL_BLOCK), lds, s, a, ncells, wpc, d_tile, tile_capacity, d_tile_overflow, d_slot_meta, d_slot_of, d_cpos, d_block_group, row_pieces, ex);
        }
    else if (fixed_images)
        hipLaunchKernelGGL((nlist_binned_kernel<ST, FB, 2, false>), grid, dim3(NL_BLOCK), lds, s, a, ncells, wpc, nullptr, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, ex);
    else if (hmd_box_is_ortho(a.box))
        hipLaunchKernelGGL((nlist_binned_kernel<ST, FB, 1, false>), grid, dim3(NL_BLOCK), lds, s, a, ncells, wpc, nullptr, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, ex);
    else
        hipLaunchKernelGGL((nlist_binned_kernel<ST, FB, 0, false>), grid, dim3(NL_BLOCK), lds, s, a, ncells, wpc, nullptr, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, ex);
    return hmd_launch_status();
    }

int nlist_binned_driver(const hmd_nlist_binned_args_t& a, hipStream_t s, unsigned short* d_tile, unsigned int tile_capacity, unsigned int* d_tile_overflow,
                        hmd_uint4* d_slot_meta, unsigned int* d_slot_of = nullptr, hmd_scalar4* d_cpos = nullptr, const unsigned int* d_block_group = nullptr,
                        unsigned int row_pieces = 0, nl_excl_t ex = nl_excl_t {nullptr, nullptr, hmd_index2d_t {0, 0}})
    {
    const bool single = (a.ntypes == 1);
    const bool fb = a.filter_body != 0;
    const unsigned int ncells = a.ci.w * a.ci.h * a.ci.d;
    // waves per cell: one fills the chip once there are several cells per SIMD slot
    unsigned int wpc = 1;
    if (ncells < 8192)
        wpc = 2;
    if (ncells < 2048)
        wpc = 4;
    const size_t lds = (sizeof(unsigned int) * 256 + sizeof(double) * 3 * 64) * NL_WAVES + (single ? 0 : sizeof(double) * a.ntypes * a.ntypes);
    if (single && !fb)
        return launch_nlist<true, false>(a, ncells, wpc, lds, s, d_tile, tile_capacity, d_tile_overflow, d_slot_meta, d_slot_of, d_cpos, d_block_group, row_pieces, ex);
    if (single && fb)
        return launch_nlist<true, true>(a, ncells, wpc, lds, s, d_tile, tile_capacity, d_tile_overflow, d_slot_meta, d_slot_of, d_cpos, d_block_group, row_pieces, ex);
    if (!single && !fb)
        return launch_nlist<false, false>(a, ncells, wpc, lds, s, d_tile, tile_capacity, d_tile_overflow, d_slot_meta, d_slot_of, d_cpos, d_block_group, row_pieces, ex);
    return launch_nlist<false, true>(a, ncells, wpc, lds, s, d_tile, tile_capacity, d_tile_overflow, d_slot_meta, d_slot_of, d_cpos, d_block_group, row_pieces, ex);
    }
    } // namespace

extern "C" int hmd_compute_nlist_binned(const hmd_nlist_binned_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_compute_nlist_binned: args is NULL");
    const hmd_nlist_binned_args_t& a = *args;
    HMD_REQUIRE(a.d_nlist && a.d_n_neigh && a.d_conditions && a.d_Nmax && a.d_head_list && a.d_pos, HMD_ERR_INVALID_ARGUMENT,
                "hmd_compute_nlist_binned: nlist, n_neigh, conditions, Nmax, head_list and pos are required");
    HMD_REQUIRE(a.d_cell_size && a.d_cell_xyzf && a.d_cell_adj && a.d_r_cut, HMD_ERR_INVALID_ARGUMENT,
                "hmd_compute_nlist_binned: cell_size, cell_xyzf (w = particle index), cell_adj and r_cut are required");
    HMD_REQUIRE(a.ntypes >= 1, HMD_ERR_INVALID_ARGUMENT, "hmd_compute_nlist_binned: ntypes must be >= 1");
    const bool single = (a.ntypes == 1);
    const bool fb = a.filter_body != 0;
    HMD_REQUIRE((single && !fb) || a.d_cell_type_body, HMD_ERR_INVALID_ARGUMENT,
                "hmd_compute_nlist_binned: d_cell_type_body is required with more than one type or a body filter");
    const unsigned int ncells = a.ci.w * a.ci.h * a.ci.d;
    HMD_REQUIRE(ncells > 0 && a.cli.h == ncells && a.cadji.h == ncells, HMD_ERR_INVALID_ARGUMENT,
                "hmd_compute_nlist_binned: indexers disagree on the number of cells");
    HMD_REQUIRE(a.cadji.w >= 1 && a.cadji.w <= 64, HMD_ERR_UNSUPPORTED,
                "hmd_compute_nlist_binned: at most 64 adjacent cells per cell (radius-1 stencils have 27)");
    HMD_REQUIRE((size_t)a.ntypes * a.ntypes * 8 <= 48 * 1024, HMD_ERR_TOO_LARGE,
                "hmd_compute_nlist_binned: r_list table exceeds 48 KiB of LDS (ntypes > 78)");
    if (a.N == 0)
        return HMD_OK;

    HMD_REQUIRE(!a.d_n_ex_idx || (a.d_ex_list_idx && a.exli.w >= a.N), HMD_ERR_INVALID_ARGUMENT,
                "hmd_compute_nlist_binned: exclusions need d_ex_list_idx and an indexer at least N wide");
    const nl_excl_t ex = {a.d_n_ex_idx, a.d_ex_list_idx, a.exli};
    return nlist_binned_driver(a, (hipStream_t)stream, nullptr, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, ex);
    }

// Tile rows from the same kernel (declared with the tile path in the header).
extern "C" int hmd_compute_nlist_tiles(const hmd_nlist_tiles_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_compute_nlist_tiles: args is NULL");
    const hmd_nlist_tiles_args_t& t = *args;
    HMD_REQUIRE(t.geom.d_cell_size && t.geom.d_cell_idx && t.d_cell_xyzf && t.d_cell_adj && t.d_tile_nlist && t.d_n_neigh && t.d_conditions
                    && t.d_Nmax && t.d_head_list && t.d_pos && t.d_r_cut && t.d_tile_overflow,
                HMD_ERR_INVALID_ARGUMENT, "hmd_compute_nlist_tiles: a required array is NULL");
    HMD_REQUIRE(t.ntypes == 1 || t.d_cell_type_body, HMD_ERR_INVALID_ARGUMENT, "hmd_compute_nlist_tiles: d_cell_type_body is required with more than one type");
    unsigned int cap = 0;
    const int rc = hmd_nlist_tiles_supported(&t.geom.ci, &t.box, t.geom.cli.w, t.ntypes, &cap);
    if (rc != HMD_OK)
        return rc;
    HMD_REQUIRE(t.geom.capacity >= 1 && t.geom.capacity <= cap, HMD_ERR_INVALID_ARGUMENT,
                "hmd_compute_nlist_tiles: geom.capacity must be in [1, 64 * cli.w]");
    const unsigned int ncells = t.geom.ci.w * t.geom.ci.h * t.geom.ci.d;
    HMD_REQUIRE(ncells > 0 && t.geom.cli.h == ncells && t.cadji.h == ncells && t.cadji.w == 27, HMD_ERR_INVALID_ARGUMENT,
                "hmd_compute_nlist_tiles: indexers disagree on the number of cells, or the stencil is not 27 cells");
    if (t.N == 0)
        return HMD_OK;
    hmd_nlist_binned_args_t a;
    a = hmd_nlist_binned_args_t {};
    a.d_n_neigh = t.d_n_neigh;
    a.d_last_updated_pos = t.d_last_updated_pos;
    a.d_conditions = t.d_conditions;
    a.d_Nmax = t.d_Nmax;
    a.d_head_list = t.d_head_list;
    a.d_pos = t.d_pos;
    a.N = t.N;
    a.d_cell_size = t.geom.d_cell_size;
    a.d_cell_xyzf = t.d_cell_xyzf;
    a.d_cell_type_body = t.d_cell_type_body;
    a.d_cell_adj = t.d_cell_adj;
    a.ci = t.geom.ci;
    a.cli = t.geom.cli;
    a.cadji = t.cadji;
    a.box = t.box;
    a.d_r_cut = t.d_r_cut;
    a.r_buff = t.r_buff;
    a.ntypes = t.ntypes;
    HMD_REQUIRE(!t.d_slot_of || (t.d_slot_meta && t.geom.d_cpos), HMD_ERR_INVALID_ARGUMENT, "hmd_compute_nlist_tiles: d_slot_of needs d_slot_meta and geom.d_cpos");
    HMD_REQUIRE(t.geom.d_block_group && t.geom.row_pieces >= 1, HMD_ERR_INVALID_ARGUMENT,
                "hmd_compute_nlist_tiles: geom.d_block_group and geom.row_pieces are required");
    const size_t groups_fit = t.tile_nlist_entries / hmd::tile_group_entries(t.geom.row_pieces);
    HMD_REQUIRE(groups_fit > hmd::TL_SLACK_GROUPS && groups_fit - hmd::TL_SLACK_GROUPS <= 0xffffffffull, HMD_ERR_INVALID_ARGUMENT,
                "hmd_compute_nlist_tiles: tile_nlist_entries holds no group of rows (size it with hmd_nlist_tiles_storage_entries)");
    const unsigned int tile_groups = (unsigned int)(groups_fit - hmd::TL_SLACK_GROUPS);
    // the groups are numbered first (that kernel also clears d_tile_overflow), and the build gives up when they do not fit
    const size_t bg_lds = (size_t)hmd::tile_num_blocks(t.geom.ci) * sizeof(unsigned int);
    // (HOOMD_AMD_BG_LDS_LIMIT: the size in bytes up to which the counts are kept in LDS, for tests of the other path)
    const char* bg_env = std::getenv("HOOMD_AMD_BG_LDS_LIMIT");
    const size_t bg_limit = bg_env ? (size_t)std::atol(bg_env) : (size_t)60 * 1024;
    const unsigned int bg_cached = bg_lds <= std::min<size_t>(bg_limit, 60 * 1024) ? 1u : 2u;
    if (bg_cached == 2u)
        hipLaunchKernelGGL(tile_block_counts_kernel, dim3(hmd_div_up(hmd::tile_num_blocks(t.geom.ci), 256)), dim3(256), 0, (hipStream_t)stream, t.geom.d_cell_size,
                           t.geom.ci, t.geom.cli.w, const_cast<unsigned int*>(t.geom.d_block_group));
    hipLaunchKernelGGL(tile_block_groups_kernel, dim3(1), dim3(BG_THREADS), bg_cached == 1u ? bg_lds : 0, (hipStream_t)stream, t.geom.d_cell_size, t.geom.ci,
                       t.geom.cli.w, const_cast<unsigned int*>(t.geom.d_block_group), tile_groups, t.d_tile_overflow, bg_cached);
    HMD_REQUIRE(!t.d_n_ex_idx || (t.d_ex_list_idx && t.exli.w >= t.N), HMD_ERR_INVALID_ARGUMENT,
                "hmd_compute_nlist_tiles: exclusions need d_ex_list_idx and an indexer at least N wide");
    const nl_excl_t ex = {t.d_n_ex_idx, t.d_ex_list_idx, t.exli};
    return nlist_binned_driver(a, (hipStream_t)stream, t.d_tile_nlist, t.geom.capacity, t.d_tile_overflow, t.d_slot_meta, t.d_slot_of,
                               t.d_slot_of ? const_cast<hmd_scalar4*>(t.geom.d_cpos) : nullptr, t.geom.d_block_group, t.geom.row_pieces, ex);
    }

// The build with the pre-test is the one that lays rows out by type (launch_nlist: two types, no body filter -- tile rows
// never have one --, orthorhombic, the 27-cell stencil with at least three cells per periodic direction that
// hmd_compute_nlist_tiles requires anyway).
extern "C" int hmd_nlist_tiles_rows_by_type(const hmd_nlist_tiles_args_t* args)
    {
    if (!args)
        return 0;
    return (args->ntypes == 2 && hmd_box_is_ortho(args->box) && !std::getenv("HOOMD_AMD_NLIST_EXACT_ONLY")) ? 1 : 0;
    }

extern "C" size_t hmd_nlist_tiles_storage_entries(const hmd_index3d_t* ci, unsigned int n_members, unsigned int row_pieces)
    {
    if (!ci)
        return 0;
    // every block rounds its members up to whole groups
    const size_t groups = (size_t)n_members / hmd::TL_GROUP + hmd::tile_num_blocks(*ci) + 1 + hmd::TL_SLACK_GROUPS;
    return groups * hmd::tile_group_entries(row_pieces);
    }

extern "C" int hmd_nlist_needs_update_check(const hmd_nlist_check_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_nlist_needs_update_check: args is NULL");
    HMD_REQUIRE(args->d_result && args->d_last_pos && args->d_pos && args->d_rcut_max, HMD_ERR_INVALID_ARGUMENT,
                "hmd_nlist_needs_update_check: result, last_pos, pos and rcut_max are required");
    if (args->N != 0)
        {
        const dim3 grid(hmd_div_up(args->N, 256));
        if (hmd_box_is_ortho(args->box))
            hipLaunchKernelGGL((nlist_check_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, *args);
        else
            hipLaunchKernelGGL((nlist_check_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, *args);
        }
    if (args->d_published)
        hipLaunchKernelGGL(nlist_publish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, args->d_result, args->d_published, args->stamp);
    return hmd_launch_status();
    }

__global__ void publish_words_kernel(const unsigned int* d_words, unsigned int n, unsigned long long* d_published, unsigned int stamp)
    {
    if (threadIdx.x < n)
        __hip_atomic_store(d_published + threadIdx.x, ((unsigned long long)stamp << 32) | (unsigned long long)d_words[threadIdx.x], __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }

extern "C" int hmd_publish_words(const unsigned int* d_words, unsigned int n, unsigned long long* d_published, unsigned int stamp, void* stream)
    {
    HMD_REQUIRE(d_words && d_published, HMD_ERR_INVALID_ARGUMENT, "hmd_publish_words: d_words and d_published are required");
    HMD_REQUIRE(n >= 1 && n <= 64, HMD_ERR_INVALID_ARGUMENT, "hmd_publish_words: n must be in [1, 64]");
    hipLaunchKernelGGL(publish_words_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_words, n, d_published, stamp);
    return hmd_launch_status();
    }

extern "C" int hmd_publish_word(const unsigned int* d_word, unsigned long long* d_published, unsigned int stamp, void* stream)
    {
    HMD_REQUIRE(d_word && d_published, HMD_ERR_INVALID_ARGUMENT, "hmd_publish_word: d_word and d_published are required");
    hipLaunchKernelGGL(nlist_publish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_word, d_published, stamp);
    return hmd_launch_status();
    }

extern "C" size_t hmd_nlist_head_list_workspace_bytes(unsigned int N)
    {
    return sizeof(unsigned long long) * (size_t)(hmd_div_up(N, HL_TILE) + 1);
    }

extern "C" int hmd_nlist_build_head_list(const hmd_head_list_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_nlist_build_head_list: args is NULL");
    HMD_REQUIRE(args->d_head_list && args->d_req_size_nlist && args->d_Nmax && args->d_pos, HMD_ERR_INVALID_ARGUMENT,
                "hmd_nlist_build_head_list: head_list, req_size_nlist, Nmax and pos are required");
    hipStream_t s = (hipStream_t)stream;
    if (args->N == 0)
        {
        hipError_t e = hipMemsetAsync(args->d_req_size_nlist, 0, sizeof(size_t), s);
        return (int)e;
        }
    if (args->ntypes == 1)
        {
        hipLaunchKernelGGL(head_uniform_kernel, dim3(hmd_div_up(args->N, HL_BLOCK)), dim3(HL_BLOCK), 0, s, args->d_head_list,
                           args->d_req_size_nlist, args->d_Nmax, args->N);
        return hmd_launch_status();
        }
    const unsigned int ntiles = hmd_div_up(args->N, HL_TILE);
    HMD_REQUIRE(args->d_workspace && args->workspace_bytes >= hmd_nlist_head_list_workspace_bytes(args->N), HMD_ERR_INVALID_ARGUMENT,
                "hmd_nlist_build_head_list: workspace smaller than hmd_nlist_head_list_workspace_bytes(N)");
    unsigned long long* tiles = reinterpret_cast<unsigned long long*>(args->d_workspace);
    hipLaunchKernelGGL(head_tile_sums_kernel, dim3(ntiles), dim3(HL_BLOCK), 0, s, tiles, args->d_Nmax, args->d_pos, args->N);
    hipLaunchKernelGGL(head_scan_tiles_kernel, dim3(1), dim3(HL_BLOCK), 0, s, tiles, ntiles, args->d_req_size_nlist);
    hipLaunchKernelGGL(head_write_kernel, dim3(ntiles), dim3(HL_BLOCK), 0, s, args->d_head_list, tiles, args->d_Nmax, args->d_pos, args->N);
    return hmd_launch_status();
    }

// ---- the head list for callers that may not allocate ----
// The tile sums of the scan live in a file-scope device array instead of a caller-provided workspace.
// `scratch` names the array in device code only; host code uses the address hipGetSymbolAddress returned
// for the current device (every GPU holds its own copy at its own address).
constexpr unsigned int HL_NOALLOC_MAX_N = 1u << 24;
constexpr unsigned int HL_NOALLOC_TILES = HL_NOALLOC_MAX_N / HL_TILE;
static __device__ __align__(256) unsigned long long scratch[HL_NOALLOC_TILES + 1];
constexpr int HL_NOALLOC_MAX_DEVICES = 64;
static unsigned long long* g_hl_scratch[HL_NOALLOC_MAX_DEVICES];
static std::mutex g_hl_scratch_mutex;

extern "C" int hmd_nlist_build_head_list_noalloc(const hmd_head_list_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_nlist_build_head_list_noalloc: args is NULL");
    HMD_REQUIRE(args->d_head_list && args->d_req_size_nlist && args->d_Nmax && args->d_pos, HMD_ERR_INVALID_ARGUMENT,
                "hmd_nlist_build_head_list_noalloc: head_list, req_size_nlist, Nmax and pos are required");
    HMD_REQUIRE(args->N <= HL_NOALLOC_MAX_N, HMD_ERR_TOO_LARGE,
                "hmd_nlist_build_head_list_noalloc: N exceeds 16,777,216 particles, the size the static scratch array is built for "
                "(use hmd_nlist_build_head_list with a workspace)");
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return (int)e;
    HMD_REQUIRE(dev >= 0 && dev < HL_NOALLOC_MAX_DEVICES, HMD_ERR_TOO_LARGE, "hmd_nlist_build_head_list_noalloc: device ordinal above 63");
    if (stream)
        {
        // every GPU has its own copy of the array: the one used must be the copy of the GPU the stream runs on
        int sdev = -1;
        e = hipStreamGetDevice((hipStream_t)stream, &sdev);
        if (e != hipSuccess)
            return (int)e;
        HMD_REQUIRE(sdev == dev, HMD_ERR_INVALID_ARGUMENT,
                    "hmd_nlist_build_head_list_noalloc: the stream belongs to another GPU than the current one (make the stream's device "
                    "current before the call)");
        }
    unsigned long long* tiles = nullptr;
        {
        std::lock_guard<std::mutex> lock(g_hl_scratch_mutex);
        if (!g_hl_scratch[dev])
            {
            void* p = nullptr;
            e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !p)
                {
                hmd_set_error("hmd_nlist_build_head_list_noalloc: hipGetSymbolAddress failed for the scratch array");
                return e != hipSuccess ? (int)e : HMD_ERR_UNSUPPORTED;
                }
            g_hl_scratch[dev] = reinterpret_cast<unsigned long long*>(p);
            }
        tiles = g_hl_scratch[dev];
        }
    hipStream_t s = (hipStream_t)stream;
    if (args->N == 0)
        return (int)hipMemsetAsync(args->d_req_size_nlist, 0, sizeof(size_t), s);
    // always the scan (also with one type, where the other entry point has a closed form): every tile sum
    // the later kernels read is written by the first one, so the array's old contents never matter
    const unsigned int ntiles = hmd_div_up(args->N, HL_TILE);
    hipLaunchKernelGGL(head_tile_sums_kernel, dim3(ntiles), dim3(HL_BLOCK), 0, s, tiles, args->d_Nmax, args->d_pos, args->N);
    hipLaunchKernelGGL(head_scan_tiles_kernel, dim3(1), dim3(HL_BLOCK), 0, s, tiles, ntiles, args->d_req_size_nlist);
    hipLaunchKernelGGL(head_write_kernel, dim3(ntiles), dim3(HL_BLOCK), 0, s, args->d_head_list, tiles, args->d_Nmax, args->d_pos, args->N);
    return hmd_launch_status();
    }

extern "C" int hmd_nlist_filter(const hmd_nlist_filter_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_nlist_filter: args is NULL");
    HMD_REQUIRE(args->d_n_neigh && args->d_nlist && args->d_head_list && args->d_n_ex && args->d_ex_list, HMD_ERR_INVALID_ARGUMENT,
                "hmd_nlist_filter: all array arguments are required");
    if (args->N == 0)
        return HMD_OK;
    hipLaunchKernelGGL(nlist_filter_kernel, dim3(hmd_div_up(args->N, 256)), dim3(256), 0, (hipStream_t)stream, *args);
    return hmd_launch_status();
    }

extern "C" int hmd_update_exclusion_list(const hmd_update_exclusion_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_update_exclusion_list: args is NULL");
    HMD_REQUIRE(args->d_tag && args->d_rtag && args->d_n_ex_tag && args->d_ex_list_tag && args->d_n_ex_idx && args->d_ex_list_idx,
                HMD_ERR_INVALID_ARGUMENT, "hmd_update_exclusion_list: all array arguments are required");
    if (args->N == 0)
        return HMD_OK;
    hipLaunchKernelGGL(update_exclusion_kernel, dim3(hmd_div_up(args->N, 256)), dim3(256), 0, (hipStream_t)stream, *args);
    return hmd_launch_status();
    }
