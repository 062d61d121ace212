// pair_force_tiles.cuh -- pair forces from LDS tiles.
//
// Same per-pair arithmetic and outputs as pair_force.cuh (hoomd/md/PotentialPairGPU.cuh:140-404); the
// neighbor rows hold 16-bit candidate numbers of the workgroup's region (tile.cuh), so a neighbor's
// coordinates are three LDS reads and the periodic image is already applied.  TPP lanes share a
// home particle; each lane takes four consecutive row entries (one 8-byte load) per trip.
// Row entries are tile slots (tile_slot() in tile.cuh): 8 * entry is the byte offset of x in the tile,
// y and z follow 256 and 512 bytes later.
#ifndef HMD_PAIR_FORCE_TILES_CUH
#define HMD_PAIR_FORCE_TILES_CUH

#include "pair_force.cuh"
#include "tile.cuh"

namespace hmd
    {
// per home particle, kept in LDS so that no wave waits on a chain of dependent global loads
struct pft_home_t
    {
    unsigned int idx;
    unsigned int row;   // offset of the particle's row in d_tile_nlist, in units of four entries
    unsigned int nn_lh; // low half: number of neighbors; high half: tile slot of the particle itself
    };
static_assert(sizeof(pft_home_t) == 12, "LDS budget");
__device__ __forceinline__ pft_home_t pft_home(unsigned int idx, unsigned int nn, unsigned int row, unsigned int lh)
    {
    pft_home_t e = {idx, row, min(nn, 0xffffu) | (lh << 16)};
    return e;
    }

// bytes of the home table: one entry per slot of the home cells and an empty one, padded to 16
__host__ __device__ constexpr size_t pft_home_bytes(unsigned int cell_Nmax)
    {
    return (((size_t)TL_NH * cell_Nmax + 1) * sizeof(pft_home_t) + 15) / 16 * 16;
    }

typedef __attribute__((address_space(3))) const double pft_lds_double;

// byte offset 8 * (16-bit half of w) in one instruction
template<int HALF> __device__ __forceinline__ unsigned int pft_slot_bytes(unsigned int w, unsigned int three)
    {
    unsigned int r;
    if (HALF == 0)
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(three), "v"(w));
    else
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(three), "v"(w));
    return r;
    }

constexpr int PFT_THREADS = 256;              // 4 waves share one staged region
constexpr size_t PFT_TABLES = (sizeof(tile_tables) + 16 + 15) / 16 * 16; // tile_tables and the shared home counter
constexpr int PFT_WAVES = PFT_THREADS / 64;
constexpr int PFT_CELLS = (TL_RC + PFT_WAVES - 1) / PFT_WAVES; // region cells staged by one wave
constexpr int PFT_BATCH = (PFT_CELLS + 1) / 2;                  // ... in two rounds of this many loads in flight

template<class Ev, unsigned int SHIFT, bool VIRIAL, int TPP, bool SINGLE_TYPE>
__global__ void __launch_bounds__(PFT_THREADS)
    pair_force_tiles_kernel(hmd_pair_args_t a, hmd_tile_geom_t g, const unsigned short* d_tile_nlist, const typename Ev::param_type* d_params)
    {
    typedef typename Ev::param_type param_t;
    // All LDS of this kernel is dynamic and the tile comes first, so that the tile starts at LDS
    // address 0 and 8 * (row entry) is the address of a neighbor's x without adding a base.
    extern __shared__ __align__(16) unsigned char s_dyn[];
    const unsigned int cap = g.capacity;
    // candidate c sits at txyz[s], txyz[s + 32], txyz[s + 64], s = tile_slot(c); candidate `total` is a dummy
    // far away that the padding of the rows points to
    double* txyz = reinterpret_cast<double*>(s_dyn);
    const unsigned int tile_len = tile_doubles(cap);
    tile_tables& T = *reinterpret_cast<tile_tables*>(txyz + tile_len);
    // s_home: one entry per home particle and an empty one after the last
    unsigned int& s_next = *reinterpret_cast<unsigned int*>(s_dyn + (size_t)tile_len * 8 + PFT_TABLES - 16);
    unsigned int& s_word = *reinterpret_cast<unsigned int*>(s_dyn + (size_t)tile_len * 8 + PFT_TABLES - 12);
    pft_home_t* s_home = reinterpret_cast<pft_home_t*>(s_dyn + (size_t)tile_len * 8 + PFT_TABLES);
    if ((unsigned int)(size_t)(__attribute__((address_space(3))) void*)txyz != 0u)
        __builtin_trap(); // the toolchain placed something before the dynamic LDS
    // One thread reads the predicate word for the whole workgroup (another stream may be raising it at this
    // very moment: the rebuild vote of other ranks arriving; a workgroup must not split over it); the others
    // learn it behind the barrier that follows the set-up below.
    if (a.d_skip_flag && threadIdx.x == 0)
        {
        const unsigned int word = __hip_atomic_load(a.d_skip_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.d_published && blockIdx.x == 0)
            __hip_atomic_store(a.d_published, ((unsigned long long)a.publish_stamp << 32) | (unsigned long long)word, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        s_word = word;
        }
    const unsigned int np = a.ntypes * a.ntypes;
    // (the 12-byte home entries end on any multiple of four: the tables of doubles behind them start 16-aligned)
    double* s_rcutsq = reinterpret_cast<double*>(s_dyn + (size_t)tile_len * 8 + PFT_TABLES + pft_home_bytes(g.cli.w));
    double* s_ronsq = s_rcutsq + (SINGLE_TYPE ? 0 : np);
    param_t* s_params = reinterpret_cast<param_t*>(s_ronsq + (SINGLE_TYPE ? 0 : np));
    unsigned char* ttype = reinterpret_cast<unsigned char*>(s_params + (SINGLE_TYPE ? 0 : np));

    // which block of the grid: all of them in launch order, or those inside / outside a window
    unsigned int vb = tile_xcd_remap(blockIdx.x, gridDim.x);
    if (a.block_window_mode)
        {
        const unsigned int nbx = (g.ci.w + TB_X - 1) / TB_X, nby = (g.ci.h + TB_Y - 1) / TB_Y;
        if (a.block_window_mode == 1)
            {
            // the launch covers the window only
            const unsigned int wx = a.block_hi[0] - a.block_lo[0] + 1, wy = a.block_hi[1] - a.block_lo[1] + 1;
            const unsigned int bx = a.block_lo[0] + vb % wx, by = a.block_lo[1] + (vb / wx) % wy, bz = a.block_lo[2] + vb / (wx * wy);
            vb = (bz * nby + by) * nbx + bx;
            }
        else
            {
            // the launch covers the blocks outside the window: six slabs, numbered one after the other
            const unsigned int nbz = (g.ci.d + TB_Z - 1) / TB_Z;
            const unsigned int wy = a.block_hi[1] - a.block_lo[1] + 1, wz = a.block_hi[2] - a.block_lo[2] + 1;
            unsigned int v = vb, bx, by, bz;
            const unsigned int nA = nbx * nby * a.block_lo[2], nB = nbx * nby * (nbz - 1 - a.block_hi[2]);
            const unsigned int nC = nbx * a.block_lo[1] * wz, nD = nbx * (nby - 1 - a.block_hi[1]) * wz;
            const unsigned int nE = a.block_lo[0] * wy * wz;
            if (v < nA)
                {
                bx = v % nbx, by = (v / nbx) % nby, bz = v / (nbx * nby);
                }
            else if ((v -= nA) < nB)
                {
                bx = v % nbx, by = (v / nbx) % nby, bz = a.block_hi[2] + 1 + v / (nbx * nby);
                }
            else if ((v -= nB) < nC)
                {
                bx = v % nbx, by = (v / nbx) % a.block_lo[1], bz = a.block_lo[2] + v / (nbx * a.block_lo[1]);
                }
            else if ((v -= nC) < nD)
                {
                const unsigned int hy = nby - 1 - a.block_hi[1];
                bx = v % nbx, by = a.block_hi[1] + 1 + (v / nbx) % hy, bz = a.block_lo[2] + v / (nbx * hy);
                }
            else if ((v -= nD) < nE)
                {
                bx = v % a.block_lo[0], by = a.block_lo[1] + (v / a.block_lo[0]) % wy, bz = a.block_lo[2] + v / (a.block_lo[0] * wy);
                }
            else
                {
                v -= nE;
                const unsigned int hx = nbx - 1 - a.block_hi[0];
                bx = a.block_hi[0] + 1 + v % hx, by = a.block_lo[1] + (v / hx) % wy, bz = a.block_lo[2] + v / (hx * wy);
                }
            vb = (bz * nby + by) * nbx + bx;
            }
        }
    int cx0, cy0, cz0;
    tile_setup(g, a.box, T, cx0, cy0, cz0, vb);
    param_t p0;
    double rcutsq0 = 0.0, ronsq0 = 0.0;
    if (SINGLE_TYPE)
        {
        p0 = d_params[0];
        rcutsq0 = a.d_rcutsq[0];
        if (SHIFT == 2)
            ronsq0 = a.d_ronsq[0];
        }
    else
        for (unsigned int t = threadIdx.x; t < np; t += PFT_THREADS)
            {
            s_params[t] = d_params[t];
            s_rcutsq[t] = a.d_rcutsq[t];
            s_ronsq[t] = (SHIFT == 2) ? a.d_ronsq[t] : 0.0;
            }
    if (threadIdx.x == 0)
        s_next = 2 * (PFT_THREADS / TPP);
    __syncthreads();
    if (a.d_skip_flag && s_word == a.skip_value)
        return;
    const unsigned int total = T.off[64];
    if (total > cap || total == 0)
        return;

    // ---- home particles: members of the (up to 8) home cells, numbered consecutively ----
    unsigned int hbeg[TL_NH], hcum[TL_NH + 1];
    hcum[0] = 0;
#pragma unroll
    for (int hc = 0; hc < TL_NH; hc++)
        {
        const int hx = hc % TB_X, hy = (hc / TB_X) % TB_Y, hz = hc / (TB_X * TB_Y);
        const unsigned int rc = tile_home_rc(hc);
        const bool home = tile_is_home(g, cx0, cy0, cz0, hx, hy, hz);
        hbeg[hc] = T.off[rc];
        hcum[hc + 1] = hcum[hc] + (home ? T.off[rc + 1] - T.off[rc] : 0u);
        }
    const unsigned int H = hcum[TL_NH];
    auto home_local = [&](unsigned int h, unsigned int& rc_home) -> unsigned int
    {
        unsigned int lh = 0;
        rc_home = 0;
#pragma unroll
        for (int c = 0; c < TL_NH; c++)
            if (h >= hcum[c] && h < hcum[c + 1])
                {
                lh = hbeg[c] + (h - hcum[c]);
                rc_home = tile_home_rc(c);
                }
        return lh;
    };
    const unsigned int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    if (g.d_cpos && g.d_slot_meta)
        {
        // ---- fast staging: positions and row info are kept in cell-list slot order, so one round of
        // loads with every lane busy fills the tile and the home table ----
        for (unsigned int h = threadIdx.x; h < H; h += PFT_THREADS)
            {
            unsigned int rc_home;
            const unsigned int lh = home_local(h, rc_home);
            const hmd_uint4 m = g.d_slot_meta[(unsigned int)T.cell[rc_home] * g.cli.w + (lh - T.off[rc_home])];
            s_home[h] = pft_home(m.x, m.y, m.z, tile_slot(lh));
            }
        // Sixteen lanes take one region cell, so a wave covers four cells per trip and the whole
        // region in PFT_TRIPS trips, all of whose loads are in flight together.
        constexpr int PFT_TRIPS = (TL_RC + 4 * PFT_WAVES - 1) / (4 * PFT_WAVES);
        const double reach_sq = tile_reach_sq(a.box);
        const unsigned int k16 = lane % 16;
        double4 p[PFT_TRIPS];
        unsigned int rcs[PFT_TRIPS], size[PFT_TRIPS];
#pragma unroll
        for (int u = 0; u < PFT_TRIPS; u++)
            {
            rcs[u] = (u * PFT_WAVES + wave) * 4 + lane / 16;
            size[u] = rcs[u] < (unsigned int)TL_RC ? T.off[rcs[u] + 1] - T.off[rcs[u]] : 0u;
            rcs[u] = min(rcs[u], (unsigned int)TL_RC - 1);
            const unsigned int slot = size[u] ? (unsigned int)T.cell[rcs[u]] * g.cli.w + min(k16, size[u] - 1) : 0u;
            p[u] = hmd_ld4(g.d_cpos + slot);
            }
#pragma unroll
        for (int u = 0; u < PFT_TRIPS; u++)
            {
            const unsigned int rc = rcs[u];
            const tile_cell_image ci = tile_cell_setup(a.box, T, rc);
            const unsigned int first = T.off[rc];
            const hmd_scalar4* cell_pos = g.d_cpos + (size[u] ? (unsigned int)T.cell[rc] * g.cli.w : 0u);
            double4 pp = p[u];
            for (unsigned int k = k16; k < size[u]; k += 16)
                {
                if (k >= 16)
                    pp = hmd_ld4(cell_pos + k);
                double x = pp.x, y = pp.y, z = pp.z;
                const double dx = x - ci.qx, dy = y - ci.qy, dz = z - ci.qz;
                if (dx * dx + dy * dy + dz * dz < reach_sq)
                    {
                    x += ci.sx;
                    y += ci.sy;
                    z += ci.sz;
                    }
                else
                    tile_place(a.box, T, rc, x, y, z);
                const unsigned int v = tile_slot(first + k);
                txyz[v] = x;
                txyz[v + 32] = y;
                txyz[v + 64] = z;
                if (!SINGLE_TYPE)
                    ttype[v] = (unsigned char)hmd_double_as_uint(pp.w);
                }
            }
        }
    else
        {
        // ---- stage the region (one wave per region cell, eight cells in flight) and the home table.
        // Loads are issued in two dependent rounds only: member indices, then positions / row info. ----
        unsigned int my_h = threadIdx.x, my_idx = 0, my_lh = 0;
        if (my_h < H)
            {
            unsigned int rc_home;
            my_lh = home_local(my_h, rc_home);
            my_idx = g.d_cell_idx[(unsigned int)T.cell[rc_home] * g.cli.w + (my_lh - T.off[rc_home])];
            }
    #pragma unroll
        for (int round = 0; round < 2; round++)
            {
            const unsigned int rc0 = wave * PFT_CELLS + round * PFT_BATCH;
            unsigned int idx[PFT_BATCH];
            bool on[PFT_BATCH];
    #pragma unroll
            for (int q = 0; q < PFT_BATCH; q++)
                {
                const unsigned int rc = rc0 + q;
                on[q] = rc < min((wave + 1) * PFT_CELLS, (unsigned int)TL_RC) && lane < T.off[rc + 1] - T.off[rc];
                idx[q] = on[q] ? g.d_cell_idx[(unsigned int)T.cell[rc] * g.cli.w + lane] : 0u;
                }
            double4 p[PFT_BATCH];
    #pragma unroll
            for (int q = 0; q < PFT_BATCH; q++)
                p[q] = hmd_ld4(a.d_pos + idx[q]);
            if (round == 0 && my_h < H)
                {
                unsigned int nn = 0, row = 0;
                if (my_idx < a.N)
                    {
                    nn = a.d_n_neigh[my_idx];
                    row = (unsigned int)(a.d_head_list[my_idx] / 4);
                    }
                s_home[my_h] = pft_home(my_idx, nn, row, tile_slot(my_lh));
                }
    #pragma unroll
            for (int q = 0; q < PFT_BATCH; q++)
                {
                const unsigned int rc = rc0 + q;
                if (on[q])
                    {
                    const unsigned int v = tile_slot(T.off[rc] + lane);
                    double x = p[q].x, y = p[q].y, z = p[q].z;
                    tile_place(a.box, T, rc, x, y, z);
                    txyz[v] = x;
                    txyz[v + 32] = y;
                    txyz[v + 64] = z;
                    if (!SINGLE_TYPE)
                        ttype[v] = (unsigned char)hmd_double_as_uint(p[q].w);
                    }
                }
            }
        // the uncommon leftovers: cells with more than 64 members, blocks with more than 256 home particles
        for (unsigned int rc = wave * PFT_CELLS; rc < min((wave + 1) * PFT_CELLS, (unsigned int)TL_RC); rc++)
            {
            const unsigned int size = T.off[rc + 1] - T.off[rc];
            for (unsigned int k = lane + 64; k < size; k += 64)
                {
                const unsigned int j = g.d_cell_idx[(unsigned int)T.cell[rc] * g.cli.w + k];
                const double4 pp = hmd_ld4(a.d_pos + j);
                const unsigned int v = tile_slot(T.off[rc] + k);
                double x = pp.x, y = pp.y, z = pp.z;
                tile_place(a.box, T, rc, x, y, z);
                txyz[v] = x;
                txyz[v + 32] = y;
                txyz[v + 64] = z;
                if (!SINGLE_TYPE)
                    ttype[v] = (unsigned char)hmd_double_as_uint(pp.w);
                }
            }
        for (unsigned int h = threadIdx.x + PFT_THREADS; h < H; h += PFT_THREADS)
            {
            unsigned int rc_home;
            const unsigned int lh = home_local(h, rc_home);
            const unsigned int idx = g.d_cell_idx[(unsigned int)T.cell[rc_home] * g.cli.w + (lh - T.off[rc_home])];
            unsigned int nn = 0, row = 0;
            if (idx < a.N)
                {
                nn = a.d_n_neigh[idx];
                row = (unsigned int)(a.d_head_list[idx] / 4);
                }
            s_home[h] = pft_home(idx, nn, row, tile_slot(lh));
            }
        }
    if (threadIdx.x == 0)
        {
        s_home[H] = pft_home_t {0u, 0u, 0u};
        const unsigned int v = tile_slot(total);
        txyz[v] = 1e30;
        txyz[v + 32] = 1e30;
        txyz[v + 64] = 1e30;
        if (!SINGLE_TYPE)
            ttype[v] = 0;
        }
    __syncthreads();

    // ---- forces: groups of TPP lanes take home particles from a shared counter.  A row is fetched
    // whole (PFT_CH chunks of four entries per lane) two home particles ahead of its use: rows stream
    // from HBM exactly once, so what bounds this phase is the number of bytes in flight. ----
    constexpr int PFT_CH = 4;
#ifdef HMD_PFT_STAGE_ONLY
    if (total != 0xdeadbeefu)
        return;
#endif
    const unsigned int sub = threadIdx.x % TPP;
    const unsigned int leader = (lane / TPP) * TPP;
    const unsigned char* rows = reinterpret_cast<const unsigned char*>(d_tile_nlist);
    // Rows are padded to a multiple of four entries with the dummy's slot.  Chunks past the end of a
    // row read the following rows or the 512-byte slack after the last one; never used.
    const unsigned int three = 3u;
    auto fetch = [&](const pft_home_t& e, uint2 (&c)[PFT_CH])
    {
        const uint2* row = reinterpret_cast<const uint2*>(rows) + e.row;
#pragma unroll
        for (int q = 0; q < PFT_CH; q++)
            c[q] = row[sub + TPP * q];
    };
    auto grab = [&](unsigned int& h, pft_home_t& e, uint2 (&c)[PFT_CH])
    {
        unsigned int nh = 0;
        if (sub == 0)
            nh = atomicAdd(&s_next, 1u);
        h = __shfl(nh, leader, 64);
        e = s_home[min(h, H)]; // s_home[H] is an empty entry
        fetch(e, c);
    };
    auto process = [&](const pft_home_t& e, const uint2 (&cur)[PFT_CH])
    {
        const unsigned int lh = e.nn_lh >> 16;
        pf_accum acc;
        const unsigned int nn = e.nn_lh & 0xffffu;
        const double xi = txyz[lh], yi = txyz[lh + 32], zi = txyz[lh + 64];
        const unsigned int ti = SINGLE_TYPE ? 0u : (unsigned int)ttype[lh];
        auto chunk = [&](uint2 w)
        {
#ifdef HMD_PFT_SKELETON
            acc.fx += (double)(w.x ^ w.y);
            return;
#endif
            const unsigned int j[4] = {pft_slot_bytes<0>(w.x, three), pft_slot_bytes<1>(w.x, three), pft_slot_bytes<0>(w.y, three), pft_slot_bytes<1>(w.y, three)};
            double xj[4], yj[4], zj[4];
#pragma unroll
            for (int m = 0; m < 4; m++)
                {
                // three 8-byte reads: paired 16-byte LDS reads run at half the rate (volatile keeps them apart)
                pft_lds_double* c = (pft_lds_double*)(size_t)j[m];
                xj[m] = c[0];
                yj[m] = *(volatile pft_lds_double*)(c + 32);
                zj[m] = *(volatile pft_lds_double*)(c + 64);
                }
#pragma unroll
            for (int m = 0; m < 4; m++)
                {
                double rcutsq, ronsq = 0.0;
                const param_t* prm;
                if (SINGLE_TYPE)
                    {
                    rcutsq = rcutsq0;
                    ronsq = ronsq0;
                    prm = &p0;
                    }
                else
                    {
                    const unsigned int tp = (unsigned int)ttype[j[m] >> 3] * a.ntypes + ti;
                    rcutsq = s_rcutsq[tp];
                    if (SHIFT == 2)
                        ronsq = s_ronsq[tp];
                    prm = s_params + tp;
                    }
                pf_pair<Ev, SHIFT, VIRIAL>(acc, xi - xj[m], yi - yj[m], zi - zj[m], rcutsq, ronsq, *prm, 0.0, 0.0);
                }
        };
#pragma unroll
        for (int q = 0; q < PFT_CH; q++)
            if (4 * (sub + TPP * q) < nn)
                chunk(cur[q]);
        // rows longer than PFT_CH * 4 * TPP entries
        for (unsigned int k = 4 * (sub + TPP * PFT_CH); k < nn; k += 4 * TPP)
            chunk((reinterpret_cast<const uint2*>(rows) + e.row)[k / 4]);
        pf_finish<VIRIAL, TPP>(acc, a, e.idx, e.idx < a.N && sub == 0);
    };
    // Three row buffers in rotation: while one home particle is computed, the rows of the next two
    // are on their way.  The counter hands out increasing numbers, so a group is done at the first
    // number past the end.
    constexpr unsigned int groups = PFT_THREADS / TPP;
    unsigned int h0 = threadIdx.x / TPP, h1 = h0 + groups, h2;
    pft_home_t e0 = s_home[min(h0, H)], e1 = s_home[min(h1, H)], e2;
    uint2 b0[PFT_CH], b1[PFT_CH], b2[PFT_CH];
    fetch(e0, b0);
    fetch(e1, b1);
    for (;;)
        {
        if (h0 >= H)
            break;
        grab(h2, e2, b2);
        process(e0, b0);
        if (h1 >= H)
            break;
        grab(h0, e0, b0);
        process(e1, b1);
        if (h2 >= H)
            break;
        grab(h1, e1, b1);
        process(e2, b2);
        }
    }

template<class Ev> static inline size_t pft_lds_bytes(unsigned int cap, unsigned int ntypes, unsigned int cell_Nmax)
    {
    size_t b = (size_t)tile_doubles(cap) * 8 + PFT_TABLES + pft_home_bytes(cell_Nmax);
    if (ntypes > 1)
        b += (size_t)ntypes * ntypes * (16 + sizeof(typename Ev::param_type)) + tile_doubles(cap);
    return (b + 15) / 16 * 16;
    }

template<class Ev, unsigned int SHIFT, bool VIRIAL>
int pft_launch2(const hmd_pair_args_t& a, const hmd_tile_geom_t& g, const unsigned short* d_tile_nlist,
                const typename Ev::param_type* d_params, unsigned int tpp, hipStream_t s)
    {
    const size_t lds = pft_lds_bytes<Ev>(g.capacity, a.ntypes, g.cli.w);
    unsigned int nblocks = tile_num_blocks(g.ci);
    if (a.block_window_mode)
        {
        const unsigned int inside = (a.block_hi[0] - a.block_lo[0] + 1) * (a.block_hi[1] - a.block_lo[1] + 1) * (a.block_hi[2] - a.block_lo[2] + 1);
        nblocks = a.block_window_mode == 1 ? inside : nblocks - inside;
        if (nblocks == 0)
            return HMD_OK; // the window is the whole grid: nothing outside it
        }
    const dim3 grid(nblocks), block(PFT_THREADS);
#define HMD_PFT_LAUNCH(TPPV)                                                                                               \
    if (a.ntypes == 1)                                                                                                     \
        {                                                                                                                  \
        tile_allow_big_lds(pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, TPPV, true>, lds);                                   \
        hipLaunchKernelGGL((pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, TPPV, true>), grid, block, lds, s, a, g, d_tile_nlist, d_params);  \
        }                                                                                                                  \
    else                                                                                                                   \
        {                                                                                                                  \
        tile_allow_big_lds(pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, TPPV, false>, lds);                                  \
        hipLaunchKernelGGL((pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, TPPV, false>), grid, block, lds, s, a, g, d_tile_nlist, d_params); \
        }
    switch (tpp)
        {
    case 4:
        HMD_PFT_LAUNCH(4) break;
    case 16:
        HMD_PFT_LAUNCH(16) break;
    default:
        HMD_PFT_LAUNCH(8) break;
        }
#undef HMD_PFT_LAUNCH
    return hmd_launch_status();
    }

template<class Ev>
int launch_pair_forces_tiles(const hmd_pair_args_t* args, const hmd_tile_geom_t* geom, const unsigned short* d_tile_nlist,
                             const typename Ev::param_type* d_params, void* stream)
    {
    HMD_REQUIRE(args && geom && d_tile_nlist && d_params, HMD_ERR_INVALID_ARGUMENT,
                "pair forces (tiles): args, geom, d_tile_nlist and d_params are required");
    const hmd_pair_args_t& a = *args;
    const hmd_tile_geom_t& g = *geom;
    HMD_REQUIRE(a.d_force && a.d_pos && a.d_n_neigh && a.d_head_list && a.d_rcutsq, HMD_ERR_INVALID_ARGUMENT,
                "pair forces (tiles): force, pos, n_neigh, head_list and rcutsq are required");
    HMD_REQUIRE(g.d_cell_size && g.d_cell_idx, HMD_ERR_INVALID_ARGUMENT, "pair forces (tiles): the cell arrays are required");
    HMD_REQUIRE((reinterpret_cast<uintptr_t>(d_tile_nlist) & 7) == 0, HMD_ERR_INVALID_ARGUMENT,
                "pair forces (tiles): d_tile_nlist must be 8-byte aligned");
    HMD_REQUIRE(a.size_neigh_list / 4 <= 0xffffffffull, HMD_ERR_TOO_LARGE, "pair forces (tiles): more than 2^34 row entries");
    HMD_REQUIRE(a.shift_mode <= 2, HMD_ERR_INVALID_ARGUMENT, "pair forces (tiles): shift_mode must be 0, 1 or 2");
    HMD_REQUIRE(a.shift_mode != 2 || a.d_ronsq, HMD_ERR_INVALID_ARGUMENT, "pair forces (tiles): xplor needs d_ronsq");
    HMD_REQUIRE(!a.compute_virial || a.d_virial, HMD_ERR_INVALID_ARGUMENT, "pair forces (tiles): compute_virial needs d_virial");
    HMD_REQUIRE(!Ev::needsCharge(), HMD_ERR_UNSUPPORTED, "pair forces (tiles): evaluators that need charges use the row-based entry point");
    unsigned int cap = 0;
    const int rc = hmd_nlist_tiles_supported(&g.ci, &a.box, g.cli.w, a.ntypes, &cap);
    if (rc != HMD_OK)
        return rc;
    HMD_REQUIRE(g.capacity >= 1 && g.capacity <= cap, HMD_ERR_INVALID_ARGUMENT, "pair forces (tiles): geom.capacity must be in [1, 64 * cli.w]");
    HMD_REQUIRE(pft_lds_bytes<Ev>(g.capacity, a.ntypes, g.cli.w) <= TL_MAX_LDS, HMD_ERR_TOO_LARGE, "pair forces (tiles): tile exceeds the LDS budget");
    HMD_REQUIRE(a.block_window_mode <= 2, HMD_ERR_INVALID_ARGUMENT, "pair forces (tiles): block_window_mode must be 0, 1 or 2");
    if (a.block_window_mode)
        {
        const unsigned int nb[3] = {(g.ci.w + TB_X - 1) / TB_X, (g.ci.h + TB_Y - 1) / TB_Y, (g.ci.d + TB_Z - 1) / TB_Z};
        for (int d = 0; d < 3; d++)
            HMD_REQUIRE(a.block_lo[d] <= a.block_hi[d] && a.block_hi[d] < nb[d], HMD_ERR_INVALID_ARGUMENT,
                        "pair forces (tiles): the block window must lie inside the grid of blocks and hold at least one block");
        }
    const unsigned int tpp = a.threads_per_particle ? a.threads_per_particle : 8;
    HMD_REQUIRE(tpp == 4 || tpp == 8 || tpp == 16, HMD_ERR_INVALID_ARGUMENT, "pair forces (tiles): threads_per_particle must be 4, 8 or 16");
    if (a.N == 0)
        return HMD_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool vir = a.compute_virial != 0;
    switch (a.shift_mode)
        {
    case 0:
        return vir ? pft_launch2<Ev, 0, true>(a, g, d_tile_nlist, d_params, tpp, s) : pft_launch2<Ev, 0, false>(a, g, d_tile_nlist, d_params, tpp, s);
    case 1:
        return vir ? pft_launch2<Ev, 1, true>(a, g, d_tile_nlist, d_params, tpp, s) : pft_launch2<Ev, 1, false>(a, g, d_tile_nlist, d_params, tpp, s);
    default:
        return vir ? pft_launch2<Ev, 2, true>(a, g, d_tile_nlist, d_params, tpp, s) : pft_launch2<Ev, 2, false>(a, g, d_tile_nlist, d_params, tpp, s);
        }
    }
    } // namespace hmd

#endif
