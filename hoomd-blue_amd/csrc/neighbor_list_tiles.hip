// neighbor_list_tiles.hip -- limits of the tile path and expansion of tile rows (16-bit region-local
// candidate numbers, see tile.cuh) into the particle-index rows of the reference layout
// (hoomd/md/NeighborList.h:662-739).  The tile rows themselves are written by the binned build in
// neighbor_list.hip.
#pragma clang fp contract(off)
#include "tile.cuh"
#include <algorithm>

namespace
    {
using namespace hmd;

__global__ void __launch_bounds__(TL_THREADS) tiles_to_rows_kernel(hmd_tiles_to_rows_args_t a)
    {
    __shared__ tile_tables T;
    extern __shared__ __align__(16) unsigned char s_dyn[];
    unsigned int* tidx = reinterpret_cast<unsigned int*>(s_dyn);
    int cx0, cy0, cz0;
    const unsigned int vb = tile_xcd_remap(blockIdx.x, gridDim.x);
    tile_setup(a.geom, a.box, T, cx0, cy0, cz0, vb);
    __syncthreads();
    if (T.off[64] > a.geom.capacity)
        return;
    tile_stage<false, true, false>(a.geom, a.box, T, nullptr, nullptr, nullptr, nullptr, tidx, nullptr);
    __syncthreads();
    const unsigned int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    const unsigned int grp0 = a.geom.d_block_group[vb];
    const unsigned int slot_mask = a.ntypes > 1 ? TL_SLOT_MASK : 0xffffu; // (rows for several types: type in the top bits)
    unsigned int hfirst = 0; // number of the home cell's first member among the block's home particles
    for (unsigned int hc = 0; hc < (unsigned int)TL_NH; hc++)
        {
        const int hx = hc % TB_X, hy = (hc / TB_X) % TB_Y, hz = hc / (TB_X * TB_Y);
        if (!tile_is_home(a.geom, cx0, cy0, cz0, hx, hy, hz))
            continue;
        const unsigned int rc_home = tile_home_rc((int)hc);
        const unsigned int n_cell = T.off[rc_home + 1] - T.off[rc_home];
        for (unsigned int s = wave; s < n_cell; s += 4)
            {
            const unsigned int i = tidx[T.off[rc_home] + s];
            if (i >= a.N)
                continue;
            const unsigned int h = hfirst + s;
            const unsigned short* trow = a.d_tile_nlist + (size_t)(grp0 + h / TL_GROUP) * tile_group_entries(a.geom.row_pieces) + (h % TL_GROUP) * TL_PIECE;
            const unsigned int cap = a.geom.row_pieces * TL_PIECE;
            const unsigned int n = min(a.d_n_neigh[i], cap);
            const size_t head = a.d_head_list[i];
            // Rows laid out by type hold up to three entries of the dummy candidate between their two parts: the n listed
            // neighbors are the first n entries that are not the dummy (what follows them is padding or stale).
            const unsigned int dummy = tile_slot(T.off[64]);
            unsigned int out = 0;
            for (unsigned int k0 = 0; k0 < min(n + 3u, cap) && out < n; k0 += 64)
                {
                const unsigned int k = k0 + lane;
                unsigned int slot = dummy;
                if (k < min(n + 3u, cap))
                    slot = (unsigned int)trow[((k & ~15u) << 4) | (k & 15u)] & slot_mask;
                const bool real = slot != dummy;
                const unsigned long long m = __ballot(real);
                const unsigned int pos = out + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
                if (real && pos < n)
                    a.d_nlist[head + pos] = tidx[tile_unslot(slot)];
                out += (unsigned int)__popcll(m);
                }
            }
        hfirst += n_cell;
        }
    }

__global__ void __launch_bounds__(256) tile_slots_kernel(hmd_tile_geom_t g, unsigned int* d_slot_of, unsigned int nslots)
    {
    const unsigned int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nslots)
        return;
    const unsigned int cell = w / g.cli.w, k = w % g.cli.w;
    if (k < min(g.d_cell_size[cell], g.cli.w))
        d_slot_of[g.d_cell_idx[w]] = w;
    }

__global__ void __launch_bounds__(256) tile_refresh_kernel(const hmd_scalar4* d_pos, const unsigned int* d_slot_of, hmd_scalar4* d_cpos, unsigned int n)
    {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const double4 p = hmd_ld4(d_pos + i);
    hmd_st4(d_cpos + d_slot_of[i], p.x, p.y, p.z, p.w);
    }

size_t build_lds_bytes(unsigned int cap, unsigned int ntypes)
    {
    size_t b = (size_t)cap * 24 + ((size_t)cap + (cap & 1)) * 4;
    if (ntypes > 1)
        b += (size_t)ntypes * ntypes * 8 + cap;
    return (b + 15) / 16 * 16;
    }
    } // namespace

extern "C" int hmd_nlist_tiles_supported(const hmd_index3d_t* ci, const hmd_box_t* box, unsigned int cell_Nmax, unsigned int ntypes,
                                         unsigned int* capacity_out)
    {
    HMD_REQUIRE(ci && box && capacity_out, HMD_ERR_INVALID_ARGUMENT, "hmd_nlist_tiles_supported: ci, box and capacity_out are required");
    const unsigned int dim[3] = {ci->w, ci->h, ci->d};
    for (int d = 0; d < 3; d++)
        {
        HMD_REQUIRE(dim[d] >= 1, HMD_ERR_INVALID_ARGUMENT, "hmd_nlist_tiles_supported: empty cell grid");
        HMD_REQUIRE(!box->periodic[d] || dim[d] >= 3, HMD_ERR_UNSUPPORTED,
                    "tile path needs at least 3 cells along every periodic direction");
        }
    HMD_REQUIRE(ntypes >= 1 && ntypes <= hmd::TL_MAX_ROW_TYPES, HMD_ERR_UNSUPPORTED, "tile path: rows carry type ids in three bits (at most 8 types)");
    size_t cap = (size_t)hmd::TL_RC * cell_Nmax;
    if (ntypes > 1)
        {
        // typed rows address 8192 tile slots: the largest capacity whose dummy candidate still has one
        size_t typed_cap = 0;
        while (hmd::tile_slot((unsigned int)(typed_cap + 1)) <= hmd::TL_SLOT_MASK)
            typed_cap++;
        cap = std::min(cap, typed_cap);
        }
    HMD_REQUIRE(cap >= 1 && hmd::tile_doubles((unsigned int)std::min<size_t>(cap, 1u << 20)) <= 65536 && build_lds_bytes((unsigned int)cap, ntypes) <= hmd::TL_MAX_LDS, HMD_ERR_UNSUPPORTED,
                "tile path: 64 cells of this occupancy do not fit the LDS tile");
    *capacity_out = (unsigned int)cap;
    return HMD_OK;
    }

extern "C" int hmd_nlist_tiles_to_rows(const hmd_tiles_to_rows_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_nlist_tiles_to_rows: args is NULL");
    const hmd_tiles_to_rows_args_t& a = *args;
    HMD_REQUIRE(a.geom.d_cell_size && a.geom.d_cell_idx && a.d_tile_nlist && a.d_n_neigh && a.d_head_list && a.d_nlist && a.geom.d_block_group
                    && a.geom.row_pieces >= 1,
                HMD_ERR_INVALID_ARGUMENT, "hmd_nlist_tiles_to_rows: a required array is NULL");
    unsigned int cap = 0;
    const int rc = hmd_nlist_tiles_supported(&a.geom.ci, &a.box, a.geom.cli.w, 1, &cap);
    if (rc != HMD_OK)
        return rc;
    HMD_REQUIRE(a.geom.capacity >= 1 && a.geom.capacity <= cap, HMD_ERR_INVALID_ARGUMENT,
                "hmd_nlist_tiles_to_rows: geom.capacity must be in [1, 64 * cli.w]");
    if (a.N == 0)
        return HMD_OK;
    const size_t lds = ((size_t)a.geom.capacity * 4 + 15) / 16 * 16;
    hmd::tile_allow_big_lds(tiles_to_rows_kernel, lds);
    hipLaunchKernelGGL(tiles_to_rows_kernel, dim3(hmd::tile_num_blocks(a.geom.ci)), dim3(hmd::TL_THREADS), lds, (hipStream_t)stream, a);
    return hmd_launch_status();
    }

extern "C" int hmd_tile_slots(const hmd_tile_geom_t* geom, unsigned int* d_slot_of, void* stream)
    {
    HMD_REQUIRE(geom && d_slot_of && geom->d_cell_size && geom->d_cell_idx, HMD_ERR_INVALID_ARGUMENT,
                "hmd_tile_slots: geom (cell_size, cell_idx) and d_slot_of are required");
    const size_t nslots = (size_t)geom->cli.w * geom->cli.h;
    HMD_REQUIRE(nslots <= 0xffffffffull, HMD_ERR_TOO_LARGE, "hmd_tile_slots: more than 2^32 cell-list slots");
    if (nslots == 0)
        return HMD_OK;
    hipLaunchKernelGGL(tile_slots_kernel, dim3(hmd_div_up((unsigned int)nslots, 256)), dim3(256), 0, (hipStream_t)stream, *geom, d_slot_of,
                       (unsigned int)nslots);
    return hmd_launch_status();
    }

extern "C" int hmd_tile_refresh(const hmd_scalar4* d_pos, const unsigned int* d_slot_of, hmd_scalar4* d_cpos, unsigned int n, void* stream)
    {
    HMD_REQUIRE(d_pos && d_slot_of && d_cpos, HMD_ERR_INVALID_ARGUMENT, "hmd_tile_refresh: pos, slot_of and cpos are required");
    if (n == 0)
        return HMD_OK;
    hipLaunchKernelGGL(tile_refresh_kernel, dim3(hmd_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, d_pos, d_slot_of, d_cpos, n);
    return hmd_launch_status();
    }
