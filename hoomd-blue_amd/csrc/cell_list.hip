// cell_list.hip -- cell-list binning for gfx950.
//
// Contract: hoomd/CellListGPU.cu:42-244 (gpu_compute_cell_list) with the binning arithmetic of
// hoomd/CellList.cc:444-538, evaluated without fused multiply-adds so that the cell of every
// particle equals the one the CPU path computes.
//
// Design: one lane per particle.  Neighbouring lanes usually fall into the same few cells (the
// particle arrays are kept in cell order by the sorter), so the slot counter is not bumped with
// one global atomic per particle: lanes of a wave that target the same cell elect a leader
// (match-any by iterated readfirstlane), the leader reserves `count` slots with ONE atomicAdd and
// every lane takes base + its rank.  That keeps the HBM-side atomic traffic at one request per
// (wave, distinct cell) instead of one per particle.
// Results of this unit are compared bit for bit with the CPU path: no product may be fused into a
// following sum (the CPU path rounds every product and sum separately).
#pragma clang fp contract(off)
#include "common.cuh"

namespace
    {
__global__ void __launch_bounds__(256) cell_list_kernel(hmd_cell_list_args_t a)
    {
    const unsigned int ntot = a.N + a.n_ghost;
    const unsigned int idx = blockIdx.x * blockDim.x + threadIdx.x;

    bool active = idx < ntot;
    unsigned int bin = 0xffffffffu;
    double4 p = make_double4(0, 0, 0, 0);

    // (what goes into the cell besides the position is asked for here, with the position: behind the atomic below it
    // would be a second round trip to memory for every wave)
    unsigned int body = HMD_NO_BODY;
    double charge = 0.0;
    if (active)
        {
        p = hmd_ld4(a.d_pos + idx);
        if (a.d_type_body && a.d_body)
            body = a.d_body[idx];
        if (a.flag_charge)
            charge = a.d_charge[idx];
        if (isnan(p.x) || isnan(p.y) || isnan(p.z))
            {
            a.d_conditions->y = idx + 1;
            active = false;
            }
        }
    if (active)
        {
        const double3 f = hmd_make_fraction(a.box, p.x, p.y, p.z, a.ghost_width);
        if ((f.x < -0.00001 || f.x >= 1.00001) || (f.y < -0.00001 || f.y >= 1.00001) || (f.z < -0.00001 || f.z >= 1.00001))
            {
            if (idx < a.N)
                a.d_conditions->z = idx + 1;
            active = false;
            }
        else
            {
            int ib = (int)HMD_MUL(f.x, (double)a.ci.w);
            int jb = (int)HMD_MUL(f.y, (double)a.ci.h);
            int kb = (int)HMD_MUL(f.z, (double)a.ci.d);
            if (ib == (int)a.ci.w && a.box.periodic[0])
                ib = 0;
            if (jb == (int)a.ci.h && a.box.periodic[1])
                jb = 0;
            if (kb == (int)a.ci.d && a.box.periodic[2])
                kb = 0;
            if (ib < 0 || ib >= (int)a.ci.w || jb < 0 || jb >= (int)a.ci.h || kb < 0 || kb >= (int)a.ci.d)
                {
                if (idx < a.N)
                    atomicMax(&a.d_conditions->z, idx + 1);
                active = false;
                }
            else
                bin = hmd_idx3(a.ci, (unsigned)ib, (unsigned)jb, (unsigned)kb);
            }
        }

    // wave-aggregated slot reservation: the lanes are grouped by cell first (no memory access), then the leaders of
    // all groups reserve their slots with one atomic instruction, so a wave waits for one round trip to the counters
    // and not for one per distinct cell
    unsigned int slot = 0, group_size = 0;
    int my_leader = 0;
    unsigned long long todo = __ballot(active);
    while (todo)
        {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned int lbin = __shfl(bin, leader);
        const bool mine = active && (bin == lbin);
        const unsigned long long peers = __ballot(mine);
        if (mine)
            {
            slot = hmd_rank_in_mask(peers);
            group_size = (unsigned int)__popcll(peers);
            my_leader = leader;
            }
        todo &= ~peers;
        }
    unsigned int base = 0;
    if (active && (int)(threadIdx.x & 63) == my_leader)
        base = atomicAdd(&a.d_cell_size[bin], group_size);
    slot += __shfl(base, my_leader);

    if (!active)
        return;

    if (slot < a.Nmax)
        {
        const unsigned int w = hmd_idx2(a.cli, slot, bin);
        double flag;
        if (a.flag_charge)
            flag = charge;
        else if (a.flag_type)
            flag = p.w;
        else
            flag = hmd_uint_as_double(idx);
        if (a.d_xyzf)
            hmd_st4(a.d_xyzf + w, p.x, p.y, p.z, flag);
        if (a.d_type_body)
            {
            *reinterpret_cast<uint2*>(a.d_type_body + w) = make_uint2(hmd_double_as_uint(p.w), body);
            }
        if (a.d_cell_orientation)
            *reinterpret_cast<double4*>(a.d_cell_orientation + w) = hmd_ld4(a.d_orientation + idx);
        if (a.d_cell_idx)
            a.d_cell_idx[w] = idx;
        }
    else
        atomicMax(&a.d_conditions->x, slot + 1);
    }

// Orders the members of each cell by particle index with an LDS rank sort: one wave per cell.
__global__ void __launch_bounds__(64) sort_cell_kernel(hmd_sort_cell_list_args_t a, unsigned int ncells)
    {
    extern __shared__ __align__(16) unsigned char s_raw[];
    const unsigned int cell = blockIdx.x;
    if (cell >= ncells)
        return;
    const unsigned int size = min(a.d_cell_size[cell], a.cli.w);
    const unsigned int lane = threadIdx.x;

    unsigned int* s_key = reinterpret_cast<unsigned int*>(s_raw);
    double4* s_xyzf = reinterpret_cast<double4*>(s_raw + ((size_t)a.cli.w * 4 + 15) / 16 * 16);
    uint2* s_tb = reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(s_xyzf) + (size_t)a.cli.w * 32);
    unsigned int* s_idx = reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned char*>(s_tb) + (size_t)a.cli.w * 8);

    for (unsigned int s = lane; s < size; s += 64)
        {
        const unsigned int w = hmd_idx2(a.cli, s, cell);
        const double4 v = hmd_ld4(a.d_xyzf + w);
        s_xyzf[s] = v;
        s_key[s] = a.flag_is_index ? hmd_double_as_uint(v.w) : a.d_cell_idx[w];
        if (a.d_type_body)
            s_tb[s] = *reinterpret_cast<const uint2*>(a.d_type_body + w);
        if (a.d_cell_idx)
            s_idx[s] = a.d_cell_idx[w];
        }
    __syncthreads();
    for (unsigned int s = lane; s < size; s += 64)
        {
        const unsigned int key = s_key[s];
        unsigned int rank = 0;
        for (unsigned int t = 0; t < size; t++)
            {
            const unsigned int k2 = s_key[t];
            rank += (k2 < key || (k2 == key && t < s)) ? 1u : 0u;
            }
        const unsigned int w = hmd_idx2(a.cli, rank, cell);
        *reinterpret_cast<double4*>(a.d_xyzf + w) = s_xyzf[s];
        if (a.d_type_body)
            *reinterpret_cast<uint2*>(a.d_type_body + w) = s_tb[s];
        if (a.d_cell_idx)
            a.d_cell_idx[w] = s_idx[s];
        }
    }
    } // namespace

extern "C" int hmd_compute_cell_list(const hmd_cell_list_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_compute_cell_list: args is NULL");
    HMD_REQUIRE(args->d_cell_size && args->d_conditions && args->d_pos, HMD_ERR_INVALID_ARGUMENT,
                "hmd_compute_cell_list: d_cell_size, d_conditions and d_pos are required");
    HMD_REQUIRE(args->ci.w && args->ci.h && args->ci.d, HMD_ERR_INVALID_ARGUMENT,
                "hmd_compute_cell_list: cell indexer has a zero dimension");
    HMD_REQUIRE(args->cli.w == args->Nmax && args->cli.h == args->ci.w * args->ci.h * args->ci.d,
                HMD_ERR_INVALID_ARGUMENT, "hmd_compute_cell_list: cli must be Index2D(Nmax, ncells)");
    HMD_REQUIRE(!args->flag_charge || args->d_charge, HMD_ERR_INVALID_ARGUMENT,
                "hmd_compute_cell_list: flag_charge needs d_charge");
    HMD_REQUIRE(!args->d_cell_orientation || args->d_orientation, HMD_ERR_INVALID_ARGUMENT,
                "hmd_compute_cell_list: d_cell_orientation needs d_orientation");
    hipStream_t s = (hipStream_t)stream;
    if (args->clear_cell_size)
        {
        hipError_t e = hipMemsetAsync(args->d_cell_size, 0, sizeof(unsigned int) * (size_t)args->cli.h, s);
        if (e != hipSuccess)
            return (int)e;
        }
    const unsigned int ntot = args->N + args->n_ghost;
    if (ntot == 0)
        return HMD_OK;
    const unsigned int bs = 256;
    hipLaunchKernelGGL(cell_list_kernel, dim3(hmd_div_up(ntot, bs)), dim3(bs), 0, s, *args);
    return hmd_launch_status();
    }

extern "C" int hmd_sort_cell_list(const hmd_sort_cell_list_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_sort_cell_list: args is NULL");
    HMD_REQUIRE(args->d_cell_size && args->d_xyzf, HMD_ERR_INVALID_ARGUMENT,
                "hmd_sort_cell_list: d_cell_size and d_xyzf are required");
    HMD_REQUIRE(args->flag_is_index || args->d_cell_idx, HMD_ERR_INVALID_ARGUMENT,
                "hmd_sort_cell_list: without an index flag in xyzf.w, d_cell_idx is required");
    HMD_REQUIRE(args->cli.w <= 1024, HMD_ERR_TOO_LARGE, "hmd_sort_cell_list: more than 1024 slots per cell");
    const unsigned int ncells = args->ci.w * args->ci.h * args->ci.d;
    if (ncells == 0 || args->cli.w == 0)
        return HMD_OK;
    const size_t lds = ((size_t)args->cli.w * 4 + 15) / 16 * 16 + (size_t)args->cli.w * (32 + 8 + 4);
    hipLaunchKernelGGL(sort_cell_kernel, dim3(ncells), dim3(64), lds, (hipStream_t)stream, *args, ncells);
    return hmd_launch_status();
    }
