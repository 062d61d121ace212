// comm.hip -- device side of the halo exchange (the CommunicatorGPU role).
//
// Contracts: hoomd/CommunicatorGPU.cu:306-372 (ghost plan), :778-875 (pack / pack with periodic
// shift).  The exchange itself (RCCL all-to-all over xGMI) is driven from the host layer.
// Results of this unit are compared bit for bit with the CPU path: no product may be fused into a
// following sum (the CPU path rounds every product and sum separately).
#pragma clang fp contract(off)
#include "common.cuh"

#include <algorithm>

namespace
    {
__global__ void __launch_bounds__(256) ghost_plan_kernel(hmd_ghost_plan_args_t a)
    {
    const unsigned int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.N)
        return;
    const double4 p = hmd_ld4(a.d_pos + idx);
    const double rg = a.d_r_ghost[hmd_double_as_uint(p.w)];
    const double gx = rg / a.box.nearest_plane[0], gy = rg / a.box.nearest_plane[1], gz = rg / a.box.nearest_plane[2];
    hmd_scalar3 zero;
    zero.x = zero.y = zero.z = 0.0;
    const double3 f = hmd_make_fraction(a.box, p.x, p.y, p.z, zero);
    unsigned int plan = 0;
    if (f.x >= HMD_SUB(1.0, gx))
        plan |= 1u;
    if (f.x < gx)
        plan |= 2u;
    if (f.y >= HMD_SUB(1.0, gy))
        plan |= 4u;
    if (f.y < gy)
        plan |= 8u;
    if (f.z >= HMD_SUB(1.0, gz))
        plan |= 16u;
    if (f.z < gz)
        plan |= 32u;
    a.d_plan[idx] = plan & a.mask;
    }

__device__ __forceinline__ unsigned int ghost_plan_of(const hmd_ghost_plan_args_t& a, unsigned int idx)
    {
    const double4 p = hmd_ld4(a.d_pos + idx);
    const double rg = a.d_r_ghost[hmd_double_as_uint(p.w)];
    const double gx = rg / a.box.nearest_plane[0], gy = rg / a.box.nearest_plane[1], gz = rg / a.box.nearest_plane[2];
    hmd_scalar3 zero;
    zero.x = zero.y = zero.z = 0.0;
    const double3 f = hmd_make_fraction(a.box, p.x, p.y, p.z, zero);
    unsigned int plan = 0;
    if (f.x >= HMD_SUB(1.0, gx))
        plan |= 1u;
    if (f.x < gx)
        plan |= 2u;
    if (f.y >= HMD_SUB(1.0, gy))
        plan |= 4u;
    if (f.y < gy)
        plan |= 8u;
    if (f.z >= HMD_SUB(1.0, gz))
        plan |= 16u;
    if (f.z < gz)
        plan |= 32u;
    return plan & a.mask;
    }

__global__ void __launch_bounds__(256) ghost_send_list_kernel(hmd_ghost_send_list_args_t a)
    {
    const unsigned int idx = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int plan = 0;
    if (idx < a.ghost_plan.N)
        {
        plan = ghost_plan_of(a.ghost_plan, idx);
        if (a.ghost_plan.d_plan)
            a.ghost_plan.d_plan[idx] = plan;
        }
    if (!__ballot(plan != 0))
        return; // no boundary particle in this wave
    const unsigned int lane = threadIdx.x & 63;
    for (unsigned int k = 0; k < a.n_dir; k++)
        {
        const unsigned int m = a.d_dir_mask[k];
        const bool match = plan != 0 && (plan & m) == m;
        const unsigned long long ballot = __ballot(match);
        if (!ballot)
            continue;
        // one reservation per wave and direction
        unsigned int base = 0;
        if (lane == 0)
            {
            base = atomicAdd(a.d_total, (unsigned int)__popcll(ballot));
            atomicAdd(a.d_dir_count + k, (unsigned int)__popcll(ballot));
            }
        base = __shfl(base, 0, 64);
        if (match)
            {
            const unsigned int slot = base + hmd_rank_in_mask(ballot);
            if (slot < a.capacity)
                a.d_keys[slot] = ((unsigned long long)k << 32) | idx;
            }
        }
    }

__global__ void __launch_bounds__(256) migrate_dest_kernel(hmd_migrate_dest_args_t a)
    {
    const unsigned int idx = blockIdx.x * blockDim.x + threadIdx.x;
    bool leaves = false;
    unsigned int dest = a.rank;
    if (idx < a.N)
        {
        const double4 p = hmd_ld4(a.d_pos + idx);
        const double x[3] = {p.x, p.y, p.z};
        unsigned int c[3];
#pragma unroll
        for (int d = 0; d < 3; d++)
            {
            const double f = HMD_SUB(x[d], a.global_lo[d]) / a.global_L[d];
            const double w = a.decomposed[d] ? floor(f) : 0.0;
            const double g = floor(HMD_MUL(HMD_SUB(f, w), (double)a.grid[d]));
            const long long gi = (long long)g;
            c[d] = (unsigned int)min(max(gi, 0ll), (long long)a.grid[d] - 1);
            }
        dest = (c[2] * a.grid[1] + c[1]) * a.grid[0] + c[0];
        a.d_dest[idx] = (int)dest;
        leaves = dest != a.rank;
        }
    const unsigned long long m = __ballot(leaves);
    if (!m)
        return;
    // one reservation per wave
    unsigned int base = 0;
    if ((threadIdx.x & 63) == 0)
        base = atomicAdd(a.d_n_leave, (unsigned int)__popcll(m));
    if (a.d_keys)
        {
        base = __shfl(base, 0, 64);
        if (leaves)
            {
            const unsigned int slot = base + hmd_rank_in_mask(m);
            if (slot < a.capacity)
                a.d_keys[slot] = ((unsigned long long)dest << 32) | idx;
            atomicAdd(a.d_rank_count + dest, 1u);
            }
        }
    }

__device__ __forceinline__ long long bits_of(double v)
    {
    return __double_as_longlong(v);
    }

__global__ void __launch_bounds__(256) migrate_pack_kernel(hmd_migrate_pack_args_t a)
    {
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (a.slot_rows && k < a.n_ranks)
        {
        // header of the message to rank k
        unsigned int most = 0;
        for (unsigned int r = 0; r < a.n_ranks; r++)
            most = max(most, a.send_off[r + 1] - a.send_off[r]);
        long long* h = a.d_out + (size_t)k * (a.slot_rows + 1) * HMD_MIGRATE_ROW_WORDS;
        h[0] = (long long)(a.send_off[k + 1] - a.send_off[k]);
        h[1] = (long long)most;
        h[2] = (long long)a.slot_rows;
        }
    if (k >= a.n)
        return;
    const unsigned long long key = a.d_keys[k];
    const unsigned int i = (unsigned int)(key & 0xffffffffull);
    size_t out_row = k;
    if (a.slot_rows)
        {
        const unsigned int dest = (unsigned int)(key >> 32);
        const unsigned int local = k - a.send_off[dest];
        if (dest >= a.n_ranks || local >= a.slot_rows)
            return; // does not fit this round's messages
        out_row = (size_t)dest * (a.slot_rows + 1) + 1 + local;
        }
    const double4 p = hmd_ld4(a.arrays.d_pos + i);
    const double4 v = hmd_ld4(a.arrays.d_vel + i);
    const hmd_scalar3 acc = a.arrays.d_accel[i];
    const hmd_int3 img = a.arrays.d_image[i];
    double x[3] = {p.x, p.y, p.z};
    long long im[3] = {img.x, img.y, img.z};
#pragma unroll
    for (int d = 0; d < 3; d++)
        {
        // the wrap the owner decision used (migrate_dest_kernel): same arithmetic, same result
        const double f = HMD_SUB(x[d], a.global_lo[d]) / a.global_L[d];
        const double w = a.decomposed[d] ? floor(f) : 0.0;
        x[d] = HMD_SUB(x[d], HMD_MUL(w, a.global_L[d]));
        im[d] += (long long)w;
        }
    long long* row = a.d_out + out_row * HMD_MIGRATE_ROW_WORDS;
    row[0] = bits_of(x[0]);
    row[1] = bits_of(x[1]);
    row[2] = bits_of(x[2]);
    row[3] = bits_of(p.w);
    row[4] = bits_of(v.x);
    row[5] = bits_of(v.y);
    row[6] = bits_of(v.z);
    row[7] = bits_of(v.w);
    row[8] = bits_of(acc.x);
    row[9] = bits_of(acc.y);
    row[10] = bits_of(acc.z);
    row[11] = im[0];
    row[12] = im[1];
    row[13] = im[2];
    row[14] = (long long)(int)a.arrays.d_tag[i];
    row[15] = (long long)(int)a.arrays.d_body[i];
    row[16] = bits_of(a.arrays.d_charge[i]);
    row[17] = bits_of(a.arrays.d_diameter[i]);
    }

__global__ void __launch_bounds__(256) migrate_unpack_kernel(hmd_migrate_unpack_args_t a)
    {
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n)
        return;
    size_t in_row = k;
    if (a.slot_rows)
        {
        unsigned int r = 0;
        while (a.recv_off[r + 1] <= k)
            r++;
        in_row = (size_t)r * (a.slot_rows + 1) + 1 + (k - a.recv_off[r]);
        }
    const long long* row = a.d_in + in_row * HMD_MIGRATE_ROW_WORDS;
    const unsigned int i = a.first + k;
    hmd_st4(a.arrays.d_pos + i, __longlong_as_double(row[0]), __longlong_as_double(row[1]), __longlong_as_double(row[2]), __longlong_as_double(row[3]));
    hmd_st4(a.arrays.d_vel + i, __longlong_as_double(row[4]), __longlong_as_double(row[5]), __longlong_as_double(row[6]), __longlong_as_double(row[7]));
    hmd_scalar3 acc;
    acc.x = __longlong_as_double(row[8]);
    acc.y = __longlong_as_double(row[9]);
    acc.z = __longlong_as_double(row[10]);
    a.arrays.d_accel[i] = acc;
    hmd_int3 img;
    img.x = (int)row[11];
    img.y = (int)row[12];
    img.z = (int)row[13];
    a.arrays.d_image[i] = img;
    a.arrays.d_tag[i] = (unsigned int)row[14];
    a.arrays.d_body[i] = (unsigned int)row[15];
    a.arrays.d_charge[i] = __longlong_as_double(row[16]);
    a.arrays.d_diameter[i] = __longlong_as_double(row[17]);
    }

__global__ void __launch_bounds__(256) split_keys_kernel(const unsigned long long* d_keys, unsigned int n, unsigned int* d_send_idx, unsigned char* d_send_dir)
    {
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    const unsigned long long key = d_keys[k];
    d_send_idx[k] = (unsigned int)(key & 0xffffffffull);
    d_send_dir[k] = (unsigned char)(key >> 32);
    }

__global__ void __launch_bounds__(256) pack_init_kernel(hmd_pack_pos_args_t a, const unsigned int* d_tag)
    {
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n_send)
        return;
    const unsigned int i = a.d_send_idx[k];
    const double4 p = hmd_ld4(a.d_pos + i);
    const hmd_scalar3 sh = a.d_shift_table[a.d_send_shift_id[k]];
    const double w = __hiloint2double((int)d_tag[i], (int)hmd_double_as_uint(p.w));
    hmd_st4(a.d_out + k, HMD_ADD(p.x, sh.x), HMD_ADD(p.y, sh.y), HMD_ADD(p.z, sh.z), w);
    }

__global__ void __launch_bounds__(256) unpack_init_kernel(hmd_unpack_ghost_init_args_t a)
    {
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n)
        return;
    const double4 p = hmd_ld4(a.d_in + k);
    const unsigned int i = a.first + k;
    hmd_st4(a.arrays.d_pos + i, p.x, p.y, p.z, hmd_uint_as_double(hmd_double_as_uint(p.w)));
    a.arrays.d_tag[i] = (unsigned int)__double2hiint(p.w);
    hmd_int3 zero;
    zero.x = zero.y = zero.z = 0;
    a.arrays.d_image[i] = zero;
    a.arrays.d_body[i] = 0xffffffffu;
    a.arrays.d_charge[i] = 0.0;
    a.arrays.d_diameter[i] = 1.0;
    }

__global__ void __launch_bounds__(64) migrate_headers_kernel(const long long* d_in, unsigned int slot_rows, unsigned int n_ranks, unsigned int* d_words)
    {
    const unsigned int r = threadIdx.x;
    if (r >= n_ranks)
        return;
    const long long* h = d_in + (size_t)r * (slot_rows + 1) * HMD_MIGRATE_ROW_WORDS;
    d_words[2 * r] = (unsigned int)h[0];
    d_words[2 * r + 1] = (unsigned int)h[1];
    }

// rank whose range [off[r], off[r + 1]) holds k (k < off[n_ranks])
__device__ __forceinline__ unsigned int slot_rank_of(const hmd_ghost_slots_t& s, unsigned int k)
    {
    unsigned int r = 0;
    while (r + 1 < s.n_ranks && s.off[r + 1] <= k)
        r++;
    return r;
    }

__global__ void __launch_bounds__(256) pack_init_slots_kernel(hmd_pack_pos_args_t a, const unsigned int* d_tag, hmd_ghost_slots_t s)
    {
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < a.n_send)
        {
        const unsigned int r = slot_rank_of(s, k), local = k - s.off[r];
        if (local < s.cap[r])
            {
            const unsigned int i = a.d_send_idx[k];
            const double4 p = hmd_ld4(a.d_pos + i);
            const hmd_scalar3 sh = a.d_shift_table[a.d_send_shift_id[k]];
            const double w = __hiloint2double((int)d_tag[i], (int)hmd_double_as_uint(p.w));
            hmd_st4(a.d_out + (size_t)s.row0[r] + 1 + local, HMD_ADD(p.x, sh.x), HMD_ADD(p.y, sh.y), HMD_ADD(p.z, sh.z), w);
            }
        }
    else if (k < a.n_send + s.n_ranks)
        {
        const unsigned int r = k - a.n_send;
        hmd_st4(a.d_out + s.row0[r], hmd_uint_as_double(s.off[r + 1] - s.off[r]), hmd_uint_as_double(s.flag), 0.0, 0.0);
        }
    }

__global__ void __launch_bounds__(64) slot_headers_kernel(const hmd_scalar4* d_in, hmd_ghost_slots_t s, unsigned int* d_words)
    {
    const unsigned int r = threadIdx.x;
    if (r >= s.n_ranks)
        return;
    const double4 p = hmd_ld4(d_in + s.row0[r]);
    d_words[2 * r] = hmd_double_as_uint(p.x);
    d_words[2 * r + 1] = hmd_double_as_uint(p.y);
    }

__global__ void __launch_bounds__(256) unpack_init_slots_kernel(hmd_unpack_ghost_init_args_t a, hmd_ghost_slots_t s)
    {
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n)
        return;
    const unsigned int r = slot_rank_of(s, k);
    const double4 p = hmd_ld4(a.d_in + (size_t)s.row0[r] + 1 + (k - s.off[r]));
    const unsigned int i = a.first + k;
    hmd_st4(a.arrays.d_pos + i, p.x, p.y, p.z, hmd_uint_as_double(hmd_double_as_uint(p.w)));
    a.arrays.d_tag[i] = (unsigned int)__double2hiint(p.w);
    hmd_int3 zero;
    zero.x = zero.y = zero.z = 0;
    a.arrays.d_image[i] = zero;
    a.arrays.d_body[i] = 0xffffffffu;
    a.arrays.d_charge[i] = 0.0;
    a.arrays.d_diameter[i] = 1.0;
    }

__global__ void __launch_bounds__(256) vote_plan_kernel(hmd_vote_plan_args_t a)
    {
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int n_send = a.send_off[a.n_ranks], n_recv = a.recv_off[a.n_ranks];
    // rows of the outgoing buffer: message to rank r starts after the data rows of the ranks before it and
    // one vote row for each of those that is not this rank
    if (t < n_send)
        {
        unsigned int r = 0;
        while (a.send_off[r + 1] <= t)
            r++;
        a.d_out_row[t] = t + r - (r > a.rank ? 1u : 0u);
        }
    if (t < a.n_ranks && t != a.rank)
        {
        const unsigned int j = t - (t > a.rank ? 1u : 0u);
        a.d_flag_rows[j] = a.send_off[t + 1] + j;
        }
    const unsigned int n_in = n_recv + a.n_ranks - 1;
    if (t < n_in)
        {
        // incoming row t: find the rank whose message holds it
        unsigned int r = 0;
        for (;; r++)
            {
            const unsigned int end = a.recv_off[r + 1] + (r + 1) - ((r + 1) > a.rank ? 1u : 0u); // rows of the messages of ranks <= r
            if (t < end)
                break;
            }
        const unsigned int begin = a.recv_off[r] + r - (r > a.rank ? 1u : 0u);
        const unsigned int local = t - begin;
        a.d_in_map[t] = local < a.recv_off[r + 1] - a.recv_off[r] ? (int)(a.recv_off[r] + local) : -1;
        }
    }

constexpr unsigned int MF_THREADS = 1024;

// Appends value(j) for every j < n with flag(j) to `out` in order of j; returns how many.  One workgroup.
template<class Flag, class Value>
__device__ unsigned int block_compact(unsigned int n, unsigned int* out, unsigned int* s_wave, Flag flag, Value value)
    {
    const unsigned int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int base = 0;
    for (unsigned int j0 = 0; j0 < n; j0 += MF_THREADS)
        {
        const unsigned int j = j0 + threadIdx.x;
        const bool f = j < n && flag(j);
        const unsigned long long m = __ballot(f);
        if (lane == 0)
            s_wave[wave] = (unsigned int)__popcll(m);
        __syncthreads();
        unsigned int before = 0, total = 0;
        for (unsigned int w = 0; w < MF_THREADS / 64; w++)
            {
            const unsigned int c = s_wave[w];
            before += w < wave ? c : 0u;
            total += c;
            }
        if (f)
            out[base + before + hmd_rank_in_mask(m)] = value(j);
        base += total;
        __syncthreads();
        }
    return base;
    }

__global__ void __launch_bounds__(MF_THREADS) migrate_fill_kernel(hmd_migrate_fill_args_t a)
    {
    __shared__ unsigned int s_wave[MF_THREADS / 64];
    const unsigned int n_keep = a.N - a.n_leave;
    unsigned int* holes = a.d_scratch;
    unsigned int* movers = a.d_scratch + a.n_leave;
    // places in front of the boundary that a leaver frees, in the order of the (sorted) keys
    const unsigned int nh = block_compact(
        a.n_leave, holes, s_wave, [&](unsigned int j) { return (unsigned int)(a.d_keys[j] & 0xffffffffull) < n_keep; },
        [&](unsigned int j) { return (unsigned int)(a.d_keys[j] & 0xffffffffull); });
    // stayers behind the boundary, in index order
    block_compact(
        a.n_leave, movers, s_wave, [&](unsigned int j) { return a.d_dest[n_keep + j] == (int)a.rank; }, [&](unsigned int j) { return n_keep + j; });
    __syncthreads(); // (the lists live in global memory: written and read by this one workgroup)
    for (unsigned int k = threadIdx.x; k < nh; k += MF_THREADS)
        {
        const unsigned int src = movers[k], dst = holes[k];
        const double4 p = hmd_ld4(a.arrays.d_pos + src);
        hmd_st4(a.arrays.d_pos + dst, p.x, p.y, p.z, p.w);
        const double4 v = hmd_ld4(a.arrays.d_vel + src);
        hmd_st4(a.arrays.d_vel + dst, v.x, v.y, v.z, v.w);
        a.arrays.d_accel[dst] = a.arrays.d_accel[src];
        a.arrays.d_image[dst] = a.arrays.d_image[src];
        a.arrays.d_tag[dst] = a.arrays.d_tag[src];
        a.arrays.d_body[dst] = a.arrays.d_body[src];
        a.arrays.d_charge[dst] = a.arrays.d_charge[src];
        a.arrays.d_diameter[dst] = a.arrays.d_diameter[src];
        }
    }

// ---- the same packs with their sizes read on the device ----
// The counts a pack needs (how many particles leave, how many go to each rank; how many ghost rows go in each
// direction) are where the kernels before it left them, in device memory.  Reading them there instead of waiting for
// the host to fetch them and pass them back as arguments takes a round trip to the host out of each half of the
// exchange: the host looks at the counts once, after the messages have crossed, together with the headers.
__global__ void __launch_bounds__(256) migrate_pack_dev_kernel(hmd_migrate_pack_args_t a, const unsigned int* d_n_leave, const unsigned int* d_rank_count,
                                                               unsigned int capacity)
    {
    __shared__ unsigned int s_off[66];
    if (threadIdx.x == 0)
        {
        unsigned int run = 0, most = 0;
        for (unsigned int r = 0; r < a.n_ranks; r++)
            {
            s_off[r] = run;
            const unsigned int c = d_rank_count[r];
            run += c;
            most = max(most, c);
            }
        s_off[a.n_ranks] = run;
        // more leavers than the key array holds: the list is incomplete, which every receiver learns from a count no
        // message can have (all ranks then start over, this one with a larger array)
        s_off[65] = *d_n_leave > capacity ? 0xffffffffu : most;
        }
    __syncthreads();
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < a.n_ranks)
        {
        long long* h = a.d_out + (size_t)k * (a.slot_rows + 1) * HMD_MIGRATE_ROW_WORDS;
        h[0] = (long long)(s_off[k + 1] - s_off[k]);
        h[1] = (long long)s_off[65];
        h[2] = (long long)a.slot_rows;
        }
    const unsigned int n = min(*d_n_leave, capacity);
    if (k >= n)
        return;
    const unsigned long long key = a.d_keys[k];
    const unsigned int i = (unsigned int)(key & 0xffffffffull);
    const unsigned int dest = (unsigned int)(key >> 32);
    if (dest >= a.n_ranks)
        return; // (an unused place of the key array: only when more particles leave than it holds, and then the host starts over)
    const unsigned int local = k - s_off[dest];
    if (local >= a.slot_rows)
        return; // does not fit this round's messages
    const size_t out_row = (size_t)dest * (a.slot_rows + 1) + 1 + local;
    const double4 p = hmd_ld4(a.arrays.d_pos + i);
    const double4 v = hmd_ld4(a.arrays.d_vel + i);
    const hmd_scalar3 acc = a.arrays.d_accel[i];
    const hmd_int3 img = a.arrays.d_image[i];
    double x[3] = {p.x, p.y, p.z};
    long long im[3] = {img.x, img.y, img.z};
#pragma unroll
    for (int d = 0; d < 3; d++)
        {
        const double f = HMD_SUB(x[d], a.global_lo[d]) / a.global_L[d];
        const double w = a.decomposed[d] ? floor(f) : 0.0;
        x[d] = HMD_SUB(x[d], HMD_MUL(w, a.global_L[d]));
        im[d] += (long long)w;
        }
    long long* row = a.d_out + out_row * HMD_MIGRATE_ROW_WORDS;
    row[0] = bits_of(x[0]);
    row[1] = bits_of(x[1]);
    row[2] = bits_of(x[2]);
    row[3] = bits_of(p.w);
    row[4] = bits_of(v.x);
    row[5] = bits_of(v.y);
    row[6] = bits_of(v.z);
    row[7] = bits_of(v.w);
    row[8] = bits_of(acc.x);
    row[9] = bits_of(acc.y);
    row[10] = bits_of(acc.z);
    row[11] = im[0];
    row[12] = im[1];
    row[13] = im[2];
    row[14] = (long long)(int)a.arrays.d_tag[i];
    row[15] = (long long)(int)a.arrays.d_body[i];
    row[16] = bits_of(a.arrays.d_charge[i]);
    row[17] = bits_of(a.arrays.d_diameter[i]);
    }

// (the word also goes, stamped, to mapped pinned memory the host polls, when the caller names it: hmd_publish_words
// without a launch of its own)
__device__ __forceinline__ void comm_word_out(unsigned int* d_words, unsigned long long* d_published, unsigned int stamp, unsigned int i, unsigned int v)
    {
    d_words[i] = v;
    if (d_published)
        __hip_atomic_store(d_published + i, ((unsigned long long)stamp << 32) | (unsigned long long)v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }

__global__ void __launch_bounds__(64) migrate_headers_and_kernel(const long long* d_in, unsigned int slot_rows, unsigned int n_ranks, const unsigned int* d_extra,
                                                                 unsigned int n_extra, unsigned int* d_words, unsigned long long* d_published, unsigned int stamp)
    {
    const unsigned int r = threadIdx.x;
    if (r < n_ranks)
        {
        const long long* h = d_in + (size_t)r * (slot_rows + 1) * HMD_MIGRATE_ROW_WORDS;
        comm_word_out(d_words, d_published, stamp, 2 * r, (unsigned int)h[0]);
        comm_word_out(d_words, d_published, stamp, 2 * r + 1, (unsigned int)h[1]);
        }
    if (r < n_extra)
        comm_word_out(d_words, d_published, stamp, 2 * n_ranks + r, d_extra[r]);
    }

__global__ void __launch_bounds__(256) pack_init_slots_dev_kernel(hmd_pack_pos_args_t a, const unsigned int* d_tag, hmd_ghost_slots_t s, const unsigned int* d_total,
                                                                  const unsigned int* d_dir_count, const unsigned int* d_dir_dest, unsigned int n_dir,
                                                                  unsigned int capacity, const unsigned long long* d_sorted_keys, unsigned int* d_send_idx_out,
                                                                  unsigned char* d_send_dir_out)
    {
    // rows per destination rank from the rows per direction (the send list is sorted by direction, and the directions
    // by destination rank), their prefix sums, and whether any message is too small for its rows
    __shared__ unsigned int s_off[HMD_GHOST_SLOTS_MAX_RANKS + 2];
    if (threadIdx.x == 0)
        {
        unsigned int rows[HMD_GHOST_SLOTS_MAX_RANKS];
        for (unsigned int r = 0; r < s.n_ranks; r++)
            rows[r] = 0;
        for (unsigned int k = 0; k < n_dir; k++)
            rows[d_dir_dest[k]] += d_dir_count[k];
        unsigned int run = 0, flag = 0;
        for (unsigned int r = 0; r < s.n_ranks; r++)
            {
            s_off[r] = run;
            run += rows[r];
            if (rows[r] > s.cap[r])
                flag = 1;
            }
        s_off[s.n_ranks] = run;
        // bit 1: the send list outgrew its key array and is incomplete (all ranks start over, this one with a larger array)
        if (*d_total > capacity)
            flag |= 2u;
        s_off[HMD_GHOST_SLOTS_MAX_RANKS + 1] = flag;
        }
    __syncthreads();
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < s.n_ranks)
        hmd_st4(a.d_out + s.row0[k], hmd_uint_as_double(s_off[k + 1] - s_off[k]), hmd_uint_as_double(s_off[HMD_GHOST_SLOTS_MAX_RANKS + 1]), 0.0, 0.0);
    const unsigned int n_send = min(*d_total, capacity);
    if (k >= n_send)
        return;
    // the send list either exists (d_send_idx, d_send_shift_id) or is taken from the sorted keys here and stored for the
    // exchanges of the steps to come (hmd_ghost_split_keys without a launch of its own)
    unsigned int i, dir;
    if (d_sorted_keys)
        {
        const unsigned long long key = d_sorted_keys[k];
        i = (unsigned int)(key & 0xffffffffull);
        dir = (unsigned int)(key >> 32) & 0xffu;
        d_send_idx_out[k] = i;
        d_send_dir_out[k] = (unsigned char)dir;
        }
    else
        {
        i = a.d_send_idx[k];
        dir = a.d_send_shift_id[k];
        }
    unsigned int r = 0;
    while (r + 1 < s.n_ranks && s_off[r + 1] <= k)
        r++;
    const unsigned int local = k - s_off[r];
    if (local >= s.cap[r])
        return;
    const double4 p = hmd_ld4(a.d_pos + i);
    const hmd_scalar3 sh = a.d_shift_table[dir];
    const double w = __hiloint2double((int)d_tag[i], (int)hmd_double_as_uint(p.w));
    hmd_st4(a.d_out + (size_t)s.row0[r] + 1 + local, HMD_ADD(p.x, sh.x), HMD_ADD(p.y, sh.y), HMD_ADD(p.z, sh.z), w);
    }

__global__ void __launch_bounds__(64) slot_headers_and_kernel(const hmd_scalar4* d_in, hmd_ghost_slots_t s, const unsigned int* d_extra, unsigned int n_extra,
                                                              unsigned int* d_words, unsigned long long* d_published, unsigned int stamp)
    {
    const unsigned int r = threadIdx.x;
    if (r < s.n_ranks)
        {
        const double4 p = hmd_ld4(d_in + s.row0[r]);
        comm_word_out(d_words, d_published, stamp, 2 * r, hmd_double_as_uint(p.x));
        comm_word_out(d_words, d_published, stamp, 2 * r + 1, hmd_double_as_uint(p.y));
        }
    if (r < n_extra)
        comm_word_out(d_words, d_published, stamp, 2 * s.n_ranks + r, d_extra[r]);
    }

// zeroes a few counters and fills a key array with the padding that sorts behind every key: one launch for the two
// memsets that precede a counting kernel
__global__ void __launch_bounds__(256) comm_prep_kernel(unsigned int* d_counters, unsigned int n_counters, unsigned long long* d_keys, unsigned int n_keys)
    {
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_counters)
        d_counters[k] = 0u;
    if (k < n_keys)
        d_keys[k] = ~0ull;
    }

__global__ void __launch_bounds__(256) pack_pos_kernel(hmd_pack_pos_args_t a)
    {
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n_send)
        return;
    const double4 p = hmd_ld4(a.d_pos + a.d_send_idx[k]);
    const hmd_scalar3 sh = a.d_shift_table[a.d_send_shift_id[k]];
    hmd_st4(a.d_out + k, HMD_ADD(p.x, sh.x), HMD_ADD(p.y, sh.y), HMD_ADD(p.z, sh.z), p.w);
    }
    } // namespace

extern "C" int hmd_make_ghost_exchange_plan(const hmd_ghost_plan_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_make_ghost_exchange_plan: args is NULL");
    HMD_REQUIRE(args->d_plan && args->d_pos && args->d_r_ghost, HMD_ERR_INVALID_ARGUMENT,
                "hmd_make_ghost_exchange_plan: plan, pos and r_ghost are required");
    if (args->N == 0)
        return HMD_OK;
    hipLaunchKernelGGL(ghost_plan_kernel, dim3(hmd_div_up(args->N, 256)), dim3(256), 0, (hipStream_t)stream, *args);
    return hmd_launch_status();
    }

__global__ void __launch_bounds__(256) pack_vote_kernel(hmd_pack_vote_args_t a)
    {
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < a.pack.n_send)
        {
        const double4 p = hmd_ld4(a.pack.d_pos + a.pack.d_send_idx[k]);
        const hmd_scalar3 sh = a.pack.d_shift_table[a.pack.d_send_shift_id[k]];
        hmd_st4(a.pack.d_out + a.d_out_row[k], HMD_ADD(p.x, sh.x), HMD_ADD(p.y, sh.y), HMD_ADD(p.z, sh.z), p.w);
        }
    else if (k < a.pack.n_send + a.n_flag_rows)
        {
        const double w = hmd_uint_as_double(*a.d_flag_word);
        hmd_st4(a.pack.d_out + a.d_flag_rows[k - a.pack.n_send], w, 0.0, 0.0, 0.0);
        }
    }

__global__ void __launch_bounds__(256) unpack_vote_kernel(hmd_unpack_vote_args_t a)
    {
    const unsigned int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_rows)
        return;
    const double4 p = hmd_ld4(a.d_in + r);
    const int m = a.d_map[r];
    if (m >= 0)
        hmd_st4(a.d_ghost_pos + m, p.x, p.y, p.z, p.w);
    else
        atomicMax(a.d_flag_word, hmd_double_as_uint(p.x));
    }

extern "C" int hmd_ghost_send_list(const hmd_ghost_send_list_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_ghost_send_list: args is NULL");
    const hmd_ghost_plan_args_t& g = args->ghost_plan;
    HMD_REQUIRE(g.d_pos && g.d_r_ghost && g.ntypes >= 1, HMD_ERR_INVALID_ARGUMENT, "hmd_ghost_send_list: pos and r_ghost are required");
    HMD_REQUIRE(args->d_total && args->d_dir_count, HMD_ERR_INVALID_ARGUMENT, "hmd_ghost_send_list: total and dir_count are required");
    HMD_REQUIRE(args->n_dir <= 26 && (args->n_dir == 0 || args->d_dir_mask), HMD_ERR_INVALID_ARGUMENT, "hmd_ghost_send_list: at most 26 directions");
    HMD_REQUIRE(args->capacity == 0 || args->d_keys, HMD_ERR_INVALID_ARGUMENT, "hmd_ghost_send_list: d_keys is required");
    if (g.N == 0)
        return HMD_OK;
    hipLaunchKernelGGL(ghost_send_list_kernel, dim3(hmd_div_up(g.N, 256)), dim3(256), 0, (hipStream_t)stream, *args);
    return hmd_launch_status();
    }

extern "C" int hmd_pack_ghost_positions_vote(const hmd_pack_vote_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_pack_ghost_positions_vote: args is NULL");
    const hmd_pack_pos_args_t& p = args->pack;
    HMD_REQUIRE(p.d_out && p.d_pos && p.d_shift_table, HMD_ERR_INVALID_ARGUMENT, "hmd_pack_ghost_positions_vote: out, pos and shift_table are required");
    HMD_REQUIRE(p.n_send == 0 || (p.d_send_idx && p.d_send_shift_id && args->d_out_row), HMD_ERR_INVALID_ARGUMENT,
                "hmd_pack_ghost_positions_vote: send_idx, send_shift_id and out_row are required");
    HMD_REQUIRE(args->n_flag_rows == 0 || (args->d_flag_word && args->d_flag_rows), HMD_ERR_INVALID_ARGUMENT,
                "hmd_pack_ghost_positions_vote: flag_word and flag_rows are required");
    const unsigned int n = p.n_send + args->n_flag_rows;
    if (n == 0)
        return HMD_OK;
    hipLaunchKernelGGL(pack_vote_kernel, dim3(hmd_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, *args);
    return hmd_launch_status();
    }

extern "C" int hmd_unpack_ghost_positions(const hmd_unpack_vote_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_unpack_ghost_positions: args is NULL");
    if (args->n_rows == 0)
        return HMD_OK;
    HMD_REQUIRE(args->d_in && args->d_map && args->d_ghost_pos && args->d_flag_word, HMD_ERR_INVALID_ARGUMENT,
                "hmd_unpack_ghost_positions: in, map, ghost_pos and flag_word are required");
    hipLaunchKernelGGL(unpack_vote_kernel, dim3(hmd_div_up(args->n_rows, 256)), dim3(256), 0, (hipStream_t)stream, *args);
    return hmd_launch_status();
    }

extern "C" int hmd_migrate_dest(const hmd_migrate_dest_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_migrate_dest: args is NULL");
    HMD_REQUIRE(args->d_pos && args->d_dest && args->d_n_leave, HMD_ERR_INVALID_ARGUMENT, "hmd_migrate_dest: pos, dest and n_leave are required");
    for (int d = 0; d < 3; d++)
        HMD_REQUIRE(args->grid[d] >= 1 && args->global_L[d] > 0.0, HMD_ERR_INVALID_ARGUMENT, "hmd_migrate_dest: empty grid or box");
    HMD_REQUIRE(args->rank < args->grid[0] * args->grid[1] * args->grid[2], HMD_ERR_INVALID_ARGUMENT, "hmd_migrate_dest: rank outside the grid");
    HMD_REQUIRE(!args->d_keys || (args->d_rank_count && args->n_ranks == args->grid[0] * args->grid[1] * args->grid[2]), HMD_ERR_INVALID_ARGUMENT,
                "hmd_migrate_dest: d_keys needs d_rank_count with one entry per rank of the grid");
    if (args->N == 0)
        return HMD_OK;
    hipLaunchKernelGGL(migrate_dest_kernel, dim3(hmd_div_up(args->N, 256)), dim3(256), 0, (hipStream_t)stream, *args);
    return hmd_launch_status();
    }

extern "C" int hmd_pack_ghost_positions(const hmd_pack_pos_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_pack_ghost_positions: args is NULL");
    HMD_REQUIRE(args->d_out && args->d_pos && args->d_send_idx && args->d_send_shift_id && args->d_shift_table, HMD_ERR_INVALID_ARGUMENT,
                "hmd_pack_ghost_positions: all array arguments are required");
    if (args->n_send == 0)
        return HMD_OK;
    hipLaunchKernelGGL(pack_pos_kernel, dim3(hmd_div_up(args->n_send, 256)), dim3(256), 0, (hipStream_t)stream, *args);
    return hmd_launch_status();
    }

static bool pdata_arrays_complete(const hmd_pdata_arrays_t& p)
    {
    return p.d_pos && p.d_vel && p.d_accel && p.d_image && p.d_tag && p.d_body && p.d_charge && p.d_diameter;
    }

extern "C" int hmd_migrate_pack(const hmd_migrate_pack_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_migrate_pack: args is NULL");
    if (args->slot_rows)
        {
        HMD_REQUIRE(args->n_ranks >= 1 && args->n_ranks <= 64 && args->d_out, HMD_ERR_INVALID_ARGUMENT,
                    "hmd_migrate_pack: fixed-size messages need 1 to 64 ranks and d_out");
        HMD_REQUIRE(args->send_off[0] == 0 && args->send_off[args->n_ranks] == args->n, HMD_ERR_INVALID_ARGUMENT,
                    "hmd_migrate_pack: send_off must start at 0 and end at n");
        for (unsigned int r = 0; r < args->n_ranks; r++)
            HMD_REQUIRE(args->send_off[r] <= args->send_off[r + 1], HMD_ERR_INVALID_ARGUMENT, "hmd_migrate_pack: send_off must not decrease");
        }
    else if (args->n == 0)
        return HMD_OK;
    HMD_REQUIRE(args->n == 0 || (pdata_arrays_complete(args->arrays) && args->d_keys && args->d_out), HMD_ERR_INVALID_ARGUMENT,
                "hmd_migrate_pack: all particle arrays, d_keys and d_out are required");
    for (int d = 0; d < 3; d++)
        HMD_REQUIRE(args->global_L[d] > 0.0, HMD_ERR_INVALID_ARGUMENT, "hmd_migrate_pack: empty box");
    const unsigned int threads = args->slot_rows ? std::max(args->n, args->n_ranks) : args->n;
    hipLaunchKernelGGL(migrate_pack_kernel, dim3(hmd_div_up(threads, 256)), dim3(256), 0, (hipStream_t)stream, *args);
    return hmd_launch_status();
    }

extern "C" int hmd_migrate_fill(const hmd_migrate_fill_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_migrate_fill: args is NULL");
    if (args->n_leave == 0)
        return HMD_OK;
    HMD_REQUIRE(args->n_leave <= args->N, HMD_ERR_INVALID_ARGUMENT, "hmd_migrate_fill: more leavers than particles");
    HMD_REQUIRE(pdata_arrays_complete(args->arrays) && args->d_dest && args->d_keys && args->d_scratch, HMD_ERR_INVALID_ARGUMENT,
                "hmd_migrate_fill: all particle arrays, d_dest, d_keys and d_scratch are required");
    hipLaunchKernelGGL(migrate_fill_kernel, dim3(1), dim3(MF_THREADS), 0, (hipStream_t)stream, *args);
    return hmd_launch_status();
    }

extern "C" int hmd_migrate_unpack(const hmd_migrate_unpack_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_migrate_unpack: args is NULL");
    if (args->n == 0)
        return HMD_OK;
    HMD_REQUIRE(pdata_arrays_complete(args->arrays) && args->d_in, HMD_ERR_INVALID_ARGUMENT, "hmd_migrate_unpack: all particle arrays and d_in are required");
    if (args->slot_rows)
        {
        HMD_REQUIRE(args->n_ranks >= 1 && args->n_ranks <= 64, HMD_ERR_INVALID_ARGUMENT, "hmd_migrate_unpack: fixed-size messages need 1 to 64 ranks");
        HMD_REQUIRE(args->recv_off[0] == 0 && args->recv_off[args->n_ranks] == args->n, HMD_ERR_INVALID_ARGUMENT,
                    "hmd_migrate_unpack: recv_off must start at 0 and end at n");
        for (unsigned int r = 0; r < args->n_ranks; r++)
            HMD_REQUIRE(args->recv_off[r] <= args->recv_off[r + 1] && args->recv_off[r + 1] - args->recv_off[r] <= args->slot_rows, HMD_ERR_INVALID_ARGUMENT,
                        "hmd_migrate_unpack: recv_off must not decrease, and no message holds more than slot_rows particles");
        }
    hipLaunchKernelGGL(migrate_unpack_kernel, dim3(hmd_div_up(args->n, 256)), dim3(256), 0, (hipStream_t)stream, *args);
    return hmd_launch_status();
    }

extern "C" int hmd_ghost_split_keys(const unsigned long long* d_keys, unsigned int n, unsigned int* d_send_idx, unsigned char* d_send_dir, void* stream)
    {
    if (n == 0)
        return HMD_OK;
    HMD_REQUIRE(d_keys && d_send_idx && d_send_dir, HMD_ERR_INVALID_ARGUMENT, "hmd_ghost_split_keys: all arrays are required");
    hipLaunchKernelGGL(split_keys_kernel, dim3(hmd_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, d_keys, n, d_send_idx, d_send_dir);
    return hmd_launch_status();
    }

extern "C" int hmd_pack_ghost_init(const hmd_pack_pos_args_t* args, const unsigned int* d_tag, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_pack_ghost_init: args is NULL");
    if (args->n_send == 0)
        return HMD_OK;
    HMD_REQUIRE(args->d_out && args->d_pos && args->d_send_idx && args->d_send_shift_id && args->d_shift_table && d_tag, HMD_ERR_INVALID_ARGUMENT,
                "hmd_pack_ghost_init: all array arguments are required");
    hipLaunchKernelGGL(pack_init_kernel, dim3(hmd_div_up(args->n_send, 256)), dim3(256), 0, (hipStream_t)stream, *args, d_tag);
    return hmd_launch_status();
    }

extern "C" int hmd_unpack_ghost_init(const hmd_unpack_ghost_init_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_unpack_ghost_init: args is NULL");
    if (args->n == 0)
        return HMD_OK;
    const hmd_pdata_arrays_t& p = args->arrays;
    HMD_REQUIRE(args->d_in && p.d_pos && p.d_image && p.d_tag && p.d_body && p.d_charge && p.d_diameter, HMD_ERR_INVALID_ARGUMENT,
                "hmd_unpack_ghost_init: d_in and the pos, image, tag, body, charge and diameter arrays are required");
    hipLaunchKernelGGL(unpack_init_kernel, dim3(hmd_div_up(args->n, 256)), dim3(256), 0, (hipStream_t)stream, *args);
    return hmd_launch_status();
    }

extern "C" int hmd_migrate_slot_headers(const long long* d_in, unsigned int slot_rows, unsigned int n_ranks, unsigned int* d_words, void* stream)
    {
    HMD_REQUIRE(d_in && d_words && slot_rows >= 1 && n_ranks >= 1 && n_ranks <= 64, HMD_ERR_INVALID_ARGUMENT,
                "hmd_migrate_slot_headers: d_in, d_words, slot_rows >= 1 and 1 to 64 ranks are required");
    hipLaunchKernelGGL(migrate_headers_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_in, slot_rows, n_ranks, d_words);
    return hmd_launch_status();
    }

static int ghost_slots_check(const hmd_ghost_slots_t* s, const char* who)
    {
    HMD_REQUIRE(s && s->n_ranks >= 1 && s->n_ranks <= HMD_GHOST_SLOTS_MAX_RANKS, HMD_ERR_INVALID_ARGUMENT, who);
    HMD_REQUIRE(s->off[0] == 0, HMD_ERR_INVALID_ARGUMENT, who);
    for (unsigned int r = 0; r < s->n_ranks; r++)
        {
        HMD_REQUIRE(s->off[r] <= s->off[r + 1], HMD_ERR_INVALID_ARGUMENT, who);
        // messages must not overlap in the buffer
        HMD_REQUIRE(r + 1 == s->n_ranks || s->row0[r] + s->cap[r] + 1 <= s->row0[r + 1], HMD_ERR_INVALID_ARGUMENT, who);
        }
    return HMD_OK;
    }

extern "C" int hmd_pack_ghost_init_slots(const hmd_pack_pos_args_t* args, const unsigned int* d_tag, const hmd_ghost_slots_t* slots, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_pack_ghost_init_slots: args is NULL");
    if (const int rc = ghost_slots_check(slots, "hmd_pack_ghost_init_slots: 1 to 64 ranks, offsets from 0 that do not decrease, messages that do not overlap"))
        return rc;
    HMD_REQUIRE(slots->off[slots->n_ranks] == args->n_send, HMD_ERR_INVALID_ARGUMENT, "hmd_pack_ghost_init_slots: the offsets must end at n_send");
    HMD_REQUIRE(args->d_out, HMD_ERR_INVALID_ARGUMENT, "hmd_pack_ghost_init_slots: d_out is required");
    HMD_REQUIRE(args->n_send == 0 || (args->d_pos && args->d_send_idx && args->d_send_shift_id && args->d_shift_table && d_tag), HMD_ERR_INVALID_ARGUMENT,
                "hmd_pack_ghost_init_slots: all array arguments are required");
    hipLaunchKernelGGL(pack_init_slots_kernel, dim3(hmd_div_up(args->n_send + slots->n_ranks, 256)), dim3(256), 0, (hipStream_t)stream, *args, d_tag, *slots);
    return hmd_launch_status();
    }

extern "C" int hmd_ghost_slot_headers(const hmd_scalar4* d_in, const hmd_ghost_slots_t* slots, unsigned int* d_words, void* stream)
    {
    HMD_REQUIRE(d_in && d_words && slots && slots->n_ranks >= 1 && slots->n_ranks <= HMD_GHOST_SLOTS_MAX_RANKS, HMD_ERR_INVALID_ARGUMENT,
                "hmd_ghost_slot_headers: d_in, d_words and 1 to 64 ranks are required");
    hipLaunchKernelGGL(slot_headers_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_in, *slots, d_words);
    return hmd_launch_status();
    }

extern "C" int hmd_unpack_ghost_init_slots(const hmd_unpack_ghost_init_args_t* args, const hmd_ghost_slots_t* slots, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_unpack_ghost_init_slots: args is NULL");
    if (const int rc = ghost_slots_check(slots, "hmd_unpack_ghost_init_slots: 1 to 64 ranks, offsets from 0 that do not decrease, messages that do not overlap"))
        return rc;
    HMD_REQUIRE(slots->off[slots->n_ranks] == args->n, HMD_ERR_INVALID_ARGUMENT, "hmd_unpack_ghost_init_slots: the offsets must end at n");
    for (unsigned int r = 0; r < slots->n_ranks; r++)
        HMD_REQUIRE(slots->off[r + 1] - slots->off[r] <= slots->cap[r], HMD_ERR_INVALID_ARGUMENT,
                    "hmd_unpack_ghost_init_slots: a message cannot hold more rows than it has room for");
    if (args->n == 0)
        return HMD_OK;
    const hmd_pdata_arrays_t& p = args->arrays;
    HMD_REQUIRE(args->d_in && p.d_pos && p.d_image && p.d_tag && p.d_body && p.d_charge && p.d_diameter, HMD_ERR_INVALID_ARGUMENT,
                "hmd_unpack_ghost_init_slots: d_in and the pos, image, tag, body, charge and diameter arrays are required");
    hipLaunchKernelGGL(unpack_init_slots_kernel, dim3(hmd_div_up(args->n, 256)), dim3(256), 0, (hipStream_t)stream, *args, *slots);
    return hmd_launch_status();
    }

extern "C" int hmd_migrate_pack_slots_dev(const hmd_migrate_pack_args_t* args, const unsigned int* d_n_leave, const unsigned int* d_rank_count,
                                          unsigned int capacity, void* stream)
    {
    HMD_REQUIRE(args && d_n_leave && d_rank_count, HMD_ERR_INVALID_ARGUMENT, "hmd_migrate_pack_slots_dev: args, d_n_leave and d_rank_count are required");
    HMD_REQUIRE(args->slot_rows >= 1 && args->n_ranks >= 1 && args->n_ranks <= 64 && args->d_out, HMD_ERR_INVALID_ARGUMENT,
                "hmd_migrate_pack_slots_dev: fixed-size messages (slot_rows >= 1) for 1 to 64 ranks and d_out are required");
    HMD_REQUIRE(capacity == 0 || (pdata_arrays_complete(args->arrays) && args->d_keys), HMD_ERR_INVALID_ARGUMENT,
                "hmd_migrate_pack_slots_dev: all particle arrays and d_keys are required");
    for (int d = 0; d < 3; d++)
        HMD_REQUIRE(args->global_L[d] > 0.0, HMD_ERR_INVALID_ARGUMENT, "hmd_migrate_pack_slots_dev: empty box");
    hipLaunchKernelGGL(migrate_pack_dev_kernel, dim3(hmd_div_up(std::max(capacity, args->n_ranks), 256)), dim3(256), 0, (hipStream_t)stream, *args, d_n_leave,
                       d_rank_count, capacity);
    return hmd_launch_status();
    }

extern "C" int hmd_comm_prep(unsigned int* d_counters, unsigned int n_counters, unsigned long long* d_keys, unsigned int n_keys, void* stream)
    {
    HMD_REQUIRE((n_counters == 0 || d_counters) && (n_keys == 0 || d_keys), HMD_ERR_INVALID_ARGUMENT, "hmd_comm_prep: an array is NULL");
    const unsigned int n = std::max(n_counters, n_keys);
    if (n == 0)
        return HMD_OK;
    hipLaunchKernelGGL(comm_prep_kernel, dim3(hmd_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, d_counters, n_counters, d_keys, n_keys);
    return hmd_launch_status();
    }

extern "C" int hmd_migrate_slot_headers_and(const long long* d_in, unsigned int slot_rows, unsigned int n_ranks, const unsigned int* d_extra,
                                            unsigned int n_extra, unsigned int* d_words, unsigned long long* d_published, unsigned int stamp, void* stream)
    {
    HMD_REQUIRE(d_in && d_words && slot_rows >= 1 && n_ranks >= 1 && n_ranks <= 64, HMD_ERR_INVALID_ARGUMENT,
                "hmd_migrate_slot_headers_and: d_in, d_words, slot_rows >= 1 and 1 to 64 ranks are required");
    HMD_REQUIRE(n_extra <= 64 && (n_extra == 0 || d_extra), HMD_ERR_INVALID_ARGUMENT, "hmd_migrate_slot_headers_and: at most 64 extra words");
    hipLaunchKernelGGL(migrate_headers_and_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_in, slot_rows, n_ranks, d_extra, n_extra, d_words, d_published,
                       stamp);
    return hmd_launch_status();
    }

extern "C" int hmd_pack_ghost_init_slots_dev(const hmd_pack_pos_args_t* args, const unsigned int* d_tag, const hmd_ghost_slots_t* slots,
                                             const unsigned int* d_total, const unsigned int* d_dir_count, const unsigned int* d_dir_dest, unsigned int n_dir,
                                             unsigned int capacity, const unsigned long long* d_sorted_keys, void* stream)
    {
    HMD_REQUIRE(args && slots && d_total, HMD_ERR_INVALID_ARGUMENT, "hmd_pack_ghost_init_slots_dev: args, slots and d_total are required");
    HMD_REQUIRE(slots->n_ranks >= 1 && slots->n_ranks <= HMD_GHOST_SLOTS_MAX_RANKS, HMD_ERR_INVALID_ARGUMENT, "hmd_pack_ghost_init_slots_dev: 1 to 64 ranks");
    for (unsigned int r = 0; r + 1 < slots->n_ranks; r++)
        HMD_REQUIRE(slots->row0[r] + slots->cap[r] + 1 <= slots->row0[r + 1], HMD_ERR_INVALID_ARGUMENT,
                    "hmd_pack_ghost_init_slots_dev: messages must not overlap in the buffer");
    HMD_REQUIRE(n_dir <= 26 && (n_dir == 0 || (d_dir_count && d_dir_dest)), HMD_ERR_INVALID_ARGUMENT,
                "hmd_pack_ghost_init_slots_dev: at most 26 directions, with their counts and destination ranks");
    HMD_REQUIRE(args->d_out, HMD_ERR_INVALID_ARGUMENT, "hmd_pack_ghost_init_slots_dev: d_out is required");
    HMD_REQUIRE(capacity == 0 || (args->d_pos && args->d_send_idx && args->d_send_shift_id && args->d_shift_table && d_tag), HMD_ERR_INVALID_ARGUMENT,
                "hmd_pack_ghost_init_slots_dev: all array arguments are required");
    hipLaunchKernelGGL(pack_init_slots_dev_kernel, dim3(hmd_div_up(std::max(capacity, slots->n_ranks), 256)), dim3(256), 0, (hipStream_t)stream, *args, d_tag,
                       *slots, d_total, d_dir_count, d_dir_dest, n_dir, capacity, d_sorted_keys, const_cast<unsigned int*>(args->d_send_idx),
                       const_cast<unsigned char*>(args->d_send_shift_id));
    return hmd_launch_status();
    }

extern "C" int hmd_ghost_slot_headers_and(const hmd_scalar4* d_in, const hmd_ghost_slots_t* slots, const unsigned int* d_extra, unsigned int n_extra,
                                          unsigned int* d_words, unsigned long long* d_published, unsigned int stamp, void* stream)
    {
    HMD_REQUIRE(d_in && d_words && slots && slots->n_ranks >= 1 && slots->n_ranks <= HMD_GHOST_SLOTS_MAX_RANKS, HMD_ERR_INVALID_ARGUMENT,
                "hmd_ghost_slot_headers_and: d_in, d_words and 1 to 64 ranks are required");
    HMD_REQUIRE(n_extra <= 64 && (n_extra == 0 || d_extra), HMD_ERR_INVALID_ARGUMENT, "hmd_ghost_slot_headers_and: at most 64 extra words");
    hipLaunchKernelGGL(slot_headers_and_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_in, *slots, d_extra, n_extra, d_words, d_published, stamp);
    return hmd_launch_status();
    }

extern "C" int hmd_vote_plan(const hmd_vote_plan_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_vote_plan: args is NULL");
    HMD_REQUIRE(args->n_ranks >= 1 && args->n_ranks <= HMD_VOTE_PLAN_MAX_RANKS && args->rank < args->n_ranks, HMD_ERR_INVALID_ARGUMENT,
                "hmd_vote_plan: between 1 and 64 ranks");
    for (unsigned int r = 0; r < args->n_ranks; r++)
        HMD_REQUIRE(args->send_off[r] <= args->send_off[r + 1] && args->recv_off[r] <= args->recv_off[r + 1], HMD_ERR_INVALID_ARGUMENT,
                    "hmd_vote_plan: offsets must not decrease");
    HMD_REQUIRE(args->send_off[0] == 0 && args->recv_off[0] == 0, HMD_ERR_INVALID_ARGUMENT, "hmd_vote_plan: offsets start at 0");
    HMD_REQUIRE(args->d_out_row && args->d_flag_rows && args->d_in_map, HMD_ERR_INVALID_ARGUMENT, "hmd_vote_plan: all outputs are required");
    const unsigned int n = std::max(std::max(args->send_off[args->n_ranks], args->recv_off[args->n_ranks] + args->n_ranks - 1), args->n_ranks);
    hipLaunchKernelGGL(vote_plan_kernel, dim3(hmd_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, *args);
    return hmd_launch_status();
    }

// ---- does a stream run beside the default stream? ----
// HIP multiplexes its streams over a handful of hardware queues; a created stream that happens to share the default
// stream's queue runs strictly behind whatever the default stream holds, however small its own kernels are.  The probe
// puts a long kernel on the default stream (LDS-bound occupancy like the force kernel: half of every CU's wave slots
// stay free) and a one-wave kernel on the candidate, queued behind an event recorded just before the long one: on a
// queue of its own the small kernel is done after a few microseconds, on the shared queue only after the long one.
namespace
    {
__global__ void __launch_bounds__(256) probe_busy_kernel(unsigned long long* d_out, unsigned long long ticks)
    {
    extern __shared__ unsigned long long s_probe[];
    s_probe[threadIdx.x] = ticks;
    const unsigned long long t0 = wall_clock64();
    unsigned long long t = t0;
    while (t - t0 < ticks) // wall_clock64 counts at 100 MHz: every wave leaves after `ticks` * 10 ns
        t = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0)
        d_out[0] = t + s_probe[0];
    }
__global__ void probe_tiny_kernel(unsigned long long* d_out)
    {
    d_out[0] = wall_clock64();
    }
    } // namespace

extern "C" int hmd_stream_runs_beside_default(void* stream, unsigned long long* d_scratch2, int* overlaps)
    {
    HMD_REQUIRE(stream && d_scratch2 && overlaps, HMD_ERR_INVALID_ARGUMENT, "hmd_stream_runs_beside_default: a created stream, two device words and overlaps are required");
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t b0 = nullptr, b1 = nullptr, t1 = nullptr, ready = nullptr;
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    step(hipEventCreate(&b0));
    step(hipEventCreate(&b1));
    step(hipEventCreate(&t1));
    step(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    int cus = 256;
    step(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    if (e == hipSuccess)
        {
        step(hipDeviceSynchronize());
        hipLaunchKernelGGL(probe_tiny_kernel, dim3(1), dim3(64), 0, nullptr, d_scratch2);
        step(hipEventRecord(ready, nullptr));
        step(hipEventRecord(b0, nullptr));
        // two generations of four workgroups per CU, 60 us each
        hipLaunchKernelGGL(probe_busy_kernel, dim3((unsigned int)cus * 8u), dim3(256), 36 * 1024, nullptr, d_scratch2, 6000ull);
        step(hipEventRecord(b1, nullptr));
        step(hipStreamWaitEvent(s, ready, 0));
        hipLaunchKernelGGL(probe_tiny_kernel, dim3(1), dim3(64), 0, s, d_scratch2 + 1);
        step(hipEventRecord(t1, s));
        step(hipStreamSynchronize(s));
        step(hipDeviceSynchronize());
        float long_ms = 0.f, tiny_ms = 0.f;
        step(hipEventElapsedTime(&long_ms, b0, b1));
        step(hipEventElapsedTime(&tiny_ms, b0, t1));
        *overlaps = (e == hipSuccess && tiny_ms < 0.5f * long_ms) ? 1 : 0;
        }
    for (hipEvent_t ev : {b0, b1, t1, ready})
        if (ev)
            (void)hipEventDestroy(ev);
    return e == hipSuccess ? hmd_launch_status() : (int)e;
    }
