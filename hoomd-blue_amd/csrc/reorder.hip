// reorder.hip -- cell-order permutation of the particle arrays (data locality for the pair-force
// gathers).  Stands in for the reference's space-filling-curve sorter kernels
// (hoomd/SFCPackTunerGPU.cu:26-160): instead of binning on a separate 256^3 grid and running a
// global key sort, the permutation is read off the cell list that the neighbor build already
// made -- an exclusive scan of the cell occupancies gives each cell's first new index.
// Results of this unit are compared bit for bit with the CPU path: no product may be fused into a
// following sum (the CPU path rounds every product and sum separately).
#pragma clang fp contract(off)
#include "common.cuh"

namespace
    {
constexpr unsigned int RO_BLOCK = 256;
constexpr unsigned int RO_ITEMS = 8;
constexpr unsigned int RO_TILE = RO_BLOCK * RO_ITEMS;

__device__ __forceinline__ unsigned int ro_block_scan(unsigned int v, unsigned int* s_wave, unsigned int& total)
    {
    const unsigned int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
        {
        const unsigned int t = __shfl_up(incl, off);
        if ((int)lane >= off)
            incl += t;
        }
    if (lane == 63)
        s_wave[wave] = incl;
    __syncthreads();
    unsigned int base = 0;
    total = 0;
    for (unsigned int w = 0; w < RO_BLOCK / 64; w++)
        {
        if (w < wave)
            base += s_wave[w];
        total += s_wave[w];
        }
    __syncthreads();
    return base + incl - v;
    }

// number of OWNED members (index < N) of a cell
__device__ __forceinline__ unsigned int owned_in_cell(const hmd_cell_order_args_t& a, unsigned int cell)
    {
    const unsigned int size = min(a.d_cell_size[cell], a.cli.w);
    unsigned int n = 0;
    for (unsigned int s = 0; s < size; s++)
        n += (hmd_double_as_uint(a.d_cell_xyzf[hmd_idx2(a.cli, s, cell)].w) < a.N) ? 1u : 0u;
    return n;
    }

__global__ void __launch_bounds__(RO_BLOCK) ro_tile_sums(hmd_cell_order_args_t a, unsigned int ncells, unsigned int* d_tile)
    {
    __shared__ unsigned int s_wave[RO_BLOCK / 64];
    unsigned int v = 0;
    const unsigned int base = blockIdx.x * RO_TILE + threadIdx.x * RO_ITEMS;
    for (unsigned int k = 0; k < RO_ITEMS; k++)
        if (base + k < ncells)
            {
            const unsigned int n = owned_in_cell(a, base + k);
            a.d_cell_offset[base + k] = n; // stash the count; overwritten by the offset below
            v += n;
            }
    unsigned int total;
    ro_block_scan(v, s_wave, total);
    if (threadIdx.x == 0)
        d_tile[blockIdx.x] = total;
    }

__global__ void __launch_bounds__(RO_BLOCK) ro_scan_tiles(unsigned int* d_tile, unsigned int ntiles)
    {
    __shared__ unsigned int s_wave[RO_BLOCK / 64];
    unsigned int carry = 0;
    for (unsigned int start = 0; start < ntiles; start += RO_BLOCK)
        {
        const unsigned int t = start + threadIdx.x;
        const unsigned int v = (t < ntiles) ? d_tile[t] : 0u;
        unsigned int total;
        const unsigned int ex = ro_block_scan(v, s_wave, total);
        if (t < ntiles)
            d_tile[t] = carry + ex;
        carry += total;
        }
    }

__global__ void __launch_bounds__(RO_BLOCK) ro_write_order(hmd_cell_order_args_t a, unsigned int ncells, const unsigned int* d_tile)
    {
    __shared__ unsigned int s_wave[RO_BLOCK / 64];
    unsigned int cnt[RO_ITEMS];
    unsigned int v = 0;
    const unsigned int base = blockIdx.x * RO_TILE + threadIdx.x * RO_ITEMS;
#pragma unroll
    for (unsigned int k = 0; k < RO_ITEMS; k++)
        {
        cnt[k] = (base + k < ncells) ? a.d_cell_offset[base + k] : 0u;
        v += cnt[k];
        }
    unsigned int total;
    unsigned int run = d_tile[blockIdx.x] + ro_block_scan(v, s_wave, total);
#pragma unroll
    for (unsigned int k = 0; k < RO_ITEMS; k++)
        {
        if (base + k < ncells)
            {
            const unsigned int cell = base + k;
            a.d_cell_offset[cell] = run;
            const unsigned int size = min(a.d_cell_size[cell], a.cli.w);
            unsigned int o = run;
            for (unsigned int s = 0; s < size; s++)
                {
                const unsigned int j = hmd_double_as_uint(a.d_cell_xyzf[hmd_idx2(a.cli, s, cell)].w);
                if (j < a.N)
                    a.d_order[o++] = j;
                }
            }
        run += cnt[k];
        }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == RO_BLOCK - 1)
        a.d_cell_offset[ncells] = run;
    }

template<typename T> __global__ void __launch_bounds__(256) gather_kernel(T* out, const T* in, const unsigned int* order, unsigned int N)
    {
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < N)
        out[k] = in[order[k]];
    }

struct b12
    {
    unsigned int a, b, c;
    };
struct b24
    {
    double a, b, c;
    };

__global__ void __launch_bounds__(256) rtag_kernel(unsigned int* d_rtag, const unsigned int* d_tag, unsigned int N)
    {
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < N)
        d_rtag[d_tag[k]] = k;
    }
    } // namespace

extern "C" size_t hmd_cell_order_workspace_bytes(unsigned int ncells)
    {
    return sizeof(unsigned int) * (size_t)(hmd_div_up(ncells, RO_TILE) + 1);
    }

extern "C" int hmd_cell_order(const hmd_cell_order_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_cell_order: args is NULL");
    const hmd_cell_order_args_t& a = *args;
    HMD_REQUIRE(a.d_order && a.d_cell_offset && a.d_cell_size && a.d_cell_xyzf && a.d_workspace, HMD_ERR_INVALID_ARGUMENT,
                "hmd_cell_order: order, cell_offset, cell_size, cell_xyzf and workspace are required");
    const unsigned int ncells = a.ci.w * a.ci.h * a.ci.d;
    HMD_REQUIRE(ncells > 0 && a.cli.h == ncells, HMD_ERR_INVALID_ARGUMENT, "hmd_cell_order: indexers disagree on the number of cells");
    HMD_REQUIRE(a.workspace_bytes >= hmd_cell_order_workspace_bytes(ncells), HMD_ERR_INVALID_ARGUMENT,
                "hmd_cell_order: workspace smaller than hmd_cell_order_workspace_bytes(ncells)");
    hipStream_t s = (hipStream_t)stream;
    unsigned int* tiles = reinterpret_cast<unsigned int*>(a.d_workspace);
    const unsigned int ntiles = hmd_div_up(ncells, RO_TILE);
    hipLaunchKernelGGL(ro_tile_sums, dim3(ntiles), dim3(RO_BLOCK), 0, s, a, ncells, tiles);
    hipLaunchKernelGGL(ro_scan_tiles, dim3(1), dim3(RO_BLOCK), 0, s, tiles, ntiles);
    hipLaunchKernelGGL(ro_write_order, dim3(ntiles), dim3(RO_BLOCK), 0, s, a, ncells, tiles);
    return hmd_launch_status();
    }

extern "C" int hmd_gather_rows(void* d_out, const void* d_in, const unsigned int* d_order, unsigned int N, unsigned int elem_bytes, void* stream)
    {
    HMD_REQUIRE(d_out && d_in && d_order, HMD_ERR_INVALID_ARGUMENT, "hmd_gather_rows: out, in and order are required");
    HMD_REQUIRE(d_out != d_in, HMD_ERR_INVALID_ARGUMENT, "hmd_gather_rows: in-place gathers are not supported");
    if (N == 0)
        return HMD_OK;
    const dim3 grid(hmd_div_up(N, 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (elem_bytes)
        {
    case 4:
        hipLaunchKernelGGL((gather_kernel<unsigned int>), grid, block, 0, s, (unsigned int*)d_out, (const unsigned int*)d_in, d_order, N);
        break;
    case 8:
        hipLaunchKernelGGL((gather_kernel<double>), grid, block, 0, s, (double*)d_out, (const double*)d_in, d_order, N);
        break;
    case 12:
        hipLaunchKernelGGL((gather_kernel<b12>), grid, block, 0, s, (b12*)d_out, (const b12*)d_in, d_order, N);
        break;
    case 16:
        hipLaunchKernelGGL((gather_kernel<double2>), grid, block, 0, s, (double2*)d_out, (const double2*)d_in, d_order, N);
        break;
    case 24:
        hipLaunchKernelGGL((gather_kernel<b24>), grid, block, 0, s, (b24*)d_out, (const b24*)d_in, d_order, N);
        break;
    case 32:
        hipLaunchKernelGGL((gather_kernel<double4>), grid, block, 0, s, (double4*)d_out, (const double4*)d_in, d_order, N);
        break;
    default:
        hmd_set_error("hmd_gather_rows: elem_bytes must be 4, 8, 12, 16, 24 or 32");
        return HMD_ERR_INVALID_ARGUMENT;
        }
    return hmd_launch_status();
    }

extern "C" int hmd_build_rtag(unsigned int* d_rtag, const unsigned int* d_tag, unsigned int N, void* stream)
    {
    HMD_REQUIRE(d_rtag && d_tag, HMD_ERR_INVALID_ARGUMENT, "hmd_build_rtag: rtag and tag are required");
    if (N == 0)
        return HMD_OK;
    hipLaunchKernelGGL(rtag_kernel, dim3(hmd_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, d_rtag, d_tag, N);
    return hmd_launch_status();
    }
