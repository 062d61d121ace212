// tile.cuh -- geometry and LDS staging shared by the tile kernels (neighbor-list build from tiles,
// expansion of tile rows into particle indices, pair forces from tiles).
//
// The cell grid of the cell list (hoomd/CellList.cc:83-131) is cut into blocks of 2x2x2 cells.  A
// workgroup owns one block: its "home" cells.  Every neighbor of a home particle lies in the 4x4x4
// region of cells around the block, so the workgroup copies that region into LDS once -- periodic
// images shifted by the box vectors up front -- and all pair work afterwards reads LDS: no
// minimum-image arithmetic per pair (hoomd/BoxDim.h:306-325) and no scattered global loads.
//
// Region cell rc = (rz*4 + ry)*4 + rx (for a 4x4x4 region) is grid cell (2*bx - 1 + rx, ...), wrapped
// along periodic directions (needs >= 3 cells there: then the 3x3x3 neighborhood of any home cell
// never holds two images of one cell) and empty outside the grid otherwise.  Candidate numbers are
// off[rc] + slot: the exclusive prefix sum of the region's cell sizes plus the slot in the cell list.
#ifndef HMD_TILE_CUH
#define HMD_TILE_CUH

#include "common.cuh"

namespace hmd
    {
constexpr int TL_THREADS = 256;
// home cells per block along x, y, z and the region around them (one cell wider on every side)
#ifndef HMD_TILE_BZ
#define HMD_TILE_BZ 2
#endif
constexpr int TB_X = 2, TB_Y = 2, TB_Z = HMD_TILE_BZ;
constexpr int TR_X = TB_X + 2, TR_Y = TB_Y + 2, TR_Z = TB_Z + 2;
constexpr int TL_RC = TR_X * TR_Y * TR_Z; // region cells, at most 64 (one wave scans them)
constexpr int TL_NH = TB_X * TB_Y * TB_Z; // home cells
static_assert(TL_RC <= 64, "the region must fit one wave");

// region cell number of region coordinates, and of home cell hc = (hz*TB_Y + hy)*TB_X + hx
__host__ __device__ constexpr int tile_rc(int rx, int ry, int rz)
    {
    return (rz * TR_Y + ry) * TR_X + rx;
    }
__host__ __device__ constexpr int tile_home_rc(int hc)
    {
    return tile_rc(hc % TB_X + 1, (hc / TB_X) % TB_Y + 1, hc / (TB_X * TB_Y) + 1);
    }

struct tile_tables
    {
    unsigned int off[65];        // exclusive prefix of the region's cell sizes; off[TL_RC..64] = total
    int cell[TL_RC];             // grid cell of a region cell, -1 when outside a non-periodic grid
    // centre of the region cell in fractional box coordinates, from its UNWRAPPED grid coordinate
    // (below 0 or above 1 for a periodic image): a member at fraction f is seen in image rint(fc - f)
    double fcx[TL_RC], fcy[TL_RC], fcz[TL_RC];
    };

__device__ __forceinline__ unsigned int tile_xcd_remap(unsigned int b, unsigned int nb)
    {
    const unsigned int q = nb / 8, r = nb % 8, x = b % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
    }

static inline unsigned int tile_num_blocks(const hmd_index3d_t& ci)
    {
    return ((ci.w + TB_X - 1) / TB_X) * ((ci.h + TB_Y - 1) / TB_Y) * ((ci.d + TB_Z - 1) / TB_Z);
    }

// Region cell t (0..63) of the block whose first home cell is (cx0,cy0,cz0): its grid cell, or -1
// when it lies outside a non-periodic grid, and the periodic image it is seen in.
__device__ __forceinline__ int tile_region_cell(const hmd_index3d_t& ci, const hmd_box_t& box, int cx0, int cy0, int cz0, int t, int (&img)[3])
    {
    int c[3] = {cx0 - 1 + t % TR_X, cy0 - 1 + (t / TR_X) % TR_Y, cz0 - 1 + t / (TR_X * TR_Y)};
    const int dim[3] = {(int)ci.w, (int)ci.h, (int)ci.d};
    bool valid = true;
#pragma unroll
    for (int d = 0; d < 3; d++)
        {
        img[d] = 0;
        if (c[d] < 0)
            {
            if (box.periodic[d])
                {
                c[d] += dim[d];
                img[d] = -1;
                }
            else
                valid = false;
            }
        else if (c[d] >= dim[d])
            {
            if (box.periodic[d])
                {
                c[d] -= dim[d];
                img[d] = 1;
                }
            else
                valid = false;
            }
        // a region wider than the grid would wrap twice: such a cell is no neighbor of any home cell
        if (c[d] < 0 || c[d] >= dim[d])
            valid = false;
        }
    return valid ? (c[2] * dim[1] + c[1]) * dim[0] + c[0] : -1;
    }

// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ unsigned int tile_wave_scan(unsigned int v, int lane)
    {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
        {
        const unsigned int u = __shfl_up(v, o);
        if (lane >= o)
            v += u;
        }
    return v;
    }

// Fills the tables (threads 0..63 of the workgroup) and returns the block's cell coordinates of its
// first home cell.  Callers __syncthreads() afterwards.
__device__ __forceinline__ void tile_setup(const hmd_tile_geom_t& g, const hmd_box_t& box, tile_tables& T, int& cx0, int& cy0, int& cz0)
    {
    const unsigned int nbx = (g.ci.w + TB_X - 1) / TB_X, nby = (g.ci.h + TB_Y - 1) / TB_Y;
    const unsigned int vb = tile_xcd_remap(blockIdx.x, gridDim.x);
    cx0 = TB_X * (int)(vb % nbx);
    cy0 = TB_Y * (int)((vb / nbx) % nby);
    cz0 = TB_Z * (int)(vb / (nbx * nby));
    if (threadIdx.x < 64)
        {
        const int t = (int)threadIdx.x;
        int img[3];
        const int cell = t < TL_RC ? tile_region_cell(g.ci, box, cx0, cy0, cz0, t, img) : -1;
        const unsigned int size = cell >= 0 ? min(g.d_cell_size[cell], g.cli.w) : 0u;
        const unsigned int incl = tile_wave_scan(size, t);
        T.off[t] = incl - size;
        if (t == 63)
            T.off[64] = incl;
        if (t < TL_RC)
            {
            T.fcx[t] = (cx0 - 1 + t % TR_X + 0.5) / (double)g.ci.w;
            T.fcy[t] = (cy0 - 1 + (t / TR_X) % TR_Y + 0.5) / (double)g.ci.h;
            T.fcz[t] = (cz0 - 1 + t / (TR_X * TR_Y) + 0.5) / (double)g.ci.d;
            T.cell[t] = cell;
            }
        }
    }

// Coordinates of a member of region cell rc as the home cells see it: the periodic image nearest to
// the region cell.  Decided per particle, not per cell: between two builds the integrator wraps a
// particle that leaves the box, which moves it by a box vector without changing its cell-list slot.
// image = r + ix*a1 + iy*a2 + iz*a3, a1 = (Lx,0,0), a2 = (xy*Ly, Ly, 0), a3 = (xz*Lz, yz*Lz, Lz)
// (hoomd/BoxDim.h:391-458).
__device__ __forceinline__ void tile_place(const hmd_box_t& box, const tile_tables& T, unsigned int rc, double& x, double& y, double& z)
    {
    hmd_scalar3 zero;
    zero.x = zero.y = zero.z = 0.0;
    const double3 f = hmd_make_fraction(box, x, y, z, zero);
    const double ix = box.periodic[0] ? rint(T.fcx[rc] - f.x) : 0.0;
    const double iy = box.periodic[1] ? rint(T.fcy[rc] - f.y) : 0.0;
    const double iz = box.periodic[2] ? rint(T.fcz[rc] - f.z) : 0.0;
    x += ix * box.L[0] + iy * (box.L[1] * box.xy) + iz * (box.L[2] * box.xz);
    y += iy * box.L[1] + iz * (box.L[2] * box.yz);
    z += iz * box.L[2];
    }

// region cell that holds candidate v (off[rc] <= v < off[rc+1]); empty cells are skipped
__device__ __forceinline__ unsigned int tile_find_cell(const tile_tables& T, unsigned int v)
    {
    // off[] has 65 entries whatever TL_RC is; the tail repeats the total, so v < total never lands there
    unsigned int lo = 0;
#pragma unroll
    for (unsigned int s = 32; s > 0; s >>= 1)
        if (T.off[lo + s] <= v)
            lo += s;
    return lo;
    }

// is region cell (rx,ry,rz) a home cell of a block whose first home cell is (cx0,cy0,cz0)?
__device__ __forceinline__ bool tile_is_home(const hmd_tile_geom_t& g, int cx0, int cy0, int cz0, int hx, int hy, int hz)
    {
    return cx0 + hx < (int)g.ci.w && cy0 + hy < (int)g.ci.h && cz0 + hz < (int)g.ci.d;
    }

// Copies the region's members into LDS.  tx/ty/tz: shifted coordinates; tidx: particle index
// (optional); ttype: type id (optional).
template<bool WITH_POS, bool WITH_IDX, bool WITH_TYPE>
__device__ __forceinline__ void tile_stage(const hmd_tile_geom_t& g, const hmd_box_t& box, const tile_tables& T, const hmd_scalar4* d_pos,
                                           double* tx, double* ty, double* tz, unsigned int* tidx, unsigned char* ttype)
    {
    const unsigned int total = T.off[64];
    for (unsigned int v = threadIdx.x; v < total; v += TL_THREADS)
        {
        const unsigned int rc = tile_find_cell(T, v);
        const unsigned int idx = g.d_cell_idx[(unsigned int)T.cell[rc] * g.cli.w + (v - T.off[rc])];
        if (WITH_IDX)
            tidx[v] = idx;
        if (WITH_POS || WITH_TYPE)
            {
            const double4 p = hmd_ld4(d_pos + idx);
            if (WITH_POS)
                {
                double x = p.x, y = p.y, z = p.z;
                tile_place(box, T, rc, x, y, z);
                tx[v] = x;
                ty[v] = y;
                tz[v] = z;
                }
            if (WITH_TYPE)
                ttype[v] = (unsigned char)hmd_double_as_uint(p.w);
            }
        }
    }

// Largest dynamic LDS a tile kernel may ask for (gfx950 has 160 KiB per CU).
constexpr size_t TL_MAX_LDS = 128 * 1024;

template<class K> static inline void tile_allow_big_lds(K kernel, size_t bytes)
    {
    if (bytes > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    }
    } // namespace hmd

#endif
