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

// Where candidate v lives in an LDS tile of doubles: blocks of 32 candidates, each block 32 x, then
// 32 y, then 32 z.  Lanes that read neighboring candidates hit neighboring banks, and one 16-bit
// number (tile rows store it) addresses all three coordinates: x at [s], y at [s + 32], z at [s + 64].
__host__ __device__ constexpr unsigned int tile_slot(unsigned int v)
    {
    return v + 64u * (v >> 5);
    }
__host__ __device__ constexpr unsigned int tile_unslot(unsigned int s)
    {
    return (s / 96u) * 32u + s % 96u;
    }
// doubles in a tile that holds candidates 0..cap (cap itself is the dummy entry)
__host__ __device__ constexpr unsigned int tile_doubles(unsigned int cap)
    {
    return (cap / 32u + 1u) * 96u;
    }

// Row storage: rows of TL_GROUP home particles are interleaved in pieces of TL_PIECE entries (32 bytes), so the
// TL_GROUP * TL_PIECE / 4 = 64 lanes of a wave -- four per home particle, four entries per lane -- read 512
// consecutive bytes per trip.  Entry k of the member `g` of a group sits at this 16-bit offset from the group's
// first entry when rows have room for P pieces.
constexpr unsigned int TL_GROUP = 16, TL_PIECE = 16;
__host__ __device__ constexpr size_t tile_row_entry(unsigned int g, unsigned int k)
    {
    return ((size_t)(k / TL_PIECE) * TL_GROUP + g) * TL_PIECE + k % TL_PIECE;
    }
__host__ __device__ constexpr size_t tile_group_entries(unsigned int P)
    {
    return (size_t)P * TL_GROUP * TL_PIECE;
    }
// pieces past the last group that the force kernel's prefetch may read
constexpr unsigned int TL_SLACK_GROUPS = 2;
// Rows built for more than one particle type carry the candidate's type in the top bits of the 16-bit entry (slot in
// the low 13 bits): the force kernel then needs no table of types beside the tile, which is 6 KB of LDS at a capacity
// of 2000 candidates and decides how many workgroups a compute unit holds.  Such rows need tile_slot(capacity) < 8192
// and at most eight types (hmd_nlist_tiles_supported says so).
constexpr unsigned int TL_TYPE_SHIFT = 13, TL_SLOT_MASK = (1u << TL_TYPE_SHIFT) - 1u, TL_MAX_ROW_TYPES = 1u << (16u - TL_TYPE_SHIFT);

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
    // Per axis d and region coordinate r (rc = (rz*TR_Y + ry)*TR_X + rx):
    // fc: centre of the region cell in fractional box coordinates, from its UNWRAPPED grid coordinate
    // (below 0 or above 1 for a periodic image): a member at fraction f is seen in image rint(fc - f);
    // img: the image a member that has not crossed the box boundary since binning is seen in.
    double fc[3][4], img[3][4];
    };
static_assert(TR_X <= 4 && TR_Y <= 4 && TR_Z <= 4, "tile_tables holds four region coordinates per axis");

__device__ __forceinline__ void tile_rc_coords(unsigned int rc, unsigned int& rx, unsigned int& ry, unsigned int& rz)
    {
    rx = rc % TR_X;
    ry = (rc / TR_X) % TR_Y;
    rz = rc / (TR_X * TR_Y);
    }

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
// vb: number of the block in the grid of blocks (x fastest)
__device__ __forceinline__ void tile_setup(const hmd_tile_geom_t& g, const hmd_box_t& box, tile_tables& T, int& cx0, int& cy0, int& cz0, unsigned int vb)
    {
    const unsigned int nbx = (g.ci.w + TB_X - 1) / TB_X, nby = (g.ci.h + TB_Y - 1) / TB_Y;
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
            T.cell[t] = cell;
        if (t < 12)
            {
            const int d = t / 4, r = t % 4;
            const int c = (d == 0 ? cx0 : d == 1 ? cy0 : cz0) - 1 + r;
            const int dim = (int)(d == 0 ? g.ci.w : d == 1 ? g.ci.h : g.ci.d);
            T.fc[d][r] = (c + 0.5) / (double)dim;
            T.img[d][r] = !box.periodic[d] ? 0.0 : c < 0 ? -1.0 : c >= dim ? 1.0 : 0.0;
            }
        }
    }

__device__ __forceinline__ void tile_setup(const hmd_tile_geom_t& g, const hmd_box_t& box, tile_tables& T, int& cx0, int& cy0, int& cz0)
    {
    tile_setup(g, box, T, cx0, cy0, cz0, tile_xcd_remap(blockIdx.x, gridDim.x));
    }

// Coordinates of a member of region cell rc as the home cells see it: the periodic image nearest to
// the region cell.  Decided per particle, not per cell: between two builds the integrator wraps a
// particle that leaves the box, which moves it by a box vector without changing its cell-list slot.
// image = r + ix*a1 + iy*a2 + iz*a3, a1 = (Lx,0,0), a2 = (xy*Ly, Ly, 0), a3 = (xz*Lz, yz*Lz, Lz)
// (hoomd/BoxDim.h:391-458).
__device__ __forceinline__ void tile_shift(const hmd_box_t& box, double ix, double iy, double iz, double& x, double& y, double& z)
    {
    x += ix * box.L[0] + iy * (box.L[1] * box.xy) + iz * (box.L[2] * box.xz);
    y += iy * box.L[1] + iz * (box.L[2] * box.yz);
    z += iz * box.L[2];
    }

__device__ __forceinline__ void tile_place(const hmd_box_t& box, const tile_tables& T, unsigned int rc, double& x, double& y, double& z)
    {
    unsigned int rx, ry, rz;
    tile_rc_coords(rc, rx, ry, rz);
    // fractional coordinates (hoomd/BoxDim.h:260-268 without ghost width); only their rounding matters here
    const double fx = ((x - box.lo[0]) - ((box.xz - box.yz * box.xy) * z + box.xy * y)) * box.Linv[0];
    const double fy = ((y - box.lo[1]) - box.yz * z) * box.Linv[1];
    const double fz = (z - box.lo[2]) * box.Linv[2];
    const double ix = box.periodic[0] ? rint(T.fc[0][rx] - fx) : 0.0;
    const double iy = box.periodic[1] ? rint(T.fc[1][ry] - fy) : 0.0;
    const double iz = box.periodic[2] ? rint(T.fc[2][rz] - fz) : 0.0;
    tile_shift(box, ix, iy, iz, x, y, z);
    }

// What tile_place computes for every member of one region cell, split into the part that depends on
// the cell only.  A member that has not crossed the box boundary since binning sits in image
// T.img of the cell; that is also what tile_place returns for it whenever the shifted member lies
// within `reach` of the cell's centre, reach < half the smallest distance between opposite box faces:
// then every fractional coordinate of (member - centre) is below 1/2 in magnitude and rounds to zero.
struct tile_cell_image
    {
    double sx, sy, sz; // shift of the cell's image
    double qx, qy, qz; // centre of the region cell minus that shift
    };

__device__ __forceinline__ tile_cell_image tile_cell_setup(const hmd_box_t& box, const tile_tables& T, unsigned int rc)
    {
    unsigned int rx, ry, rz;
    tile_rc_coords(rc, rx, ry, rz);
    tile_cell_image c;
    c.sx = c.sy = c.sz = 0.0;
    tile_shift(box, T.img[0][rx], T.img[1][ry], T.img[2][rz], c.sx, c.sy, c.sz);
    // centre in Cartesian coordinates (hoomd/BoxDim.h:282-288)
    const double vz = box.lo[2] + T.fc[2][rz] * box.L[2];
    const double vy = box.lo[1] + T.fc[1][ry] * box.L[1];
    c.qx = box.lo[0] + T.fc[0][rx] * box.L[0] + (box.xy * vy + box.xz * vz);
    c.qy = vy + box.yz * vz;
    c.qz = vz;
    c.qx -= c.sx;
    c.qy -= c.sy;
    c.qz -= c.sz;
    return c;
    }

// squared reach for tile_cell_image: (0.49 * smallest periodic face distance)^2
__device__ __forceinline__ double tile_reach_sq(const hmd_box_t& box)
    {
    double m = 1e300;
#pragma unroll
    for (int d = 0; d < 3; d++)
        if (box.periodic[d])
            m = fmin(m, box.nearest_plane[d]);
    return m >= 1e150 ? 1e300 : 0.2401 * m * m;
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
