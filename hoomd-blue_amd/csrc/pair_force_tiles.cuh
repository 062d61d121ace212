// pair_force_tiles.cuh -- pair forces from LDS tiles.
//
// Same per-pair arithmetic and outputs as pair_force.cuh (hoomd/md/PotentialPairGPU.cuh:140-404); the
// neighbor rows hold 16-bit tile slots of the workgroup's region (tile.cuh), so a neighbor's
// coordinates are three LDS reads and the periodic image is already applied.  Four lanes share a
// home particle and a wave works on the sixteen home particles of one group of rows at a time: rows are
// stored neighbor-major within a group (tile.cuh, TL_GROUP / TL_PIECE), so every trip of the wave is one
// 512-byte load, each lane taking four consecutive entries of its particle's row.  A wave's trip count is
// that of the longest of its sixteen rows, whatever the lengths of the others, and what is left per home
// particle is one read of the home table and a two-step reduction over four lanes.
// Row entries are tile slots (tile_slot() in tile.cuh): 8 * entry is the byte offset of x in the tile,
// y and z follow 256 and 512 bytes later.
#ifndef HMD_PAIR_FORCE_TILES_CUH
#define HMD_PAIR_FORCE_TILES_CUH

#include "pair_force.cuh"
#include "step_body.cuh"
#include "tile.cuh"
#include <cstring>

namespace hmd
    {
// per home particle, kept in LDS so that no wave waits on a chain of dependent global loads
struct pft_home_t
    {
    unsigned int idx;
    unsigned int nn_lh; // low half: number of neighbors; high half: tile slot of the particle itself
    };
static_assert(sizeof(pft_home_t) == 8, "LDS budget");
// (with several particle types lh carries the particle's type above the slot, like a row entry: tile.cuh, TL_TYPE_SHIFT)
__device__ __forceinline__ pft_home_t pft_home(unsigned int idx, unsigned int nn, unsigned int lh)
    {
    pft_home_t e = {idx, min(nn, 0xffffu) | (lh << 16)};
    return e;
    }

// bytes of the home table: one entry per slot of the home cells and an empty one, padded to 16
__host__ __device__ constexpr size_t pft_home_bytes(unsigned int cell_Nmax)
    {
    return (((size_t)TL_NH * cell_Nmax + 1) * sizeof(pft_home_t) + 15) / 16 * 16;
    }

typedef __attribute__((address_space(3))) const double pft_lds_double;

// position (and type id when asked for) of one slot of the slot-ordered position copy
struct pft_pos
    {
    double x, y, z;
    unsigned int t;
    };
template<bool WITH_TYPE> __device__ __forceinline__ pft_pos pft_ld_pos(const hmd_scalar4* p)
    {
    pft_pos r;
    const double2 xy = *reinterpret_cast<const double2*>(p);
    r.x = xy.x;
    r.y = xy.y;
    if (WITH_TYPE)
        {
        const double2 zw = reinterpret_cast<const double2*>(p)[1];
        r.z = zw.x;
        r.t = hmd_double_as_uint(zw.y);
        }
    else
        {
        r.z = reinterpret_cast<const double*>(p)[2];
        r.t = 0;
        }
    return r;
    }

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

constexpr int PFT_THREADS = 256;              // 4 waves share one staged region (8 where only two regions fit a CU's LDS)
constexpr int PFT_T = 4;                      // lanes per home particle: 64 / PFT_T = TL_GROUP home particles per wave
constexpr int PFT_PRE = 6;                    // pieces of every row (96 entries) fetched one group ahead
constexpr int PFT_TAIL = 2;                   // further pieces fetched at the start of a group by the lanes whose rows are that long
static_assert(64 / PFT_T == (int)TL_GROUP && PFT_T * 4 == (int)TL_PIECE, "a wave reads one piece of each of its sixteen rows per trip");
constexpr int PFT_WAVES = PFT_THREADS / 64;
constexpr int PFT_TRIPS = 4;
static_assert(TR_X == 4 && TR_Y == 4 && TR_Z == 4 && PFT_WAVES == 4 && TL_NH == 8, "staging maps (lane / 16, wave % 4, trip) to the cells of a 4x4x4 region");

// The integration pass that may ride on the kernel, in the form the kernel carries it: an orthorhombic box periodic
// in all three directions that has not changed since the neighbor list was built, no velocity rescaling -- the case
// of an NVE run (the launcher declines everything else).  The kernel argument is copied to LDS once per workgroup and
// read from there where a home particle is finished: as scalar registers these forty values would have to stay
// resident through the whole pair loop, which has none to spare.
struct pft_step_t
    {
    double dt;
    double lo[3], hi[3], L[3], Linv[3];
    double r_buff;
    hmd_scalar4* d_vel;
    hmd_scalar4* d_pos;
    hmd_int3* d_image;
    const unsigned int* d_slot_of;
    hmd_scalar4* d_cpos;
    const hmd_scalar4* d_last_pos;
    const double* d_rcut_max;
    unsigned int* d_result;
    unsigned int checkn, check;
    };

// can the pass ride?  (the conditions above; `why` receives the first one that fails)
static inline bool pft_step_supported(const hmd_pair_args_t& a, const hmd_fused_step_args_t& t, const char** why)
    {
    const char* w = nullptr;
    if (a.compute_virial || a.block_window_mode != 0)
        w = "not with virials or a window of blocks";
    else if (!(t.one.d_pos && t.one.d_vel && t.one.d_image && t.one.d_vel == t.two.d_vel))
        w = "one.d_pos, one.d_vel (= two.d_vel) and one.d_image are required";
    else if (t.one.d_group_members || t.two.d_group_members || t.one.group_size != a.N || t.two.group_size != a.N || t.one.use_limit)
        w = "the group must be every particle, without a displacement limit";
    else if (t.two.d_accel)
        w = "the accelerations are not stored (two.d_accel must be null)";
    else if (t.one.rescale_factor != 1.0 || t.one.d_rescale_ke || t.two.rescale_factor != 1.0 || t.one.deltaT != t.two.deltaT)
        w = "no velocity rescaling, one time step length";
    else if (!hmd_box_is_ortho(t.one.box) || !(t.one.box.periodic[0] && t.one.box.periodic[1] && t.one.box.periodic[2]))
        w = "the box must be orthorhombic and periodic in all directions";
    else if (t.check.d_result
             && (std::memcmp(&t.one.box, &t.check.box, sizeof(hmd_box_t)) != 0 || t.check.lambda.x != 1.0 || t.check.lambda.y != 1.0
                 || t.check.lambda.z != 1.0 || t.check.lambda_min != 1.0))
        w = "the box must be the one the neighbor list was built in";
    else if (t.check.d_result && !(t.check.d_last_pos && t.check.d_rcut_max && t.check.N == a.N))
        w = "the distance check needs d_last_pos, d_rcut_max and N";
    if (why)
        *why = w;
    return w == nullptr;
    }

// One home particle's share of the pass: hmd_fused_step_particle<true, true> (step_body.cuh) with the factors that are
// one and the tilts that are zero left out -- x * 1.0, x - 0.0 and x + 0.0 * y are exact, so the bits are the same.
typedef const __attribute__((address_space(3))) pft_step_t pft_step_lds;
typedef double pft_v4d __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) pft_v4d pft_g_double4;
typedef __attribute__((address_space(1))) int pft_g_int;
typedef __attribute__((address_space(1))) const unsigned int pft_g_cuint;
typedef __attribute__((address_space(1))) const double pft_g_cdouble;
// Everything the pass reads is fetched first, in one memory round trip (the index of the particle is all the addresses
// need; with one particle type so is the type), the arithmetic and the stores follow.
template<bool SINGLE_TYPE>
__device__ __forceinline__ bool pft_step_particle(pft_step_lds& S, unsigned int idx, double fx, double fy, double fz)
    {
#pragma clang fp contract(off)
    pft_g_double4* d_vel = (pft_g_double4*)S.d_vel;
    pft_g_double4* d_pos = (pft_g_double4*)S.d_pos;
    const bool check = S.check != 0u;
    const pft_v4d v = d_vel[idx];
    const pft_v4d p = d_pos[idx];
    const unsigned int slot = ((pft_g_cuint*)S.d_slot_of)[idx];
    pft_v4d last = {0.0, 0.0, 0.0, 0.0};
    double rmin = 0.0;
    if (check)
        {
        last = ((const pft_g_double4*)S.d_last_pos)[idx];
        if (SINGLE_TYPE)
            rmin = ((pft_g_cdouble*)S.d_rcut_max)[0];
        }
    const double dt = S.dt;
    // ---- step two (t) ----
    // (1 / 1 is 1: when every particle the wave is finishing has unit mass the division, a dozen instructions, is left out)
    double minv = 1.0;
    if (__builtin_amdgcn_readfirstlane((int)(__ballot(v.w != 1.0) != 0ull)))
        minv = 1.0 / v.w;
    const double ax = HMD_MUL(fx, minv), ay = HMD_MUL(fy, minv), az = HMD_MUL(fz, minv);
    const double hdt = HMD_MUL(0.5, dt);
    double vx = HMD_ADD(v.x, HMD_MUL(hdt, ax));
    double vy = HMD_ADD(v.y, HMD_MUL(hdt, ay));
    double vz = HMD_ADD(v.z, HMD_MUL(hdt, az));
    // ---- step one (t+1) ----
    vx = HMD_ADD(vx, HMD_MUL(HMD_MUL(0.5, ax), dt));
    vy = HMD_ADD(vy, HMD_MUL(HMD_MUL(0.5, ay), dt));
    vz = HMD_ADD(vz, HMD_MUL(HMD_MUL(0.5, az), dt));
    double x = HMD_ADD(p.x, HMD_MUL(dt, vx));
    double y = HMD_ADD(p.y, HMD_MUL(dt, vy));
    double z = HMD_ADD(p.z, HMD_MUL(dt, vz));
    // BoxDim::wrap (hoomd/BoxDim.h:391-458) without tilt
    int ix = 0, iy = 0, iz = 0;
        {
        const double L0 = S.L[0], L1 = S.L[1], L2 = S.L[2];
        if (x >= S.hi[0])
            {
            x = HMD_SUB(x, L0);
            ix++;
            }
        else if (x < S.lo[0])
            {
            x = HMD_ADD(x, L0);
            ix--;
            }
        if (y >= S.hi[1])
            {
            y = HMD_SUB(y, L1);
            iy++;
            }
        else if (y < S.lo[1])
            {
            y = HMD_ADD(y, L1);
            iy--;
            }
        if (z >= S.hi[2])
            {
            z = HMD_SUB(z, L2);
            iz++;
            }
        else if (z < S.lo[2])
            {
            z = HMD_ADD(z, L2);
            iz--;
            }
        }
    d_pos[idx] = pft_v4d {x, y, z, p.w};
    d_vel[idx] = pft_v4d {vx, vy, vz, v.w};
    ((pft_g_double4*)S.d_cpos)[slot] = pft_v4d {x, y, z, p.w};
    if (ix | iy | iz)
        {
        pft_g_int* d_image = (pft_g_int*)S.d_image + 3u * (size_t)idx;
        d_image[0] += ix;
        d_image[1] += iy;
        d_image[2] += iz;
        }
    // ---- distance check of the new position (hoomd/md/NeighborListGPU.cu:42-84) ----
    bool moved = false;
    if (check)
        {
        if (!SINGLE_TYPE)
            rmin = ((pft_g_cdouble*)S.d_rcut_max)[hmd_double_as_uint(p.w)];
        double dx = HMD_SUB(x, last.x), dy = HMD_SUB(y, last.y), dz = HMD_SUB(z, last.z);
        dz = HMD_SUB(dz, HMD_MUL(S.L[2], rint(HMD_MUL(dz, S.Linv[2]))));
        dy = HMD_SUB(dy, HMD_MUL(S.L[1], rint(HMD_MUL(dy, S.Linv[1]))));
        dx = HMD_SUB(dx, HMD_MUL(S.L[0], rint(HMD_MUL(dx, S.Linv[0]))));
        const double rmax = HMD_ADD(rmin, S.r_buff);
        const double delta_max = HMD_SUB(rmax, rmin) / 2.0;
        const double maxsq = (delta_max > 0) ? HMD_MUL(delta_max, delta_max) : 0.0;
        moved = HMD_ADD(HMD_ADD(HMD_MUL(dx, dx), HMD_MUL(dy, dy)), HMD_MUL(dz, dz)) >= maxsq;
        }
    return moved;
    }

// what the workgroup shares besides the tile and the home table
struct pft_tables
    {
    unsigned int off[65];         // exclusive prefix of the region's cell sizes; off[64] = total
    unsigned int hcum[TL_NH + 1]; // exclusive prefix of the sizes of the home cells, in home-cell order; hcum[8] = home particles
    unsigned int word;            // the predicate word, read once for the whole workgroup
    unsigned int pad_;
    pft_step_t step;              // what the integration pass needs, when it rides on the kernel
    };
constexpr size_t PFT_TABLES = (sizeof(pft_tables) + 15) / 16 * 16;

// launch helpers computed on the host (not part of the C ABI): the grid of blocks and reciprocals, so that no
// integer or fp64 division runs on the device (q = n * ceil(2^40 / d) >> 40 is exact for n < 2^24, d < 2^16)
struct pft_launch_t
    {
    unsigned int nbx, nby;
    unsigned long long inv_nbx, inv_nby;
    double inv_dim[3];
    };

// one axis of a region cell: grid coordinate (wrapped along a periodic axis), the image it is seen in, whether
// it exists, and whether a member may have been wrapped by the integrator since binning (only members of the
// first and last cells along a periodic axis can cross the box boundary between two builds)
struct pft_axis
    {
    int c, img;
    bool valid, edge;
    };
__device__ __forceinline__ pft_axis pft_axis_cell(int c, int dim, bool periodic)
    {
    pft_axis r = {c, 0, true, false};
    if (c < 0)
        {
        if (periodic)
            {
            r.c = c + dim;
            r.img = -1;
            }
        else
            r.valid = false;
        }
    else if (c >= dim)
        {
        if (periodic)
            {
            r.c = c - dim;
            r.img = 1;
            }
        else
            r.valid = false;
        }
    // a region wider than the grid would wrap twice: such a cell is no neighbor of any home cell
    if (r.c < 0 || r.c >= dim)
        r.valid = false;
    r.edge = periodic && (r.c == 0 || r.c == dim - 1);
    return r;
    }

// STEP: the integration pass rides on the kernel.  The lane that holds a home particle's finished force goes straight
// on with the second half of the time step, the first half of the next one, the distance check of the new position
// and the slot-ordered copy (step_body.cuh) -- the force never makes the round trip through memory to a separate
// pass, and a time step is this one kernel.  The new positions go to the particle array, which this kernel does not
// read, and to a second slot-ordered buffer (st.d_cpos must not be g.d_cpos: other workgroups are still staging from it).
// WAVES: 4, or 8 for regions so large that only two workgroups fit the LDS of a CU -- the same sixteen waves per CU then
// (the waves of the second half take the upper two layers of the region in the staging phase).
// BY_TYPE (two types): the rows are laid out by type (hmd_tile_geom_t::rows_by_type), every chunk of four entries has one
// type and its first entry tells which -- cutoff and parameters are chosen once per chunk, not once per pair.
template<class Ev, unsigned int SHIFT, bool VIRIAL, bool SINGLE_TYPE, bool TWO_TYPES = false, bool STEP = false, int WAVES = PFT_WAVES, bool BY_TYPE = false>
__global__ void __launch_bounds__(64 * WAVES, 4) // four waves per SIMD (128 registers): what the LDS tile allows anyway
    pair_force_tiles_kernel(hmd_pair_args_t a, hmd_tile_geom_t g, const unsigned short* d_tile_nlist, const typename Ev::param_type* d_params,
                            pft_launch_t lp, pft_step_t st)
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
    pft_tables& T = *reinterpret_cast<pft_tables*>(txyz + tile_len);
    // s_home: one entry per home particle and an empty one after the last
    pft_home_t* s_home = reinterpret_cast<pft_home_t*>(s_dyn + (size_t)tile_len * 8 + PFT_TABLES);
    if ((unsigned int)(size_t)(__attribute__((address_space(3))) void*)txyz != 0u)
        __builtin_trap(); // the toolchain placed something before the dynamic LDS
    // Launches over the whole grid are queued behind whatever writes the predicate word, on the same stream: the word
    // is final, every wave reads it for itself and a switched-off launch ends here, before any load is issued.
    if (a.d_skip_flag && a.block_window_mode == 0)
        {
        const unsigned int word = __builtin_amdgcn_readfirstlane(__hip_atomic_load(a.d_skip_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (word == a.skip_value)
            {
            if (a.d_published && blockIdx.x == 0 && threadIdx.x == 0)
                __hip_atomic_store(a.d_published, ((unsigned long long)a.publish_stamp << 32) | (unsigned long long)word, __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
            return;
            }
        }
    // (A window launch whose word already says "skip" when a wave first looks ends there as well, before any load: the
    // word only ever moves towards that value, so no later look can disagree.  Only a word that does not say so yet has
    // to be read once for the whole workgroup, below.)
    if (a.d_skip_flag && a.block_window_mode != 0)
        {
        const unsigned int word = __builtin_amdgcn_readfirstlane(__hip_atomic_load(a.d_skip_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (word == a.skip_value)
            {
            if (a.d_published && blockIdx.x == 0 && threadIdx.x == 0)
                __hip_atomic_store(a.d_published, ((unsigned long long)a.publish_stamp << 32) | (unsigned long long)word, __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
            return;
            }
        }
    // Launches over a window of blocks run beside the exchange: one thread reads the predicate word for the whole
    // workgroup (another stream may be raising it at this very moment: the rebuild vote of other ranks arriving; a
    // workgroup must not split over it); the others learn it behind the barrier that follows the set-up below.
    if (a.d_skip_flag && threadIdx.x == 0)
        {
        const unsigned int word = __hip_atomic_load(a.d_skip_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.d_published && blockIdx.x == 0)
            __hip_atomic_store(a.d_published, ((unsigned long long)a.publish_stamp << 32) | (unsigned long long)word, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        T.word = word;
        }
    if (STEP && threadIdx.x == 64)
        T.step = st;
    const unsigned int np = a.ntypes * a.ntypes;
    // (the 8-byte home entries end on a multiple of eight: pft_home_bytes pads to 16 for the tables of doubles behind them)
    double* s_rcutsq = reinterpret_cast<double*>(s_dyn + (size_t)tile_len * 8 + PFT_TABLES + pft_home_bytes(g.cli.w));
    double* s_ronsq = s_rcutsq + (SINGLE_TYPE ? 0 : np);
    param_t* s_params = reinterpret_cast<param_t*>(s_ronsq + (SINGLE_TYPE ? 0 : np));

    // ---- which block of the grid: all of them in launch order, or those inside / outside a window ----
    const unsigned int nbx = lp.nbx, nby = lp.nby;
    unsigned int bx, by, bz;
        {
        const unsigned int v0 = tile_xcd_remap(blockIdx.x, gridDim.x);
        if (a.block_window_mode == 0)
            {
            const unsigned int q = (unsigned int)(((unsigned long long)v0 * lp.inv_nbx) >> 40); // v0 / nbx
            bx = v0 - q * nbx;
            bz = (unsigned int)(((unsigned long long)q * lp.inv_nby) >> 40);
            by = q - bz * nby;
            }
        else if (a.block_window_mode == 1)
            {
            // the launch covers the window only
            const unsigned int wx = a.block_hi[0] - a.block_lo[0] + 1, wy = a.block_hi[1] - a.block_lo[1] + 1;
            bx = a.block_lo[0] + v0 % wx, by = a.block_lo[1] + (v0 / wx) % wy, bz = a.block_lo[2] + v0 / (wx * wy);
            }
        else
            {
            // the launch covers the blocks outside the window: six slabs, numbered one after the other
            const unsigned int nbz = (g.ci.d + TB_Z - 1) / TB_Z;
            const unsigned int wy = a.block_hi[1] - a.block_lo[1] + 1, wz = a.block_hi[2] - a.block_lo[2] + 1;
            unsigned int v = v0;
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
            }
        }
    const unsigned int vb = (bz * nby + by) * nbx + bx;
    const int cx0 = TB_X * (int)bx, cy0 = TB_Y * (int)by, cz0 = TB_Z * (int)bz;
    const int dimx = (int)g.ci.w, dimy = (int)g.ci.h, dimz = (int)g.ci.d;

    // ---- staging, first half: every load of the region is issued before anything is known about the
    // cells' populations.  Sixteen lanes take one region cell: rx = lane / 16, ry = wave, rz = trip, so a wave
    // covers a 4x1x4 slab of the region in four trips (of eight waves, wave and wave + 4 share a slab, two trips each); slots k16 and k16 + 16 of every cell are fetched
    // unconditionally (slots past a cell's population hold stale data that is never used), and with them the
    // row info of the home cells' members.  The cell sizes arrive in the same round trip (wave 0 below):
    // staging costs one memory latency, not three.
    const unsigned int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), lane = threadIdx.x % 64;
    // The staging phase runs at raised wave priority: its few instructions (addresses, loads, LDS stores) otherwise queue
    // behind the arithmetic of the three other workgroups on the compute unit, and a workgroup that is slow to get its
    // region in place holds a quarter of the unit's LDS and wave slots without computing (pair kernel 125 -> 119 us).
    __builtin_amdgcn_s_setprio(2);
    static_assert(WAVES == 4 || WAVES == 8, "four or eight waves");
    constexpr int TRIPS = 4 * PFT_WAVES / WAVES;
    const unsigned int rx = lane / 16, k16 = lane % 16;
    const unsigned int ry = WAVES == 4 ? wave : wave % 4u;
    const int u0 = WAVES == 4 ? 0 : 2 * (int)(wave / 4u); // first layer of the region this wave stages
    const unsigned int cellNmax = g.cli.w;
    const pft_axis ax = pft_axis_cell(cx0 - 1 + (int)rx, dimx, a.box.periodic[0]);
    const pft_axis ay = pft_axis_cell(cy0 - 1 + (int)ry, dimy, a.box.periodic[1]);
    const bool home_xy = (rx == 1 || rx == 2) && (ry == 1 || ry == 2) && cx0 + (int)rx - 1 < dimx && cy0 + (int)ry - 1 < dimy;
    // (layers 1 and 2 hold the home cells: with four trips per wave their row info goes to mA[0], mA[1]; with two trips
    // a wave meets one of them, mA[0])
    pft_pos pA[TRIPS], pB[TRIPS];
    uint2 mA[2], mB[2];
    pft_axis az[TRIPS];
        {
        // byte offsets fit 32 bits (checked by the launcher): one scalar base plus a 32-bit lane offset per load
        const unsigned int oA = min(k16, cellNmax - 1u) * 32u, oB = min(k16 + 16u, cellNmax - 1u) * 32u;
        const char* cpos = reinterpret_cast<const char*>(g.d_cpos);
        const char* meta = reinterpret_cast<const char*>(g.d_slot_meta);
#pragma unroll
        for (int t = 0; t < TRIPS; t++)
            {
            const int u = u0 + t;
            az[t] = pft_axis_cell(cz0 - 1 + u, dimz, a.box.periodic[2]);
            const bool valid = ax.valid && ay.valid && az[t].valid;
            const unsigned int base = valid ? (unsigned int)((az[t].c * dimy + ay.c) * dimx + ax.c) * cellNmax * 32u : 0u;
            pA[t] = pft_ld_pos<!SINGLE_TYPE>(reinterpret_cast<const hmd_scalar4*>(cpos + (base + oA)));
            pB[t] = pft_ld_pos<!SINGLE_TYPE>(reinterpret_cast<const hmd_scalar4*>(cpos + (base + oB)));
            if ((u == 1 || u == 2) && home_xy)
                {
                mA[WAVES == 4 ? t - 1 : 0] = *reinterpret_cast<const uint2*>(meta + ((base + oA) >> 1));
                mB[WAVES == 4 ? t - 1 : 0] = *reinterpret_cast<const uint2*>(meta + ((base + oB) >> 1));
                }
            }
        }
    // ---- wave 0: populations of the 64 region cells -> prefix sums; home cells -> numbering of the home particles ----
    if (wave == 0)
        {
        const pft_axis tx = pft_axis_cell(cx0 - 1 + (int)(lane % 4), dimx, a.box.periodic[0]);
        const pft_axis ty = pft_axis_cell(cy0 - 1 + (int)((lane / 4) % 4), dimy, a.box.periodic[1]);
        const pft_axis tz = pft_axis_cell(cz0 - 1 + (int)(lane / 16), dimz, a.box.periodic[2]);
        unsigned int size = 0;
        if (tx.valid && ty.valid && tz.valid)
            size = min(g.d_cell_size[(tz.c * dimy + ty.c) * dimx + tx.c], cellNmax);
        const unsigned int incl = tile_wave_scan(size, (int)lane);
        T.off[lane] = incl - size;
        if (lane == 63)
            T.off[64] = incl;
        // lane hc < 8: the size of home cell hc (region cell (1 + hx, 1 + hy, 1 + hz)) when the grid has it
        const unsigned int hc = lane % TL_NH;
        const int hx = hc % TB_X, hy = (hc / TB_X) % TB_Y, hz = hc / (TB_X * TB_Y);
        const unsigned int hsize = __shfl(size, tile_rc(1 + hx, 1 + hy, 1 + hz), 64);
        const bool home = lane < (unsigned int)TL_NH && cx0 + hx < dimx && cy0 + hy < dimy && cz0 + hz < dimz;
        const unsigned int hincl = tile_wave_scan(home ? hsize : 0u, (int)lane);
        if (lane < (unsigned int)TL_NH)
            T.hcum[lane + 1] = hincl;
        if (lane == 0)
            T.hcum[0] = 0;
        }
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
        for (unsigned int t = threadIdx.x; t < np; t += 64u * WAVES)
            {
            s_params[t] = d_params[t];
            s_rcutsq[t] = a.d_rcutsq[t];
            s_ronsq[t] = (SHIFT == 2) ? a.d_ronsq[t] : 0.0;
            }
    // first group of this block's rows (wave-uniform; in flight while the region is staged)
    const unsigned int grp0 = g.d_block_group[vb];
    __syncthreads();
    if (a.d_skip_flag && T.word == a.skip_value)
        return;
    const unsigned int total = T.off[64];
    if (total > cap || total == 0)
        return;
    const unsigned int H = T.hcum[TL_NH];
    const unsigned int ngroups = (H + TL_GROUP - 1) / TL_GROUP;
    const unsigned int P = g.row_pieces;
    // piece p of group j of this block: 64 uint2 (four entries each) at rows[(j * P + p) * 64], lane order
    const uint2* rows = reinterpret_cast<const uint2*>(d_tile_nlist) + (size_t)grp0 * P * 64 + lane;
    auto fetch = [&](unsigned int grp, uint2 (&c)[PFT_PRE])
    {
        const uint2* rp = rows + (size_t)grp * P * 64;
#pragma unroll
        for (int q = 0; q < PFT_PRE; q++)
            c[q] = rp[min((unsigned int)q, P - 1u) * 64u]; // rows with room for fewer pieces: read the last one again
    };
    // the rows of this wave's first group are asked for here, so that they arrive while the region is put in place
    uint2 b0[PFT_PRE], b1[PFT_PRE];
#ifndef HMD_PFT_LATE_FIRST_FETCH
    if (wave < ngroups)
        fetch(wave, b0);
#endif

    // ---- staging, second half: place what arrived.  Members of cells away from the periodic faces of the box are
    // copied as they are; the others go to the periodic image nearest to their region cell, decided per particle
    // (the integrator wraps a particle that leaves the box, which moves it by a box vector without changing its
    // cell-list slot): image = rint(centre of the region cell - position) in fractional coordinates
    // (hoomd/BoxDim.h:260-268, 391-458). ----
#pragma unroll
    for (int t = 0; t < TRIPS; t++)
        {
        const int u = u0 + t;
        const int mi = WAVES == 4 ? t - 1 : 0;
        const unsigned int rc = (unsigned int)tile_rc((int)rx, (int)ry, u);
        const unsigned int first = T.off[rc], size = T.off[rc + 1] - first;
        if (__builtin_amdgcn_readfirstlane(__ballot(size != 0) == 0))
            continue;
        const bool edge = ax.edge || ay.edge || az[t].edge;
        // centre of the region cell in fractional coordinates, from its UNWRAPPED grid coordinate
        const double fcx = ((double)(cx0 - 1 + (int)rx) + 0.5) * lp.inv_dim[0];
        const double fcy = ((double)(cy0 - 1 + (int)ry) + 0.5) * lp.inv_dim[1];
        const double fcz = ((double)(cz0 - 1 + u) + 0.5) * lp.inv_dim[2];
        auto place = [&](const pft_pos& pp, unsigned int k)
        {
            double x = pp.x, y = pp.y, z = pp.z;
            if (edge)
                {
                const double fx = ((x - a.box.lo[0]) - ((a.box.xz - a.box.yz * a.box.xy) * z + a.box.xy * y)) * a.box.Linv[0];
                const double fy = ((y - a.box.lo[1]) - a.box.yz * z) * a.box.Linv[1];
                const double fz = (z - a.box.lo[2]) * a.box.Linv[2];
                const double ix = a.box.periodic[0] ? rint(fcx - fx) : 0.0;
                const double iy = a.box.periodic[1] ? rint(fcy - fy) : 0.0;
                const double iz = a.box.periodic[2] ? rint(fcz - fz) : 0.0;
                tile_shift(a.box, ix, iy, iz, x, y, z);
                }
            const unsigned int v = tile_slot(first + k);
            txyz[v] = x;
            txyz[v + 32] = y;
            txyz[v + 64] = z;
        };
        if (k16 < size)
            place(pA[t], k16);
        if (k16 + 16u < size)
            place(pB[t], k16 + 16u);
        // members of a home cell of this block get their entry of the home table
        const bool home_cell = (u == 1 || u == 2) && home_xy && cz0 + u - 1 < dimz;
        unsigned int hfirst = 0;
        if (u == 1 || u == 2)
            {
            if (home_cell)
                hfirst = T.hcum[((u - 1) * TB_Y + (ry - 1)) * TB_X + (rx - 1)];
            if (home_cell && k16 < size)
                s_home[hfirst + k16] = pft_home(mA[mi].x, mA[mi].y, tile_slot(first + k16) | (SINGLE_TYPE ? 0u : pA[t].t << TL_TYPE_SHIFT));
            if (home_cell && k16 + 16u < size)
                s_home[hfirst + k16 + 16u] = pft_home(mB[mi].x, mB[mi].y, tile_slot(first + k16 + 16u) | (SINGLE_TYPE ? 0u : pB[t].t << TL_TYPE_SHIFT));
            }
        // cells with more than 32 members (uncommon): the rest, one dependent load per trip
        if (__builtin_amdgcn_readfirstlane(__ballot(size > 32u) != 0))
            {
            const unsigned int base = (unsigned int)((az[t].c * dimy + ay.c) * dimx + ax.c) * cellNmax;
            for (unsigned int k = k16 + 32u; k < size; k += 16u)
                {
                const pft_pos pk = pft_ld_pos<!SINGLE_TYPE>(g.d_cpos + base + k);
                place(pk, k);
                if (home_cell)
                    {
                    const uint2 m = *reinterpret_cast<const uint2*>(g.d_slot_meta + base + k);
                    s_home[hfirst + k] = pft_home(m.x, m.y, tile_slot(first + k) | (SINGLE_TYPE ? 0u : pk.t << TL_TYPE_SHIFT));
                    }
                }
            }
        }
    if (threadIdx.x == 0)
        {
        s_home[H] = pft_home_t {0u, 0u};
        const unsigned int v = tile_slot(total);
        txyz[v] = 1e30;
        txyz[v + 32] = 1e30;
        txyz[v + 64] = 1e30;
        }
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);

    // ---- forces: a wave takes the block's groups of sixteen rows in turn (wave w: groups w, w + 4, ...).  The
    // first PFT_PRE pieces of every row of the next group are fetched while the present one is computed: rows
    // stream from HBM exactly once, so what bounds this phase is the number of bytes in flight. ----
#ifdef HMD_PFT_STAGE_ONLY
    if (total != 0xdeadbeefu)
        return;
#endif
    const unsigned int s4 = lane % PFT_T, gl = lane / PFT_T;
    // (STEP) the particle whose finished force this lane keeps for the integration pass, and the wave's count of groups
    double kfx = 0.0, kfy = 0.0, kfz = 0.0;
    unsigned int kidx = 0xffffffffu, kept = 0;
    const unsigned int three = 3u;
    auto process = [&](unsigned int grp, const uint2 (&cur)[PFT_PRE])
    {
        const unsigned int h = grp * TL_GROUP + gl;
        const pft_home_t e = s_home[min(h, H)]; // s_home[H] is an empty entry
        const unsigned int lh = SINGLE_TYPE ? e.nn_lh >> 16 : (e.nn_lh >> 16) & TL_SLOT_MASK;
        const unsigned int nn = min(e.nn_lh & 0xffffu, P * TL_PIECE);
        pf_accum acc;
        const double xi = txyz[lh], yi = txyz[lh + 32], zi = txyz[lh + 64];
        const unsigned int ti = SINGLE_TYPE ? 0u : e.nn_lh >> (16u + TL_TYPE_SHIFT);
        // Two particle types: the two parameter sets a pair with this home particle can have sit in registers, chosen
        // per neighbor by its type byte -- one LDS read per pair besides the coordinates instead of four in a
        // dependent chain (type, then the table row), which made the binary mixture LDS-bound.
        param_t prm2[2];
        double rc2[2] = {0.0, 0.0}, ro2[2] = {0.0, 0.0};
        if (TWO_TYPES)
            {
#pragma unroll
            for (int tj = 0; tj < 2; tj++)
                {
                prm2[tj] = s_params[tj * 2 + ti];
                rc2[tj] = s_rcutsq[tj * 2 + ti];
                if (SHIFT == 2)
                    ro2[tj] = s_ronsq[tj * 2 + ti];
                }
            }
        // coordinates of the four neighbors of one row chunk, read from the tile
        struct quad
            {
            double x[4], y[4], z[4];
            unsigned int j[4];
            unsigned int t[SINGLE_TYPE ? 1 : 4]; // several particle types: the neighbors' types, from the top bits of the entries
            };
        auto read_quad = [&](uint2 w, quad& c)
        {
            if (SINGLE_TYPE)
                {
                c.j[0] = pft_slot_bytes<0>(w.x, three);
                c.j[1] = pft_slot_bytes<1>(w.x, three);
                c.j[2] = pft_slot_bytes<0>(w.y, three);
                c.j[3] = pft_slot_bytes<1>(w.y, three);
                }
            else
                {
                c.j[0] = __builtin_amdgcn_ubfe(w.x, 0u, TL_TYPE_SHIFT) << 3;
                c.j[1] = __builtin_amdgcn_ubfe(w.x, 16u, TL_TYPE_SHIFT) << 3;
                c.j[2] = __builtin_amdgcn_ubfe(w.y, 0u, TL_TYPE_SHIFT) << 3;
                c.j[3] = __builtin_amdgcn_ubfe(w.y, 16u, TL_TYPE_SHIFT) << 3;
                c.t[0 % (SINGLE_TYPE ? 1 : 4)] = __builtin_amdgcn_ubfe(w.x, TL_TYPE_SHIFT, 16u - TL_TYPE_SHIFT);
                c.t[1 % (SINGLE_TYPE ? 1 : 4)] = w.x >> (16u + TL_TYPE_SHIFT);
                c.t[2 % (SINGLE_TYPE ? 1 : 4)] = __builtin_amdgcn_ubfe(w.y, TL_TYPE_SHIFT, 16u - TL_TYPE_SHIFT);
                c.t[3 % (SINGLE_TYPE ? 1 : 4)] = w.y >> (16u + TL_TYPE_SHIFT);
                }
#pragma unroll
            for (int m = 0; m < 4; m++)
                {
                // three 8-byte reads: paired 16-byte LDS reads run at half the rate (volatile keeps them apart)
                pft_lds_double* t = (pft_lds_double*)(size_t)c.j[m];
                c.x[m] = t[0];
                c.y[m] = *(volatile pft_lds_double*)(t + 32);
                c.z[m] = *(volatile pft_lds_double*)(t + 64);
                }
        };
        // the evaluator's four-pair form where it exists (with the virial sums it would not fit the registers)
#ifdef HMD_PFT_NO_BATCH4
        constexpr bool BATCH4 = false;
#else
        constexpr bool BATCH4 = SHIFT == 0 && !VIRIAL && has_branch_free4<Ev>::value && (SINGLE_TYPE || TWO_TYPES);
#endif
        auto pairs = [&](const quad& c)
        {
#ifdef HMD_PFT_SKELETON
            acc.fx += c.x[0] + c.y[1] + c.z[2] + c.x[3];
            return;
#endif
            if constexpr (BATCH4)
                {
                // the four pairs of a chunk at once (they share work inside the evaluator)
                double dx[4], dy[4], dz[4], rc[4];
                const param_t* pp[4];
#pragma unroll
                for (int m = 0; m < 4; m++)
                    {
                    dx[m] = xi - c.x[m], dy[m] = yi - c.y[m], dz[m] = zi - c.z[m];
                    if (SINGLE_TYPE)
                        {
                        rc[m] = rcutsq0;
                        pp[m] = &p0;
                        }
                    else
                        {
                        const bool second = c.t[(BY_TYPE ? 0 : m) % (SINGLE_TYPE ? 1 : 4)] != 0u;
                        rc[m] = second ? rc2[1] : rc2[0];
                        pp[m] = second ? &prm2[1] : &prm2[0];
                        }
                    }
                pf_pair4<Ev, VIRIAL>(acc, dx, dy, dz, rc, pp);
                return;
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
                else if (TWO_TYPES)
                    {
                    const bool second = c.t[m % (SINGLE_TYPE ? 1 : 4)] != 0u;
                    rcutsq = second ? rc2[1] : rc2[0];
                    if (SHIFT == 2)
                        ronsq = second ? ro2[1] : ro2[0];
                    const param_t sel = second ? prm2[1] : prm2[0];
                    pf_pair<Ev, SHIFT, VIRIAL>(acc, xi - c.x[m], yi - c.y[m], zi - c.z[m], rcutsq, ronsq, sel, 0.0, 0.0);
                    continue;
                    }
                else
                    {
                    const unsigned int tp = c.t[m % (SINGLE_TYPE ? 1 : 4)] * a.ntypes + ti;
                    rcutsq = s_rcutsq[tp];
                    if (SHIFT == 2)
                        ronsq = s_ronsq[tp];
                    prm = s_params + tp;
                    }
                pf_pair<Ev, SHIFT, VIRIAL>(acc, xi - c.x[m], yi - c.y[m], zi - c.z[m], rcutsq, ronsq, *prm, 0.0, 0.0);
                }
        };
        // Rows are padded to a multiple of four entries with the dummy's slot; pieces past the end of a row hold
        // whatever was there: their lanes are switched off.  The tile reads of the next chunk are issued before the
        // arithmetic of the present one, so a wave does not sit idle through the LDS latency at every chunk.
        // (only where the registers allow it without spilling: the plain potential without virials, evaluated pair by
        // pair; the four-pair form keeps four pairs in flight by itself and needs the registers)
#ifdef HMD_PFT_NO_PIPE
        constexpr bool PIPELINED = false;
#else
        constexpr bool PIPELINED = (SHIFT == 0 && !VIRIAL && !TWO_TYPES && !BATCH4);
#endif
#ifndef HMD_PFT_NO_TAIL_PREFETCH
        uint2 tail[PFT_TAIL];
#pragma unroll
        for (int t = 0; t < PFT_TAIL; t++)
            {
            tail[t] = make_uint2(0u, 0u);
            if ((unsigned int)(TL_PIECE * (PFT_PRE + t)) + 4u * s4 < nn)
                tail[t] = (rows + (size_t)grp * P * 64)[(PFT_PRE + t) * 64u];
            }
#endif
        quad qa;
        if (PIPELINED)
            {
            quad qb;
            bool on = 4u * s4 < nn;
            if (on)
                read_quad(cur[0], qa);
#pragma unroll
            for (int q = 0; q < PFT_PRE; q++)
                {
                quad& now = (q & 1) ? qb : qa;
                quad& next = (q & 1) ? qa : qb;
                const bool on_next = q + 1 < PFT_PRE && (unsigned int)(TL_PIECE * (q + 1)) + 4u * s4 < nn;
                if (on_next)
                    read_quad(cur[q + 1], next);
                if (on)
                    pairs(now);
                on = on_next;
                }
            }
        else
            {
#pragma unroll
            for (int q = 0; q < PFT_PRE; q++)
                if ((unsigned int)(TL_PIECE * q) + 4u * s4 < nn)
                    {
                    read_quad(cur[q], qa);
                    pairs(qa);
                    }
            }
        // rows longer than PFT_PRE pieces: the next PFT_TAIL pieces were asked for before the arithmetic above (only by
        // the lanes whose rows reach that far), so they have arrived by now; what lies beyond is read on the spot
#ifndef HMD_PFT_NO_TAIL_PREFETCH
#pragma unroll
        for (int t = 0; t < PFT_TAIL; t++)
            {
            const bool more = (unsigned int)(TL_PIECE * (PFT_PRE + t)) + 4u * s4 < nn;
            if (__any(more) && more)
                {
                read_quad(tail[t], qa);
                pairs(qa);
                }
            }
        constexpr int tail_done = PFT_PRE + PFT_TAIL;
#else
        constexpr int tail_done = PFT_PRE;
#endif
        if (__any((unsigned int)(TL_PIECE * tail_done) < nn))
            {
            const uint2* rp = rows + (size_t)grp * P * 64;
            for (unsigned int q = tail_done; q < P; q++)
                {
                const bool more = TL_PIECE * q + 4u * s4 < nn;
                if (!__any(more))
                    break;
                if (more)
                    {
                    read_quad(rp[q * 64u], qa);
                    pairs(qa);
                    }
                }
            }
        const bool writer = h < H && e.idx < a.N && s4 == 0;
        const double4 f = pf_finish<VIRIAL, PFT_T>(acc, a, e.idx, writer);
        if constexpr (STEP)
            {
            // The finished force is complete in all four lanes of the home particle: the lane whose number among the
            // four equals the count of groups this wave has finished keeps it, so after four groups every lane of the
            // wave holds one particle and the integration pass runs once with all lanes busy (flush below), instead
            // of sixteen lanes at the end of every group.
            if (s4 == (kept & 3u))
                {
                kfx = f.x;
                kfy = f.y;
                kfz = f.z;
                kidx = (h < H && e.idx < a.N) ? e.idx : 0xffffffffu;
                }
            kept++;
            }
    };
    auto flush = [&]()
    {
        if constexpr (STEP)
            {
            // (the offset passes through an empty asm statement so that the parameters are read here, in one batch, and
            // not ahead of the loop, where they would sit in registers all the way)
            unsigned int soff = tile_len * 8u + (unsigned int)offsetof(pft_tables, step);
            asm volatile("" : "+v"(soff));
            pft_step_lds& S = *(pft_step_lds*)(size_t)soff;
            bool moved = false;
            if (kidx != 0xffffffffu)
                moved = pft_step_particle<SINGLE_TYPE>(S, kidx, kfx, kfy, kfz);
            if (__ballot(moved) && lane == 0)
                atomicMax(S.d_result, S.checkn);
            kidx = 0xffffffffu;
            }
    };
    unsigned int grp = wave;
#ifdef HMD_PFT_LATE_FIRST_FETCH
    if (grp < ngroups)
        fetch(grp, b0);
#endif
    while (grp < ngroups)
        {
        if (grp + WAVES < ngroups)
            fetch(grp + WAVES, b1);
        process(grp, b0);
        grp += WAVES;
        if (grp >= ngroups)
            break;
        if (grp + WAVES < ngroups)
            fetch(grp + WAVES, b0);
        process(grp, b1);
        grp += WAVES;
        if (STEP && (kept & 3u) == 0u)
            flush();
        }
    if (STEP && (kept & 3u) != 0u)
        flush();
    }

template<class Ev> static inline size_t pft_lds_bytes(unsigned int cap, unsigned int ntypes, unsigned int cell_Nmax)
    {
    size_t b = (size_t)tile_doubles(cap) * 8 + PFT_TABLES + pft_home_bytes(cell_Nmax);
    if (ntypes > 1)
        b += (size_t)ntypes * ntypes * (16 + sizeof(typename Ev::param_type));
    return (b + 15) / 16 * 16;
    }

template<class Ev, unsigned int SHIFT, bool VIRIAL, bool STEP = false>
int pft_launch2(const hmd_pair_args_t& a, const hmd_tile_geom_t& g, const unsigned short* d_tile_nlist,
                const typename Ev::param_type* d_params, hipStream_t s, const hmd_fused_step_args_t* step = nullptr)
    {
    pft_step_t st;
    memset(&st, 0, sizeof(st));
    if (STEP)
        {
        st.dt = step->one.deltaT;
        for (int d = 0; d < 3; d++)
            {
            st.lo[d] = step->one.box.lo[d];
            st.hi[d] = step->one.box.hi[d];
            st.L[d] = step->one.box.L[d];
            st.Linv[d] = step->one.box.Linv[d];
            }
        st.r_buff = step->check.r_buff;
        st.d_vel = step->one.d_vel;
        st.d_pos = step->one.d_pos;
        st.d_image = step->one.d_image;
        st.d_slot_of = step->d_slot_of;
        st.d_cpos = step->d_cpos;
        st.d_last_pos = step->check.d_last_pos;
        st.d_rcut_max = step->check.d_rcut_max;
        st.d_result = step->check.d_result;
        st.checkn = step->check.checkn;
        st.check = step->check.d_result != nullptr;
        }
    const size_t lds = pft_lds_bytes<Ev>(g.capacity, a.ntypes, g.cli.w);
    unsigned int nblocks = tile_num_blocks(g.ci);
    if (a.block_window_mode)
        {
        const unsigned int inside = (a.block_hi[0] - a.block_lo[0] + 1) * (a.block_hi[1] - a.block_lo[1] + 1) * (a.block_hi[2] - a.block_lo[2] + 1);
        nblocks = a.block_window_mode == 1 ? inside : nblocks - inside;
        if (nblocks == 0)
            return HMD_OK; // the window is the whole grid: nothing outside it
        }
    // Eight waves per workgroup where the region is so large that only two workgroups fit the LDS of a CU (densely
    // filled cells): the CU then holds sixteen waves as it does with four workgroups of four.  (The kernel with the
    // integration pass riding on it exists with four waves only.)
    // (A/B switch: the kernel with per-pair parameter selects also on rows laid out by type)
    const bool by_type = g.rows_by_type && !std::getenv("HOOMD_AMD_PFT_NO_BY_TYPE");
    bool eight = !STEP && 3 * (lds + 1024) > 160 * 1024;
    if (const char* w = std::getenv("HOOMD_AMD_PFT_WAVES"))
        eight = !STEP && std::atoi(w) == 8;
    const dim3 grid(nblocks), block(eight ? 512 : PFT_THREADS);
    pft_launch_t lp;
    lp.nbx = (g.ci.w + TB_X - 1) / TB_X;
    lp.nby = (g.ci.h + TB_Y - 1) / TB_Y;
    lp.inv_nbx = ((1ull << 40) + lp.nbx - 1) / lp.nbx;
    lp.inv_nby = ((1ull << 40) + lp.nby - 1) / lp.nby;
    lp.inv_dim[0] = 1.0 / (double)g.ci.w;
    lp.inv_dim[1] = 1.0 / (double)g.ci.h;
    lp.inv_dim[2] = 1.0 / (double)g.ci.d;
    if (a.ntypes == 1)
        {
        if constexpr (!STEP)
            if (eight)
                {
                tile_allow_big_lds(pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, true, false, false, 8>, lds);
                hipLaunchKernelGGL((pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, true, false, false, 8>), grid, block, lds, s, a, g, d_tile_nlist, d_params, lp, st);
                return hmd_launch_status();
                }
        tile_allow_big_lds(pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, true, false, STEP>, lds);
        hipLaunchKernelGGL((pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, true, false, STEP>), grid, block, lds, s, a, g, d_tile_nlist, d_params, lp, st);
        }
    else if (a.ntypes == 2)
        {
        if constexpr (!STEP)
            if (eight)
                {
                if constexpr (SHIFT == 0 && !VIRIAL && has_branch_free4<Ev>::value)
                    if (by_type)
                        {
                        tile_allow_big_lds(pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, false, true, false, 8, true>, lds);
                        hipLaunchKernelGGL((pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, false, true, false, 8, true>), grid, block, lds, s, a, g, d_tile_nlist, d_params, lp, st);
                        return hmd_launch_status();
                        }
                tile_allow_big_lds(pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, false, true, false, 8>, lds);
                hipLaunchKernelGGL((pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, false, true, false, 8>), grid, block, lds, s, a, g, d_tile_nlist, d_params, lp, st);
                return hmd_launch_status();
                }
        // rows laid out by type: the kernel that picks parameters per chunk (the plain potential without virials)
        if constexpr (!STEP && SHIFT == 0 && !VIRIAL && has_branch_free4<Ev>::value)
            if (by_type)
                {
                tile_allow_big_lds(pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, false, true, false, 4, true>, lds);
                hipLaunchKernelGGL((pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, false, true, false, 4, true>), grid, block, lds, s, a, g, d_tile_nlist, d_params, lp, st);
                return hmd_launch_status();
                }
        tile_allow_big_lds(pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, false, true, STEP>, lds);
        hipLaunchKernelGGL((pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, false, true, STEP>), grid, block, lds, s, a, g, d_tile_nlist, d_params, lp, st);
        }
    else
        {
        if constexpr (!STEP)
            if (eight)
                {
                tile_allow_big_lds(pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, false, false, false, 8>, lds);
                hipLaunchKernelGGL((pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, false, false, false, 8>), grid, block, lds, s, a, g, d_tile_nlist, d_params, lp, st);
                return hmd_launch_status();
                }
        tile_allow_big_lds(pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, false, false, STEP>, lds);
        hipLaunchKernelGGL((pair_force_tiles_kernel<Ev, SHIFT, VIRIAL, false, false, STEP>), grid, block, lds, s, a, g, d_tile_nlist, d_params, lp, st);
        }
    return hmd_launch_status();
    }

// step: the integration pass to carry (hmd_fused_step_args_t as for hmd_nve_step_two_one_check), or null
template<class Ev>
int launch_pair_forces_tiles(const hmd_pair_args_t* args, const hmd_tile_geom_t* geom, const unsigned short* d_tile_nlist,
                             const typename Ev::param_type* d_params, void* stream, const hmd_fused_step_args_t* step = nullptr)
    {
    HMD_REQUIRE(args && geom && d_tile_nlist && d_params, HMD_ERR_INVALID_ARGUMENT,
                "pair forces (tiles): args, geom, d_tile_nlist and d_params are required");
    const hmd_pair_args_t& a = *args;
    const hmd_tile_geom_t& g = *geom;
    HMD_REQUIRE(a.d_force && a.d_pos && a.d_n_neigh && a.d_rcutsq, HMD_ERR_INVALID_ARGUMENT,
                "pair forces (tiles): force, pos, n_neigh and rcutsq are required");
    HMD_REQUIRE(g.d_block_group && g.row_pieces >= 1, HMD_ERR_INVALID_ARGUMENT, "pair forces (tiles): geom.d_block_group and geom.row_pieces are required");
    HMD_REQUIRE(g.d_cell_size && g.d_cell_idx, HMD_ERR_INVALID_ARGUMENT, "pair forces (tiles): the cell arrays are required");
    HMD_REQUIRE((size_t)g.cli.w * g.cli.h * 32 <= 0xffffffffull && tile_num_blocks(g.ci) < (1u << 24) && g.ci.w < 65536 && g.ci.h < 65536,
                HMD_ERR_TOO_LARGE, "pair forces (tiles): the cell list exceeds 2^27 slots or the grid 2^24 blocks");
    HMD_REQUIRE(g.d_cpos && g.d_slot_meta, HMD_ERR_INVALID_ARGUMENT,
                "pair forces (tiles): geom.d_cpos (hmd_tile_refresh) and geom.d_slot_meta (hmd_compute_nlist_tiles) are required");
    HMD_REQUIRE((reinterpret_cast<uintptr_t>(d_tile_nlist) & 7) == 0, HMD_ERR_INVALID_ARGUMENT,
                "pair forces (tiles): d_tile_nlist must be 8-byte aligned");
    HMD_REQUIRE(a.shift_mode <= 2, HMD_ERR_INVALID_ARGUMENT, "pair forces (tiles): shift_mode must be 0, 1 or 2");
    HMD_REQUIRE(a.shift_mode != 2 || a.d_ronsq, HMD_ERR_INVALID_ARGUMENT, "pair forces (tiles): xplor needs d_ronsq");
    HMD_REQUIRE(!a.compute_virial || a.d_virial, HMD_ERR_INVALID_ARGUMENT, "pair forces (tiles): compute_virial needs d_virial");
    HMD_REQUIRE(!pf_needs_charge<Ev>((hipStream_t)stream), HMD_ERR_UNSUPPORTED,
                "pair forces (tiles): evaluators that need charges use the row-based entry point");
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
    HMD_REQUIRE(a.threads_per_particle == 0 || a.threads_per_particle == PFT_T, HMD_ERR_INVALID_ARGUMENT,
                "pair forces (tiles): threads_per_particle must be 0 or 4");
    if (step)
        {
        const char* why = nullptr;
        if (!pft_step_supported(a, *step, &why))
            {
            static thread_local char msg[256];
            snprintf(msg, sizeof(msg), "pair forces (tiles) with the integration pass: %s", why);
            hmd_set_error(msg);
            return HMD_ERR_UNSUPPORTED;
            }
        HMD_REQUIRE(step->d_cpos && step->d_slot_of && step->d_cpos != g.d_cpos, HMD_ERR_INVALID_ARGUMENT,
                    "pair forces (tiles) with the integration pass: d_slot_of and a d_cpos other than geom.d_cpos are required");
        }
    if (a.N == 0)
        return HMD_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool vir = a.compute_virial != 0;
    if (step)
        switch (a.shift_mode)
            {
        case 0:
            return pft_launch2<Ev, 0, false, true>(a, g, d_tile_nlist, d_params, s, step);
        case 1:
            return pft_launch2<Ev, 1, false, true>(a, g, d_tile_nlist, d_params, s, step);
        default:
            return pft_launch2<Ev, 2, false, true>(a, g, d_tile_nlist, d_params, s, step);
            }
    switch (a.shift_mode)
        {
    case 0:
        return vir ? pft_launch2<Ev, 0, true>(a, g, d_tile_nlist, d_params, s) : pft_launch2<Ev, 0, false>(a, g, d_tile_nlist, d_params, s);
    case 1:
        return vir ? pft_launch2<Ev, 1, true>(a, g, d_tile_nlist, d_params, s) : pft_launch2<Ev, 1, false>(a, g, d_tile_nlist, d_params, s);
    default:
        return vir ? pft_launch2<Ev, 2, true>(a, g, d_tile_nlist, d_params, s) : pft_launch2<Ev, 2, false>(a, g, d_tile_nlist, d_params, s);
        }
    }
    } // namespace hmd

#endif
