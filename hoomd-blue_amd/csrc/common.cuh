// common.cuh -- shared device helpers of the gfx950 kernel-driver library.
#ifndef HMD_COMMON_CUH
#define HMD_COMMON_CUH

#include <hip/hip_runtime.h>

#include "hoomd_amd.h"

#define HMD_WAVE 64

void hmd_set_error(const char* msg);

#define HMD_REQUIRE(cond, code, msg) \
    do                               \
        {                            \
        if (!(cond))                 \
            {                        \
            hmd_set_error(msg);      \
            return (code);           \
            }                        \
        } while (0)

static inline int hmd_launch_status()
    {
    hipError_t e = hipGetLastError();
    return (int)e;
    }

static inline unsigned int hmd_div_up(unsigned int a, unsigned int b)
    {
    return (a + b - 1) / b;
    }

// ---- bit stuffing of integers in the low word of a double (hoomd/HOOMDMath.h:150-217) ----
__device__ __forceinline__ unsigned int hmd_double_as_uint(double v)
    {
    return (unsigned int)__double2loint(v);
    }
__device__ __forceinline__ double hmd_uint_as_double(unsigned int a)
    {
    return __hiloint2double(0, (int)a);
    }

// ---- indexers ----
__device__ __forceinline__ unsigned int hmd_idx2(const hmd_index2d_t& ix, unsigned int i, unsigned int j)
    {
    return j * ix.w + i;
    }
__device__ __forceinline__ unsigned int hmd_idx3(const hmd_index3d_t& ix, unsigned int i, unsigned int j, unsigned int k)
    {
    return (k * ix.h + j) * ix.w + i;
    }

// ---- 16-byte vector access to Scalar4 arrays ----
__device__ __forceinline__ double4 hmd_ld4(const hmd_scalar4* p)
    {
    return *reinterpret_cast<const double4*>(p);
    }
__device__ __forceinline__ void hmd_st4(hmd_scalar4* p, double x, double y, double z, double w)
    {
    *reinterpret_cast<double4*>(p) = make_double4(x, y, z, w);
    }

// The functions below implement bit-exact comparisons against the CPU path, so no product may be
// fused into a following sum.  HIP's __dmul_rn/__dadd_rn are plain operators and can be contracted,
// so the guarantee comes from the translation unit: every unit that uses these helpers for a
// bit-exact result sets `#pragma clang fp contract(off)` before including this header and is
// compiled with -ffp-contract=off (csrc/Makefile).  The macros only mark the places that rely on it.
// The box helpers below also carry the pragma in their bodies, so they round the same way when a unit
// that allows fused multiply-adds calls them (the integration pass that rides on the pair-force kernel).
#define HMD_MUL(a, b) ((a) * (b))
#define HMD_ADD(a, b) ((a) + (b))
#define HMD_SUB(a, b) ((a) - (b))

// BoxDim::makeFraction (hoomd/BoxDim.h:260-268), products and sums rounded separately.
__device__ __forceinline__ double3 hmd_make_fraction(const hmd_box_t& b, double x, double y, double z, const hmd_scalar3& gw)
    {
#pragma clang fp contract(off)
    double dx = HMD_SUB(x, b.lo[0]);
    double dy = HMD_SUB(y, b.lo[1]);
    double dz = HMD_SUB(z, b.lo[2]);
    dx = HMD_SUB(dx, HMD_ADD(HMD_MUL(HMD_SUB(b.xz, HMD_MUL(b.yz, b.xy)), z), HMD_MUL(b.xy, y)));
    dy = HMD_SUB(dy, HMD_MUL(b.yz, z));
    const double gfx = gw.x / b.nearest_plane[0];
    const double gfy = gw.y / b.nearest_plane[1];
    const double gfz = gw.z / b.nearest_plane[2];
    double3 f;
    f.x = HMD_ADD(HMD_MUL(dx, b.Linv[0]), gfx) / HMD_ADD(1.0, HMD_MUL(2.0, gfx));
    f.y = HMD_ADD(HMD_MUL(dy, b.Linv[1]), gfy) / HMD_ADD(1.0, HMD_MUL(2.0, gfy));
    f.z = HMD_ADD(HMD_MUL(dz, b.Linv[2]), gfz) / HMD_ADD(1.0, HMD_MUL(2.0, gfz));
    return f;
    }

// BoxDim::minImage, device form (hoomd/BoxDim.h:306-325).  Within one image of the box the result
// equals the host form (:326-373) bit for bit: the image count is -1, 0 or +1 and L*count is exact.
template<bool ORTHO> __device__ __forceinline__ void hmd_min_image(const hmd_box_t& b, double& x, double& y, double& z)
    {
#pragma clang fp contract(off)
    if (b.periodic[2])
        {
        const double img = rint(HMD_MUL(z, b.Linv[2]));
        z = HMD_SUB(z, HMD_MUL(b.L[2], img));
        if (!ORTHO)
            {
            y = HMD_SUB(y, HMD_MUL(HMD_MUL(b.L[2], b.yz), img));
            x = HMD_SUB(x, HMD_MUL(HMD_MUL(b.L[2], b.xz), img));
            }
        }
    if (b.periodic[1])
        {
        const double img = rint(HMD_MUL(y, b.Linv[1]));
        y = HMD_SUB(y, HMD_MUL(b.L[1], img));
        if (!ORTHO)
            x = HMD_SUB(x, HMD_MUL(HMD_MUL(b.L[1], b.xy), img));
        }
    if (b.periodic[0])
        {
        const double img = rint(HMD_MUL(x, b.Linv[0]));
        x = HMD_SUB(x, HMD_MUL(b.L[0], img));
        }
    }

// BoxDim::wrap with flags (0,0,0) (hoomd/BoxDim.h:391-458)
__device__ __forceinline__ void hmd_wrap(const hmd_box_t& b, double& x, double& y, double& z, int& ix, int& iy, int& iz)
    {
#pragma clang fp contract(off)
    double oy = (b.hi[1] + b.lo[1]) / 2.0;
    double oz = (b.hi[2] + b.lo[2]) / 2.0;
    if (!b.periodic[1])
        oy = 0.0;
    if (!b.periodic[2])
        oz = 0.0;
    if (b.periodic[0])
        {
        const double tilt_x = HMD_ADD(HMD_MUL(HMD_SUB(b.xz, HMD_MUL(b.xy, b.yz)), HMD_SUB(z, oz)), HMD_MUL(b.xy, HMD_SUB(y, oy)));
        if (x >= HMD_ADD(b.hi[0], tilt_x))
            {
            x = HMD_SUB(x, b.L[0]);
            ix++;
            }
        else if (x < HMD_ADD(b.lo[0], tilt_x))
            {
            x = HMD_ADD(x, b.L[0]);
            ix--;
            }
        }
    if (b.periodic[1])
        {
        const double tilt_y = HMD_MUL(b.yz, HMD_SUB(z, oz));
        if (y >= HMD_ADD(b.hi[1], tilt_y))
            {
            y = HMD_SUB(y, b.L[1]);
            x = HMD_SUB(x, HMD_MUL(b.L[1], b.xy));
            iy++;
            }
        else if (y < HMD_ADD(b.lo[1], tilt_y))
            {
            y = HMD_ADD(y, b.L[1]);
            x = HMD_ADD(x, HMD_MUL(b.L[1], b.xy));
            iy--;
            }
        }
    if (b.periodic[2])
        {
        if (z >= b.hi[2])
            {
            z = HMD_SUB(z, b.L[2]);
            y = HMD_SUB(y, HMD_MUL(b.L[2], b.yz));
            x = HMD_SUB(x, HMD_MUL(b.L[2], b.xz));
            iz++;
            }
        else if (z < b.lo[2])
            {
            z = HMD_ADD(z, b.L[2]);
            y = HMD_ADD(y, HMD_MUL(b.L[2], b.yz));
            x = HMD_ADD(x, HMD_MUL(b.L[2], b.xz));
            iz--;
            }
        }
    }

static inline bool hmd_box_is_ortho(const hmd_box_t& b)
    {
    return b.xy == 0.0 && b.xz == 0.0 && b.yz == 0.0;
    }

// the factor of the first half step: the one the host passed, or sqrt(target energy / kinetic energy) with the kinetic
// energy read where the thermo sums left it (every lane reads the same word)
__device__ __forceinline__ double hmd_rescale_factor(const hmd_step_one_args_t& a)
    {
    return a.d_rescale_ke ? sqrt(a.rescale_gamma / *a.d_rescale_ke) : a.rescale_factor;
    }

// number of lanes below this one in a 64-bit ballot mask
__device__ __forceinline__ unsigned int hmd_rank_in_mask(unsigned long long mask)
    {
    return __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
    }

#endif
