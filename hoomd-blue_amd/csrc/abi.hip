// abi.hip -- host-only pieces of the C ABI: box construction, version, error text.
#include "common.cuh"

#include <cmath>
#include <cstring>

static thread_local char g_err[256] = "";

void hmd_set_error(const char* msg)
    {
    std::strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
    }

extern "C" const char* hmd_last_error(void)
    {
    return g_err;
    }

extern "C" int hmd_abi_version(void)
    {
    return HMD_ABI_VERSION;
    }

// BoxDim::getNearestPlaneDistance (hoomd/BoxDim.h:536-552)
static void set_nearest_plane(hmd_box_t* b)
    {
    const double t = b->xy * b->yz - b->xz;
    b->nearest_plane[0] = b->L[0] * (1.0 / std::sqrt(1.0 + b->xy * b->xy + t * t));
    b->nearest_plane[1] = b->L[1] * (1.0 / std::sqrt(1.0 + b->yz * b->yz));
    b->nearest_plane[2] = b->L[2];
    if (b->L[2] == 0.0)
        b->nearest_plane[2] = 1.0;
    }

extern "C" void hmd_box_init(hmd_box_t* b, double Lx, double Ly, double Lz, double xy, double xz, double yz)
    {
    std::memset(b, 0, sizeof(*b));
    const double L[3] = {Lx, Ly, Lz};
    for (int d = 0; d < 3; d++)
        {
        b->hi[d] = L[d] / 2.0;
        b->lo[d] = -b->hi[d];
        b->Linv[d] = 1.0 / L[d];
        b->L[d] = L[d];
        b->periodic[d] = 1;
        }
    if (Lz == 0.0)
        b->Linv[2] = 0.0;
    b->xy = xy;
    b->xz = xz;
    b->yz = yz;
    set_nearest_plane(b);
    }

extern "C" void hmd_box_set_lo_hi(hmd_box_t* b, const double lo[3], const double hi[3])
    {
    for (int d = 0; d < 3; d++)
        {
        b->hi[d] = hi[d];
        b->lo[d] = lo[d];
        b->Linv[d] = 1.0 / (hi[d] - lo[d]);
        b->L[d] = hi[d] - lo[d];
        }
    if (hi[2] == 0.0 && lo[2] == 0.0)
        b->Linv[2] = 0.0;
    set_nearest_plane(b);
    }
