// ref_harness.cc -- TEST INFRASTRUCTURE ONLY.
//
// Compiles the reference's own header-only arithmetic WHERE IT LIES under /root/reference
// (nothing is copied into this repository) and prints golden vectors as JSON on stdout:
//   hoomd/BoxDim.h            makeFraction / minImage (host form) / wrap / getNearestPlaneDistance
//   hoomd/Index1D.h           Index2D / Index3D
//   hoomd/md/EvaluatorPairLJ.h                              evalForceAndEnergy, LRC integrals
//   example_plugins/pair_plugin/EvaluatorPairExample.h      evalForceAndEnergy
//   hoomd/md/EvaluatorBondFENE.h                            evalForceAndEnergy
// Doubles are written as C99 hex floats so the vectors are bit-exact.
//
// Build recipe: oracle/Makefile target `ref` (binary lands in oracle/_ref/, git-ignored).
// The generated file is committed as tests/golden/ref_header_vectors.json by
// tests/golden/make_golden.py.

#include "hoomd/BoxDim.h"
#include "hoomd/HOOMDMath.h"
#include "hoomd/Index1D.h"
#include "hoomd/md/EvaluatorBondFENE.h"
#include "hoomd/md/EvaluatorPairLJ.h"
#include "example_plugins/pair_plugin/EvaluatorPairExample.h"

#include <cstdint>
#include <cstdio>
#include <vector>

using namespace hoomd;
using namespace hoomd::md;

// small deterministic generator (splitmix64) so the inputs are reproducible anywhere
static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint64_t next_u64()
    {
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
    }
static double uni(double lo, double hi)
    {
    return lo + (hi - lo) * (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0);
    }

static void pd(double v)
    {
    printf("\"%a\"", v);
    }
static void p3(const Scalar3& v)
    {
    printf("[");
    pd(v.x);
    printf(",");
    pd(v.y);
    printf(",");
    pd(v.z);
    printf("]");
    }

int main()
    {
    printf("{\n");

    // ---------------- evaluators ----------------
    printf("\"lj\": [\n");
    const int n_lj = 400;
    for (int i = 0; i < n_lj; i++)
        {
        double sigma = uni(0.5, 1.5), eps = uni(0.0, 2.0);
        if (i % 37 == 0)
            eps = 0.0; // lj1 == 0 branch
        double rcut = uni(1.0, 3.5);
        double r = uni(0.7, 3.6);
        int shift = i & 1;
        EvaluatorPairLJ::param_type p(sigma, eps);
        EvaluatorPairLJ ev(r * r, rcut * rcut, p);
        Scalar f = 0, e = 0;
        bool ok = ev.evalForceAndEnergy(f, e, shift);
        EvaluatorPairLJ ev0(0.0, rcut * rcut, p);
        printf("{\"sigma\":");
        pd(sigma);
        printf(",\"epsilon\":");
        pd(eps);
        printf(",\"sigma_6\":");
        pd(p.sigma_6);
        printf(",\"epsilon_x_4\":");
        pd(p.epsilon_x_4);
        printf(",\"rsq\":");
        pd(r * r);
        printf(",\"rcutsq\":");
        pd(rcut * rcut);
        printf(",\"shift\":%d,\"evaluated\":%d,\"force_divr\":", shift, (int)ok);
        pd(f);
        printf(",\"pair_eng\":");
        pd(e);
        printf(",\"lrc_energy\":");
        pd(ev0.evalEnergyLRCIntegral());
        printf(",\"lrc_pressure\":");
        pd(ev0.evalPressureLRCIntegral());
        printf("}%s\n", i + 1 < n_lj ? "," : "");
        }
    printf("],\n");

    printf("\"example\": [\n");
    const int n_ex = 200;
    for (int i = 0; i < n_ex; i++)
        {
        EvaluatorPairExample::param_type p;
        p.k = uni(0.1, 50.0);
        p.sigma = uni(0.3, 2.0);
        double rcut = uni(0.5, 3.0);
        double r = uni(0.05, 3.2);
        int shift = i & 1;
        EvaluatorPairExample ev(r * r, rcut * rcut, p);
        Scalar f = 0, e = 0;
        bool ok = ev.evalForceAndEnergy(f, e, shift);
        printf("{\"k\":");
        pd(p.k);
        printf(",\"sigma\":");
        pd(p.sigma);
        printf(",\"rsq\":");
        pd(r * r);
        printf(",\"rcutsq\":");
        pd(rcut * rcut);
        printf(",\"shift\":%d,\"evaluated\":%d,\"force_divr\":", shift, (int)ok);
        pd(f);
        printf(",\"pair_eng\":");
        pd(e);
        printf("}%s\n", i + 1 < n_ex ? "," : "");
        }
    printf("],\n");

    // ---------------- boxes ----------------
    struct BoxSpec
        {
        double Lx, Ly, Lz, xy, xz, yz;
        int px, py, pz;
        };
    std::vector<BoxSpec> specs = {{42.394, 42.394, 42.394, 0, 0, 0, 1, 1, 1},
                                  {135.66, 135.66, 67.83, 0, 0, 0, 1, 1, 1},
                                  {10.0, 5.5, 7.25, 0, 0, 0, 1, 1, 1},
                                  {20.0, 20.0, 20.0, 0, 0, 0, 1, 0, 0},
                                  {12.0, 9.0, 11.0, 0.3, -0.2, 0.15, 1, 1, 1},
                                  {8.0, 8.0, 0.0, 0.1, 0, 0, 1, 1, 0}};
    printf("\"boxes\": [\n");
    for (size_t b = 0; b < specs.size(); b++)
        {
        const BoxSpec& s = specs[b];
        BoxDim box(make_scalar3(s.Lx, s.Ly, s.Lz));
        box.setTiltFactors(s.xy, s.xz, s.yz);
        box.setPeriodic(make_uchar3(s.px, s.py, s.pz));
        Scalar3 npd = box.getNearestPlaneDistance();
        printf("{\"L\":[");
        pd(s.Lx);
        printf(",");
        pd(s.Ly);
        printf(",");
        pd(s.Lz);
        printf("],\"tilt\":[");
        pd(s.xy);
        printf(",");
        pd(s.xz);
        printf(",");
        pd(s.yz);
        printf("],\"periodic\":[%d,%d,%d],\"nearest_plane_distance\":", s.px, s.py, s.pz);
        p3(npd);
        printf(",\n \"cases\":[\n");
        const int n_cases = 64;
        for (int c = 0; c < n_cases; c++)
            {
            // a point up to ~0.9 box lengths from the origin (within one image)
            Scalar3 v = make_scalar3(uni(-0.9, 0.9) * s.Lx, uni(-0.9, 0.9) * s.Ly, uni(-0.9, 0.9) * s.Lz);
            if (c % 8 == 0)
                v.x = s.Lx / 2.0; // exactly on the hi face
            if (c % 8 == 1)
                v.y = -s.Ly / 2.0; // exactly on the lo face
            Scalar3 gw = (c & 1) ? make_scalar3(0.7, 1.1, s.Lz == 0.0 ? 0.0 : 0.4) : make_scalar3(0, 0, 0);
            // fractional coordinate of a point inside (+- a little) the box
            Scalar3 vin = make_scalar3(v.x * 0.55, v.y * 0.55, v.z * 0.55);
            Scalar3 f = box.makeFraction(vin, gw);
            Scalar3 mi = box.minImage(v);
            // wrap: point up to 1.4 half-lengths out
            Scalar3 w = make_scalar3(v.x * 0.75, v.y * 0.75, v.z * 0.75);
            Scalar3 w_in = w;
            int3 img = make_int3(c % 3 - 1, 0, 2);
            box.wrap(w, img);
            printf("  {\"v\":");
            p3(v);
            printf(",\"ghost_width\":");
            p3(gw);
            printf(",\"frac_in\":");
            p3(vin);
            printf(",\"frac\":");
            p3(f);
            printf(",\"min_image\":");
            p3(mi);
            printf(",\"wrap_in\":");
            p3(w_in);
            printf(",\"wrap_img_in\":[%d,%d,%d],\"wrap_out\":", c % 3 - 1, 0, 2);
            p3(w);
            printf(",\"wrap_img_out\":[%d,%d,%d]}%s\n", img.x, img.y, img.z, c + 1 < n_cases ? "," : "");
            }
        printf(" ]}%s\n", b + 1 < specs.size() ? "," : "");
        }
    printf("],\n");

    // ---------------- indexers ----------------
    printf("\"index2d\": [");
    for (int i = 0; i < 16; i++)
        {
        unsigned int w = 1 + (unsigned)(next_u64() % 97), h = 1 + (unsigned)(next_u64() % 53);
        unsigned int a = (unsigned)(next_u64() % w), c = (unsigned)(next_u64() % h);
        Index2D idx(w, h);
        printf("[%u,%u,%u,%u,%u,%u]%s", w, h, a, c, idx(a, c), idx.getNumElements(), i < 15 ? "," : "");
        }
    printf("],\n\"index3d\": [");
    for (int i = 0; i < 16; i++)
        {
        unsigned int w = 1 + (unsigned)(next_u64() % 47), h = 1 + (unsigned)(next_u64() % 46),
                     d = 1 + (unsigned)(next_u64() % 23);
        unsigned int a = (unsigned)(next_u64() % w), c = (unsigned)(next_u64() % h),
                     e = (unsigned)(next_u64() % d);
        Index3D idx(w, h, d);
        printf("[%u,%u,%u,%u,%u,%u,%u,%u]%s", w, h, d, a, c, e, idx(a, c, e), idx.getNumElements(),
               i < 15 ? "," : "");
        }
    printf("],\n");

    // ---------------- FENE + WCA bond evaluator (last: the sections above keep their random inputs) ----
    printf("\"fene\": [\n");
    const int n_fene = 300;
    for (int i = 0; i < n_fene; i++)
        {
        EvaluatorBondFENE::param_type p;
        p.k = (i % 29 == 0) ? 0.0 : uni(1.0, 40.0);
        p.r_0 = uni(1.2, 2.5);
        const double sigma = uni(0.7, 1.3), eps = (i % 31 == 0) ? 0.0 : uni(0.2, 2.0);
        p.sigma_6 = sigma * sigma * sigma * sigma * sigma * sigma;
        p.epsilon_x_4 = 4.0 * eps;
        p.delta = (i % 3 == 0) ? 0.0 : uni(-0.5, 0.3);
        // bond lengths from deep inside the WCA core to beyond r_0 (evaluated == false there)
        const double r = uni(0.75, 1.15) * ((i % 17 == 0) ? p.r_0 + p.delta : 1.0) + ((p.delta > 0) ? p.delta : 0.0);
        EvaluatorBondFENE ev(r * r, p);
        Scalar f = 0, e = 0;
        bool ok = ev.evalForceAndEnergy(f, e);
        printf("{\"k\":");
        pd(p.k);
        printf(",\"r_0\":");
        pd(p.r_0);
        printf(",\"sigma_6\":");
        pd(p.sigma_6);
        printf(",\"epsilon_x_4\":");
        pd(p.epsilon_x_4);
        printf(",\"delta\":");
        pd(p.delta);
        printf(",\"rsq\":");
        pd(r * r);
        printf(",\"evaluated\":%d,\"force_divr\":", (int)ok);
        pd(f);
        printf(",\"bond_eng\":");
        pd(e);
        printf("}%s\n", i + 1 < n_fene ? "," : "");
        }
    printf("]\n}\n");
    return 0;
    }
