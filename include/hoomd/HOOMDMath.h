// hoomd/HOOMDMath.h -- the numeric vocabulary an evaluator header is written in.
//
// A pair-potential evaluator (hoomd/md/EvaluatorPairLJ.h:35-81 documents the contract,
// example_plugins/pair_plugin/EvaluatorPairExample.h:33-167 is the reference's minimal one) is ONE header that the HIP
// compiler reads for the device code of the force kernels and the host compiler reads for the Python bindings, the
// tail-correction integrals and the host-side pair energies.  That header includes this file and nothing else of the
// framework, under the include path the reference uses, so an evaluator written for the reference compiles here
// unchanged.  What it may rely on:
//   Scalar (double), the fast:: functions, ENABLE_HIP, HOOMD_LONGREAL_SIZE, and, in the host pass, pybind11 and
//   <string>/<stdexcept> already included.
#ifndef HOOMD_AMD_HOOMDMATH_H
#define HOOMD_AMD_HOOMDMATH_H

#include <cmath>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define HOOMD_AMD_HD __host__ __device__
#else
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <stdexcept>
#include <string>
#define HOOMD_AMD_HD
#endif

#ifndef ENABLE_HIP
#define ENABLE_HIP 1
#endif
// every real number of this build is a double
#define HOOMD_LONGREAL_SIZE 64
#define HOOMD_SHORTREAL_SIZE 64

namespace hoomd
    {
typedef double Scalar;
typedef double LongReal;
typedef double ShortReal;

namespace fast
    {
// Reciprocal of a positive, finite double.  Device: the hardware estimate (v_rcp_f64, measured relative error 4.6e-8 on
// MI355X) refined by one Newton step, which leaves 2.2e-15 (measured over 4M inputs in [0.3, 9.3]) -- nine orders of
// magnitude inside the 1e-6 force tolerance, at a quarter of the instruction count of an IEEE-rounded division.
HOOMD_AMD_HD inline Scalar rcp(Scalar x)
    {
#ifdef __HIP_DEVICE_COMPILE__
    const Scalar r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
#else
    return 1.0 / x;
#endif
    }
HOOMD_AMD_HD inline Scalar sqrt(Scalar x)
    {
    return ::sqrt(x);
    }
HOOMD_AMD_HD inline Scalar rsqrt(Scalar x)
    {
    return 1.0 / ::sqrt(x);
    }
HOOMD_AMD_HD inline Scalar pow(Scalar x, Scalar y)
    {
    return ::pow(x, y);
    }
HOOMD_AMD_HD inline Scalar exp(Scalar x)
    {
    return ::exp(x);
    }
HOOMD_AMD_HD inline Scalar log(Scalar x)
    {
    return ::log(x);
    }
HOOMD_AMD_HD inline Scalar sin(Scalar x)
    {
    return ::sin(x);
    }
HOOMD_AMD_HD inline Scalar cos(Scalar x)
    {
    return ::cos(x);
    }
HOOMD_AMD_HD inline Scalar tanh(Scalar x)
    {
    return ::tanh(x);
    }
HOOMD_AMD_HD inline Scalar erfc(Scalar x)
    {
    return ::erfc(x);
    }
    } // namespace fast
    } // namespace hoomd

#endif
