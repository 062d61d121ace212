// evaluators.cuh -- pair-potential evaluators usable with pair_force.cuh.
//
// The pair-force kernel is a template over a class that follows the reference's evaluator
// contract (documented in hoomd/md/EvaluatorPairLJ.h:35-81 and shown by
// example_plugins/pair_plugin/EvaluatorPairExample.h:33-167):
//
//   struct param_type            plain data, one per type pair, copied to LDS by the kernel
//   Evaluator(rsq, rcutsq, const param_type&)
//   static bool needsCharge();   void setCharge(qi, qj);
//   bool evalForceAndEnergy(double& force_divr, double& pair_eng, bool energy_shift);
//
// A plugin adds a class of this shape and one translation unit that instantiates
// hmd::launch_pair_forces<Evaluator> behind an extern "C" symbol (see pair_force_example.hip).
#ifndef HMD_EVALUATORS_CUH
#define HMD_EVALUATORS_CUH

#include "hoomd_amd.h"
#include <hip/hip_runtime.h>
#include <utility>

namespace hmd
    {
// Reciprocal of a positive, finite double: the hardware estimate (v_rcp_f64, measured relative
// error 4.6e-8 on MI355X) refined by one Newton step, which leaves 2.2e-15 (measured over 4M
// inputs in [0.3, 9.3], scripts/exp/rcp_accuracy.hip) -- nine orders of magnitude inside the 1e-6
// force tolerance, at a quarter of the instruction count of an IEEE-rounded division.
__device__ __forceinline__ double fast_rcp(double x)
    {
    const double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    }

// 12-6 Lennard-Jones.  Formula and cutoff test as hoomd/md/EvaluatorPairLJ.h:162-211:
//   F/r = r^-2 r^-6 (12 lj1 r^-6 - 6 lj2),  E = r^-6 (lj1 r^-6 - lj2),  lj1 = 4 eps sigma^12, lj2 = 4 eps sigma^6
class EvaluatorPairLJ
    {
    public:
    typedef hmd_lj_params_t param_type;

    __device__ __forceinline__ EvaluatorPairLJ(double rsq_, double rcutsq_, const param_type& p)
        : rsq(rsq_), rcutsq(rcutsq_), lj1(p.epsilon_x_4 * p.sigma_6 * p.sigma_6), lj2(p.epsilon_x_4 * p.sigma_6)
        {
        }
    __host__ __device__ static constexpr bool needsCharge()
        {
        return false;
        }
    __device__ __forceinline__ void setCharge(double, double) { }

    __device__ __forceinline__ bool evalForceAndEnergy(double& force_divr, double& pair_eng, bool energy_shift) const
        {
        if (!(rsq < rcutsq && lj1 != 0.0))
            return false;
        const double r2inv = fast_rcp(rsq);
        const double r6inv = r2inv * r2inv * r2inv;
        force_divr = r2inv * r6inv * (12.0 * lj1 * r6inv - 6.0 * lj2);
        pair_eng = r6inv * (lj1 * r6inv - lj2);
        if (energy_shift)
            {
            const double rc2inv = fast_rcp(rcutsq);
            const double rc6inv = rc2inv * rc2inv * rc2inv;
            pair_eng -= rc6inv * (lj1 * rc6inv - lj2);
            }
        return true;
        }

    // Optional extension of the contract (detected by the kernels, not required from a plugin):
    // force and energy without a branch for the unshifted potential.  Outside the cutoff rsq is
    // replaced by 1e300, so r^-6 underflows to zero and both results vanish; eps = 0 gives
    // lj1 = lj2 = 0 and zero results as well.
    __device__ __forceinline__ void evalUnshiftedBranchFree(double& force_divr, double& pair_eng) const
        {
        // outside the cutoff rsq becomes about 1e300 and every term below underflows to zero; only the
        // high word needs replacing for that
        const int hi = (rsq < rcutsq) ? __double2hiint(rsq) : 0x7e37e43c;
        const double r2inv = fast_rcp(__hiloint2double(hi, __double2loint(rsq)));
        const double r6inv = r2inv * r2inv * r2inv;
        force_divr = r2inv * r6inv * (12.0 * lj1 * r6inv - 6.0 * lj2);
        pair_eng = r6inv * (lj1 * r6inv - lj2);
        }

    private:
    double rsq, rcutsq, lj1, lj2;
    };

// does the evaluator offer evalUnshiftedBranchFree()?
template<class Ev, class = void> struct has_branch_free
    {
    static constexpr bool value = false;
    };
template<class Ev> struct has_branch_free<Ev, decltype(std::declval<const Ev&>().evalUnshiftedBranchFree(std::declval<double&>(), std::declval<double&>()))>
    {
    static constexpr bool value = true;
    };

// Harmonic repulsion/attraction about sigma, the plugin example
// (example_plugins/pair_plugin/EvaluatorPairExample.h:83-133): E = k (sigma - r)^2 / 2, F/r = k (sigma - r)/r
class EvaluatorPairExample
    {
    public:
    typedef hmd_example_params_t param_type;

    __device__ __forceinline__ EvaluatorPairExample(double rsq_, double rcutsq_, const param_type& p)
        : rsq(rsq_), rcutsq(rcutsq_), k(p.k), sigma(p.sigma)
        {
        }
    __host__ __device__ static constexpr bool needsCharge()
        {
        return false;
        }
    __device__ __forceinline__ void setCharge(double, double) { }

    __device__ __forceinline__ bool evalForceAndEnergy(double& force_divr, double& pair_eng, bool energy_shift) const
        {
        if (!(rsq < rcutsq))
            return false;
        const double r = sqrt(rsq);
        const double overlap = sigma - r;
        force_divr = k * overlap / r;
        pair_eng = 0.5 * k * overlap * overlap;
        if (energy_shift)
            {
            const double co = sigma - sqrt(rcutsq);
            pair_eng -= 0.5 * k * co * co;
            }
        return true;
        }

    private:
    double rsq, rcutsq, k, sigma;
    };
    } // namespace hmd

#endif
