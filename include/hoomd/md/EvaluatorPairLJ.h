// hoomd/md/EvaluatorPairLJ.h -- the 12-6 Lennard-Jones evaluator, in the one-header form every evaluator has
// (contract: hoomd/md/EvaluatorPairLJ.h:35-81 of the reference; formula and cutoff test :162-211, tail-correction
// integrals :213-248, parameters from epsilon and sigma :119-133).
//
//   F/r = r^-2 r^-6 (12 lj1 r^-6 - 6 lj2),  E = r^-6 (lj1 r^-6 - lj2),  lj1 = 4 eps sigma^12, lj2 = 4 eps sigma^6
#ifndef HOOMD_AMD_EVALUATOR_PAIR_LJ_H
#define HOOMD_AMD_EVALUATOR_PAIR_LJ_H

#include "hoomd/HOOMDMath.h"
#include "hoomd_amd.h"

#ifdef __HIPCC__
#define DEVICE __device__
#define HOSTDEVICE __host__ __device__
#else
#define DEVICE
#define HOSTDEVICE
#endif

namespace hoomd
    {
namespace md
    {
class EvaluatorPairLJ
    {
    public:
    // the parameter block the C entry points take (sigma^6 and 4 epsilon), with the conversions of the Python side
    struct param_type : public hmd_lj_params_t
        {
        DEVICE void load_shared(char*& ptr, unsigned int& available_bytes) { }
        HOSTDEVICE void allocate_shared(char*& ptr, unsigned int& available_bytes) const { }
        HOSTDEVICE param_type()
            {
            sigma_6 = 0;
            epsilon_x_4 = 0;
            }
#ifndef __HIPCC__
        param_type(pybind11::dict v, bool managed = false)
            {
            const Scalar sigma = v["sigma"].cast<Scalar>(), epsilon = v["epsilon"].cast<Scalar>();
            sigma_6 = sigma * sigma * sigma * sigma * sigma * sigma;
            epsilon_x_4 = Scalar(4.0) * epsilon;
            }
        pybind11::dict asDict()
            {
            pybind11::dict v;
            v["sigma"] = std::pow(sigma_6, 1. / 6.);
            v["epsilon"] = epsilon_x_4 / 4.0;
            return v;
            }
#endif
        };

    DEVICE EvaluatorPairLJ(Scalar _rsq, Scalar _rcutsq, const param_type& _params)
        : rsq(_rsq), rcutsq(_rcutsq), lj1(_params.epsilon_x_4 * _params.sigma_6 * _params.sigma_6), lj2(_params.epsilon_x_4 * _params.sigma_6)
        {
        }
    HOSTDEVICE static constexpr bool needsCharge()
        {
        return false;
        }
    DEVICE void setCharge(Scalar qi, Scalar qj) { }

    DEVICE bool evalForceAndEnergy(Scalar& force_divr, Scalar& pair_eng, bool energy_shift)
        {
        if (!(rsq < rcutsq && lj1 != 0.0))
            return false;
        const Scalar r2inv = fast::rcp(rsq);
        const Scalar r6inv = r2inv * r2inv * r2inv;
        force_divr = r2inv * r6inv * (12.0 * lj1 * r6inv - 6.0 * lj2);
        pair_eng = r6inv * (lj1 * r6inv - lj2);
        if (energy_shift)
            {
            const Scalar rc2inv = fast::rcp(rcutsq);
            const Scalar rc6inv = rc2inv * rc2inv * rc2inv;
            pair_eng -= rc6inv * (lj1 * rc6inv - lj2);
            }
        return true;
        }

#ifdef __HIPCC__
    // Optional extension of the contract (detected by the kernels, not required from an evaluator): force and energy of
    // the unshifted potential without a branch.  Outside the cutoff rsq is replaced by about 1e300 (only its high word
    // needs replacing for that), so r^-6 underflows to zero and both results vanish; eps = 0 gives lj1 = lj2 = 0 and
    // zero results as well.
    __device__ __forceinline__ void evalUnshiftedBranchFree(Scalar& force_divr, Scalar& pair_eng) const
        {
        const int hi = (rsq < rcutsq) ? __double2hiint(rsq) : 0x7e37e43c;
        const Scalar r2inv = fast::rcp(__hiloint2double(hi, __double2loint(rsq)));
        const Scalar r6inv = r2inv * r2inv * r2inv;
        force_divr = r2inv * r6inv * (12.0 * lj1 * r6inv - 6.0 * lj2);
        pair_eng = r6inv * (lj1 * r6inv - lj2);
        }

    // Optional extension, four pairs at once: ONE reciprocal serves all four, 1/a = (b c d) / (a b c d) -- the hardware
    // reciprocal runs at a quarter of the multiply rate, three more of them cost more than the nine multiplications
    // that replace them.  Every r^-2 carries the refined reciprocal's error plus three roundings (about 3e-15
    // relative).  Outside the cutoff the high word of r^-2 is cleared, which leaves a subnormal whose cube is zero.
    // The arguments must be positive and their product finite (squared distances up to 1e60 are).
    __device__ __forceinline__ static void evalUnshiftedBranchFree4(const Scalar (&rsq)[4], const Scalar (&rcutsq)[4], const param_type* const (&prm)[4],
                                                                    Scalar (&force_divr)[4], Scalar (&pair_eng)[4])
        {
        const Scalar ab = rsq[0] * rsq[1], cd = rsq[2] * rsq[3];
        const Scalar inv = fast::rcp(ab * cd);
        const Scalar iab = inv * cd, icd = inv * ab;
        const Scalar r2[4] = {iab * rsq[1], iab * rsq[0], icd * rsq[3], icd * rsq[2]};
#pragma unroll
        for (int m = 0; m < 4; m++)
            {
            const Scalar lj2 = prm[m]->epsilon_x_4 * prm[m]->sigma_6, lj1 = lj2 * prm[m]->sigma_6;
            const int hi = (rsq[m] < rcutsq[m]) ? __double2hiint(r2[m]) : 0;
            const Scalar r2inv = __hiloint2double(hi, __double2loint(r2[m]));
            const Scalar r6inv = r2inv * r2inv * r2inv;
            force_divr[m] = r2inv * r6inv * (12.0 * lj1 * r6inv - 6.0 * lj2);
            pair_eng[m] = r6inv * (lj1 * r6inv - lj2);
            }
        }
#endif

    // integrals of the tail corrections from r_cut to infinity (the evaluator is constructed with rsq = 0 for them)
    DEVICE Scalar evalPressureLRCIntegral()
        {
        if (rcutsq == 0)
            return 0;
        const Scalar rc3inv = 1.0 / fast::pow(rcutsq, 1.5), rc9inv = rc3inv * rc3inv * rc3inv;
        return lj1 * 4.0 / 3.0 * rc9inv - lj2 * 2.0 * rc3inv;
        }
    DEVICE Scalar evalEnergyLRCIntegral()
        {
        if (rcutsq == 0)
            return 0;
        const Scalar rc3inv = 1.0 / fast::pow(rcutsq, 1.5), rc9inv = rc3inv * rc3inv * rc3inv;
        return lj1 / 9.0 * rc9inv - lj2 / 3.0 * rc3inv;
        }

#ifndef __HIPCC__
    static std::string getName()
        {
        return std::string("lj");
        }
    std::string getShapeSpec() const
        {
        throw std::runtime_error("Shape definition not supported for this pair potential.");
        }
#endif

    protected:
    Scalar rsq, rcutsq, lj1, lj2;
    };
    } // namespace md
    } // namespace hoomd

#undef DEVICE
#undef HOSTDEVICE
#endif
