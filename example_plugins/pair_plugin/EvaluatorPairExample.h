// EvaluatorPairExample.h -- everything the framework needs to know about one pair potential, in one header.
//
// Harmonic well about sigma, cut at r_cut:   V(r) = k/2 (sigma - r)^2,   F(r)/r = k (sigma - r) / r
//
// The HIP compiler reads this header for the force kernels (PotentialPairExampleGPUKernel.hip), the host compiler
// for the Python class, the host-side pair energies and the tail-correction integrals (module.cc).  The shape is the
// reference's evaluator contract (hoomd/md/EvaluatorPairLJ.h:35-81; its own example is
// example_plugins/pair_plugin/EvaluatorPairExample.h:33-167, whose potential this one repeats so that the reference's
// test values apply): a header written for the reference compiles against this framework unchanged, and this one
// against the reference.
#ifndef PAIR_PLUGIN_EVALUATOR_PAIR_EXAMPLE_H
#define PAIR_PLUGIN_EVALUATOR_PAIR_EXAMPLE_H

#include "hoomd/HOOMDMath.h"

// device functions for the HIP compiler, plain functions for the host compiler
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
class EvaluatorPairExample
    {
    public:
    // One of these per type pair.  Plain data: the kernels copy the table into LDS, so the two shared-memory hooks
    // of the contract have nothing to do.
    struct param_type
        {
        Scalar k;     // stiffness
        Scalar sigma; // position of the minimum

        DEVICE void load_shared(char*& ptr, unsigned int& available_bytes) { }
        HOSTDEVICE void allocate_shared(char*& ptr, unsigned int& available_bytes) const { }

#ifndef __HIPCC__
        param_type() : k(0), sigma(0) { }
        // from and to the dictionary of the Python side: pair.params[(A, B)] = dict(k=..., sigma=...)
        param_type(pybind11::dict v, bool managed = false) : k(v["k"].cast<Scalar>()), sigma(v["sigma"].cast<Scalar>()) { }
        pybind11::dict asDict()
            {
            pybind11::dict v;
            v["k"] = k;
            v["sigma"] = sigma;
            return v;
            }
#endif
        } __attribute__((aligned(16)));

    DEVICE EvaluatorPairExample(Scalar _rsq, Scalar _rcutsq, const param_type& _params)
        : rsq(_rsq), rcutsq(_rcutsq), k(_params.k), sigma(_params.sigma)
        {
        }

    // no charges in this potential
    DEVICE static bool needsCharge()
        {
        return false;
        }
    DEVICE void setCharge(Scalar qi, Scalar qj) { }

    // force_divr = F(r)/r and pair_eng = V(r) [- V(r_cut) when energy_shift]; false beyond the cutoff
    DEVICE bool evalForceAndEnergy(Scalar& force_divr, Scalar& pair_eng, bool energy_shift)
        {
        if (!(rsq < rcutsq))
            return false;
        const Scalar r = fast::sqrt(rsq);
        const Scalar stretch = sigma - r;
        force_divr = k * stretch / r;
        pair_eng = Scalar(0.5) * k * stretch * stretch;
        if (energy_shift)
            {
            const Scalar at_cut = sigma - fast::sqrt(rcutsq);
            pair_eng -= Scalar(0.5) * k * at_cut * at_cut;
            }
        return true;
        }

    // a potential that ends at its cutoff has no tail to correct for
    DEVICE Scalar evalPressureLRCIntegral()
        {
        return 0;
        }
    DEVICE Scalar evalEnergyLRCIntegral()
        {
        return 0;
        }

#ifndef __HIPCC__
    static std::string getName()
        {
        return std::string("example_pair");
        }
    std::string getShapeSpec() const
        {
        throw std::runtime_error("Shape definition not supported for this pair potential.");
        }
#endif

    protected:
    Scalar rsq, rcutsq, k, sigma;
    };
    } // namespace md
    } // namespace hoomd

#endif
