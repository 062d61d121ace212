// hoomd/md/PotentialPair.h -- PotentialPair<evaluator>: the host side every pair potential shares
// (hoomd/md/PotentialPair.h:80-1036 of the reference): per-type-pair parameters, r_cut and r_on tables, shift mode,
// tail corrections from the evaluator's integrals, host-side pair energies between two sets of particles, and the
// Python export.  The force computation itself exists on the GPU only: PotentialPairGPU<evaluator>
// (hoomd/md/PotentialPairGPU.h) derives from this class and launches the kernels; an object of this class alone refuses
// to compute.
//
// Header-only so that a plugin built outside the source tree can instantiate it for its own evaluator; it needs the
// host classes of hoomd_host.h, which the module hoomd_amd/_hoomd_amd exports.
#ifndef HOOMD_AMD_POTENTIAL_PAIR_H
#define HOOMD_AMD_POTENTIAL_PAIR_H

#include "hoomd/HOOMDMath.h"
#include "hoomd_host.h"

#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

namespace hoomd
    {
namespace md
    {
using hoomd_amd::ForceCompute;
using hoomd_amd::SystemDefinition;
typedef hoomd_amd::NeighborListGPUBinned NeighborList;

template<class evaluator> class PotentialPair : public ForceCompute
    {
    public:
    typedef typename evaluator::param_type param_type;
    enum energyShiftMode
        {
        no_shift = 0,
        shift,
        xplor
        };

    PotentialPair(std::shared_ptr<SystemDefinition> sysdef, std::shared_ptr<NeighborList> nlist)
        : ForceCompute(sysdef), m_nlist(nlist), m_ntypes(m_pdata->getNTypes())
        {
        const size_t np = (size_t)m_ntypes * m_ntypes;
        m_params.assign(np, param_type {});
        m_rcutsq.assign(np, 0.0);
        m_ronsq.assign(np, 0.0);
        m_r_cut_nlist = std::make_shared<std::vector<Scalar>>(np, 0.0);
        m_nlist->addRCutMatrix(m_r_cut_nlist); // hoomd/md/PotentialPair.h:369-371
        }
    virtual ~PotentialPair()
        {
        if (m_nlist)
            m_nlist->removeRCutMatrix(m_r_cut_nlist);
        }

    // hoomd/md/PotentialPair.h:467-482 (symmetric assignment)
    virtual void setParams(unsigned int typ1, unsigned int typ2, const param_type& p)
        {
        validateTypes(typ1, typ2, "setting params");
        m_params[typ2 * m_ntypes + typ1] = p;
        m_params[typ1 * m_ntypes + typ2] = p;
        m_tables_dirty = true;
        }
    void setParamsPython(pybind11::tuple typ, pybind11::dict params)
        {
        setParams(typ[0].cast<unsigned int>(), typ[1].cast<unsigned int>(), param_type(params, false));
        }
    pybind11::dict getParams(pybind11::tuple typ)
        {
        const unsigned int typ1 = typ[0].cast<unsigned int>(), typ2 = typ[1].cast<unsigned int>();
        validateTypes(typ1, typ2, "getting params");
        return m_params[typ2 * m_ntypes + typ1].asDict();
        }
    // hoomd/md/PotentialPair.h:512-539
    virtual void setRcut(unsigned int typ1, unsigned int typ2, Scalar rcut)
        {
        validateTypes(typ1, typ2, "setting r_cut");
        m_rcutsq[typ2 * m_ntypes + typ1] = m_rcutsq[typ1 * m_ntypes + typ2] = rcut * rcut;
        (*m_r_cut_nlist)[typ2 * m_ntypes + typ1] = (*m_r_cut_nlist)[typ1 * m_ntypes + typ2] = rcut;
        m_nlist->notifyRCutMatrixChange();
        m_tables_dirty = true;
        }
    void setRCutPython(pybind11::tuple types, Scalar r_cut)
        {
        setRcut(types[0].cast<unsigned int>(), types[1].cast<unsigned int>(), r_cut);
        }
    Scalar getRCut(pybind11::tuple types)
        {
        const unsigned int typ1 = types[0].cast<unsigned int>(), typ2 = types[1].cast<unsigned int>();
        validateTypes(typ1, typ2, "getting r_cut");
        return (*m_r_cut_nlist)[typ2 * m_ntypes + typ1];
        }
    // hoomd/md/PotentialPair.h:553-580
    virtual void setRon(unsigned int typ1, unsigned int typ2, Scalar ron)
        {
        validateTypes(typ1, typ2, "setting r_on");
        m_ronsq[typ2 * m_ntypes + typ1] = m_ronsq[typ1 * m_ntypes + typ2] = ron * ron;
        m_tables_dirty = true;
        }
    void setROnPython(pybind11::tuple types, Scalar r_on)
        {
        setRon(types[0].cast<unsigned int>(), types[1].cast<unsigned int>(), r_on);
        }
    Scalar getROn(pybind11::tuple types)
        {
        const unsigned int typ1 = types[0].cast<unsigned int>(), typ2 = types[1].cast<unsigned int>();
        validateTypes(typ1, typ2, "getting r_on");
        return std::sqrt(m_ronsq[typ2 * m_ntypes + typ1]);
        }
    void setShiftMode(energyShiftMode mode)
        {
        m_shift_mode = mode;
        }
    energyShiftMode getShiftMode() const
        {
        return m_shift_mode;
        }
    void setTailCorrectionEnabled(bool enable)
        {
        m_tail = enable;
        }
    bool getTailCorrectionEnabled() const
        {
        return m_tail;
        }
    std::shared_ptr<NeighborList> getNeighborList() const
        {
        return m_nlist;
        }

    //! U = sum over i in tags1, j in tags2 of V_ij (hoomd/md/PotentialPair.h:833-991): a host-side double loop over
    //! the two sets with the minimum image in the global box, the evaluator run by the host compiler
    Scalar computeEnergyBetweenSets(const std::vector<unsigned int>& tags1, const std::vector<unsigned int>& tags2);

    protected:
    // no CPU force loop in this build, by design: forces come from PotentialPairGPU
    void computeForces(uint64_t timestep) override
        {
        throw std::runtime_error("PotentialPair<" + std::string(evaluator::getName()) + ">: forces are computed on the GPU only; construct the class "
                                 "exported by export_PotentialPairGPU");
        }
    void validateTypes(unsigned int typ1, unsigned int typ2, const std::string& action) const
        {
        if (typ1 >= m_ntypes || typ2 >= m_ntypes)
            throw std::invalid_argument("Error " + action + " in PotentialPair: pair." + std::string(evaluator::getName()) + ": invalid type");
        }
    //! hoomd/md/PotentialPair.h:255-341: corrections from the GLOBAL number of particles of every type
    //! (:283-337, counted once and again only when the particle set is replaced; summed over ranks under domain
    //! decomposition as in :430), added on rank 0 only so that the sum over ranks holds them once
    void computeTailCorrection();
    std::vector<double> m_type_count;
    uint64_t m_type_count_generation = ~0ull;

    std::shared_ptr<NeighborList> m_nlist;
    unsigned int m_ntypes;
    energyShiftMode m_shift_mode = no_shift;
    bool m_tail = false, m_tables_dirty = true;
    std::vector<param_type> m_params;
    std::vector<Scalar> m_rcutsq, m_ronsq;
    std::shared_ptr<std::vector<Scalar>> m_r_cut_nlist;
    };

template<class evaluator>
Scalar PotentialPair<evaluator>::computeEnergyBetweenSets(const std::vector<unsigned int>& tags1, const std::vector<unsigned int>& tags2)
    {
    using namespace hoomd_amd;
    if (tags1.empty() || tags2.empty())
        return 0;
    if (m_sysdef->isDomainDecomposed())
        throw std::runtime_error("computeEnergyBetweenSets is not available under domain decomposition");
    const unsigned int N = m_pdata->getN(), n_tags = m_pdata->getNGlobal();
    std::vector<Scalar4> pos(N);
    std::vector<unsigned int> rtag(n_tags);
    hipCheck(hipMemcpy(pos.data(), m_pdata->pos(), sizeof(Scalar4) * N, hipMemcpyDeviceToHost), "pos D2H");
    hipCheck(hipMemcpy(rtag.data(), m_pdata->rtag(), sizeof(unsigned int) * n_tags, hipMemcpyDeviceToHost), "rtag D2H");
    const hmd_box_t b = m_pdata->getGlobalBox().b;
    Scalar energy = 0;
    for (unsigned int t1 : tags1)
        {
        if (t1 >= n_tags)
            throw std::invalid_argument("computeEnergyBetweenSets: tag out of range");
        const unsigned int i = rtag[t1];
        if (i >= N)
            continue;
        unsigned int typei;
        std::memcpy(&typei, &pos[i].w, sizeof(unsigned int));
        for (unsigned int t2 : tags2)
            {
            if (t2 >= n_tags)
                throw std::invalid_argument("computeEnergyBetweenSets: tag out of range");
            const unsigned int j = rtag[t2];
            if (j >= N)
                continue;
            Scalar dx = pos[i].x - pos[j].x, dy = pos[i].y - pos[j].y, dz = pos[i].z - pos[j].z;
            // minimum image (hoomd/BoxDim.h:301-375)
            if (b.periodic[2])
                {
                const Scalar img = std::rint(dz * b.Linv[2]);
                dz -= b.L[2] * img;
                dy -= b.L[2] * b.yz * img;
                dx -= b.L[2] * b.xz * img;
                }
            if (b.periodic[1])
                {
                const Scalar img = std::rint(dy * b.Linv[1]);
                dy -= b.L[1] * img;
                dx -= b.L[1] * b.xy * img;
                }
            if (b.periodic[0])
                dx -= b.L[0] * std::rint(dx * b.Linv[0]);
            const Scalar rsq = dx * dx + dy * dy + dz * dz;
            unsigned int typej;
            std::memcpy(&typej, &pos[j].w, sizeof(unsigned int));
            const size_t tp = (size_t)typej * m_ntypes + typei;
            const Scalar rcutsq = m_rcutsq[tp], ronsq = m_shift_mode == xplor ? m_ronsq[tp] : 0.0;
            const bool energy_shift = m_shift_mode == shift || (m_shift_mode == xplor && ronsq > rcutsq);
            Scalar force_divr = 0, pair_eng = 0;
            evaluator eval(rsq, rcutsq, m_params[tp]);
            if (!eval.evalForceAndEnergy(force_divr, pair_eng, energy_shift))
                continue;
            if (m_shift_mode == xplor && rsq >= ronsq && rsq < rcutsq)
                {
                const Scalar d = rcutsq - ronsq, m = rsq - rcutsq;
                pair_eng *= m * m * (rcutsq + 2.0 * rsq - 3.0 * ronsq) / (d * d * d);
                }
            energy += pair_eng;
            }
        }
    return energy;
    }

template<class evaluator> void PotentialPair<evaluator>::computeTailCorrection()
    {
    using namespace hoomd_amd;
    if (!m_tail)
        return;
    if (m_shift_mode != no_shift)
        throw std::runtime_error("Pair potential shift mode must be \"none\" to apply tail corrections.");
    const bool decomposed = m_sysdef->isDomainDecomposed();
    if (m_type_count_generation != m_pdata->getCompositionGeneration() || m_type_count.size() != m_ntypes)
        {
        const unsigned int n = m_pdata->getN();
        std::vector<Scalar4> pos(n);
        if (n)
            hipCheck(hipMemcpy(pos.data(), m_pdata->pos(), sizeof(Scalar4) * n, hipMemcpyDeviceToHost), "pos D2H");
        m_type_count.assign(m_ntypes, 0.0);
        for (auto& p : pos)
            {
            unsigned int t;
            std::memcpy(&t, &p.w, sizeof(unsigned int));
            m_type_count[t] += 1.0;
            }
        if (decomposed)
            m_sysdef->getCommunicator()->allReduceSum(m_type_count.data(), (int)m_ntypes);
        m_type_count_generation = m_pdata->getCompositionGeneration();
        }
    const std::vector<double>& count = m_type_count;
    if (decomposed && m_sysdef->getCommunicator()->getRank() != 0)
        {
        m_external_energy = 0;
        for (auto& v : m_external_virial)
            v = 0;
        return;
        }
    const bool twod = m_sysdef->getNDimensions() == 2;
    const Scalar volume = m_pdata->getGlobalBox().getVolume(twod);
    m_external_energy = 0;
    for (unsigned int i = 0; i < m_ntypes; i++)
        for (unsigned int j = 0; j < m_ntypes; j++)
            {
            const Scalar rho_j = count[j] / volume;
            evaluator eval(Scalar(0.0), m_rcutsq[j * m_ntypes + i], m_params[j * m_ntypes + i]);
            m_external_energy += 2.0 * count[i] * M_PI * rho_j * eval.evalEnergyLRCIntegral();
            }
    if (m_pdata->getFlag(pressure_tensor))
        {
        for (auto& v : m_external_virial)
            v = 0;
        const int dimension = (int)m_sysdef->getNDimensions();
        for (unsigned int i = 0; i < m_ntypes; i++)
            for (unsigned int j = 0; j < m_ntypes; j++)
                {
                const Scalar rho_i = count[i] / volume, rho_j = count[j] / volume;
                evaluator eval(Scalar(0.0), m_rcutsq[j * m_ntypes + i], m_params[j * m_ntypes + i]);
                const Scalar dP = 4.0 / 6.0 * rho_i * rho_j * M_PI * eval.evalPressureLRCIntegral();
                const Scalar dW = dimension * volume * dP / 3.0;
                m_external_virial[0] += dW;
                m_external_virial[3] += dW;
                m_external_virial[5] += dW;
                }
        }
    }

namespace detail
    {
// hoomd/md/PotentialPair.h:1017-1036
template<class T> void export_PotentialPair(pybind11::module& m, const std::string& name)
    {
    // the base classes are registered by the core module
    pybind11::module::import("hoomd_amd._hoomd_amd");
    if (hoomd_amd::hostLayoutFingerprintOfTheseHeaders() != hoomd_amd::hostLayoutFingerprint())
        throw std::runtime_error("pair potential " + name + ": built against other hoomd_amd headers than the installed core module; rebuild it");
    pybind11::class_<PotentialPair<T>, ForceCompute, std::shared_ptr<PotentialPair<T>>>(m, name.c_str())
        .def(pybind11::init<std::shared_ptr<SystemDefinition>, std::shared_ptr<NeighborList>>())
        .def("setParams", &PotentialPair<T>::setParamsPython)
        .def("getParams", &PotentialPair<T>::getParams)
        .def("setRCut", &PotentialPair<T>::setRCutPython)
        .def("getRCut", &PotentialPair<T>::getRCut)
        .def("setROn", &PotentialPair<T>::setROnPython)
        .def("getROn", &PotentialPair<T>::getROn)
        .def_property("mode", [](PotentialPair<T>& self) { return (int)self.getShiftMode(); },
                      [](PotentialPair<T>& self, int mode)
                      {
                      if (mode < 0 || mode > 2)
                          throw std::invalid_argument("mode must be 0 (none), 1 (shift) or 2 (xplor)");
                      self.setShiftMode((typename PotentialPair<T>::energyShiftMode)mode);
                      })
        .def_property("tail_correction", &PotentialPair<T>::getTailCorrectionEnabled, &PotentialPair<T>::setTailCorrectionEnabled)
        .def("computeEnergyBetweenSets", &PotentialPair<T>::computeEnergyBetweenSets);
    }
    } // namespace detail
    } // namespace md
    } // namespace hoomd

#endif
