// forces.cc -- ForceCompute, PotentialPairGPU<evaluator>, ComputeThermoGPU and the thermostats.
// Follows hoomd/ForceCompute.cc:32-110,511-523, hoomd/md/PotentialPair.h:255-341,349-580,
// hoomd/md/PotentialPairGPU.h:89-160, hoomd/md/ComputeThermo.cc:99-320, hoomd/md/Thermostat.h:337-420.
#include "hoomd_host.h"

#include <algorithm>
#include <cmath>
#include <random>

namespace hoomd_amd
    {
ForceCompute::ForceCompute(std::shared_ptr<SystemDefinition> sysdef) : Compute(sysdef)
    {
    reallocate();
    m_sort_slot = m_pdata->getParticleSortSignal().connect([this]() { m_particles_sorted = true; });
    m_maxn_slot = m_pdata->getMaxParticleNumberChangeSignal().connect([this]() { reallocate(); });
    }

ForceCompute::~ForceCompute()
    {
    m_pdata->getParticleSortSignal().disconnect(m_sort_slot);
    m_pdata->getMaxParticleNumberChangeSignal().disconnect(m_maxn_slot);
    }

void ForceCompute::reallocate()
    {
    const size_t n = std::max(m_pdata->getMaxN(), 1u);
    m_force.resize(n, false);
    m_virial_pitch = n;
    m_virial.resize(6 * n, false);
    m_particles_sorted = true; // contents are gone: recompute on the next request
    }

// hoomd/ForceCompute.cc:511-523
void ForceCompute::compute(uint64_t timestep)
    {
    if (m_particles_sorted || shouldCompute(timestep) || m_pdata->getFlags() != m_computed_flags)
        computeForces(timestep);
    m_particles_sorted = false;
    m_computed_flags = m_pdata->getFlags();
    }

std::vector<std::array<Scalar, 4>> ForceCompute::getForcesAndEnergies()
    {
    const unsigned int n = m_pdata->getN();
    auto f = m_force.toHost(n);
    std::vector<unsigned int> tag(n);
    if (n)
        hipCheck(hipMemcpy(tag.data(), m_pdata->tag(), sizeof(unsigned int) * n, hipMemcpyDeviceToHost), "tags D2H");
    std::vector<unsigned int> order(n);
    for (unsigned int i = 0; i < n; i++)
        order[i] = i;
    std::sort(order.begin(), order.end(), [&](unsigned int a, unsigned int b) { return tag[a] < tag[b]; });
    std::vector<std::array<Scalar, 4>> out(n);
    for (unsigned int k = 0; k < n; k++)
        out[k] = {f[order[k]].x, f[order[k]].y, f[order[k]].z, f[order[k]].w};
    return out;
    }

std::vector<std::array<Scalar, 6>> ForceCompute::getVirials()
    {
    const unsigned int n = m_pdata->getN();
    auto v = m_virial.toHost();
    std::vector<unsigned int> tag(n);
    if (n)
        hipCheck(hipMemcpy(tag.data(), m_pdata->tag(), sizeof(unsigned int) * n, hipMemcpyDeviceToHost), "tags D2H");
    std::vector<unsigned int> order(n);
    for (unsigned int i = 0; i < n; i++)
        order[i] = i;
    std::sort(order.begin(), order.end(), [&](unsigned int a, unsigned int b) { return tag[a] < tag[b]; });
    std::vector<std::array<Scalar, 6>> out(n);
    for (unsigned int k = 0; k < n; k++)
        for (unsigned int c = 0; c < 6; c++)
            out[k][c] = v[c * m_virial_pitch + order[k]];
    return out;
    }

// hoomd/ForceCompute.cc:196-226
Scalar ForceCompute::calcEnergySum()
    {
    const unsigned int n = m_pdata->getN();
    auto f = m_force.toHost(n);
    double pe = m_external_energy;
    for (unsigned int i = 0; i < n; i++)
        pe += f[i].w;
    return pe;
    }

// ---------------------------------------------------------------------------------------------
// hoomd/md/EvaluatorPairLJ.h:213-248
Scalar EvaluatorLJHost::energyLRC(Scalar rcutsq, const param_type& p)
    {
    if (rcutsq == 0)
        return 0;
    const Scalar lj1 = p.epsilon_x_4 * p.sigma_6 * p.sigma_6, lj2 = p.epsilon_x_4 * p.sigma_6;
    const Scalar rc3inv = 1.0 / std::pow(rcutsq, 1.5), rc9inv = rc3inv * rc3inv * rc3inv;
    return lj1 / 9.0 * rc9inv - lj2 / 3.0 * rc3inv;
    }

Scalar EvaluatorLJHost::pressureLRC(Scalar rcutsq, const param_type& p)
    {
    if (rcutsq == 0)
        return 0;
    const Scalar lj1 = p.epsilon_x_4 * p.sigma_6 * p.sigma_6, lj2 = p.epsilon_x_4 * p.sigma_6;
    const Scalar rc3inv = 1.0 / std::pow(rcutsq, 1.5), rc9inv = rc3inv * rc3inv * rc3inv;
    return lj1 * 4.0 / 3.0 * rc9inv - lj2 * 2.0 * rc3inv;
    }

template<class Ev>
PotentialPairGPU<Ev>::PotentialPairGPU(std::shared_ptr<SystemDefinition> sysdef, std::shared_ptr<NeighborListGPUBinned> nlist)
    : ForceCompute(sysdef), m_nlist(nlist), m_ntypes(m_pdata->getNTypes())
    {
    const size_t np = (size_t)m_ntypes * m_ntypes;
    m_params.assign(np, param_type {});
    m_rcutsq.assign(np, 0.0);
    m_ronsq.assign(np, 0.0);
    m_r_cut_nlist = std::make_shared<std::vector<Scalar>>(np, 0.0);
    m_nlist->addRCutMatrix(m_r_cut_nlist); // hoomd/md/PotentialPair.h:369-371
    // queue the force kernel behind the distance check, predicated on "list still valid": on the steps
    // without a rebuild (most of them) the GPU does not idle while the host reads the check's result
    m_nlist->setSpeculationHook([this](uint64_t timestep, const unsigned int* d_flag, unsigned int stale, unsigned long long* d_published,
                                       unsigned int stamp)
                                {
                                m_spec_tstep = timestep;
                                // the interior blocks of this step may already be running on the side stream
                                return launchKernel(d_flag, stale, d_published, stamp, m_interior_tstep == timestep ? 2u : 0u);
                                });
    m_nlist->setInteriorHook([this](uint64_t timestep, const unsigned int* d_flag, unsigned int stale)
                             { return launchInterior(timestep, d_flag, stale); });
    m_d_params.resize(np);
    m_d_rcutsq.resize(np);
    m_d_ronsq.resize(np);
    }

template<class Ev> PotentialPairGPU<Ev>::~PotentialPairGPU()
    {
    if (m_nlist)
        {
        m_nlist->setSpeculationHook(nullptr);
        m_nlist->setInteriorHook(nullptr);
        m_nlist->removeRCutMatrix(m_r_cut_nlist);
        }
    }

// Queue the tile blocks that see no ghosts, predicated on the local result of the distance check, before the
// halo exchange is queued (on another stream): they run while the ghost positions travel.  When another rank
// votes for a rebuild their result is simply overwritten by the launch that follows the rebuild.
template<class Ev> bool PotentialPairGPU<Ev>::launchInterior(uint64_t timestep, const unsigned int* d_flag, unsigned int stale)
    {
    if (!m_nlist->tilesActive() || m_pdata->getN() == 0)
        return false;
    if (!m_nlist->getInteriorBlockWindow(timestep, m_win_lo, m_win_hi))
        return false;
    launchKernel(d_flag, stale, nullptr, 0, 1u);
    m_interior_tstep = timestep;
    return true;
    }

// hoomd/md/PotentialPair.h:467-482 (symmetric assignment)
template<class Ev> void PotentialPairGPU<Ev>::setParams(unsigned int typ1, unsigned int typ2, const param_type& p)
    {
    if (typ1 >= m_ntypes || typ2 >= m_ntypes)
        throw std::invalid_argument("pair." + std::string(Ev::getName()) + ": invalid type");
    m_params[typ2 * m_ntypes + typ1] = p;
    m_params[typ1 * m_ntypes + typ2] = p;
    m_tables_dirty = true;
    }

// hoomd/md/PotentialPair.h:512-539
template<class Ev> void PotentialPairGPU<Ev>::setRcut(unsigned int typ1, unsigned int typ2, Scalar rcut)
    {
    if (typ1 >= m_ntypes || typ2 >= m_ntypes)
        throw std::invalid_argument("pair." + std::string(Ev::getName()) + ": invalid type");
    m_rcutsq[typ2 * m_ntypes + typ1] = m_rcutsq[typ1 * m_ntypes + typ2] = rcut * rcut;
    (*m_r_cut_nlist)[typ2 * m_ntypes + typ1] = (*m_r_cut_nlist)[typ1 * m_ntypes + typ2] = rcut;
    m_nlist->notifyRCutMatrixChange();
    m_tables_dirty = true;
    }

// hoomd/md/PotentialPair.h:553-580
template<class Ev> void PotentialPairGPU<Ev>::setRon(unsigned int typ1, unsigned int typ2, Scalar ron)
    {
    if (typ1 >= m_ntypes || typ2 >= m_ntypes)
        throw std::invalid_argument("pair." + std::string(Ev::getName()) + ": invalid type");
    m_ronsq[typ2 * m_ntypes + typ1] = m_ronsq[typ1 * m_ntypes + typ2] = ron * ron;
    m_tables_dirty = true;
    }

template<class Ev> Scalar PotentialPairGPU<Ev>::getROn(unsigned int typ1, unsigned int typ2) const
    {
    return std::sqrt(m_ronsq[typ2 * m_ntypes + typ1]);
    }

// hoomd/md/PotentialPairGPU.h:89-160
template<class Ev> void PotentialPairGPU<Ev>::computeForces(uint64_t timestep)
    {
    // The speculation hook may already have queued this step's kernel: from inside compute() below, or,
    // under domain decomposition, from the communicator's look at the rebuild decision before this call.
    m_nlist->compute(timestep);
    if (m_nlist->getStorageMode() == NeighborListGPUBinned::half)
        throw std::runtime_error("PotentialPairGPU cannot handle a half neighborlist");
    // the speculative launch stands when the list was not rebuilt at this step; it is used once
    const bool queued = m_spec_tstep == timestep && m_nlist->getLastBuildTimestep() != timestep;
    m_spec_tstep = ~0ull;
    if (!queued)
        launchKernel(nullptr, 0);
    computeTailCorrection();
    }

template<class Ev>
bool PotentialPairGPU<Ev>::launchKernel(const unsigned int* d_skip_flag, unsigned int skip_value, unsigned long long* d_published, unsigned int stamp,
                                        unsigned int part)
    {
    if (m_tables_dirty)
        {
        m_d_params.upload(m_params.data(), m_params.size());
        m_d_rcutsq.upload(m_rcutsq.data(), m_rcutsq.size());
        m_d_ronsq.upload(m_ronsq.data(), m_ronsq.size());
        m_tables_dirty = false;
        }
    hmd_pair_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.d_force = m_force.get();
    a.d_virial = m_virial.get();
    a.virial_pitch = m_virial_pitch;
    a.N = m_pdata->getN();
    a.n_max = m_pdata->getMaxN();
    a.d_pos = m_pdata->pos();
    a.d_charge = m_pdata->charge();
    a.box = m_pdata->getBox().b;
    const bool tiles = m_nlist->tilesActive();
    a.d_n_neigh = m_nlist->getNNeighArray().get();
    a.d_nlist = tiles ? nullptr : m_nlist->getNListArray().get();
    a.d_head_list = m_nlist->getHeadList().get();
    a.d_rcutsq = m_d_rcutsq.get();
    a.d_ronsq = m_d_ronsq.get();
    a.size_neigh_list = tiles ? m_nlist->getTileNlist().size() : m_nlist->getNListArray().size();
    a.ntypes = m_ntypes;
    a.shift_mode = (unsigned int)m_shift_mode;
    a.compute_virial = m_pdata->getFlag(pressure_tensor) ? 1u : 0u;
    a.threads_per_particle = m_tpp;
    if (tiles)
        a.threads_per_particle = 0; // the tile kernel always puts four lanes on a home particle (its row layout is built for that)
    if (!tiles && m_tpp == 0)
        {
        // the reference tunes this at run time (hoomd/md/PotentialPairGPU.h:120-145); here: rows of at most 32
        // entries (short cutoffs, e.g. WCA) are walked by 4 lanes, longer rows by 8 (measured at 11 and 84 neighbors)
        unsigned int longest = 0;
        for (unsigned int n : m_nlist->getNmaxPerType())
            longest = std::max(longest, n);
        a.threads_per_particle = longest <= 32 ? 4u : 8u;
        }
    a.d_skip_flag = d_skip_flag;
    a.skip_value = skip_value;
    a.d_published = d_published;
    a.publish_stamp = stamp;
    if (part)
        {
        a.block_window_mode = part;
        for (int d = 0; d < 3; d++)
            {
            a.block_lo[d] = m_win_lo[d];
            a.block_hi[d] = m_win_hi[d];
            }
        }
    if (tiles && part != 1) // (the interior blocks read owned particles only, whose copies are current)
        m_nlist->refreshTilePositions();
    ScopedKernelTimer t(part == 1 ? "pair_force_interior" : "pair_force");
    if (tiles)
        {
        const hmd_tile_geom_t g = m_nlist->getTileGeom();
        abiCheck(Ev::launchTiles(&a, &g, m_nlist->getTileNlist().get(), m_d_params.get(), nullptr), "gpu_compute_pair_forces (tiles)");
        }
    else
        abiCheck(Ev::launch(&a, m_d_params.get(), nullptr), "gpu_compute_pair_forces");
    return a.N != 0; // the drivers launch nothing for an empty system
    }

// hoomd/md/PotentialPair.h:833-991
template<class Ev> Scalar PotentialPairGPU<Ev>::computeEnergyBetweenSets(const std::vector<unsigned int>& tags1, const std::vector<unsigned int>& tags2)
    {
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
            Scalar pair_eng = 0;
            if (!Ev::evalEnergy(rsq, rcutsq, m_params[tp], energy_shift, pair_eng))
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

// hoomd/md/PotentialPair.h:255-341 (single rank: the per-type counts come from the type ids)
template<class Ev> void PotentialPairGPU<Ev>::computeTailCorrection()
    {
    if (!m_tail)
        return;
    if (m_shift_mode != no_shift)
        throw std::runtime_error("Pair potential shift mode must be \"none\" to apply tail corrections.");
    const unsigned int n = m_pdata->getN();
    std::vector<Scalar4> pos(n);
    if (n)
        hipCheck(hipMemcpy(pos.data(), m_pdata->pos(), sizeof(Scalar4) * n, hipMemcpyDeviceToHost), "pos D2H");
    std::vector<unsigned int> count(m_ntypes, 0);
    for (auto& p : pos)
        {
        unsigned int t;
        std::memcpy(&t, &p.w, sizeof(unsigned int));
        count[t]++;
        }
    const bool twod = m_sysdef->getNDimensions() == 2;
    const Scalar volume = m_pdata->getGlobalBox().getVolume(twod);
    m_external_energy = 0;
    for (unsigned int i = 0; i < m_ntypes; i++)
        for (unsigned int j = 0; j < m_ntypes; j++)
            {
            const Scalar rho_j = count[j] / volume;
            m_external_energy += 2.0 * count[i] * M_PI * rho_j * Ev::energyLRC(m_rcutsq[j * m_ntypes + i], m_params[j * m_ntypes + i]);
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
                const Scalar dP = 4.0 / 6.0 * rho_i * rho_j * M_PI * Ev::pressureLRC(m_rcutsq[j * m_ntypes + i], m_params[j * m_ntypes + i]);
                const Scalar dW = dimension * volume * dP / 3.0;
                m_external_virial[0] += dW;
                m_external_virial[3] += dW;
                m_external_virial[5] += dW;
                }
        }
    }

template class PotentialPairGPU<EvaluatorLJHost>;
template class PotentialPairGPU<EvaluatorExampleHost>;

// ---------------------------------------------------------------------------------------------
ComputeThermoGPU::ComputeThermoGPU(std::shared_ptr<SystemDefinition> sysdef, std::shared_ptr<ParticleGroup> group) : Compute(sysdef), m_group(group)
    {
    m_out.resize(8);
    m_scratch.resize(hmd_thermo_scratch_bytes());
    }

void ComputeThermoGPU::compute(uint64_t timestep)
    {
    if (!shouldCompute(timestep))
        return;
    hmd_thermo_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.d_out = m_out.get();
    a.d_scratch = m_scratch.get();
    a.d_vel = m_pdata->vel();
    a.d_net_force = m_pdata->netForceForIntegration();
    a.d_net_virial = m_pdata->getFlag(pressure_tensor) ? m_pdata->netVirial() : nullptr;
    a.net_virial_pitch = m_pdata->netVirialPitch();
    a.d_group_members = m_group->getIndexArray();
    a.group_size = m_group->getNumMembers();
        {
        ScopedKernelTimer t("thermo");
        abiCheck(hmd_compute_thermo(&a, nullptr), "hmd_compute_thermo");
        }
    m_out.download(m_vals.data(), 8);
    m_vals[1] += m_pdata->getExternalEnergy();
    m_vals[2] += m_pdata->getExternalVirial(0) + m_pdata->getExternalVirial(3) + m_pdata->getExternalVirial(5);
    if (m_allreduce_sum)
        m_allreduce_sum(m_vals.data(), 6);
    }

// hoomd/md/ComputeThermo.h:190-215: P = (2 K_trans + W) / (D V)
Scalar ComputeThermoGPU::getPressure() const
    {
    const unsigned int D = m_sysdef->getNDimensions();
    const Scalar V = m_pdata->getGlobalBox().getVolume(D == 2);
    return (2.0 * m_vals[0] + m_vals[2]) / (Scalar(D) * V);
    }

// hoomd/md/Thermostat.h:394-420
std::array<Scalar, 2> BerendsenThermostat::getRescalingFactorsOne(uint64_t timestep, Scalar deltaT)
    {
    m_thermo->compute(timestep);
    const Scalar T_cur = m_thermo->getTranslationalTemperature();
    if (m_thermo->getTranslationalDOF() != 0 && m_thermo->getTranslationalKineticEnergy() == 0)
        throw std::runtime_error("Berendsen thermostat requires non-zero initial temperatures");
    const Scalar lambda_T = std::sqrt(1.0 + deltaT / m_tau * (m_T / T_cur - 1.0));
    return {lambda_T, 1.0};
    }

// hoomd/md/Thermostat.h:348-371
std::array<Scalar, 2> BussiThermostat::getRescalingFactorsOne(uint64_t timestep, Scalar deltaT)
    {
    m_thermo->compute(timestep);
    const Scalar ntdof = m_thermo->getTranslationalDOF();
    if (ntdof != 0 && m_thermo->getTranslationalKineticEnergy() == 0)
        throw std::runtime_error("Bussi thermostat requires non-zero initial temperatures");
    std::seed_seq seq {(uint64_t)m_sysdef->getSeed(), timestep, (uint64_t)0x42757373u};
    std::mt19937_64 rng(seq);
    std::gamma_distribution<double> gamma(ntdof / 2.0, m_T);
    return {std::sqrt(gamma(rng) / m_thermo->getTranslationalKineticEnergy()), 1.0};
    }

    } // namespace hoomd_amd
