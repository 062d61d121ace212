// forces.cc -- ForceCompute, ComputeThermoGPU and the thermostats.
// Follows hoomd/ForceCompute.cc:32-110,511-523, hoomd/md/ComputeThermo.cc:99-320, hoomd/md/Thermostat.h:337-420.
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
    if (m_sysdef->isDomainDecomposed())
        m_sysdef->getCommunicator()->allReduceSum(&pe, 1);
    return pe;
    }

// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
ComputeThermoGPU::ComputeThermoGPU(std::shared_ptr<SystemDefinition> sysdef, std::shared_ptr<ParticleGroup> group) : Compute(sysdef), m_group(group)
    {
    *reinterpret_cast<double*>(m_host_copy.get()) = 1.0;
    m_out.resize(8);
    m_scratch.resize(hmd_thermo_scratch_bytes());
    }

void ComputeThermoGPU::fillArgs(hmd_thermo_args_t& a)
    {
    std::memset(&a, 0, sizeof(a));
    a.d_out = m_out.get();
    a.d_scratch = m_scratch.get();
    a.d_vel = m_pdata->vel();
    a.d_net_force = m_pdata->netForceForIntegration();
    a.d_net_virial = m_pdata->getFlag(pressure_tensor) ? m_pdata->netVirial() : nullptr;
    a.net_virial_pitch = m_pdata->netVirialPitch();
    a.d_group_members = m_group->getIndexArray();
    a.group_size = m_group->getNumMembers();
    a.d_out_host = reinterpret_cast<double*>(m_host_copy.deviceView());
    }

void ComputeThermoGPU::launch()
    {
    hmd_thermo_args_t a;
    fillArgs(a);
    ScopedKernelTimer t("thermo");
    abiCheck(hmd_compute_thermo(&a, nullptr), "hmd_compute_thermo");
    m_host_stale = true;
    }

void ComputeThermoGPU::fetch()
    {
    m_out.download(m_vals.data(), 8);
    m_host_stale = false;
    m_vals[1] += m_pdata->getExternalEnergy();
    m_vals[2] += m_pdata->getExternalVirial(0) + m_pdata->getExternalVirial(3) + m_pdata->getExternalVirial(5);
    if (m_allreduce_sum)
        m_allreduce_sum(m_vals.data(), 6);
    }

void ComputeThermoGPU::compute(uint64_t timestep)
    {
    if (shouldCompute(timestep))
        launch();
    if (m_host_stale)
        fetch();
    }

void ComputeThermoGPU::computeAsync(uint64_t timestep)
    {
    if (shouldCompute(timestep))
        launch();
    }

void ComputeThermoGPU::adoptQueuedValues(uint64_t timestep)
    {
    m_first_compute = false;
    m_force_compute = false;
    m_last_computed = timestep;
    m_host_stale = true;
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
    return {std::sqrt(drawTargetEnergy(timestep) / m_thermo->getTranslationalKineticEnergy()), 1.0};
    }

// the kinetic energy the thermostat sets: gamma distributed with ntdof / 2 degrees of freedom and scale T
double BussiThermostat::drawTargetEnergy(uint64_t timestep)
    {
    std::seed_seq seq {(uint64_t)m_sysdef->getSeed(), timestep, (uint64_t)0x42757373u};
    std::mt19937_64 rng(seq);
    std::gamma_distribution<double> gamma(m_thermo->getTranslationalDOF() / 2.0, m_T);
    return gamma(rng);
    }

// The same factor without the host waiting for the kinetic energy: the sums are queued, the random target energy does
// not depend on them, and the first half step divides the two on the device.  A kinetic energy of zero is reported from
// the latest sums that have arrived, so the error of the reference comes a time step late at most.
bool BussiThermostat::getDeferredFactorOne(uint64_t timestep, Scalar deltaT, const double*& d_ke, double& gamma)
    {
    if (m_thermo->reducesOverRanks() || std::getenv("HOOMD_AMD_NO_DEFERRED_THERMOSTAT"))
        return false;
    m_thermo->computeAsync(timestep);
    if (m_thermo->getTranslationalDOF() != 0 && m_thermo->peekLatestKineticEnergy() == 0)
        throw std::runtime_error("Bussi thermostat requires non-zero initial temperatures");
    gamma = drawTargetEnergy(timestep);
    d_ke = m_thermo->deviceValues();
    return true;
    }

    } // namespace hoomd_amd
