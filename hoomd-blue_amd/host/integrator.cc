// integrator.cc -- TwoStepConstantVolumeGPU, IntegratorTwoStep, ParticleSorter and the System loop.
// Follows hoomd/md/TwoStepConstantVolumeGPU.cc:46-110, hoomd/md/IntegratorTwoStep.cc:51-130,161-196,
// 233-288, hoomd/Integrator.cc:106-158,344-560, hoomd/System.cc:80-190.
#include "hoomd_host.h"

#include <algorithm>
#include <chrono>
#include <cmath>

namespace hoomd_amd
    {
// hoomd/md/IntegrationMethodTwoStep.cc:73-108
Scalar IntegrationMethodTwoStep::getTranslationalDOF(std::shared_ptr<ParticleGroup> query)
    {
    // members shared by the query group and the method's group
    unsigned int n;
    if (m_group->isAll())
        n = query->getNumMembersGlobal();
    else if (query->isAll())
        n = m_group->getNumMembersGlobal();
    else
        n = std::min(query->getNumMembersGlobal(), m_group->getNumMembersGlobal());
    return Scalar(m_sysdef->getNDimensions()) * Scalar(n);
    }

// hoomd/md/TwoStepConstantVolumeGPU.cc:46-82
void TwoStepConstantVolumeGPU::integrateStepOne(uint64_t timestep)
    {
    if (m_group->getNumMembersGlobal() == 0)
        throw std::runtime_error("Empty integration group.");
    // (a factor the device can work out from sums that are only queued spares the host the wait for them)
    const double* d_ke = nullptr;
    double gamma = 0.0;
    const bool deferred = m_thermostat && m_thermostat->getDeferredFactorOne(timestep, m_deltaT, d_ke, gamma);
    const auto rescale = m_thermostat && !deferred ? m_thermostat->getRescalingFactorsOne(timestep, m_deltaT) : std::array<Scalar, 2> {1., 1.};
    hmd_step_one_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.d_rescale_ke = d_ke;
    a.rescale_gamma = gamma;
    if (deferred)
        m_n_deferred++;
    a.d_pos = m_pdata->pos();
    a.d_vel = m_pdata->vel();
    a.d_accel = m_pdata->accel();
    a.d_image = m_pdata->image();
    a.d_group_members = m_group->getIndexArray();
    a.group_size = m_group->getNumMembers();
    a.box = m_pdata->getBox().b;
    a.rescale_factor = rescale[0];
    a.deltaT = m_deltaT;
    a.use_limit = m_limit >= 0;
    a.maximum_displacement = m_limit;
    // The neighbor list's distance check of the new positions and the slot-ordered position copy of the tile path ride
    // in the same pass when the neighbor list takes the offer (the positions this half step produces are those of
    // time step timestep + 1).
    const auto& hook = m_pdata->getDisplacementCheckHook();
    bool fused = false;
    if (hook && !a.use_limit && a.d_group_members == nullptr && a.group_size == m_pdata->getN())
        {
        hmd_fused_step_args_t f;
        std::memset(&f, 0, sizeof(f));
        f.one = a;
        hook(f, timestep + 1);
        if (f.check.d_result || f.d_cpos)
            {
            ScopedKernelTimer t("step_one_check");
            abiCheck(hmd_nvt_rescale_step_one_check(&f, nullptr), "hmd_nvt_rescale_step_one_check");
            fused = true;
            }
        }
    if (!fused)
        {
        ScopedKernelTimer t("step_one");
        abiCheck(hmd_nvt_rescale_step_one(&a, nullptr), "hmd_nvt_rescale_step_one");
        }
    if (m_thermostat)
        m_thermostat->advanceThermostat(timestep, m_deltaT);
    }

// hoomd/md/TwoStepConstantVolumeGPU.cc:84-110
void TwoStepConstantVolumeGPU::integrateStepTwo(uint64_t timestep)
    {
    const auto rescale = m_thermostat ? m_thermostat->getRescalingFactorsTwo(timestep, m_deltaT) : std::array<Scalar, 2> {1., 1.};
    hmd_step_two_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.d_vel = m_pdata->vel();
    a.d_accel = m_pdata->accel();
    a.d_group_members = m_group->getIndexArray();
    a.group_size = m_group->getNumMembers();
    const Scalar4 *f1 = nullptr, *f2 = nullptr;
    if (m_pdata->getNetForcePair(f1, f2))
        {
        a.d_net_force = f1;
        a.d_net_force2 = f2;
        }
    else
        a.d_net_force = m_pdata->netForceForIntegration();
    a.deltaT = m_deltaT;
    a.rescale_factor = rescale[0];
    // With a thermostat that reads the kinetic energy of the next time step from the device, the sums are taken in this
    // pass (the velocities it leaves are the ones they are over), when nothing touches the particles in between.
    if (m_thermostat && m_sums_ride && !f2)
        {
        auto thermo = m_thermostat->getThermo();
        hmd_thermo_args_t th;
        thermo->fillArgs(th);
        if (!thermo->reducesOverRanks() && !th.d_net_virial && th.d_net_force == a.d_net_force && th.d_group_members == a.d_group_members
            && th.group_size == a.group_size && !std::getenv("HOOMD_AMD_NO_DEFERRED_THERMOSTAT"))
            {
            ScopedKernelTimer t("step_two_thermo");
            abiCheck(hmd_nvt_rescale_step_two_thermo(&a, &th, nullptr), "hmd_nvt_rescale_step_two_thermo");
            thermo->adoptQueuedValues(timestep + 1);
            m_n_with_sums++;
            return;
            }
        }
    ScopedKernelTimer t("step_two");
    abiCheck(hmd_nvt_rescale_step_two(&a, nullptr), "hmd_nvt_rescale_step_two");
    }

// Step two of `timestep` fused with step one of `timestep + 1` and, when a neighbor list asks for it,
// with the distance check of the new positions (those of time step `timestep + 2`).
void TwoStepConstantVolumeGPU::integrateStepTwoOne(uint64_t timestep)
    {
    if (m_group->getNumMembersGlobal() == 0)
        throw std::runtime_error("Empty integration group.");
    hmd_fused_step_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.two.d_vel = m_pdata->vel();
    // the fused pass carries the acceleration from its second-half part to its first-half part in
    // registers; the array is brought up to date by the separate second half step that ends every
    // sequence of fused passes (IntegratorTwoStep::flush)
    a.two.d_accel = nullptr;
    a.two.d_group_members = m_group->getIndexArray();
    a.two.group_size = m_group->getNumMembers();
    const Scalar4 *f1 = nullptr, *f2 = nullptr;
    if (m_pdata->getNetForcePair(f1, f2))
        {
        a.two.d_net_force = f1;
        a.two.d_net_force2 = f2;
        }
    else
        a.two.d_net_force = m_pdata->netForceForIntegration();
    a.two.deltaT = m_deltaT;
    a.two.rescale_factor = 1.0;
    a.one.d_pos = m_pdata->pos();
    a.one.d_vel = m_pdata->vel();
    a.one.d_accel = nullptr;
    a.one.d_image = m_pdata->image();
    a.one.d_group_members = a.two.d_group_members;
    a.one.group_size = a.two.group_size;
    a.one.box = m_pdata->getBox().b;
    a.one.rescale_factor = 1.0;
    a.one.deltaT = m_deltaT;
    const auto& hook = m_pdata->getDisplacementCheckHook();
    if (hook && a.one.d_group_members == nullptr && a.one.group_size == m_pdata->getN())
        {
        // step one of timestep + 1 produces the positions of time step timestep + 2
        hook(a, timestep + 2);
        }
    ScopedKernelTimer t("step_two_one");
    abiCheck(hmd_nve_step_two_one_check(&a, nullptr), "hmd_nve_step_two_one_check");
    }

// ---------------------------------------------------------------------------------------------
IntegratorTwoStep::IntegratorTwoStep(std::shared_ptr<SystemDefinition> sysdef, Scalar deltaT)
    : m_sysdef(sysdef), m_pdata(sysdef->getParticleData()), m_deltaT(deltaT)
    {
    }

IntegratorTwoStep::~IntegratorTwoStep()
    {
    m_pdata->setNetForcePair(nullptr, nullptr, nullptr);
    m_pdata->aliasNetForce(nullptr);
    m_pdata->setStepRiderSource(nullptr);
    }

// What a force kernel needs to run the second half of time step `timestep - 1` and the first half of `timestep`
// behind the forces of `timestep` (the integrator's side of hmd_fused_step_args_t; the neighbor list adds the rest).
bool IntegratorTwoStep::fillStepRider(hmd_fused_step_args_t& a, uint64_t timestep)
    {
    if (!m_rider_armed || timestep != m_rider_tstep)
        return false;
    auto method = m_methods[0];
    a.two.d_vel = m_pdata->vel();
    a.two.d_accel = nullptr;
    a.two.d_group_members = method->getGroup()->getIndexArray();
    a.two.group_size = method->getGroup()->getNumMembers();
    a.two.d_net_force = m_pdata->netForceForIntegration();
    a.two.deltaT = m_deltaT;
    a.two.rescale_factor = 1.0;
    a.one.d_pos = m_pdata->pos();
    a.one.d_vel = m_pdata->vel();
    a.one.d_accel = nullptr;
    a.one.d_image = m_pdata->image();
    a.one.d_group_members = a.two.d_group_members;
    a.one.group_size = a.two.group_size;
    a.one.box = m_pdata->getBox().b;
    a.one.rescale_factor = 1.0;
    a.one.deltaT = m_deltaT;
    return a.one.d_group_members == nullptr && a.one.group_size == m_pdata->getN();
    }

void IntegratorTwoStep::setDeltaT(Scalar dt)
    {
    m_deltaT = dt;
    for (auto& m : m_methods)
        m->setDeltaT(dt);
    }

// hoomd/md/IntegratorTwoStep.cc:161-196
Scalar IntegratorTwoStep::getTranslationalDOF(std::shared_ptr<ParticleGroup> group)
    {
    Scalar removed = 0;
    const unsigned int N_filter = group->getNumMembersGlobal();
    const unsigned int N_particles = m_pdata->getNGlobal();
    if (m_methods.size() == 1 && m_methods[0]->isMomentumConserving() && m_methods[0]->getGroup()->getNumMembersGlobal() == N_particles)
        removed = Scalar(m_sysdef->getNDimensions()) * (Scalar(N_filter) / Scalar(N_particles));
    Scalar total = 0;
    for (auto& m : m_methods)
        total += m->getTranslationalDOF(group);
    return total - removed;
    }

// hoomd/Integrator.cc:131-158
std::array<double, 3> IntegratorTwoStep::computeLinearMomentum()
    {
    const unsigned int n = m_pdata->getN();
    std::vector<Scalar4> vel(n);
    if (n)
        hipCheck(hipMemcpy(vel.data(), m_pdata->vel(), sizeof(Scalar4) * n, hipMemcpyDeviceToHost), "vel D2H");
    std::array<double, 3> p {{0, 0, 0}};
    for (auto& v : vel)
        {
        p[0] += v.w * v.x;
        p[1] += v.w * v.y;
        p[2] += v.w * v.z;
        }
    return p;
    }

// hoomd/Integrator.cc:344-560.  With one force compute and no virial request the sum is the
// identity: the integrator then reads that force array directly instead of copying it.
void IntegratorTwoStep::computeNetForceGPU(uint64_t timestep)
    {
    m_pdata->setNetForcePair(nullptr, nullptr, nullptr); // (a sum of the previous step's forces nobody asked for)
    // Forces that only launch their kernel go first (a bond force), those that may wait for the device's rebuild decision
    // before they can launch (a pair force behind its neighbor list) after them: the device computes bonds while the host
    // looks at the flag.  The forces still add up in the order they were given (the arrays keep their places below).  Not
    // under domain decomposition, where the neighbor list's check starts the halo exchange the bond force needs.
    const bool reorder = !m_sysdef->isDomainDecomposed();
    for (auto& f : m_forces)
        if (reorder && !f->mayWaitForDevice())
            f->compute(timestep);
    for (auto& f : m_forces)
        if (!reorder || f->mayWaitForDevice())
            f->compute(timestep);

    Scalar ext_e = 0, ext_v[6] = {0, 0, 0, 0, 0, 0};
    for (auto& f : m_forces)
        {
        ext_e += f->getExternalEnergy();
        for (unsigned int k = 0; k < 6; k++)
            ext_v[k] += f->getExternalVirial(k);
        }
    m_pdata->setExternalEnergy(ext_e);
    for (unsigned int k = 0; k < 6; k++)
        m_pdata->setExternalVirial(k, ext_v[k]);

    const bool virial = m_pdata->getFlag(pressure_tensor);
    if (m_forces.size() == 1 && !virial)
        {
        m_pdata->aliasNetForce(m_forces[0]->getForceArray());
        return;
        }
    m_pdata->aliasNetForce(nullptr);

    const unsigned int nparticles = m_pdata->getN() + m_pdata->getNGhosts();
    if (m_forces.empty())
        {
        hipCheck(hipMemsetAsync(m_pdata->netForceStorage(), 0, sizeof(Scalar4) * nparticles, 0), "zero net force");
        return;
        }
    if (m_forces.size() == 2 && !virial)
        {
        // Two forces, no virial: the half steps add the two arrays where they read them (hmd_step_two_args_t::
        // d_net_force2), and the net-force array is filled only when somebody asks for it (thermo, accelerations).
        m_pdata->setNetForcePair(m_forces[0]->getForceArray(), m_forces[1]->getForceArray(), [this]() { sumNetForce(false); });
        return;
        }
    sumNetForce(virial);
    }

// hoomd/Integrator.cc:344-560: the net-force pass
void IntegratorTwoStep::sumNetForce(bool virial)
    {
    const unsigned int nparticles = m_pdata->getN() + m_pdata->getNGhosts();
    ScopedKernelTimer t("sum_net_force");
    for (size_t base = 0; base < m_forces.size(); base += 6)
        {
        hmd_sum_net_force_args_t a;
        std::memset(&a, 0, sizeof(a));
        a.d_net_force = m_pdata->netForceStorage();
        a.d_net_virial = m_pdata->netVirial();
        a.net_virial_pitch = m_pdata->netVirialPitch();
        for (size_t k = 0; k < 6 && base + k < m_forces.size(); k++)
            {
            a.force_list.f[k] = m_forces[base + k]->getForceArray();
            a.force_list.v[k] = m_forces[base + k]->getVirialArray();
            a.force_list.vpitch[k] = m_forces[base + k]->getVirialPitch();
            }
        a.nparticles = nparticles;
        a.clear = (base == 0);
        a.compute_virial = virial;
        abiCheck(hmd_integrator_sum_net_force(&a, nullptr), "hmd_integrator_sum_net_force");
        }
    }

void IntegratorTwoStep::computeAccelerations(uint64_t timestep)
    {
    abiCheck(hmd_compute_accelerations(m_pdata->accel(), m_pdata->vel(), m_pdata->netForceForIntegration(), m_pdata->getN(), nullptr),
             "hmd_compute_accelerations");
    }

// hoomd/md/IntegratorTwoStep.cc:233-288
void IntegratorTwoStep::prepRun(uint64_t timestep)
    {
    for (auto& m : m_methods)
        m->setDeltaT(m_deltaT);
    if (m_sysdef->isDomainDecomposed())
        {
        m_sysdef->getCommunicator()->forceMigrate();
        m_sysdef->getCommunicator()->communicate(timestep);
        }
    computeNetForceGPU(timestep);
    if (!m_pdata->isAccelSet())
        {
        computeAccelerations(timestep);
        m_pdata->notifyAccelSet();
        }
    m_prepared = true;
    }

// hoomd/md/IntegratorTwoStep.cc:51-130
void IntegratorTwoStep::update(uint64_t timestep)
    {
    if (!m_prepared)
        prepRun(timestep);
    // One constant-volume method without thermostat: the second half of a step and the first half of the
    // next touch the same arrays with nothing in between, so they run as one kernel.  The second half
    // stays pending until the next update() (or flush()).
    TwoStepConstantVolumeGPU* nve = nullptr;
    if (m_fuse && m_methods.size() == 1)
        {
        nve = dynamic_cast<TwoStepConstantVolumeGPU*>(m_methods[0].get());
        if (nve && !nve->canFuse())
            nve = nullptr;
        }
    if (nve && m_step_one_done_for == timestep)
        {
        // the previous step's force kernel went on with the second half of that step and the first half of this one
        m_step_one_done_for = ~0ull;
        }
    else if (nve && m_pending_two && m_pending_tstep + 1 == timestep)
        {
        nve->setDeltaT(m_deltaT);
        nve->integrateStepTwoOne(m_pending_tstep);
        m_pending_two = false;
        }
    else
        {
        flush();
        for (auto& m : m_methods)
            {
            m->setDeltaT(m_deltaT);
            m->integrateStepOne(timestep);
            }
        }
    if (m_sysdef->isDomainDecomposed())
        m_sysdef->getCommunicator()->communicate(timestep + 1);
    // The force kernel of this step may take the second half of the step and the first half of the next one along,
    // when the run loop has promised that the next update() follows with nothing in between: one method without
    // thermostat over every particle, one force, no virial, no domain decomposition.
    m_rider_armed = nve && m_rider_enabled && m_lookahead && m_forces.size() == 1 && !m_sysdef->isDomainDecomposed() && !m_pdata->getFlag(pressure_tensor);
    m_rider_tstep = timestep + 1;
    if (m_rider_armed)
        m_pdata->setStepRiderSource([this](hmd_fused_step_args_t& a, uint64_t t) { return fillStepRider(a, t); });
    computeNetForceGPU(timestep + 1);
    const bool rode = m_rider_armed && m_pdata->getStepRiderDone() == timestep + 1;
    m_rider_armed = false;
    if (rode)
        {
        m_step_one_done_for = timestep + 1;
        m_pending_two = false;
        m_n_rider_steps++;
        }
    else if (nve)
        {
        m_pending_two = true;
        m_pending_tstep = timestep;
        }
    else
        for (auto it = m_methods.rbegin(); it != m_methods.rend(); ++it)
            {
            // (the run loop has promised that the next update() follows with nothing in between: a thermostat's sums
            // of the next time step may be taken in this pass)
            (*it)->setSumsRide(m_lookahead && m_methods.size() == 1);
            (*it)->integrateStepTwo(timestep);
            (*it)->setSumsRide(false);
            }
    }

void IntegratorTwoStep::flush()
    {
    if (!m_pending_two)
        return;
    m_pending_two = false;
    for (auto it = m_methods.rbegin(); it != m_methods.rend(); ++it)
        (*it)->integrateStepTwo(m_pending_tstep);
    }

// ---------------------------------------------------------------------------------------------
ParticleSorter::ParticleSorter(std::shared_ptr<SystemDefinition> sysdef, uint64_t period)
    : m_sysdef(sysdef), m_pdata(sysdef->getParticleData()), m_cl(std::make_shared<CellListGPU>(sysdef)), m_period(period)
    {
    m_cl->setRadius(1);
    m_cl->setFlagIndex();
    m_cl->setComputeAdj(false);
    m_cl->setNmaxHeadroom(4);
    // members of a cell in index order: the new particle order is then a function of the old one, not of the order in
    // which the binning kernel's atomics were served (one sort of the cell list per reordering)
    m_cl->setSortCellList(true);
    }

void ParticleSorter::update(uint64_t timestep)
    {
    if (m_period == 0 || timestep % m_period != 0 || timestep == m_last_sorted)
        return;
    m_last_sorted = timestep;
    sortNow();
    }

void ParticleSorter::sortNow()
    {
    if (m_pdata->getN() == 0)
        return;
    // ghosts are dropped before reordering under domain decomposition: the communicator re-exchanges
    // them on the forced rebuild that the sort signal triggers
    if (m_sysdef->isDomainDecomposed())
        m_pdata->setNGhosts(0);
    if (m_width <= 0)
        {
        const auto npd = m_pdata->getBox().getNearestPlaneDistance();
        const double rho = m_pdata->getN() / (m_pdata->getBox().getVolume(m_sysdef->getNDimensions() == 2));
        m_width = std::max(std::cbrt(8.0 / std::max(rho, 1e-12)), std::min(npd[0], npd[1]) / 512.0);
        }
    m_cl->setNominalWidth(m_width);
    m_cl->forceCompute(0);
    const auto ci = m_cl->getCellIndexer();
    const unsigned int ncells = ci.w * ci.h * ci.d;
    m_order.resize(std::max(m_pdata->getMaxN(), 1u), false);
    m_offsets.resize(ncells + 1, false);
    m_workspace.resize(hmd_cell_order_workspace_bytes(ncells), false);
    hmd_cell_order_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.d_order = m_order.get();
    a.d_cell_offset = m_offsets.get();
    a.d_cell_size = m_cl->getCellSizeArray().get();
    a.d_cell_xyzf = m_cl->getXYZFArray().get();
    a.ci = ci;
    a.cli = m_cl->getCellListIndexer();
    a.N = m_pdata->getN();
    a.d_workspace = m_workspace.get();
    a.workspace_bytes = m_workspace.size();
        {
        ScopedKernelTimer t("sorter");
        abiCheck(hmd_cell_order(&a, nullptr), "hmd_cell_order");
        m_pdata->applyOrder(m_order.get());
        }
    }

// ---------------------------------------------------------------------------------------------
// hoomd/System.cc:80-190
void System::run(uint64_t nsteps, bool prepare)
    {
    if (!m_integrator)
        throw std::runtime_error("Cannot run without an integrator");
    auto pdata = m_sysdef->getParticleData();
    pdata->setFlags(m_pressure_flag ? (1u << pressure_tensor) : 0u);
    if (prepare || !m_integrator->isPrepared())
        m_integrator->prepRun(m_cur_tstep);
    hipCheck(hipStreamSynchronize(0), "hipStreamSynchronize");
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t s = 0; s < nsteps; s++)
        {
        if (m_sorter && m_sorter->isDue(m_cur_tstep))
            {
            m_integrator->flush(); // the pending half step reads forces in the old particle order
            m_sorter->update(m_cur_tstep);
            }
        // does the next update follow this one with nothing in between?
        m_integrator->setLookahead(s + 1 < nsteps && !(m_sorter && m_sorter->isDue(m_cur_tstep + 1)));
        m_integrator->update(m_cur_tstep);
        m_cur_tstep++;
        }
    m_integrator->setLookahead(false);
    m_integrator->flush();
    hipCheck(hipStreamSynchronize(0), "hipStreamSynchronize");
    m_integrator->checkForceErrors();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    m_last_walltime = dt;
    m_last_tps = (dt > 0 && nsteps > 0) ? double(nsteps) / dt : 0.0;
    }

    } // namespace hoomd_amd
