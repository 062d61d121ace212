// neighbor_list.cc -- NeighborListGPUBinned host class: r_cut bookkeeping, the rebuild state
// machine, head list / overflow protocol and exclusions.
// Follows hoomd/md/NeighborList.cc:32-246 (construction), :363-435 (compute), :452-529 (updateRList),
// :535-561 (checkBoxSize), :1313-1436 (needsUpdating + statistics), :1544-1625 (head list,
// conditions), hoomd/md/NeighborListGPU.cc:29-232 and hoomd/md/NeighborListGPUBinned.cc:19-180.
#include "hoomd_host.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace hoomd_amd
    {
NeighborListGPUBinned::NeighborListGPUBinned(std::shared_ptr<SystemDefinition> sysdef, Scalar r_buff)
    : Compute(sysdef), m_cl(std::make_shared<CellListGPU>(sysdef)), m_r_buff(r_buff), m_h_words(64)
    {
    if (m_r_buff < 0.0)
        throw std::runtime_error("Requested buffer radius is less than zero.");
    const unsigned int nt = m_pdata->getNTypes();
    m_r_cut.assign((size_t)nt * nt, 0.0);
    m_rcut_max.assign(nt, 0.0);
    m_Nmax.assign(nt, 4); // hoomd/md/NeighborList.cc:145-152
    m_update_periods.assign(100, 0);
    m_last_L = m_pdata->getGlobalBox().getNearestPlaneDistance();

    m_d_Nmax.resize(nt);
    m_d_Nmax.upload(m_Nmax.data(), nt);
    m_conditions.resize(nt + 2); // [nt] = tile capacity overflow flag, [nt + 1] = largest tile population
    m_d_r_cut.resize((size_t)nt * nt);
    m_d_rcut_max.resize(nt);
    m_flag.resize(1);
    m_req_size.resize(1);
    reallocatePerParticle();
    m_nlist.resize(4 * (size_t)std::max(m_pdata->getMaxN(), 1u), false);

    // hoomd/md/NeighborListGPUBinned.cc:19-45
    m_cl->setRadius(1);
    m_cl->setComputeTypeBody(true);
    m_cl->setComputeIdx(true); // the tile kernels number candidates through the index array
    m_cl->setFlagIndex();
    m_cl->setDeferConditions(true);

    m_pdata->setDisplacementCheckHook([this](hmd_fused_step_args_t& a, uint64_t timestep) { offerFusedPass(a, timestep); });
    m_sort_slot = m_pdata->getParticleSortSignal().connect([this]() { forceUpdate(); });
    m_box_slot = m_pdata->getBoxChangeSignal().connect([this]() { forceUpdate(); });
    m_maxn_slot = m_pdata->getMaxParticleNumberChangeSignal().connect([this]()
                                                                      {
                                                                      reallocatePerParticle();
                                                                      forceUpdate();
                                                                      });
    }

NeighborListGPUBinned::~NeighborListGPUBinned()
    {
    m_pdata->setDisplacementCheckHook(nullptr);
    m_pdata->getParticleSortSignal().disconnect(m_sort_slot);
    m_pdata->getBoxChangeSignal().disconnect(m_box_slot);
    m_pdata->getMaxParticleNumberChangeSignal().disconnect(m_maxn_slot);
    if (m_comm_stream)
        {
        (void)hipStreamSynchronize(m_comm_stream);
        (void)hipStreamDestroy(m_comm_stream);
        for (hipStream_t r : m_rejected_streams)
            (void)hipStreamDestroy(r);
        (void)hipEventDestroy(m_ev_ready);
        (void)hipEventDestroy(m_ev_comm);
        }
    }

void NeighborListGPUBinned::reallocatePerParticle()
    {
    const size_t n = std::max(m_pdata->getMaxN(), 1u);
    m_n_neigh.resize(n, false);
    m_head_list.resize(n, false);
    m_head_uniform_n = 0;
    m_last_pos.resize(n, false);
    m_n_ex_idx.resize(n, false);
    m_workspace.resize(hmd_nlist_head_list_workspace_bytes((unsigned int)n), false);
    m_ex_dirty = true;
    }

void NeighborListGPUBinned::addRCutMatrix(std::shared_ptr<std::vector<Scalar>> m)
    {
    if (m->size() != m_r_cut.size())
        throw std::invalid_argument("given r_cut_matrix is not the right size");
    m_consumer_r_cut.push_back(m);
    notifyRCutMatrixChange();
    }

void NeighborListGPUBinned::removeRCutMatrix(std::shared_ptr<std::vector<Scalar>> m)
    {
    m_consumer_r_cut.erase(std::remove(m_consumer_r_cut.begin(), m_consumer_r_cut.end(), m), m_consumer_r_cut.end());
    notifyRCutMatrixChange();
    }

void NeighborListGPUBinned::setRBuff(Scalar r_buff)
    {
    m_r_buff = r_buff;
    if (m_r_buff < 0.0)
        throw std::runtime_error("Requested buffer radius is less than zero.");
    notifyRCutMatrixChange();
    forceUpdate();
    }

void NeighborListGPUBinned::setStorageMode(storageMode m)
    {
    m_storage_mode = m;
    forceUpdate();
    }

void NeighborListGPUBinned::setDeterministic(bool d)
    {
    m_deterministic = d;
    m_cl->setSortCellList(d);
    forceUpdate();
    }

// hoomd/md/NeighborList.cc:452-529
void NeighborListGPUBinned::updateRList()
    {
    const unsigned int nt = m_pdata->getNTypes();
    std::fill(m_r_cut.begin(), m_r_cut.end(), 0.0);
    for (auto& m : m_consumer_r_cut)
        for (size_t i = 0; i < m_r_cut.size(); i++)
            m_r_cut[i] = std::max(m_r_cut[i], (*m)[i]);
    Scalar r_cut_max = 0.0;
    for (unsigned int i = 0; i < nt; i++)
        {
        Scalar mi = 0.0;
        for (unsigned int j = 0; j < nt; j++)
            mi = std::max(mi, m_r_cut[j * nt + i]);
        m_rcut_max[i] = mi;
        r_cut_max = std::max(r_cut_max, mi);
        }
    m_rcut_max_max = r_cut_max;
    Scalar r_cut_min = r_cut_max;
    for (Scalar rc : m_r_cut)
        if (rc > 0.0 && rc < r_cut_min)
            r_cut_min = rc;
    m_rcut_min = r_cut_min;
    m_d_r_cut.upload(m_r_cut.data(), m_r_cut.size());
    m_d_rcut_max.upload(m_rcut_max.data(), nt);
    m_rcut_changed = false;
    m_update_cell_size = true;
    }

// hoomd/md/NeighborList.cc:535-561
void NeighborListGPUBinned::checkBoxSize()
    {
    const BoxDim& box = m_pdata->getBox();
    const auto npd = box.getNearestPlaneDistance();
    const Scalar rmax = m_rcut_max_max + m_r_buff;
    if ((box.b.periodic[0] && npd[0] <= rmax * 2.0) || (box.b.periodic[1] && npd[1] <= rmax * 2.0)
        || (m_sysdef->getNDimensions() == 3 && box.b.periodic[2] && npd[2] <= rmax * 2.0))
        {
        std::ostringstream oss;
        oss << "nlist: Simulation box is too small, the neighbor list is searching beyond the minimum image: rmax=" << rmax;
        throw std::runtime_error(oss.str());
        }
    }

unsigned int NeighborListGPUBinned::getSmallestRebuild() const
    {
    for (unsigned int i = 0; i < m_update_periods.size(); i++)
        if (m_update_periods[i] != 0)
            return i;
    return (unsigned int)m_update_periods.size();
    }

void NeighborListGPUBinned::resetStats()
    {
    m_updates = m_forced_updates = m_dangerous_updates = 0;
    std::fill(m_update_periods.begin(), m_update_periods.end(), 0);
    }

// arguments of the distance check for the current positions; a fresh check number each time, so the
// device flag never needs clearing
void NeighborListGPUBinned::fillCheckArgs(hmd_nlist_check_args_t& a, bool fresh)
    {
    const auto L_g = m_pdata->getGlobalBox().getNearestPlaneDistance();
    std::memset(&a, 0, sizeof(a));
    a.lambda = Scalar3 {L_g[0] / m_last_L[0], L_g[1] / m_last_L[1], L_g[2] / m_last_L[2]};
    a.lambda_min = std::min(a.lambda.x, std::min(a.lambda.y, a.lambda.z));
    a.d_result = m_flag.get();
    a.d_last_pos = m_last_pos.get();
    a.d_pos = m_pdata->pos();
    a.N = m_pdata->getN();
    a.box = m_pdata->getBox().b;
    a.d_rcut_max = m_d_rcut_max.get();
    a.r_buff = m_r_buff;
    a.ntypes = m_pdata->getNTypes();
    a.checkn = fresh ? ++m_checkn : m_checkn;
    // The flag comes back through pinned memory that the host polls: no stream synchronisation.  Not from
    // the check itself: whoever consumes the check publishes (the predicated force kernel as its first
    // act, or hmd_publish_word), which saves a launch per step.
    a.d_published = nullptr;
    a.stamp = m_checkn;
    }

// The integrator offers to run the check of the positions of `timestep` inside its own pass over the
// particles.  Accepted when needsUpdating(timestep) is going to ask for exactly this check.
bool NeighborListGPUBinned::offerCheck(hmd_nlist_check_args_t& a, uint64_t timestep)
    {
    if (m_force_update || m_rcut_changed || !m_dist_check || !m_has_been_updated_once || m_r_buff < 1e-6)
        return false;
    if (timestep < m_last_updated_tstep + m_rebuild_check_delay || m_last_checked_tstep == timestep)
        return false;
    // (offered twice for one time step when the kernel that carried the check was switched off and is launched again
    // after the rebuild: the same check number then)
    fillCheckArgs(a, m_prelaunched_tstep != timestep);
    m_prelaunched_tstep = timestep;
    return true;
    }

void NeighborListGPUBinned::offerFusedPass(hmd_fused_step_args_t& a, uint64_t timestep)
    {
    if (!offerCheck(a.check, timestep))
        std::memset(&a.check, 0, sizeof(a.check));
    // valid as long as the list is not rebuilt at `timestep` (a rebuild renumbers the slots and refreshes all)
    if (m_tiles_active && !m_force_update && m_has_been_updated_once && m_slot_of.size() >= m_pdata->getN())
        {
        a.d_slot_of = m_slot_of.get();
        a.d_cpos = m_cpos.get();
        m_cpos_tstep = timestep;
        }
    }

bool NeighborListGPUBinned::offerStepRider(hmd_fused_step_args_t& a, uint64_t timestep)
    {
    if (!m_tiles_active || m_force_update || !m_has_been_updated_once || m_slot_of.size() < m_pdata->getN() || m_pdata->getNGhosts() != 0)
        return false;
    const bool had = m_prelaunched_tstep == timestep;
    if (!offerCheck(a.check, timestep))
        {
        std::memset(&a.check, 0, sizeof(a.check));
        if (had)
            m_prelaunched_tstep = ~0ull;
        }
    if (m_cpos_alt.size() != m_cpos.size())
        m_cpos_alt.resize(m_cpos.size(), false);
    a.d_slot_of = m_slot_of.get();
    a.d_cpos = m_cpos_alt.get();
    return true;
    }

void NeighborListGPUBinned::commitStepRider(uint64_t timestep)
    {
    m_cpos.swap(m_cpos_alt);
    m_cpos_tstep = timestep;
    }

// hoomd/md/NeighborListGPU.cc:29-101
bool NeighborListGPUBinned::distanceCheck(uint64_t timestep)
    {
    if (m_prelaunched_tstep == timestep)
        m_prelaunched_tstep = ~0ull; // the fused integration pass already launched check number m_checkn
    else
        {
        hmd_nlist_check_args_t a;
        fillCheckArgs(a);
        ScopedKernelTimer t("nlist_check");
        abiCheck(hmd_nlist_needs_update_check(&a, nullptr), "hmd_nlist_needs_update_check");
        }
    if (m_sysdef->isDomainDecomposed())
        {
        // the vote of all ranks is reduced on the device, queued behind the check; the communicator also
        // refreshes the ghost positions in the same breath, so work that is valid for a fresh list can be
        // queued behind the vote before the host has seen it
        auto comm = m_sysdef->getCommunicator();
        // Work that needs no ghost positions is queued first, on the default stream; the exchange then goes on
        // a stream of its own that waits only for what was queued before that work (integration, distance
        // check), so the two run side by side.  The default stream waits for the exchange before it goes on.
        bool overlapped = false;
        if (m_overlap && m_interior_hook)
            {
            if (!m_comm_stream)
                {
                // normal priority unless asked otherwise (setCommStreamHighPriority, HOOMD_AMD_COMM_PRIORITY=high):
                // the runtime keeps high-priority streams on hardware queues of their own, so the exchange's small
                // kernels are not lined up behind the force kernel of the default stream -- but a high-priority
                // queue waiting on an event has starved normal-priority queues when ranks shared a GPU, and no run
                // with peers at high priority is on record.  Ranks that share a GPU always get normal priority.
                int least = 0, greatest = 0;
                hipCheck(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
                const char* prio = std::getenv("HOOMD_AMD_COMM_PRIORITY");
                bool high = m_comm_high_priority && !m_ranks_share_gpu;
                if (prio && std::string(prio) == "normal")
                    high = false;
                else if (prio && std::string(prio) == "high" && !m_ranks_share_gpu)
                    high = true;
                hipCheck(hipStreamCreateWithPriority(&m_comm_stream, hipStreamNonBlocking, high ? greatest : 0), "hipStreamCreateWithPriority");
                // A created stream may land on the hardware queue of the default stream and then runs strictly behind
                // the force kernel it is meant to run beside.  With a GPU of its own the process tries a few streams
                // and keeps the first one whose kernels do run beside the default stream's
                // (hmd_stream_runs_beside_default); the rejected ones are kept until the end so that the runtime does
                // not hand their queue out again.  HOOMD_AMD_COMM_STREAM_PROBE=0 takes the first stream as it comes.
                const char* probe_env = std::getenv("HOOMD_AMD_COMM_STREAM_PROBE");
                m_comm_stream_probe = -1;
                if (!high && !m_ranks_share_gpu && !(probe_env && std::string(probe_env) == "0"))
                    {
                    DeviceArray<unsigned long long> words;
                    words.resize(2, false);
                    for (int attempt = 0; attempt < 5; attempt++)
                        {
                        int overlaps = 0;
                        abiCheck(hmd_stream_runs_beside_default(m_comm_stream, words.get(), &overlaps), "hmd_stream_runs_beside_default");
                        if (overlaps)
                            {
                            m_comm_stream_probe = attempt;
                            break;
                            }
                        if (attempt == 4)
                            break; // none of them: keep the last
                        m_rejected_streams.push_back(m_comm_stream);
                        hipCheck(hipStreamCreateWithPriority(&m_comm_stream, hipStreamNonBlocking, 0), "hipStreamCreateWithPriority");
                        }
                    }
                hipCheck(hipEventCreateWithFlags(&m_ev_ready, hipEventDisableTiming), "hipEventCreateWithFlags");
                hipCheck(hipEventCreateWithFlags(&m_ev_comm, hipEventDisableTiming), "hipEventCreateWithFlags");
                }
            hipCheck(hipEventRecord(m_ev_ready, nullptr), "hipEventRecord");
            overlapped = m_interior_hook(timestep, m_flag.get(), m_checkn);
            if (overlapped)
                {
                hipCheck(hipStreamWaitEvent(m_comm_stream, m_ev_ready, 0), "hipStreamWaitEvent");
                m_n_overlapped++;
                }
            }
        const uintptr_t d_vote = comm->beginRebuildVote((uintptr_t)m_flag.get(), m_checkn, overlapped ? (uintptr_t)m_comm_stream : 0);
        if (overlapped && !d_vote)
            hipCheck(hipStreamSynchronize(m_comm_stream), "hipStreamSynchronize");
        if (d_vote)
            {
            const unsigned int* d_word = reinterpret_cast<const unsigned int*>(d_vote);
            // the reduced word reaches the host like the local one does: published to pinned memory, polled
            volatile unsigned long long* hv = reinterpret_cast<volatile unsigned long long*>(m_h_words.get() + 20);
            unsigned long long* d_pub = reinterpret_cast<unsigned long long*>(m_h_words.deviceView() + 20);
            if (overlapped)
                {
                // the exchange's stream hands the vote to the host as soon as it is known, long before the
                // default stream gets to the remaining blocks
                abiCheck(hmd_publish_word(d_word, d_pub, m_checkn, m_comm_stream), "hmd_publish_word");
                // The blocks that see ghosts follow on the exchange's own stream, predicated on the vote: they start
                // as soon as the ghosts are in place and run beside the last workgroups of the interior launch, instead
                // of as a launch of their own behind it (two kernel tails one after the other).  The default stream
                // goes on when both are done.
                if (m_speculate && m_boundary_on_comm_stream)
                    m_speculate(timestep, d_word, m_checkn, nullptr, 0, m_comm_stream);
                hipCheck(hipEventRecord(m_ev_comm, m_comm_stream), "hipEventRecord");
                hipCheck(hipStreamWaitEvent(nullptr, m_ev_comm, 0), "hipStreamWaitEvent");
                if (m_speculate && !m_boundary_on_comm_stream)
                    m_speculate(timestep, d_word, m_checkn, nullptr, 0, nullptr);
                }
            else if (!(m_speculate && m_speculate(timestep, d_word, m_checkn, d_pub, m_checkn, nullptr)))
                abiCheck(hmd_publish_word(d_word, d_pub, m_checkn, nullptr), "hmd_publish_word");
            const unsigned long long v = pollStampedWord(hv, m_checkn, "rebuild vote", overlapped ? m_comm_stream : nullptr);
            return (unsigned int)v == m_checkn;
            }
        }
    volatile unsigned long long* h = reinterpret_cast<volatile unsigned long long*>(m_h_words.get() + 16);
    // (the consumer's kernel may carry the next time step's check and take a new number for it)
    const unsigned int checkn = m_checkn;
    if (!m_sysdef->isDomainDecomposed())
        {
        unsigned long long* d_pub = reinterpret_cast<unsigned long long*>(m_h_words.deviceView() + 16);
        // only on the consumer's own path (compute): nothing else touches the particles before the forces
        if (!(m_speculate && m_in_compute && m_speculate(timestep, m_flag.get(), checkn, d_pub, checkn, nullptr)))
            abiCheck(hmd_publish_word(m_flag.get(), d_pub, checkn, nullptr), "hmd_publish_word");
        }
    else
        abiCheck(hmd_publish_word(m_flag.get(), reinterpret_cast<unsigned long long*>(m_h_words.deviceView() + 16), checkn, nullptr),
                 "hmd_publish_word");
    const unsigned long long v = pollStampedWord(h, checkn, "distance check");
    bool result = ((unsigned int)v == checkn);
    if (m_sysdef->isDomainDecomposed())
        result = m_sysdef->getCommunicator()->anyRank(result);
    return result;
    }

// hoomd/md/NeighborList.cc:1325-1418
bool NeighborListGPUBinned::needsUpdating(uint64_t timestep)
    {
    m_compute_tstep = timestep;
    if (m_last_checked_tstep == timestep)
        {
        if (m_force_update)
            {
            m_force_update = false;
            return true;
            }
        return m_last_check_result;
        }
    m_last_checked_tstep = timestep;

    if (!m_force_update && timestep < (m_last_updated_tstep + m_rebuild_check_delay))
        {
        m_last_check_result = false;
        return false;
        }

    bool result = false;
    bool dangerous = m_dist_check && m_rebuild_check_delay > 1 && timestep == (m_last_updated_tstep + m_rebuild_check_delay);

    if (m_force_update)
        {
        result = true;
        m_force_update = false;
        m_forced_updates += 1;
        m_last_updated_tstep = timestep;
        dangerous = false;
        }
    else
        {
        if (m_r_buff < 1e-6
            || (!m_dist_check
                && (m_rebuild_check_delay == 0 || (m_rebuild_check_delay > 1 && timestep == (m_last_updated_tstep + m_rebuild_check_delay)))))
            result = true;
        else
            result = distanceCheck(timestep);

        if (result)
            {
            if (timestep > m_last_updated_tstep)
                {
                uint64_t period = timestep - m_last_updated_tstep;
                if (period >= m_update_periods.size())
                    period = m_update_periods.size() - 1;
                m_update_periods[period]++;
                }
            m_last_updated_tstep = timestep;
            m_updates += 1;
            }
        }
    if (result && dangerous)
        m_dangerous_updates += 1;
    m_last_check_result = result;
    return result;
    }

// hoomd/md/NeighborListGPU.cc:166-232 / hoomd/md/NeighborList.cc:1544-1592
void NeighborListGPUBinned::buildHeadList()
    {
    hmd_head_list_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.d_head_list = m_head_list.get();
    a.d_req_size_nlist = m_req_size.get();
    a.d_Nmax = m_d_Nmax.get();
    a.d_pos = m_pdata->pos();
    a.N = m_pdata->getN();
    a.ntypes = m_pdata->getNTypes();
    a.d_workspace = m_workspace.get();
    a.workspace_bytes = m_workspace.size();
    size_t req = 0;
    if (a.ntypes == 1)
        {
        // uniform rows: head[i] = i * Nmax whoever particle i is, and the total is known without a readback.  The list
        // is written for every place of the array at once and stays valid until the row length or the array changes
        // (particles come and go on every rebuild of a decomposed run: no launch for that).
        if (m_head_uniform_n < a.N || m_head_uniform_nmax != m_Nmax[0])
            {
            a.N = (unsigned int)std::min<size_t>(m_head_list.size(), 0xffffffffu);
            abiCheck(hmd_nlist_build_head_list(&a, nullptr), "hmd_nlist_build_head_list");
            m_head_uniform_n = a.N;
            m_head_uniform_nmax = m_Nmax[0];
            a.N = m_pdata->getN();
            }
        req = (size_t)a.N * m_Nmax[0];
        }
    else
        {
        abiCheck(hmd_nlist_build_head_list(&a, nullptr), "hmd_nlist_build_head_list");
        hipCheck(hipMemcpy(&req, m_req_size.get(), sizeof(size_t), hipMemcpyDeviceToHost), "req size D2H");
        }
    resizeNlist(req);
    }

// hoomd/md/NeighborList.cc:1573-1592
void NeighborListGPUBinned::resizeNlist(size_t size)
    {
    if (size > m_nlist.size())
        {
        size_t alloc = m_nlist.size() ? m_nlist.size() : 1;
        while (size > alloc)
            alloc = ((size_t)(((float)alloc) * 1.125f)) + 1;
        alloc = (alloc > 4) ? (alloc + 3) & ~(size_t)3 : 4;
        m_nlist.resize(alloc, false);
        }
    }

// hoomd/md/NeighborList.cc:1601-1625
bool NeighborListGPUBinned::checkConditions()
    {
    const unsigned int nt = m_pdata->getNTypes();
    if (nt + 2 > 64)
        throw std::runtime_error("nlist: more than 62 particle types");
    unsigned int h[64];
    m_readback.read(m_conditions.get(), nt + 2, h, "neighbor list conditions");
    bool result = false;
    if (h[nt])
        {
        // a 64-cell region outgrew the LDS tile: this build falls back to index rows
        m_tiles_overflowed = true;
        result = true;
        }
    else if (m_tiles_active && h[nt + 1])
        m_tile_capacity = h[nt + 1]; // exact: the force kernel sizes its LDS tile by it
    for (unsigned int i = 0; i < nt; i++)
        if (h[i] > m_Nmax[i])
            {
            m_Nmax[i] = (h[i] > 4) ? (h[i] + 3) & ~3u : 4;
            result = true;
            }
    if (result)
        m_d_Nmax.upload(m_Nmax.data(), nt);
    return result;
    }

void NeighborListGPUBinned::resetConditions()
    {
    m_conditions.zero();
    }

// hoomd/md/NeighborListGPUBinned.cc:49-180
void NeighborListGPUBinned::buildNlist(uint64_t timestep)
    {
    if (m_storage_mode != full)
        throw std::runtime_error("GPU neighbor lists require a full storage mode.");
    if (m_update_cell_size)
        {
        m_cl->setNominalWidth(getMaxRCut() + m_r_buff);
        m_update_cell_size = false;
        }
    m_cl->compute(timestep);

    ScopedKernelTimer t("nlist_build");
    unsigned int capacity = 0;
    if (tilesUsable(capacity))
        {
        // rows of the tile path: room for the longest row any type may have, in pieces of 16 entries, for every
        // member of the cell list (owned and ghost) plus one partly filled group per block
        const hmd_index3d_t ci = m_cl->getCellIndexer();
        const unsigned int nmax_all = *std::max_element(m_Nmax.begin(), m_Nmax.end());
        // (with two types the rows are laid out by type and hold up to three entries of the dummy candidate besides the neighbors)
        m_row_pieces = (nmax_all + (m_pdata->getNTypes() == 2 ? 3u : 0u) + 15) / 16;
        const size_t entries = hmd_nlist_tiles_storage_entries(&ci, m_pdata->getN() + m_pdata->getNGhosts(), m_row_pieces);
        if (m_tile_nlist.size() < entries)
            m_tile_nlist.resize(entries + entries / 8, false);
        const size_t nblocks = (size_t)((ci.w + 1) / 2) * ((ci.h + 1) / 2) * ((ci.d + 1) / 2);
        if (m_block_group.size() < nblocks + 1)
            m_block_group.resize(nblocks + 1, false);
        const hmd_index2d_t cli = m_cl->getCellListIndexer();
        const size_t nslots = (size_t)cli.w * cli.h;
        if (m_cpos.size() < nslots)
            {
            m_cpos.resize(nslots, false);
            m_slot_meta.resize(nslots, false);
            }
        if (m_slot_of.size() < m_pdata->getMaxN())
            m_slot_of.resize(std::max(m_pdata->getMaxN(), 1u), false);
        hmd_nlist_tiles_args_t a;
        std::memset(&a, 0, sizeof(a));
        m_tile_capacity = capacity;
        a.geom = getTileGeom();
        a.d_slot_meta = m_slot_meta.get();
        a.d_slot_of = m_slot_of.get(); // the build also fills slot_of and the slot-ordered position copy
        a.d_cell_xyzf = m_cl->getXYZFArray().get();
        a.d_cell_type_body = m_cl->getTypeBodyArray().get();
        a.d_cell_adj = m_cl->getCellAdjArray().get();
        a.cadji = m_cl->getCellAdjIndexer();
        a.d_tile_nlist = m_tile_nlist.get();
        a.d_n_neigh = m_n_neigh.get();
        a.d_last_updated_pos = m_last_pos.get();
        a.d_conditions = m_conditions.get();
        a.d_Nmax = m_d_Nmax.get();
        a.d_head_list = m_head_list.get();
        a.d_pos = m_pdata->pos();
        a.N = m_pdata->getN();
        a.box = m_pdata->getBox().b;
        a.d_r_cut = m_d_r_cut.get();
        a.r_buff = m_r_buff;
        a.ntypes = m_pdata->getNTypes();
        a.d_tile_overflow = m_conditions.get() + a.ntypes;
        a.tile_nlist_entries = m_tile_nlist.size();
        m_rows_by_type = hmd_nlist_tiles_rows_by_type(&a) != 0; // (the force launches that read these rows are told)
        if (m_exclusions_set)
            {
            // left out while the rows are written: no filter pass over tile rows (hoomd/md/NeighborListGPU.cu:158-223)
            a.d_n_ex_idx = m_n_ex_idx.get();
            a.d_ex_list_idx = m_ex_list_idx.get();
            a.exli = hmd_index2d_t {std::max(m_pdata->getMaxN(), 1u), m_ex_height};
            }
        abiCheck(hmd_compute_nlist_tiles(&a, nullptr), "hmd_compute_nlist_tiles");
        m_cpos_tstep = timestep; // current for every particle, ghosts included
        m_cpos_all_tstep = timestep;
        m_tiles_active = true;
        m_rows_valid = false;
        return;
        }
    m_tiles_active = false;
    m_rows_valid = true;
    hmd_nlist_binned_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.d_nlist = m_nlist.get();
    a.d_n_neigh = m_n_neigh.get();
    a.d_last_updated_pos = m_last_pos.get();
    a.d_conditions = m_conditions.get();
    a.d_Nmax = m_d_Nmax.get();
    a.d_head_list = m_head_list.get();
    a.d_pos = m_pdata->pos();
    a.d_body = m_pdata->body();
    a.N = m_pdata->getN();
    a.d_cell_size = m_cl->getCellSizeArray().get();
    a.d_cell_xyzf = m_cl->getXYZFArray().get();
    a.d_cell_type_body = m_cl->getTypeBodyArray().get();
    a.d_cell_adj = m_cl->getCellAdjArray().get();
    a.ci = m_cl->getCellIndexer();
    a.cli = m_cl->getCellListIndexer();
    a.cadji = m_cl->getCellAdjIndexer();
    a.box = m_pdata->getBox().b;
    a.d_r_cut = m_d_r_cut.get();
    a.r_buff = m_r_buff;
    a.ntypes = m_pdata->getNTypes();
    a.filter_body = m_filter_body;
    a.ghost_width = m_cl->getGhostWidth();
    if (m_exclusions_set)
        {
        a.d_n_ex_idx = m_n_ex_idx.get();
        a.d_ex_list_idx = m_ex_list_idx.get();
        a.exli = hmd_index2d_t {std::max(m_pdata->getMaxN(), 1u), m_ex_height};
        }
    abiCheck(hmd_compute_nlist_binned(&a, nullptr), "hmd_compute_nlist_binned");
    }

hmd_tile_geom_t NeighborListGPUBinned::getTileGeom() const
    {
    hmd_tile_geom_t g;
    std::memset(&g, 0, sizeof(g));
    g.d_cell_size = m_cl->getCellSizeArray().get();
    g.d_cell_idx = m_cl->getIndexArray().get();
    g.ci = m_cl->getCellIndexer();
    g.cli = m_cl->getCellListIndexer();
    g.capacity = m_tile_capacity;
    g.d_cpos = m_cpos.get();
    g.d_slot_meta = m_slot_meta.get();
    g.d_block_group = m_block_group.get();
    g.row_pieces = m_row_pieces;
    g.rows_by_type = m_rows_by_type ? 1u : 0u;
    return g;
    }

// A ghost lies outside the local box along at least one decomposed axis when the cell list is filled (it is
// owned by another rank), so along such an axis it sits in a cell that reaches beyond the box: cell c covers
// [c, c+1) * (L + 2 ghost_width) / dim of the padded box (hoomd/CellList.cc:447-476).  A block of 2x2x2 cells
// sees the cells 2b-1 .. 2b+2 along each axis; it is interior when all of those lie wholly inside the box
// along every decomposed axis.
bool NeighborListGPUBinned::getInteriorBlockWindow(uint64_t timestep, unsigned int (&lo)[3], unsigned int (&hi)[3]) const
    {
    if (!m_tiles_active || m_cpos_tstep != timestep || m_compute_tstep != timestep)
        return false;
    const hmd_index3d_t ci = m_cl->getCellIndexer();
    const unsigned int dim[3] = {ci.w, ci.h, ci.d};
    const Scalar3 gw3 = m_cl->getGhostWidth();
    const Scalar gw[3] = {gw3.x, gw3.y, gw3.z};
    const hmd_box_t box = m_pdata->getBox().b;
    for (int d = 0; d < 3; d++)
        {
        const unsigned int nb = (dim[d] + 1) / 2;
        if (gw[d] <= 0)
            {
            lo[d] = 0;
            hi[d] = nb - 1;
            continue;
            }
        const double W = (box.L[d] + 2.0 * gw[d]) / dim[d];
        // first and last cell wholly inside the box, with a margin far above the rounding of the binning
        const long first = (long)std::floor(gw[d] / W + 1e-3) + 1;
        const long last = (long)std::floor((gw[d] + box.L[d]) / W - 1e-3) - 1;
        // region of block b: cells max(2b-1, 0) .. min(2b+2, dim-1)
        const long blo = (first + 2) / 2;                   // smallest b with 2b-1 >= first
        const long bhi = last >= 2 ? (last - 2) / 2 : -1; // largest b with 2b+2 <= last
        if (first < 1 || bhi < blo || bhi >= (long)nb)
            return false;
        lo[d] = (unsigned int)blo;
        hi[d] = (unsigned int)bhi;
        }
    return true;
    }

void NeighborListGPUBinned::refreshTilePositions(hipStream_t stream)
    {
    const unsigned int N = m_pdata->getN(), ng = m_pdata->getNGhosts();
    // the build of this time step wrote every slot, ghosts included
    if (m_cpos_all_tstep == m_compute_tstep)
        return;
    // the integration pass may already have mirrored the owned particles of this time step
    const unsigned int first = (m_cpos_tstep == m_compute_tstep) ? N : 0u;
    if (first == N && ng == 0)
        return;
    ScopedKernelTimer t("tile_refresh", stream);
    abiCheck(hmd_tile_refresh(m_pdata->pos() + first, m_slot_of.get() + first, m_cpos.get(), N + ng - first, stream), "hmd_tile_refresh");
    }

// can this build write tile rows?  (no body filter, 3D, grid and occupancy within the tile limits; exclusions are
// applied by the build itself)
bool NeighborListGPUBinned::tilesUsable(unsigned int& capacity)
    {
    if (!m_tiles_enabled || m_tiles_overflowed || m_filter_body || m_sysdef->getNDimensions() != 3)
        return false;
    const hmd_index3d_t ci = m_cl->getCellIndexer();
    const hmd_index2d_t cadji = m_cl->getCellAdjIndexer();
    if (cadji.w != 27)
        return false;
    // Tiles stage the 64 cells around a block of 8 to serve the block's particles: that pays when cells are well
    // filled (LJ liquid, r_list 2.9: 20 per cell, tile force 137 us against 320 us from index rows at N = 2^20) and
    // loses when they are nearly empty (bead-spring melt, r_list 1.52: 3.3 per cell, 244 us against 62 us).
    const double occupancy = double(m_pdata->getN() + m_pdata->getNGhosts()) / (double(ci.w) * ci.h * ci.d);
    if (occupancy < m_tile_min_occupancy)
        return false;
    const hmd_box_t box = m_pdata->getBox().b;
    return hmd_nlist_tiles_supported(&ci, &box, m_cl->getNmax(), m_pdata->getNTypes(), &capacity) == HMD_OK;
    }

// index rows on demand: consumers other than the tile pair force (pair lists, local access, filters)
void NeighborListGPUBinned::ensureRows()
    {
    if (m_rows_valid)
        return;
    hmd_tiles_to_rows_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.geom = getTileGeom();
    a.d_tile_nlist = m_tile_nlist.get();
    a.d_n_neigh = m_n_neigh.get();
    a.d_head_list = m_head_list.get();
    a.d_nlist = m_nlist.get();
    a.N = m_pdata->getN();
    a.box = m_pdata->getBox().b;
    a.ntypes = m_pdata->getNTypes();
    abiCheck(hmd_nlist_tiles_to_rows(&a, nullptr), "hmd_nlist_tiles_to_rows");
    m_rows_valid = true;
    }

// hoomd/md/NeighborList.cc:363-435
void NeighborListGPUBinned::compute(uint64_t timestep)
    {
    if (m_rcut_changed)
        {
        updateRList();
        if (m_sysdef->isDomainDecomposed())
            m_sysdef->getCommunicator()->communicate(timestep);
        }
    if (!shouldCompute(timestep) && !m_force_update)
        return;

    if (m_force_update || m_head_dirty)
        {
        buildHeadList();
        if (m_exclusions_set)
            updateExListIdx();
        m_head_dirty = false;
        }

    m_compute_tstep = timestep;
    m_in_compute = true;
    const bool rebuild = needsUpdating(timestep);
    m_in_compute = false;
    if (rebuild)
        {
        m_last_build_tstep = timestep;
        m_cpos_tstep = ~0ull; // the slots are renumbered: refresh everything
        checkBoxSize();
        m_tiles_overflowed = false;
        bool overflowed = false;
        do
            {
            buildNlist(timestep);
            overflowed = checkConditions();
            // the cell list's own overflow check rode along (one wait of the host instead of two): when it had
            // overflowed it has been computed again by now, and the rows are built again from the complete list
            if (m_cl->finishConditions())
                overflowed = true;
            if (overflowed)
                {
                buildHeadList();
                resetConditions();
                }
            } while (overflowed);

        // (exclusions were left out by the build kernels themselves: no filter pass, hoomd/md/NeighborList.cc:423-426)

        // setLastUpdatedPos (hoomd/md/NeighborList.cc:1293-1311): the build kernel wrote last_pos
        m_last_L = m_pdata->getGlobalBox().getNearestPlaneDistance();
        m_has_been_updated_once = true;
        }
    }

// ---- exclusions ------------------------------------------------------------------------------
void NeighborListGPUBinned::clearExclusions()
    {
    m_ex_by_tag.clear();
    m_n_exclusions = 0;
    m_exclusions_set = false;
    m_ex_dirty = true;
    forceUpdate();
    }

// hoomd/md/NeighborList.cc:569-640: symmetric, duplicates ignored
// hoomd/md/NeighborList.cc:700-720
void NeighborListGPUBinned::addExclusionsFromBonds()
    {
    auto bonds = m_sysdef->getBondData();
    if (!bonds)
        return;
    for (const auto& g : bonds->getSnapshot().groups)
        addExclusion(g[0], g[1]);
    }

void NeighborListGPUBinned::addExclusion(unsigned int tag1, unsigned int tag2)
    {
    const unsigned int ng = m_pdata->getNGlobal();
    if (tag1 >= ng || tag2 >= ng)
        throw std::runtime_error("Particle tag out of bounds when attempting to add neighborlist exclusion");
    if (m_ex_by_tag.size() < ng)
        m_ex_by_tag.resize(ng);
    auto& l1 = m_ex_by_tag[tag1];
    if (std::find(l1.begin(), l1.end(), tag2) != l1.end())
        return;
    l1.push_back(tag2);
    m_ex_by_tag[tag2].push_back(tag1);
    m_n_exclusions++;
    m_exclusions_set = true;
    m_ex_dirty = true;
    forceUpdate();
    }

// hoomd/md/NeighborListGPU.cc:132-163
void NeighborListGPUBinned::updateExListIdx()
    {
    const unsigned int ng = m_pdata->getNGlobal();
    if (m_ex_dirty)
        {
        unsigned int height = 1;
        for (auto& l : m_ex_by_tag)
            height = std::max<unsigned int>(height, (unsigned int)l.size());
        m_ex_height = height;
        std::vector<unsigned int> n_ex(ng, 0), list((size_t)ng * height, 0);
        for (unsigned int t = 0; t < m_ex_by_tag.size(); t++)
            {
            n_ex[t] = (unsigned int)m_ex_by_tag[t].size();
            for (unsigned int o = 0; o < n_ex[t]; o++)
                list[(size_t)o * ng + t] = m_ex_by_tag[t][o];
            }
        m_n_ex_tag.resize(ng, false);
        m_ex_list_tag.resize((size_t)ng * height, false);
        m_n_ex_tag.upload(n_ex.data(), ng);
        m_ex_list_tag.upload(list.data(), list.size());
        m_ex_list_idx.resize((size_t)std::max(m_pdata->getMaxN(), 1u) * height, false);
        m_ex_dirty = false;
        }
    hmd_update_exclusion_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.d_tag = m_pdata->tag();
    a.d_rtag = m_pdata->rtag();
    a.d_n_ex_tag = m_n_ex_tag.get();
    a.d_ex_list_tag = m_ex_list_tag.get();
    a.ex_list_tag_indexer = hmd_index2d_t {ng, m_ex_height};
    a.d_n_ex_idx = m_n_ex_idx.get();
    a.d_ex_list_idx = m_ex_list_idx.get();
    a.ex_list_indexer = hmd_index2d_t {std::max(m_pdata->getMaxN(), 1u), m_ex_height};
    a.N = m_pdata->getN();
    abiCheck(hmd_update_exclusion_list(&a, nullptr), "hmd_update_exclusion_list");
    }

// hoomd/md/NeighborList.cc:1726-1800 (getLocalPairList / getPairList)
std::vector<std::pair<unsigned int, unsigned int>> NeighborListGPUBinned::getLocalPairList()
    {
    const unsigned int n = m_pdata->getN();
    auto n_neigh = m_n_neigh.toHost(n);
    auto head = m_head_list.toHost(n);
    ensureRows();
    auto nlist = m_nlist.toHost();
    std::vector<std::pair<unsigned int, unsigned int>> out;
    for (unsigned int i = 0; i < n; i++)
        for (unsigned int k = 0; k < n_neigh[i]; k++)
            {
            const unsigned int j = nlist[head[i] + k];
            if (m_storage_mode == half || i < j)
                out.emplace_back(i, j);
            }
    return out;
    }

std::vector<std::pair<unsigned int, unsigned int>> NeighborListGPUBinned::getPairList()
    {
    auto local = getLocalPairList();
    auto tag = m_pdata->getN() + m_pdata->getNGhosts() ? std::vector<unsigned int>(m_pdata->getN() + m_pdata->getNGhosts()) : std::vector<unsigned int>();
    if (!tag.empty())
        hipCheck(hipMemcpy(tag.data(), m_pdata->tag(), sizeof(unsigned int) * tag.size(), hipMemcpyDeviceToHost), "tags D2H");
    for (auto& p : local)
        p = std::make_pair(tag[p.first], tag[p.second]);
    return local;
    }

    } // namespace hoomd_amd
