// hoomd/md/PotentialPairGPU.h -- PotentialPairGPU<evaluator>: queues the pair-force kernels of one evaluator
// (role of hoomd/md/PotentialPairGPU.h:46-190 of the reference; kernel arguments as :89-160).
//
// The kernels are the instantiation gpu_compute_pair_forces<evaluator> that the evaluator's HIP translation unit names
// (hoomd/md/PotentialPairGPU.cuh).  Besides the plain launch after the neighbor list is current, the class hands the
// neighbor list two hooks: the launch queued behind the distance check and predicated on "list still valid" (so the
// GPU does not idle while the host reads the check's result), and, under domain decomposition, the launch of the tile
// blocks that see no ghosts ahead of the halo exchange.
#ifndef HOOMD_AMD_POTENTIAL_PAIR_GPU_H
#define HOOMD_AMD_POTENTIAL_PAIR_GPU_H

#include "hoomd/md/PotentialPair.h"
#include "hoomd/md/PotentialPairGPU.cuh"

namespace hoomd
    {
namespace md
    {
template<class evaluator> class PotentialPairGPU : public PotentialPair<evaluator>
    {
    public:
    typedef typename evaluator::param_type param_type;
    PotentialPairGPU(std::shared_ptr<SystemDefinition> sysdef, std::shared_ptr<NeighborList> nlist) : PotentialPair<evaluator>(sysdef, nlist)
        {
        this->m_nlist->setSpeculationHook(
            [this](uint64_t timestep, const unsigned int* d_flag, unsigned int stale, unsigned long long* d_published, unsigned int stamp, hipStream_t stream)
            {
            m_spec_tstep = timestep;
            // the interior blocks of this step may already be running on the default stream: then this launch takes the
            // other blocks, on the stream the halo exchange ran on
            return launchKernel(d_flag, stale, d_published, stamp, m_interior_tstep == timestep ? 2u : 0u, stream);
            });
        this->m_nlist->setInteriorHook([this](uint64_t timestep, const unsigned int* d_flag, unsigned int stale)
                                       { return launchInterior(timestep, d_flag, stale); });
        const size_t np = (size_t)this->m_ntypes * this->m_ntypes;
        m_d_params.resize(np);
        m_d_rcutsq.resize(np);
        m_d_ronsq.resize(np);
        }
    virtual ~PotentialPairGPU()
        {
        if (this->m_nlist)
            {
            this->m_nlist->setSpeculationHook(nullptr);
            this->m_nlist->setInteriorHook(nullptr);
            }
        }
    //! lanes per particle of the row kernel; 0 = chosen from the row lengths (the reference's autotuner parameter)
    void setThreadsPerParticle(unsigned int tpp)
        {
        m_tpp = tpp;
        }
    unsigned int getThreadsPerParticle() const
        {
        return m_tpp;
        }

    protected:
    void computeForces(uint64_t timestep) override
        {
        // The speculation hook may already have queued this step's kernel: from inside compute() below, or, under
        // domain decomposition, from the communicator's look at the rebuild decision before this call.
        m_force_tstep = timestep;
        this->m_nlist->compute(timestep);
        if (this->m_nlist->getStorageMode() == NeighborList::half)
            throw std::runtime_error("PotentialPairGPU cannot handle a half neighborlist");
        // the speculative launch stands when the list was not rebuilt at this step; it is used once
        const bool queued = m_spec_tstep == timestep && this->m_nlist->getLastBuildTimestep() != timestep;
        m_spec_tstep = ~0ull;
        if (!queued)
            {
            m_rider_tstep = ~0ull; // (a switched-off launch has not run its pass either)
            launchKernel(nullptr, 0);
            }
        if (m_rider_tstep == timestep)
            {
            // the kernel that stands carried the integrator's half steps: the positions of timestep + 1 exist
            this->m_nlist->commitStepRider(timestep + 1);
            this->m_pdata->notifyStepRiderDone(timestep);
            }
        m_rider_tstep = ~0ull;
        m_force_tstep = ~0ull;
        this->computeTailCorrection();
        }
    //! the time step of the force computation in progress (launchKernel is also called from the neighbor list's hooks)
    uint64_t m_force_tstep = ~0ull;
    uint64_t m_rider_tstep = ~0ull; // time step whose queued kernel carries the integrator's half steps
    //! part: 0 every particle; 1 the blocks of tiles that see no ghosts; 2 the other blocks
    bool launchKernel(const unsigned int* d_skip_flag, unsigned int skip_value, unsigned long long* d_published = nullptr, unsigned int stamp = 0,
                      unsigned int part = 0, hipStream_t stream = nullptr);
    // Queue the tile blocks that see no ghosts, predicated on the local result of the distance check, before the halo
    // exchange is queued (on another stream): they run while the ghost positions travel.  When another rank votes for
    // a rebuild their result is simply overwritten by the launch that follows the rebuild.
    bool launchInterior(uint64_t timestep, const unsigned int* d_flag, unsigned int stale)
        {
        if (!this->m_nlist->tilesActive() || this->m_pdata->getN() == 0)
            return false;
        if (!this->m_nlist->getInteriorBlockWindow(timestep, m_win_lo, m_win_hi))
            return false;
        launchKernel(d_flag, stale, nullptr, 0, 1u);
        m_interior_tstep = timestep;
        return true;
        }

    uint64_t m_spec_tstep = ~0ull;     // time step whose force kernel was queued speculatively
    uint64_t m_interior_tstep = ~0ull; // time step whose interior blocks were queued ahead of the exchange
    unsigned int m_win_lo[3] = {0, 0, 0}, m_win_hi[3] = {0, 0, 0};
    unsigned int m_tpp = 0;
    hoomd_amd::DeviceArray<param_type> m_d_params;
    hoomd_amd::DeviceArray<Scalar> m_d_rcutsq, m_d_ronsq;
    };

template<class evaluator>
bool PotentialPairGPU<evaluator>::launchKernel(const unsigned int* d_skip_flag, unsigned int skip_value, unsigned long long* d_published,
                                               unsigned int stamp, unsigned int part, hipStream_t stream)
    {
    using namespace hoomd_amd;
    auto& nlist = this->m_nlist;
    auto& pdata = this->m_pdata;
    if (this->m_tables_dirty)
        {
        m_d_params.upload(this->m_params.data(), this->m_params.size());
        m_d_rcutsq.upload(this->m_rcutsq.data(), this->m_rcutsq.size());
        m_d_ronsq.upload(this->m_ronsq.data(), this->m_ronsq.size());
        this->m_tables_dirty = false;
        }
    kernel::pair_args_t pa;
    std::memset(&pa, 0, sizeof(pa));
    hmd_pair_args_t& a = pa.args;
    pa.stream = stream;
    a.d_force = this->m_force.get();
    a.d_virial = this->m_virial.get();
    a.virial_pitch = this->m_virial_pitch;
    a.N = pdata->getN();
    a.n_max = pdata->getMaxN();
    a.d_pos = pdata->pos();
    a.d_charge = pdata->charge();
    a.box = pdata->getBox().b;
    const bool tiles = nlist->tilesActive();
    a.d_n_neigh = nlist->getNNeighArray().get();
    a.d_nlist = tiles ? nullptr : nlist->getNListArray().get();
    a.d_head_list = nlist->getHeadList().get();
    a.d_rcutsq = m_d_rcutsq.get();
    a.d_ronsq = m_d_ronsq.get();
    a.size_neigh_list = tiles ? nlist->getTileNlist().size() : nlist->getNListArray().size();
    a.ntypes = this->m_ntypes;
    a.shift_mode = (unsigned int)this->m_shift_mode;
    a.compute_virial = pdata->getFlag(pressure_tensor) ? 1u : 0u;
    a.threads_per_particle = m_tpp;
    if (tiles)
        a.threads_per_particle = 0; // the tile kernel always puts four lanes on a home particle (its row layout is built for that)
    if (!tiles && m_tpp == 0)
        {
        // the reference tunes this at run time (hoomd/md/PotentialPairGPU.h:120-145); here: rows of at most 32
        // entries (short cutoffs, e.g. WCA) are walked by 4 lanes, longer rows by 8 (measured at 11 and 84 neighbors)
        unsigned int longest = 0;
        for (unsigned int n : nlist->getNmaxPerType())
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
        nlist->refreshTilePositions(stream);
    hmd_tile_geom_t g;
    hmd_fused_step_args_t step;
    if (tiles)
        {
        g = nlist->getTileGeom();
        pa.tile_geom = &g;
        pa.d_tile_nlist = nlist->getTileNlist().get();
        // The integrator may offer its half steps around this force computation to the kernel (one NVE method, every
        // particle, the next time step certain to follow): taken when the whole grid is launched at once, without
        // virials, by the tile kernel, in the case the kernel carries (hmd_pair_forces_tiles_step_supported).
        const auto& source = pdata->getStepRiderSource();
        if (source && part == 0 && !a.compute_virial && m_force_tstep != ~0ull && a.N != 0)
            {
            std::memset(&step, 0, sizeof(step));
            if (source(step, m_force_tstep) && nlist->offerStepRider(step, m_force_tstep + 1))
                {
                if (hmd_pair_forces_tiles_step_supported(&a, &step) == HMD_OK)
                    {
                    pa.step = &step;
                    m_rider_tstep = m_force_tstep;
                    }
                else
                    nlist->withdrawStepRider(m_force_tstep + 1);
                }
            }
        }
    ScopedKernelTimer t(part == 1 ? "pair_force_interior" : (pa.step ? "pair_force_step" : "pair_force"), stream);
    if (kernel::gpu_compute_pair_forces<evaluator>(pa, m_d_params.get()) != hipSuccess)
        throw std::runtime_error(std::string("gpu_compute_pair_forces<") + std::string(evaluator::getName()) + ">: " + hmd_last_error());
    return a.N != 0; // the drivers launch nothing for an empty system
    }

namespace detail
    {
// hoomd/md/PotentialPairGPU.h:180-186
template<class T> void export_PotentialPairGPU(pybind11::module& m, const std::string& name)
    {
    pybind11::class_<PotentialPairGPU<T>, PotentialPair<T>, std::shared_ptr<PotentialPairGPU<T>>>(m, name.c_str())
        .def(pybind11::init<std::shared_ptr<SystemDefinition>, std::shared_ptr<NeighborList>>())
        .def_property("threads_per_particle", &PotentialPairGPU<T>::getThreadsPerParticle, &PotentialPairGPU<T>::setThreadsPerParticle);
    }
    } // namespace detail
    } // namespace md
    } // namespace hoomd

#endif
