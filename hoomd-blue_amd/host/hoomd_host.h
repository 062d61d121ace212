// hoomd_host.h -- C++ host layer above the C ABI (include/hoomd_amd.h).
//
// The classes keep the reference's names, virtual interfaces and host-side bookkeeping for the
// short-range MD path (ParticleData, CellList, NeighborList, ForceCompute, PotentialPair,
// IntegrationMethodTwoStep, Integrator, System, Communicator); the work itself is done by the gfx950
// kernel drivers.  Reference files are cited next to each class (paths relative to HOOMD-blue 4.1.0).
// Everything runs on one HIP stream (the default stream, as the reference does); device memory is
// owned by DeviceArray members; the only host<->device traffic per step is the 4-byte rebuild flag
// and, on rebuild steps, the small overflow/condition words.
#ifndef HOOMD_HOST_H
#define HOOMD_HOST_H

#include <hip/hip_runtime_api.h>

#include <array>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <functional>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "hoomd_amd.h"

namespace hoomd_amd
    {
typedef double Scalar;
typedef hmd_scalar4 Scalar4;
typedef hmd_scalar3 Scalar3;
typedef hmd_int3 int3;

void hipCheck(hipError_t e, const char* what);
void abiCheck(int code, const char* what);

// ---------------------------------------------------------------------------------------------
// Device memory (role of GlobalArray / GPUArray, hoomd/GlobalArray.h, hoomd/GPUArray.h)
// ---------------------------------------------------------------------------------------------
template<class T> class DeviceArray
    {
    public:
    DeviceArray() = default;
    explicit DeviceArray(size_t n)
        {
        resize(n, false);
        }
    ~DeviceArray()
        {
        release();
        }
    DeviceArray(const DeviceArray&) = delete;
    DeviceArray& operator=(const DeviceArray&) = delete;

    //! grow or shrink; contents are kept up to min(old, new) when keep is set; new memory is zeroed
    void resize(size_t n, bool keep = true)
        {
        if (n == m_n)
            return;
        T* p = nullptr;
        if (n)
            {
            hipCheck(hipMalloc((void**)&p, n * sizeof(T)), "hipMalloc");
            hipCheck(hipMemsetAsync(p, 0, n * sizeof(T), 0), "hipMemsetAsync");
            if (keep && m_p && m_n)
                hipCheck(hipMemcpyAsync(p, m_p, (n < m_n ? n : m_n) * sizeof(T), hipMemcpyDeviceToDevice, 0), "hipMemcpyAsync");
            }
        if (m_p)
            {
            hipCheck(hipStreamSynchronize(0), "hipStreamSynchronize");
            hipCheck(hipFree(m_p), "hipFree");
            }
        m_p = p;
        m_n = n;
        }
    void release()
        {
        if (m_p)
            (void)hipFree(m_p);
        m_p = nullptr;
        m_n = 0;
        }
    void swap(DeviceArray& o)
        {
        std::swap(m_p, o.m_p);
        std::swap(m_n, o.m_n);
        }
    void zero()
        {
        if (m_n)
            hipCheck(hipMemsetAsync(m_p, 0, m_n * sizeof(T), 0), "hipMemsetAsync");
        }
    T* get() const
        {
        return m_p;
        }
    size_t size() const
        {
        return m_n;
        }
    void upload(const T* h, size_t n, size_t offset = 0)
        {
        if (n)
            hipCheck(hipMemcpy(m_p + offset, h, n * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D");
        }
    void download(T* h, size_t n, size_t offset = 0) const
        {
        if (n)
            hipCheck(hipMemcpy(h, m_p + offset, n * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy D2H");
        }
    std::vector<T> toHost(size_t n = (size_t)-1) const
        {
        if (n == (size_t)-1)
            n = m_n;
        std::vector<T> v(n);
        download(v.data(), n);
        return v;
        }

    private:
    T* m_p = nullptr;
    size_t m_n = 0;
    };

//! small pinned host word block for flag readbacks
class PinnedWords
    {
    public:
    explicit PinnedWords(size_t n);
    ~PinnedWords();
    unsigned int* get()
        {
        return m_p;
        }
    //! address of the same words as the device sees them
    unsigned int* deviceView();

    private:
    unsigned int* m_p = nullptr;
    unsigned int* m_d = nullptr;
    };

//! Spins until the 8-byte word at `h` (mapped pinned memory a kernel stores to) carries `stamp` in its upper half and
//! returns the word.  The default stream and `other` (when not null) are probed for errors while waiting, and after a
//! wall-clock limit (HOOMD_AMD_POLL_TIMEOUT_S, default 120 s) the wait ends in a std::runtime_error that names `what`,
//! the stamp and which of the streams still had work queued, so a stuck exchange ends the process instead of spinning.
unsigned long long pollStampedWord(volatile unsigned long long* h, unsigned int stamp, const char* what, hipStream_t other = nullptr);
//! seconds the host has spent inside pollStampedWord since the last reset
double pollWaitSeconds(bool reset);

//! Reads a few device words without a stream synchronisation: a one-wave kernel queued behind the producer
//! stores them, stamped, to mapped pinned memory that the host polls (hmd_publish_words).
class WordReadback
    {
    public:
    WordReadback() : m_pinned(128) { }
    //! blocks until the words have arrived; n <= 64
    void read(const unsigned int* d_words, unsigned int n, unsigned int* out, const char* what);
    //! the same in two halves: begin() queues the publication behind the work that produces the words, finish() waits
    //! for it -- in between the caller queues more work, so the device does not sit idle while the host looks
    void begin(const unsigned int* d_words, unsigned int n);
    void finish(unsigned int* out, const char* what);
    //! for a kernel that publishes n words itself (stamped, as hmd_publish_words does): where to, and the stamp to use;
    //! finish() then waits for them
    unsigned long long* expect(unsigned int n, unsigned int& stamp);

    private:
    PinnedWords m_pinned; // 64 slots of 8 bytes
    unsigned int m_stamp = 0, m_n = 0;
    };

// ---------------------------------------------------------------------------------------------
// signals (role of Nano::Signal, used for particle-sort / box-change / r_cut-change notifications)
// ---------------------------------------------------------------------------------------------
class Signal
    {
    public:
    int connect(std::function<void()> f)
        {
        m_slots.emplace_back(m_next, std::move(f));
        return m_next++;
        }
    void disconnect(int id)
        {
        for (size_t i = 0; i < m_slots.size(); i++)
            if (m_slots[i].first == id)
                {
                m_slots.erase(m_slots.begin() + i);
                return;
                }
        }
    void emit() const
        {
        for (auto& s : m_slots)
            s.second();
        }

    private:
    std::vector<std::pair<int, std::function<void()>>> m_slots;
    int m_next = 0;
    };

// ---------------------------------------------------------------------------------------------
// Box (hoomd/BoxDim.h) -- thin value wrapper around hmd_box_t
// ---------------------------------------------------------------------------------------------
struct BoxDim
    {
    hmd_box_t b;
    BoxDim()
        {
        hmd_box_init(&b, 1, 1, 1, 0, 0, 0);
        }
    BoxDim(Scalar Lx, Scalar Ly, Scalar Lz, Scalar xy = 0, Scalar xz = 0, Scalar yz = 0)
        {
        hmd_box_init(&b, Lx, Ly, Lz, xy, xz, yz);
        }
    void setLoHi(const Scalar lo[3], const Scalar hi[3])
        {
        hmd_box_set_lo_hi(&b, lo, hi);
        }
    void setPeriodic(bool x, bool y, bool z)
        {
        b.periodic[0] = x;
        b.periodic[1] = y;
        b.periodic[2] = z;
        }
    std::array<Scalar, 3> getNearestPlaneDistance() const
        {
        return {b.nearest_plane[0], b.nearest_plane[1], b.nearest_plane[2]};
        }
    Scalar getVolume(bool twod = false) const
        {
        return twod ? b.L[0] * b.L[1] : b.L[0] * b.L[1] * b.L[2];
        }
    };

// ---------------------------------------------------------------------------------------------
// Snapshot exchanged with Python (hoomd/SnapshotSystemData.h, hoomd/ParticleData.h:168-262)
// ---------------------------------------------------------------------------------------------
struct SnapshotParticleData
    {
    std::vector<std::array<Scalar, 3>> pos, vel, accel;
    std::vector<unsigned int> type, body;
    std::vector<unsigned int> tag; //!< global tags of the rows; empty on input means 0..N-1
    std::vector<Scalar> mass, charge, diameter;
    std::vector<std::array<int, 3>> image;
    std::vector<std::string> type_mapping;
    void resize(unsigned int n);
    unsigned int size = 0;
    };

enum pdata_flag
    {
    pressure_tensor = 0,
    rotational_kinetic_energy,
    external_field_virial
    };

class Communicator;

// ---------------------------------------------------------------------------------------------
// ParticleData (hoomd/ParticleData.h:1308-1357 array set; only what the path touches)
// ---------------------------------------------------------------------------------------------
class ParticleData
    {
    public:
    //! n_global = 0 means the snapshot holds every particle (single rank)
    ParticleData(const SnapshotParticleData& snap, const BoxDim& global_box, unsigned int n_global = 0);

    unsigned int getN() const
        {
        return m_N;
        }
    unsigned int getNGhosts() const
        {
        return m_nghost;
        }
    unsigned int getNGlobal() const
        {
        return m_nglobal;
        }
    //! counts the times the particle set was replaced as a whole (initializeFromSnapshot): anything derived from the
    //! composition of the system -- global per-type particle counts -- is stale when it has changed
    uint64_t getCompositionGeneration() const
        {
        return m_composition_generation;
        }
    void setNGlobal(unsigned int n)
        {
        m_nglobal = n;
        }
    unsigned int getMaxN() const
        {
        return m_maxn;
        }
    unsigned int getNTypes() const
        {
        return (unsigned int)m_type_mapping.size();
        }
    const std::vector<std::string>& getTypeMapping() const
        {
        return m_type_mapping;
        }
    unsigned int getTypeByName(const std::string& name) const;

    const BoxDim& getBox() const
        {
        return m_box;
        }
    const BoxDim& getGlobalBox() const
        {
        return m_global_box;
        }
    //! local box differs from the global one under domain decomposition
    void setBoxes(const BoxDim& global_box, const BoxDim& local_box);

    //! owned-particle count changes (migration); arrays grow geometrically, contents kept
    void resizeOwned(unsigned int n);
    //! ghosts live at [N, N+nghost)
    void setNGhosts(unsigned int n);
    void reallocate(unsigned int max_n);

    // device arrays
    Scalar4* pos() const
        {
        return m_pos.get();
        }
    Scalar4* vel() const
        {
        return m_vel.get();
        }
    Scalar3* accel() const
        {
        return m_accel.get();
        }
    int3* image() const
        {
        return m_image.get();
        }
    unsigned int* tag() const
        {
        return m_tag.get();
        }
    //! index of every tag (local particles and ghosts; 0xffffffff for tags this rank does not hold); brought up to date
    //! here when an exchange has left it stale (markRtagStale)
    unsigned int* rtag() const
        {
        if (m_rtag_stale)
            refreshRtag();
        return m_rtag.get();
        }
    //! the set of local particles and ghosts has changed (migration, new ghost layer): the next reader of rtag()
    //! rebuilds it -- runs without bonds, exclusions or snapshots never pay for the two launches
    void markRtagStale()
        {
        m_rtag_stale = true;
        }
    unsigned int* body() const
        {
        return m_body.get();
        }
    Scalar* charge() const
        {
        return m_charge.get();
        }
    Scalar* diameter() const
        {
        return m_diameter.get();
        }
    Scalar4* netForce() const
        {
        materializeNetForce();
        return m_net_force.get();
        }
    //! the storage itself, for the pass that fills it
    Scalar4* netForceStorage() const
        {
        return m_net_force.get();
        }
    Scalar* netVirial() const
        {
        return m_net_virial.get();
        }
    size_t netVirialPitch() const
        {
        return m_maxn;
        }
    //! A neighbor list may ask whoever moves the particles to do two things in the same pass: run its
    //! distance check on the positions of `timestep` and mirror them into its slot-ordered copy.  The
    //! hook fills the corresponding fields of the pass's arguments (or leaves them zero).
    typedef std::function<void(hmd_fused_step_args_t&, uint64_t)> DisplacementCheckHook;
    void setDisplacementCheckHook(DisplacementCheckHook h)
        {
        m_check_hook = h;
        }
    const DisplacementCheckHook& getDisplacementCheckHook() const
        {
        return m_check_hook;
        }

    //! The integrator's offer to the force compute: take the second half of this time step and the first half of the
    //! next one into the force kernel (hmd_compute_pair_forces_lj_tiles_step).  The source fills two / one of the
    //! arguments for forces of `timestep` and says whether the offer stands; a force compute that has run the pass
    //! says so with notifyStepRiderDone(timestep), and the integrator skips its own pass.
    typedef std::function<bool(hmd_fused_step_args_t&, uint64_t)> StepRiderSource;
    void setStepRiderSource(StepRiderSource s)
        {
        m_rider_source = s;
        }
    const StepRiderSource& getStepRiderSource() const
        {
        return m_rider_source;
        }
    void notifyStepRiderDone(uint64_t timestep)
        {
        m_rider_done_tstep = timestep;
        }
    uint64_t getStepRiderDone() const
        {
        return m_rider_done_tstep;
        }

    //! when a single force compute feeds the integrator its force array IS the net force
    void aliasNetForce(Scalar4* p)
        {
        m_net_force_alias = p;
        }
    Scalar4* netForceForIntegration() const
        {
        materializeNetForce();
        return m_net_force_alias ? m_net_force_alias : m_net_force.get();
        }
    //! Two forces and no virial request: the integrator's kernels add the two arrays where they read them, and the net
    //! force array is only filled (by `sum`, the net-force pass) when somebody asks for it.
    void setNetForcePair(const Scalar4* f1, const Scalar4* f2, std::function<void()> sum)
        {
        m_pair_f1 = f1;
        m_pair_f2 = f2;
        m_pair_sum = sum;
        }
    //! the two arrays whose sum is the net force, while that sum has not been stored
    bool getNetForcePair(const Scalar4*& f1, const Scalar4*& f2) const
        {
        if (!m_pair_sum)
            return false;
        f1 = m_pair_f1;
        f2 = m_pair_f2;
        return true;
        }
    void materializeNetForce() const
        {
        if (m_pair_sum)
            {
            auto sum = std::move(m_pair_sum);
            m_pair_sum = nullptr;
            sum();
            }
        }

    bool isAccelSet() const
        {
        return m_accel_set;
        }
    void notifyAccelSet()
        {
        m_accel_set = true;
        }

    bool getFlag(pdata_flag f) const
        {
        return (m_flags >> f) & 1u;
        }
    unsigned int getFlags() const
        {
        return m_flags;
        }
    void setFlags(unsigned int f)
        {
        m_flags = f;
        }

    //! apply a permutation: new[k] = old[order[k]] to every per-particle array, rebuild rtag, emit sort signal
    void applyOrder(const unsigned int* d_order);
    void notifyParticleSort()
        {
        m_sort_count++;
        m_sort_signal.emit();
        }
    //! number of reorderings so far (lets the Python-level communicator notice a sort)
    uint64_t getSortCount() const
        {
        return m_sort_count;
        }
    Signal& getParticleSortSignal()
        {
        return m_sort_signal;
        }
    Signal& getBoxChangeSignal()
        {
        return m_box_signal;
        }
    Signal& getMaxParticleNumberChangeSignal()
        {
        return m_maxn_signal;
        }
    Signal& getGlobalParticleNumberChangeSignal()
        {
        return m_nglobal_signal;
        }

    //! owned particles in tag order (local tags only under domain decomposition)
    void takeSnapshot(SnapshotParticleData& snap) const;
    void initializeFromSnapshot(const SnapshotParticleData& snap);
    //! overwrite positions/velocities/... of owned particles given full-size tag-ordered arrays
    void setExternalEnergy(Scalar e)
        {
        m_external_energy = e;
        }
    Scalar getExternalEnergy() const
        {
        return m_external_energy;
        }
    void setExternalVirial(unsigned int i, Scalar v)
        {
        m_external_virial[i] = v;
        }
    Scalar getExternalVirial(unsigned int i) const
        {
        return m_external_virial[i];
        }

    private:
    unsigned int m_N = 0, m_nghost = 0, m_nglobal = 0, m_maxn = 0;
    BoxDim m_box, m_global_box;
    std::vector<std::string> m_type_mapping;
    DeviceArray<Scalar4> m_pos, m_vel, m_net_force, m_alt4;
    DeviceArray<Scalar3> m_accel, m_alt3;
    DeviceArray<int3> m_image, m_alti3;
    DeviceArray<unsigned int> m_tag, m_rtag, m_body, m_altu;
    mutable bool m_rtag_stale = false;
    void refreshRtag() const;
    DeviceArray<Scalar> m_charge, m_diameter, m_net_virial, m_alt1;
    Scalar4* m_net_force_alias = nullptr;
    DisplacementCheckHook m_check_hook;
    StepRiderSource m_rider_source;
    const Scalar4* m_pair_f1 = nullptr;
    const Scalar4* m_pair_f2 = nullptr;
    mutable std::function<void()> m_pair_sum;
    uint64_t m_rider_done_tstep = ~0ull;
    bool m_accel_set = false;
    unsigned int m_flags = 0;
    Scalar m_external_energy = 0;
    Scalar m_external_virial[6] = {0, 0, 0, 0, 0, 0};
    Signal m_sort_signal, m_box_signal, m_maxn_signal, m_nglobal_signal;
    uint64_t m_composition_generation = 0;
    uint64_t m_sort_count = 0;
    };

// ---------------------------------------------------------------------------------------------
// BondData: bonded pairs by particle tag and their per-particle GPU table
// (hoomd/BondedGroupData.h, group_size 2; snapshot section hoomd/BondedGroupData.h:350-420)
// ---------------------------------------------------------------------------------------------
struct SnapshotBondData
    {
    std::vector<std::array<unsigned int, 2>> groups; //!< member tags
    std::vector<unsigned int> type_id;
    std::vector<std::string> type_mapping;
    };

class BondData
    {
    public:
    BondData(std::shared_ptr<ParticleData> pdata, const SnapshotBondData& snap);
    ~BondData();
    unsigned int getNGlobal() const
        {
        return (unsigned int)m_snap.groups.size();
        }
    unsigned int getNTypes() const
        {
        return (unsigned int)m_snap.type_mapping.size();
        }
    const SnapshotBondData& getSnapshot() const
        {
        return m_snap;
        }
    //! bonds per particle (index order) and the table of (partner index, bond type); brought up to
    //! date with the particle order on request (hoomd/BondedGroupData.cc rebuildGPUTable)
    const unsigned int* getNBondsArray()
        {
        update();
        return m_n_bonds.get();
        }
    const hmd_uint2* getGPUTable()
        {
        update();
        return m_blist.get();
        }
    hmd_index2d_t getGPUTableIndexer() const
        {
        return m_blist_idx;
        }
    //! throws "bond incomplete" if the last table update met a partner that is not local
    void checkComplete();

    private:
    void update();
    std::shared_ptr<ParticleData> m_pdata;
    SnapshotBondData m_snap;
    DeviceArray<unsigned int> m_tag_n_bonds, m_n_bonds;
    DeviceArray<hmd_uint2> m_tag_blist, m_blist;
    hmd_index2d_t m_tag_blist_idx {0, 0}, m_blist_idx {0, 0};
    PinnedWords m_flags;
    bool m_dirty = true;
    int m_sort_slot, m_maxn_slot;
    };

class SystemDefinition
    {
    public:
    SystemDefinition(const SnapshotParticleData& snap, const BoxDim& box, unsigned int ndim, unsigned int seed, unsigned int n_global = 0)
        : m_pdata(std::make_shared<ParticleData>(snap, box, n_global)), m_ndim(ndim), m_seed(seed)
        {
        }
    std::shared_ptr<ParticleData> getParticleData() const
        {
        return m_pdata;
        }
    unsigned int getNDimensions() const
        {
        return m_ndim;
        }
    unsigned int getSeed() const
        {
        return m_seed;
        }
    bool isDomainDecomposed() const
        {
        return (bool)m_comm;
        }
    std::shared_ptr<Communicator> getCommunicator() const
        {
        return m_comm;
        }
    void setCommunicator(std::shared_ptr<Communicator> c)
        {
        m_comm = c;
        }
    //! bonds of the system (hoomd/SystemDefinition.h getBondData); null when there are none
    std::shared_ptr<BondData> getBondData() const
        {
        return m_bonds;
        }
    void setBondData(const SnapshotBondData& snap)
        {
        m_bonds = snap.groups.empty() && snap.type_mapping.empty() ? nullptr : std::make_shared<BondData>(m_pdata, snap);
        }

    private:
    std::shared_ptr<ParticleData> m_pdata;
    unsigned int m_ndim;
    unsigned int m_seed;
    std::shared_ptr<Communicator> m_comm;
    std::shared_ptr<BondData> m_bonds;
    };

// ---------------------------------------------------------------------------------------------
// Communicator interface (hoomd/Communicator.h:166-232).  The exchange is implemented in Python on
// top of torch.distributed (RCCL); the host classes only see this interface.
// ---------------------------------------------------------------------------------------------
class Communicator
    {
    public:
    virtual ~Communicator() { }
    //! migrate / exchange ghosts or update ghost positions (hoomd/Communicator.cc:1510-1569)
    virtual void communicate(uint64_t timestep) = 0;
    virtual void forceMigrate() = 0;
    //! largest ghost layer width requested by the subscribers
    virtual Scalar getGhostLayerMaxWidth() const = 0;
    //! global logical OR of a local flag (the rebuild vote, hoomd/md/NeighborListGPU.cc:84-98)
    virtual bool anyRank(bool local) = 0;
    //! The same vote without a host round trip in the middle: d_flag is the device word of the distance
    //! check ("stale" when it equals checkn; it only ever holds check numbers, which all ranks count alike).
    //! Queues a MAX-reduction of that word over all ranks behind the check, in place, and returns its address:
    //! afterwards it equals checkn on every rank iff the list of any rank is stale (0 = not supported: fall
    //! back to anyRank).  The caller reads the word itself (hmd_publish_word); endRebuildVote() is the slow
    //! way to the same answer.
    //! stream: HIP stream to queue the exchange on (0 = the default stream)
    virtual uintptr_t beginRebuildVote(uintptr_t d_flag, unsigned int checkn, uintptr_t stream)
        {
        return 0;
        }
    virtual bool endRebuildVote()
        {
        return false;
        }
    //! in-place sum over all ranks of n host doubles (hoomd/ForceCompute.cc:218, hoomd/md/PotentialPair.h:430)
    virtual void allReduceSum(double* v, int n) { }
    virtual unsigned int getRank() const
        {
        return 0;
        }
    };

// ---------------------------------------------------------------------------------------------
// Compute (hoomd/Compute.cc:42-76)
// ---------------------------------------------------------------------------------------------
class Compute
    {
    public:
    explicit Compute(std::shared_ptr<SystemDefinition> sysdef) : m_sysdef(sysdef), m_pdata(sysdef->getParticleData()) { }
    virtual ~Compute() { }
    virtual void compute(uint64_t timestep) { }
    void forceCompute(uint64_t timestep)
        {
        m_force_compute = true;
        compute(timestep);
        }

    protected:
    bool shouldCompute(uint64_t timestep);
    std::shared_ptr<SystemDefinition> m_sysdef;
    std::shared_ptr<ParticleData> m_pdata;
    bool m_first_compute = true, m_force_compute = false;
    uint64_t m_last_computed = 0;
    };

// ---------------------------------------------------------------------------------------------
// ParticleGroup (hoomd/ParticleGroup.h): all particles, or an explicit tag list
// ---------------------------------------------------------------------------------------------
class ParticleGroup
    {
    public:
    //! group of all particles
    explicit ParticleGroup(std::shared_ptr<SystemDefinition> sysdef);
    //! group of the given tags
    ParticleGroup(std::shared_ptr<SystemDefinition> sysdef, const std::vector<unsigned int>& tags);
    ~ParticleGroup();
    unsigned int getNumMembers();
    unsigned int getNumMembersGlobal() const
        {
        return m_is_all ? m_pdata->getNGlobal() : (unsigned int)m_tags.size();
        }
    //! device index list, nullptr for the identity group
    const unsigned int* getIndexArray();
    bool isAll() const
        {
        return m_is_all;
        }
    void setTranslationalDOF(Scalar d)
        {
        m_dof = d;
        }
    Scalar getTranslationalDOF() const
        {
        return m_dof;
        }

    private:
    void rebuild();
    std::shared_ptr<ParticleData> m_pdata;
    bool m_is_all;
    std::vector<unsigned int> m_tags;
    DeviceArray<unsigned int> m_idx;
    unsigned int m_n_local = 0;
    bool m_dirty = true;
    int m_slot = -1;
    Scalar m_dof = 0;
    };

// ---------------------------------------------------------------------------------------------
// CellList / CellListGPU (hoomd/CellList.{h,cc}, hoomd/CellListGPU.{h,cc})
// ---------------------------------------------------------------------------------------------
class CellListGPU : public Compute
    {
    public:
    explicit CellListGPU(std::shared_ptr<SystemDefinition> sysdef);
    ~CellListGPU();

    void setNominalWidth(Scalar w)
        {
        if (w != m_nominal_width)
            m_params_changed = true;
        m_nominal_width = w;
        }
    void setRadius(unsigned int r)
        {
        if (r != m_radius)
            m_params_changed = true;
        m_radius = r;
        }
    //! extra slots per cell allocated beyond the largest occupancy seen: a cell list that is rebuilt rarely
    //! on a drifting system (the particle sorter's) would otherwise reallocate at almost every use
    void setNmaxHeadroom(unsigned int h)
        {
        m_nmax_headroom = h;
        }
    //! consumers that never walk neighboring cells (the particle sorter) skip the adjacency table
    void setComputeAdj(bool b)
        {
        if (b != m_compute_adj)
            m_params_changed = true;
        m_compute_adj = b;
        }
    void setComputeTypeBody(bool b)
        {
        m_compute_type_body = b;
        m_params_changed = true;
        }
    void setComputeIdx(bool b)
        {
        m_compute_idx = b;
        m_params_changed = true;
        }
    void setFlagIndex()
        {
        m_flag_charge = m_flag_type = false;
        m_params_changed = true;
        }
    void setFlagCharge()
        {
        m_flag_charge = true;
        m_flag_type = false;
        m_params_changed = true;
        }
    void setFlagType()
        {
        m_flag_type = true;
        m_flag_charge = false;
        m_params_changed = true;
        }
    void setSortCellList(bool s)
        {
        m_sort = s;
        m_params_changed = true;
        }
    //! A consumer that checks conditions of its own right after using the list (the neighbor-list build) can take the
    //! cell list's overflow check along: compute() then queues the kernel and returns, and finishConditions() tells
    //! later whether the list had overflowed -- in which case it has been computed again with more room and whatever
    //! was built from it has to be built again.  Saves one wait of the host per rebuild.
    void setDeferConditions(bool d)
        {
        m_defer_conditions = d;
        }
    bool finishConditions();
    void setMultiple(unsigned int m)
        {
        m_multiple = m ? m : 1;
        m_params_changed = true;
        }
    std::array<unsigned int, 3> getDim() const
        {
        return m_dim;
        }
    Scalar3 getGhostWidth() const
        {
        return m_ghost_width;
        }
    unsigned int getNmax() const
        {
        return m_Nmax;
        }
    hmd_index3d_t getCellIndexer() const
        {
        return hmd_index3d_t {m_dim[0], m_dim[1], m_dim[2]};
        }
    hmd_index2d_t getCellListIndexer() const
        {
        return hmd_index2d_t {m_Nmax, m_dim[0] * m_dim[1] * m_dim[2]};
        }
    hmd_index2d_t getCellAdjIndexer() const
        {
        return hmd_index2d_t {m_n_adj, m_dim[0] * m_dim[1] * m_dim[2]};
        }
    const DeviceArray<unsigned int>& getCellSizeArray() const
        {
        return m_cell_size;
        }
    const DeviceArray<Scalar4>& getXYZFArray() const
        {
        return m_xyzf;
        }
    const DeviceArray<hmd_uint2>& getTypeBodyArray() const
        {
        return m_type_body;
        }
    const DeviceArray<unsigned int>& getIndexArray() const
        {
        return m_idx;
        }
    const DeviceArray<unsigned int>& getCellAdjArray() const
        {
        return m_cell_adj;
        }
    void compute(uint64_t timestep) override;
    std::array<unsigned int, 3> computeDimensions();

    private:
    void initializeAll();
    void initializeWidth();
    void initializeMemory();
    void initializeCellAdj();
    void computeCellList();
    bool checkConditions();
    bool evaluateConditions(const unsigned int (&h)[3]);
    bool m_defer_conditions = false, m_conditions_pending = false;

    Scalar m_nominal_width = 0;
    unsigned int m_radius = 1, m_Nmax = 0, m_multiple = 1, m_n_adj = 0, m_nmax_headroom = 0;
    WordReadback m_readback;
    std::array<unsigned int, 3> m_dim {{0, 0, 0}};
    Scalar3 m_ghost_width {0, 0, 0};
    bool m_params_changed = true, m_particles_sorted = false, m_box_changed = false, m_compute_adj = true;
    bool m_compute_type_body = false, m_compute_idx = false, m_flag_charge = false, m_flag_type = false, m_sort = false;
    DeviceArray<unsigned int> m_cell_size, m_cell_adj, m_idx;
    DeviceArray<Scalar4> m_xyzf;
    DeviceArray<hmd_uint2> m_type_body;
    DeviceArray<hmd_uint3> m_conditions;
    PinnedWords m_h_cond;
    int m_sort_slot, m_box_slot;
    };

// ---------------------------------------------------------------------------------------------
// NeighborList (hoomd/md/NeighborList.{h,cc}) + NeighborListGPU + NeighborListGPUBinned
// ---------------------------------------------------------------------------------------------
class NeighborListGPUBinned : public Compute
    {
    public:
    enum storageMode
        {
        half,
        full
        };
    NeighborListGPUBinned(std::shared_ptr<SystemDefinition> sysdef, Scalar r_buff);
    ~NeighborListGPUBinned();

    // ---- r_cut matrices (hoomd/md/NeighborList.h:230-262) ----
    void addRCutMatrix(std::shared_ptr<std::vector<Scalar>> m);
    void removeRCutMatrix(std::shared_ptr<std::vector<Scalar>> m);
    void notifyRCutMatrixChange()
        {
        m_rcut_changed = true;
        forceUpdate();
        }
    void setRBuff(Scalar r_buff);
    Scalar getRBuff() const
        {
        return m_r_buff;
        }
    void setRebuildCheckDelay(uint64_t d)
        {
        m_rebuild_check_delay = d;
        forceUpdate();
        }
    uint64_t getRebuildCheckDelay() const
        {
        return m_rebuild_check_delay;
        }
    void setDistCheck(bool b)
        {
        m_dist_check = b;
        forceUpdate();
        }
    bool getDistCheck() const
        {
        return m_dist_check;
        }
    void setStorageMode(storageMode m);
    storageMode getStorageMode() const
        {
        return m_storage_mode;
        }
    void setFilterBody(bool f)
        {
        m_filter_body = f;
        forceUpdate();
        }
    void setDeterministic(bool d);
    bool getDeterministic() const
        {
        return m_deterministic;
        }
    Scalar getMaxRCut()
        {
        if (m_rcut_changed)
            updateRList();
        return m_rcut_max_max;
        }
    Scalar getMinRCut()
        {
        if (m_rcut_changed)
            updateRList();
        return m_rcut_min;
        }
    Scalar getMaxRList()
        {
        return getMaxRCut() + m_r_buff;
        }
    Scalar getMinRList()
        {
        return getMinRCut() + m_r_buff;
        }
    //! ghost layer width the communicator must provide (hoomd/md/NeighborList.h:379-397)
    Scalar getGhostLayerWidth()
        {
        Scalar r = getMaxRCut();
        return r > 0 ? r + m_r_buff : 0;
        }

    void forceUpdate()
        {
        m_force_update = true;
        m_head_dirty = true;
        }
    uint64_t getNumUpdates() const
        {
        return m_updates + m_forced_updates;
        }
    unsigned int getSmallestRebuild() const;
    uint64_t getNumDangerousUpdates() const
        {
        return m_dangerous_updates;
        }
    void resetStats();

    // ---- exclusions (hoomd/md/NeighborList.cc:569-1211), tag based ----
    void clearExclusions();
    void addExclusion(unsigned int tag1, unsigned int tag2);
    //! hoomd/md/NeighborList.cc:700-720
    void addExclusionsFromBonds();
    unsigned int getNumExclusions() const
        {
        return m_n_exclusions;
        }

    void compute(uint64_t timestep) override;
    bool needsUpdating(uint64_t timestep);
    bool peekUpdate(uint64_t timestep)
        {
        return needsUpdating(timestep);
        }

    const DeviceArray<unsigned int>& getNNeighArray() const
        {
        return m_n_neigh;
        }
    //! rows of particle indices (hoomd/md/NeighborList.h:662-739); expanded from the tile rows on first use
    const DeviceArray<unsigned int>& getNListArray()
        {
        ensureRows();
        return m_nlist;
        }
    //! the last build wrote tile rows (16-bit candidate numbers of 4x4x4-cell regions), consumed by
    //! the tile pair-force kernel; off when exclusions, a body filter or a small grid rule it out
    bool tilesActive() const
        {
        return m_tiles_active;
        }
    //! largest number of particles any tile region held at the last build
    unsigned int getTileCapacity() const
        {
        return m_tile_capacity;
        }
    void setTileMinOccupancy(double v)
        {
        m_tile_min_occupancy = v;
        forceUpdate();
        }
    double getTileMinOccupancy() const
        {
        return m_tile_min_occupancy;
        }
    void setTilesEnabled(bool b)
        {
        m_tiles_enabled = b;
        forceUpdate();
        }
    bool getTilesEnabled() const
        {
        return m_tiles_enabled;
        }
    const DeviceArray<unsigned short>& getTileNlist() const
        {
        return m_tile_nlist;
        }
    hmd_tile_geom_t getTileGeom() const;
    //! copy the current positions (owned and ghost) into cell-list slot order for the tile force kernel
    void refreshTilePositions(hipStream_t stream = nullptr);
    void offerFusedPass(hmd_fused_step_args_t& a, uint64_t timestep);
    //! the same offer to a pass that rides on the force kernel of the tile path: the new positions go to the second
    //! slot-ordered buffer (the kernel is still staging from the first); false when the tile path is not in use
    bool offerStepRider(hmd_fused_step_args_t& a, uint64_t timestep);
    //! the kernel that carried the pass has run (it was not switched off): the second buffer is the current one now
    void commitStepRider(uint64_t timestep);
    //! the offer was not used after all: the check of `timestep` has not been launched
    void withdrawStepRider(uint64_t timestep)
        {
        if (m_prelaunched_tstep == timestep)
            m_prelaunched_tstep = ~0ull;
        }
    //! Called from compute() right after a distance check has been queued and before the host waits for
    //! its result, with the device flag and the value it takes when the list is stale: a consumer may
    //! queue work that is only valid for a fresh list, predicated on the flag.
    //! (time step, device word, value that means "stale", where to publish the word for the host, stamp):
    //! queues the consumer's kernel predicated on the word; true when a kernel that publishes was queued
    //! (time step, predicate word, value that switches the launch off, where to publish the word, stamp, stream to launch on)
    typedef std::function<bool(uint64_t, const unsigned int*, unsigned int, unsigned long long*, unsigned int, hipStream_t)> SpeculationHook;
    void setSpeculationHook(SpeculationHook h)
        {
        m_speculate = h;
        }
    //! Under domain decomposition, called after the distance check has been queued and before the halo
    //! exchange is: (time step, device word of the check, value that means "stale").  The consumer may queue
    //! work that needs no ghost positions on the default stream; the exchange is then queued on a stream of
    //! its own and runs beside that work.
    typedef std::function<bool(uint64_t, const unsigned int*, unsigned int)> InteriorHook;
    void setInteriorHook(InteriorHook h)
        {
        m_interior_hook = h;
        }
    //! Window of tile blocks (2x2x2 cells) whose 4x4x4-cell regions hold no ghost particles; false when
    //! there is none, the tile path is off, or the slot-ordered positions of the owned particles are not
    //! those of `timestep`.
    bool getInteriorBlockWindow(uint64_t timestep, unsigned int (&lo)[3], unsigned int (&hi)[3]) const;
    void setOverlapEnabled(bool b)
        {
        m_overlap = b;
        }
    bool getOverlapEnabled() const
        {
        return m_overlap;
        }
    void setBoundaryOnCommStream(bool b)
        {
        m_boundary_on_comm_stream = b;
        }
    bool getBoundaryOnCommStream() const
        {
        return m_boundary_on_comm_stream;
        }
    //! number of time steps whose interior blocks ran during the halo exchange
    uint64_t getNumOverlapped() const
        {
        return m_n_overlapped;
        }
    //! Priority of the stream the halo exchange runs on beside the interior force launch.  Normal by default.  High
    //! (HOOMD_AMD_COMM_PRIORITY=high, or this setter) keeps the exchange's small kernels on a hardware queue of their
    //! own, ahead of the force kernel's workgroups; it has only ever run with one rank per GPU and no peers, and it
    //! must never be used when several processes share a GPU: their queues then outnumber the hardware's, the
    //! scheduler maps queues by priority, and a high-priority queue that waits for an event of its own process's
    //! normal-priority queue keeps that queue (and everyone else's) from ever running.  Takes effect when the stream
    //! is created (first overlapped step).
    void setCommStreamHighPriority(bool high)
        {
        m_comm_high_priority = high;
        }
    bool getCommStreamHighPriority() const
        {
        return m_comm_high_priority && !m_ranks_share_gpu;
        }
    //! told by the decomposition when two ranks of the run use one GPU: the exchange stream then stays at normal priority
    void setRanksShareGPU(bool b)
        {
        m_ranks_share_gpu = b;
        if (b)
            m_comm_high_priority = false;
        }
    bool getRanksShareGPU() const
        {
        return m_ranks_share_gpu;
        }
    //! which of the streams tried for the exchange was kept: 0 the first, k the (k+1)-th after k that shared the default
    //! stream's hardware queue; -1 no probe was run (high priority, ranks sharing a GPU, switched off) or none passed
    int getCommStreamProbe() const
        {
        return m_comm_stream_probe;
        }

    //! time step of the last (re)build
    uint64_t getLastBuildTimestep() const
        {
        return m_last_build_tstep;
        }
    const DeviceArray<size_t>& getHeadList() const
        {
        return m_head_list;
        }
    std::array<unsigned int, 3> getDim() const
        {
        return m_cl->getDim();
        }
    unsigned int getNmax() const
        {
        return m_cl->getNmax();
        }
    std::shared_ptr<CellListGPU> getCellList() const
        {
        return m_cl;
        }
    //! (i, j) index pairs with i < j of the current list, local indices
    std::vector<std::pair<unsigned int, unsigned int>> getLocalPairList();
    //! same by tag
    std::vector<std::pair<unsigned int, unsigned int>> getPairList();
    std::vector<unsigned int> getNmaxPerType() const
        {
        return m_Nmax;
        }

    protected:
    void updateRList();
    void checkBoxSize();
    bool distanceCheck(uint64_t timestep);
    void buildNlist(uint64_t timestep);
    void buildHeadList();
    bool checkConditions();
    void resetConditions();
    void resizeNlist(size_t size);
    void updateExListIdx();
    void reallocatePerParticle();
    void ensureRows();
    bool tilesUsable(unsigned int& capacity);
    void fillCheckArgs(hmd_nlist_check_args_t& a, bool fresh = true);
    bool offerCheck(hmd_nlist_check_args_t& a, uint64_t timestep);

    std::shared_ptr<CellListGPU> m_cl;
    Scalar m_r_buff;
    bool m_filter_body = false;
    storageMode m_storage_mode = half;
    bool m_deterministic = false;
    std::vector<std::shared_ptr<std::vector<Scalar>>> m_consumer_r_cut;
    std::vector<Scalar> m_r_cut, m_rcut_max;
    Scalar m_rcut_max_max = 0, m_rcut_min = 0;
    bool m_rcut_changed = true, m_update_cell_size = true;

    uint64_t m_updates = 0, m_forced_updates = 0, m_dangerous_updates = 0;
    bool m_force_update = true, m_dist_check = true, m_has_been_updated_once = false;
    //! head list (and exclusion indices) are stale; survives a peekUpdate() that consumed m_force_update
    bool m_head_dirty = true;
    uint64_t m_last_updated_tstep = 0, m_last_checked_tstep = 0, m_rebuild_check_delay = 0;
    bool m_last_check_result = false;
    std::vector<uint64_t> m_update_periods;
    std::array<Scalar, 3> m_last_L {{0, 0, 0}};
    unsigned int m_checkn = 1;
    //! time step whose distance check was already launched by the integrator's fused pass
    uint64_t m_prelaunched_tstep = ~0ull;
    //! time step for which the owned part of the slot-ordered position copy is already current
    uint64_t m_cpos_tstep = ~0ull, m_cpos_all_tstep = ~0ull;
    uint64_t m_compute_tstep = 0;
    SpeculationHook m_speculate;
    InteriorHook m_interior_hook;
    hipStream_t m_comm_stream = nullptr;
    std::vector<hipStream_t> m_rejected_streams;
    int m_comm_stream_probe = -1;
    hipEvent_t m_ev_ready = nullptr, m_ev_comm = nullptr;
    bool m_overlap = true;
    //! the blocks that see ghosts are launched on the exchange's stream (beside the interior launch) rather than behind it
    bool m_boundary_on_comm_stream = true;
    bool m_comm_high_priority = false, m_ranks_share_gpu = false;
    uint64_t m_n_overlapped = 0;
    WordReadback m_readback;
    bool m_in_compute = false;
    uint64_t m_last_build_tstep = ~0ull;

    std::vector<unsigned int> m_Nmax;
    unsigned int m_head_uniform_n = 0, m_head_uniform_nmax = 0; // places of the head list that hold i * Nmax, and that Nmax
    DeviceArray<unsigned int> m_d_Nmax, m_conditions, m_n_neigh, m_nlist, m_flag;
    DeviceArray<size_t> m_head_list, m_req_size;
    DeviceArray<Scalar4> m_last_pos;
    DeviceArray<Scalar> m_d_r_cut, m_d_rcut_max;
    DeviceArray<unsigned char> m_workspace;
    PinnedWords m_h_words;

    // tile rows
    bool m_tiles_enabled = true, m_tiles_active = false, m_rows_valid = true, m_tiles_overflowed = false;
    double m_tile_min_occupancy = 8.0; // particles per cell from which the tile path is used
    unsigned int m_tile_capacity = 0;
    DeviceArray<unsigned short> m_tile_nlist;
    DeviceArray<unsigned int> m_block_group; // first group of rows of every block of 2x2x2 cells
    unsigned int m_row_pieces = 1;           // 16-entry pieces every tile row has room for
    DeviceArray<unsigned int> m_slot_of;   // particle index -> cell-list slot
    DeviceArray<Scalar4> m_cpos;           // positions in slot order, refreshed before each force launch
    DeviceArray<Scalar4> m_cpos_alt;       // where a pass riding on the force kernel puts the next step's positions
    DeviceArray<hmd_uint4> m_slot_meta;    // per slot: particle index, entries of the row
    bool m_rows_by_type = false;           // the rows of the last build are laid out by type (two types)

    // exclusions
    bool m_exclusions_set = false, m_ex_dirty = false;
    unsigned int m_n_exclusions = 0;
    std::vector<std::vector<unsigned int>> m_ex_by_tag;
    DeviceArray<unsigned int> m_n_ex_tag, m_ex_list_tag, m_n_ex_idx, m_ex_list_idx;
    unsigned int m_ex_height = 0;

    int m_sort_slot, m_box_slot, m_maxn_slot;
    };

// ---------------------------------------------------------------------------------------------
// ForceCompute (hoomd/ForceCompute.{h,cc})
// ---------------------------------------------------------------------------------------------
class ForceCompute : public Compute
    {
    public:
    explicit ForceCompute(std::shared_ptr<SystemDefinition> sysdef);
    ~ForceCompute();
    void compute(uint64_t timestep) override;
    Scalar4* getForceArray() const
        {
        return m_force.get();
        }
    Scalar* getVirialArray() const
        {
        return m_virial.get();
        }
    size_t getVirialPitch() const
        {
        return m_virial_pitch;
        }
    Scalar getExternalEnergy() const
        {
        return m_external_energy;
        }
    Scalar getExternalVirial(unsigned int i) const
        {
        return m_external_virial[i];
        }
    //! tag-ordered host copies (hoomd/ForceCompute.cc:628-643 getForces/getEnergies/getVirials)
    std::vector<std::array<Scalar, 4>> getForcesAndEnergies();
    std::vector<std::array<Scalar, 6>> getVirials();
    Scalar calcEnergySum();
    //! Raise what the kernels could only flag (called with the stream idle, e.g. at the end of a run)
    virtual void checkErrors() { }
    //! whether compute() may have to wait for the host to learn something from the device first (a neighbor list's
    //! rebuild decision): forces that never do are launched before those that may, so the device has work meanwhile
    virtual bool mayWaitForDevice() const
        {
        return true;
        }

    protected:
    virtual void computeForces(uint64_t timestep) = 0;
    void reallocate();
    DeviceArray<Scalar4> m_force;
    DeviceArray<Scalar> m_virial;
    size_t m_virial_pitch = 0;
    bool m_particles_sorted = false;
    unsigned int m_computed_flags = 0;
    Scalar m_external_energy = 0;
    Scalar m_external_virial[6] = {0, 0, 0, 0, 0, 0};
    int m_sort_slot, m_maxn_slot;
    };

// ---------------------------------------------------------------------------------------------
// PotentialPair<evaluator> / PotentialPairGPU<evaluator> are header-only templates for evaluators built in and out of
// the tree: include/hoomd/md/PotentialPair.h, include/hoomd/md/PotentialPairGPU.h.
// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// PotentialBondGPU<EvaluatorBondFENE> (hoomd/md/PotentialBond.h, hoomd/md/PotentialBondGPU.h)
// ---------------------------------------------------------------------------------------------
class PotentialBondFENEGPU : public ForceCompute
    {
    public:
    bool mayWaitForDevice() const override
        {
        return false;
        }
    explicit PotentialBondFENEGPU(std::shared_ptr<SystemDefinition> sysdef);
    //! k, r0, epsilon, sigma, delta of a bond type (hoomd/md/EvaluatorBondFENE.h:47-67)
    void setParams(unsigned int type, Scalar k, Scalar r0, Scalar epsilon, Scalar sigma, Scalar delta);
    std::array<Scalar, 5> getParams(unsigned int type) const;
    void checkErrors() override;

    protected:
    void computeForces(uint64_t timestep) override;

    private:
    std::shared_ptr<BondData> m_bonds;
    std::vector<hmd_fene_params_t> m_params;
    DeviceArray<hmd_fene_params_t> m_d_params;
    bool m_params_dirty = true;
    PinnedWords m_flags;
    };

// ---------------------------------------------------------------------------------------------
// ComputeThermoGPU (hoomd/md/ComputeThermo.{h,cc}, hoomd/md/ComputeThermoGPU.cu)
// ---------------------------------------------------------------------------------------------
class ComputeThermoGPU : public Compute
    {
    public:
    ComputeThermoGPU(std::shared_ptr<SystemDefinition> sysdef, std::shared_ptr<ParticleGroup> group);
    void compute(uint64_t timestep) override;
    //! Queues the sums of `timestep` (unless they are this time step's already) and returns without waiting for them:
    //! they stay on the device (deviceValues()) and reach the host when compute() of the same time step asks.
    void computeAsync(uint64_t timestep);
    //! The sums of `timestep` are already queued by somebody else with the arguments of fillArgs() (the second half
    //! step that computes them in its own pass): nothing is launched for that time step any more.
    void adoptQueuedValues(uint64_t timestep);
    void fillArgs(hmd_thermo_args_t& a);
    const double* deviceValues() const
        {
        return m_out.get();
        }
    //! kinetic energy of the latest sums that have finished on the device (never waits; 1 before the first)
    double peekLatestKineticEnergy()
        {
        return *reinterpret_cast<volatile double*>(m_host_copy.get());
        }
    bool reducesOverRanks() const
        {
        return (bool)m_allreduce_sum;
        }
    std::shared_ptr<ParticleGroup> getGroup() const
        {
        return m_group;
        }
    Scalar getTranslationalKineticEnergy() const
        {
        return m_vals[0];
        }
    Scalar getKineticEnergy() const
        {
        return m_vals[0];
        }
    Scalar getPotentialEnergy() const
        {
        return m_vals[1];
        }
    Scalar getVirialTrace() const
        {
        return m_vals[2];
        }
    std::array<Scalar, 3> getMomentum() const
        {
        return {m_vals[3], m_vals[4], m_vals[5]};
        }
    Scalar getTranslationalDOF() const
        {
        return m_group->getTranslationalDOF();
        }
    Scalar getTranslationalTemperature() const
        {
        return 2.0 * m_vals[0] / m_group->getTranslationalDOF();
        }
    Scalar getPressure() const;
    //! hook for summing over ranks (set by the Python communicator)
    std::function<void(double*, int)> m_allreduce_sum;

    private:
    std::shared_ptr<ParticleGroup> m_group;
    DeviceArray<Scalar> m_out;
    DeviceArray<unsigned char> m_scratch;
    std::array<Scalar, 8> m_vals {{0, 0, 0, 0, 0, 0, 0, 0}};
    PinnedWords m_host_copy {16}; // eight doubles the final reduction also stores to (peekLatestKineticEnergy)
    bool m_host_stale = false;    // the device holds sums the host has not fetched
    void launch();
    void fetch();
    };

// ---------------------------------------------------------------------------------------------
// Thermostats (hoomd/md/Thermostat.h)
// ---------------------------------------------------------------------------------------------
class Thermostat
    {
    public:
    Thermostat(Scalar T, std::shared_ptr<ParticleGroup> group, std::shared_ptr<ComputeThermoGPU> thermo, std::shared_ptr<SystemDefinition> sysdef)
        : m_T(T), m_group(group), m_thermo(thermo), m_sysdef(sysdef)
        {
        }
    virtual ~Thermostat() { }
    virtual std::array<Scalar, 2> getRescalingFactorsOne(uint64_t timestep, Scalar deltaT)
        {
        return {1.0, 1.0};
        }
    virtual std::array<Scalar, 2> getRescalingFactorsTwo(uint64_t timestep, Scalar deltaT)
        {
        return {1.0, 1.0};
        }
    virtual void advanceThermostat(uint64_t timestep, Scalar deltaT) { }
    //! A factor for the first half step of the form sqrt(gamma / kinetic energy) that the device evaluates, with the
    //! kinetic energy left at d_ke by sums that are queued but not waited for; false = ask getRescalingFactorsOne.
    virtual bool getDeferredFactorOne(uint64_t timestep, Scalar deltaT, const double*& d_ke, double& gamma)
        {
        return false;
        }
    std::shared_ptr<ComputeThermoGPU> getThermo() const
        {
        return m_thermo;
        }
    void setT(Scalar T)
        {
        m_T = T;
        }
    Scalar getT() const
        {
        return m_T;
        }

    protected:
    Scalar m_T;
    std::shared_ptr<ParticleGroup> m_group;
    std::shared_ptr<ComputeThermoGPU> m_thermo;
    std::shared_ptr<SystemDefinition> m_sysdef;
    };

//! hoomd/md/Thermostat.h:374-420
class BerendsenThermostat : public Thermostat
    {
    public:
    BerendsenThermostat(Scalar T, std::shared_ptr<ParticleGroup> group, std::shared_ptr<ComputeThermoGPU> thermo, std::shared_ptr<SystemDefinition> sysdef, Scalar tau)
        : Thermostat(T, group, thermo, sysdef), m_tau(tau)
        {
        }
    std::array<Scalar, 2> getRescalingFactorsOne(uint64_t timestep, Scalar deltaT) override;

    private:
    Scalar m_tau;
    };

//! hoomd/md/Thermostat.h:337-372.  The reference draws the Gamma variate from Philox (random123,
//! not vendored); this class draws it from std::mt19937_64 seeded with (seed, timestep), so the
//! ensemble is the same but the random stream is not (statistical parity only).
class BussiThermostat : public Thermostat
    {
    public:
    using Thermostat::Thermostat;
    std::array<Scalar, 2> getRescalingFactorsOne(uint64_t timestep, Scalar deltaT) override;
    bool getDeferredFactorOne(uint64_t timestep, Scalar deltaT, const double*& d_ke, double& gamma) override;

    private:
    double drawTargetEnergy(uint64_t timestep);
    };

// ---------------------------------------------------------------------------------------------
// Integration methods (hoomd/md/IntegrationMethodTwoStep.h, hoomd/md/TwoStepConstantVolume{,GPU}.*)
// ---------------------------------------------------------------------------------------------
class IntegrationMethodTwoStep
    {
    public:
    IntegrationMethodTwoStep(std::shared_ptr<SystemDefinition> sysdef, std::shared_ptr<ParticleGroup> group)
        : m_sysdef(sysdef), m_pdata(sysdef->getParticleData()), m_group(group)
        {
        }
    virtual ~IntegrationMethodTwoStep() { }
    virtual void integrateStepOne(uint64_t timestep) = 0;
    virtual void integrateStepTwo(uint64_t timestep) = 0;
    virtual bool isMomentumConserving() const
        {
        return true;
        }
    void setDeltaT(Scalar dt)
        {
        m_deltaT = dt;
        }
    //! whether the second half step may also take sums over the velocities it leaves (nothing touches the particles
    //! before the next first half step)
    void setSumsRide(bool v)
        {
        m_sums_ride = v;
        }
    std::shared_ptr<ParticleGroup> getGroup() const
        {
        return m_group;
        }
    //! hoomd/md/IntegrationMethodTwoStep.cc:73-108 (3 per member in 3D, 2 in 2D)
    Scalar getTranslationalDOF(std::shared_ptr<ParticleGroup> query);

    protected:
    std::shared_ptr<SystemDefinition> m_sysdef;
    std::shared_ptr<ParticleData> m_pdata;
    std::shared_ptr<ParticleGroup> m_group;
    Scalar m_deltaT = 0;
    bool m_sums_ride = false;
    };

class TwoStepConstantVolumeGPU : public IntegrationMethodTwoStep
    {
    public:
    TwoStepConstantVolumeGPU(std::shared_ptr<SystemDefinition> sysdef, std::shared_ptr<ParticleGroup> group, std::shared_ptr<Thermostat> thermostat)
        : IntegrationMethodTwoStep(sysdef, group), m_thermostat(thermostat)
        {
        }
    void integrateStepOne(uint64_t timestep) override;
    void integrateStepTwo(uint64_t timestep) override;
    //! step two of `timestep` and step one of `timestep + 1` in one pass; only without thermostat / limit
    bool canFuse() const
        {
        return !m_thermostat && m_limit < 0;
        }
    void integrateStepTwoOne(uint64_t timestep);
    void setThermostat(std::shared_ptr<Thermostat> t)
        {
        m_thermostat = t;
        }
    //! second half steps that took the thermostat's sums along / first half steps whose factor the device worked out
    uint64_t getNumStepsWithSums() const
        {
        return m_n_with_sums;
        }
    uint64_t getNumDeferredFactors() const
        {
        return m_n_deferred;
        }
    //! maximum displacement per step; negative disables (hoomd/md/TwoStepConstantVolume.h:57-67)
    void setLimit(Scalar l)
        {
        m_limit = l;
        }
    Scalar getLimit() const
        {
        return m_limit;
        }

    private:
    std::shared_ptr<Thermostat> m_thermostat;
    Scalar m_limit = -1;
    uint64_t m_n_with_sums = 0, m_n_deferred = 0;
    };

// ---------------------------------------------------------------------------------------------
// Integrator / IntegratorTwoStep (hoomd/Integrator.{h,cc}, hoomd/md/IntegratorTwoStep.{h,cc})
// ---------------------------------------------------------------------------------------------
class IntegratorTwoStep
    {
    public:
    IntegratorTwoStep(std::shared_ptr<SystemDefinition> sysdef, Scalar deltaT);
    ~IntegratorTwoStep();
    void update(uint64_t timestep);
    void prepRun(uint64_t timestep);
    //! run the second half step that update() left pending (it is fused with the next first half)
    void flush();
    //! after the stream has drained: let every force raise what its kernels flagged
    void checkForceErrors()
        {
        for (auto& f : m_forces)
            f->checkErrors();
        }
    void setDeltaT(Scalar dt);
    Scalar getDeltaT() const
        {
        return m_deltaT;
        }
    std::vector<std::shared_ptr<ForceCompute>>& getForces()
        {
        return m_forces;
        }
    std::vector<std::shared_ptr<IntegrationMethodTwoStep>>& getIntegrationMethods()
        {
        return m_methods;
        }
    void notifyForcesChanged()
        {
        m_prepared = false;
        }
    bool isPrepared() const
        {
        return m_prepared;
        }
    //! hoomd/md/IntegratorTwoStep.cc:161-196
    Scalar getTranslationalDOF(std::shared_ptr<ParticleGroup> group);
    void updateGroupDOF(std::shared_ptr<ParticleGroup> group)
        {
        group->setTranslationalDOF(getTranslationalDOF(group));
        }
    std::array<double, 3> computeLinearMomentum();
    void computeNetForceGPU(uint64_t timestep);
    //! pdata flags the integrator wants on a step (pressure_tensor when a consumer asked for it)
    void setRequestPressure(bool b)
        {
        m_request_pressure = b;
        }
    //! Will update(timestep + 1) follow this update() with nothing in between (no sorter, no end of the run)?  Only then
    //! may the force kernel of this step run the half steps around it (the run loop says so before every update).
    void setLookahead(bool b)
        {
        m_lookahead = b;
        }
    //! switch for the half steps riding on the force kernel (on by default; tests compare both ways)
    void setStepRider(bool b)
        {
        m_rider_enabled = b;
        }
    bool getStepRider() const
        {
        return m_rider_enabled;
        }
    //! time steps whose half steps ran inside the force kernel
    uint64_t getNumRiderSteps() const
        {
        return m_n_rider_steps;
        }

    private:
    void computeAccelerations(uint64_t timestep);
    bool fillStepRider(hmd_fused_step_args_t& a, uint64_t timestep);
    void sumNetForce(bool virial);
    std::shared_ptr<SystemDefinition> m_sysdef;
    std::shared_ptr<ParticleData> m_pdata;
    Scalar m_deltaT;
    std::vector<std::shared_ptr<ForceCompute>> m_forces;
    std::vector<std::shared_ptr<IntegrationMethodTwoStep>> m_methods;
    bool m_prepared = false, m_request_pressure = false;
    bool m_pending_two = false;       // step two of m_pending_tstep has not run yet
    uint64_t m_pending_tstep = 0;
    bool m_fuse = true;
    // (off unless asked for: measured against the two-kernel step, the joined kernel buys under one per cent)
    bool m_lookahead = false, m_rider_enabled = false;
    bool m_rider_armed = false;          // the offer to the force compute stands for the force computation in progress
    uint64_t m_rider_tstep = 0;          // ... of this time step
    uint64_t m_step_one_done_for = ~0ull; // time step whose first half step already ran inside the previous force kernel
    uint64_t m_n_rider_steps = 0;
    };

// ---------------------------------------------------------------------------------------------
// ParticleSorter: periodic cell-order reordering (role of hoomd/SFCPackTuner{,GPU}.*)
// ---------------------------------------------------------------------------------------------
class ParticleSorter
    {
    public:
    ParticleSorter(std::shared_ptr<SystemDefinition> sysdef, uint64_t period);
    void setPeriod(uint64_t p)
        {
        m_period = p;
        }
    uint64_t getPeriod() const
        {
        return m_period;
        }
    //! cell width of the sorting grid (defaults to the neighbor-list cell width when one exists)
    void setWidth(Scalar w)
        {
        m_width = w;
        }
    void update(uint64_t timestep);
    //! would update(timestep) reorder the particles?
    bool isDue(uint64_t timestep) const
        {
        return m_period != 0 && timestep % m_period == 0 && timestep != m_last_sorted;
        }
    void sortNow();

    private:
    std::shared_ptr<SystemDefinition> m_sysdef;
    std::shared_ptr<ParticleData> m_pdata;
    std::shared_ptr<CellListGPU> m_cl;
    uint64_t m_period;
    Scalar m_width = 0;
    uint64_t m_last_sorted = (uint64_t)-1;
    DeviceArray<unsigned int> m_order, m_offsets;
    DeviceArray<unsigned char> m_workspace;
    };

// ---------------------------------------------------------------------------------------------
// System (hoomd/System.{h,cc}): the step loop
// ---------------------------------------------------------------------------------------------
class System
    {
    public:
    System(std::shared_ptr<SystemDefinition> sysdef, uint64_t initial_tstep) : m_sysdef(sysdef), m_cur_tstep(initial_tstep) { }
    void setIntegrator(std::shared_ptr<IntegratorTwoStep> i)
        {
        m_integrator = i;
        }
    std::shared_ptr<IntegratorTwoStep> getIntegrator() const
        {
        return m_integrator;
        }
    void setSorter(std::shared_ptr<ParticleSorter> s)
        {
        m_sorter = s;
        }
    void setPressureFlag(bool b)
        {
        m_pressure_flag = b;
        }
    /// `prepare` = false continues a run that was cut into pieces with nothing changed in between (the caller wrote
    /// output between them): the forced migration and the force computation of a run's start are not repeated
    void run(uint64_t nsteps, bool prepare = true);
    uint64_t getCurrentTimeStep() const
        {
        return m_cur_tstep;
        }
    double getLastTPS() const
        {
        return m_last_tps;
        }
    double getLastWalltime() const
        {
        return m_last_walltime;
        }

    private:
    std::shared_ptr<SystemDefinition> m_sysdef;
    std::shared_ptr<IntegratorTwoStep> m_integrator;
    std::shared_ptr<ParticleSorter> m_sorter;
    uint64_t m_cur_tstep;
    bool m_pressure_flag = false;
    double m_last_tps = 0, m_last_walltime = 0;
    };

// ---------------------------------------------------------------------------------------------
// optional per-kernel timing with HIP events on the launch stream (bench harness)
// ---------------------------------------------------------------------------------------------
struct KernelTimer
    {
    static KernelTimer& get();
    void enable(bool on);
    //! time only the kernels whose names start with this ("" = all)
    void setFilter(const std::string& name)
        {
        m_filter = name;
        }
    //! samples shorter than this are booked under "<name>_skipped" (launches that were predicated off); a negative
    //! value is relative: that fraction of the median sample of the same name
    void setMinMs(double ms)
        {
        m_min_ms = ms;
        }
    bool enabled() const
        {
        return m_on;
        }
    //! time one launch in `n` of every name (events on the launch stream cost a few microseconds each: a sample
    //! perturbs a run less than timing every launch)
    void setEvery(unsigned int n)
        {
        m_every = std::max(n, 1u);
        }
    bool wants(const char* name)
        {
        if (!(m_on && (m_filter.empty() || std::strncmp(name, m_filter.c_str(), m_filter.size()) == 0)))
            return false;
        if (m_every == 1)
            return true;
        for (auto& c : m_seen)
            if (c.first == name)
                return (c.second++ % m_every) == 0;
        m_seen.emplace_back(name, 1u);
        return true;
        }
    void begin(const char* name, hipStream_t stream = nullptr);
    void end(const char* name);
    //! name -> (count, total ms); synchronizes the stream
    std::vector<std::tuple<std::string, uint64_t, double>> collect();
    void reset();

    private:
    bool m_on = false;
    std::string m_filter;
    double m_min_ms = 0;
    unsigned int m_every = 1;
    std::vector<std::pair<std::string, unsigned int>> m_seen;
    struct Rec
        {
        std::string name;
        hipEvent_t e0, e1;
        hipStream_t stream;
        };
    std::vector<Rec> m_open, m_done;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> m_pool;
    std::vector<std::tuple<std::string, uint64_t, double>> m_acc;
    };

struct ScopedKernelTimer
    {
    const char* n;
    bool on;
    explicit ScopedKernelTimer(const char* name, hipStream_t stream = nullptr) : n(name), on(KernelTimer::get().wants(name))
        {
        if (on)
            KernelTimer::get().begin(n, stream);
        }
    ~ScopedKernelTimer()
        {
        if (on)
            KernelTimer::get().end(n);
        }
    };

// ---------------------------------------------------------------------------------------------
// CommunicatorGPU (hoomd/Communicator.h:166-232, hoomd/CommunicatorGPU.cc): migration, ghost exchange, per-step ghost
// update and the rebuild vote, driven from C++.  The message-passing layer is an interface: the production object
// hands the device buffers to torch.distributed (RCCL), the test object stages them through gloo.
// ---------------------------------------------------------------------------------------------
class ExchangeTransport
    {
    public:
    virtual ~ExchangeTransport() { }
    //! All-to-all of rows of row_bytes bytes: send_rows[r] rows go to rank r (consecutive in d_send by rank),
    //! recv_rows[r] rows arrive from rank r in d_recv; queued on `stream` (0: the default stream).  Every rank calls.
    virtual void allToAll(uintptr_t d_send, const std::vector<unsigned int>& send_rows, uintptr_t d_recv, const std::vector<unsigned int>& recv_rows,
                          unsigned int row_bytes, uintptr_t stream)
        = 0;
    //! in-place maximum over all ranks of one 32-bit device word, queued on `stream`
    virtual void allReduceMaxWord(uintptr_t d_word, uintptr_t stream) = 0;
    //! logical OR of a host flag over all ranks
    virtual bool anyRank(bool local) = 0;
    //! sum over all ranks of a few host doubles (energies, particle counts per type); every rank calls
    virtual std::vector<double> allReduceSumHost(const std::vector<double>& v)
        {
        throw std::runtime_error("this exchange transport has no host-side sum");
        }
    };

//! The transport on RCCL itself: ncclSend / ncclRecv groups and ncclAllReduce issued from C++ on the caller's stream
//! (rccl_transport.cc).  One object per rank; all ranks construct it together (ncclCommInitRank is collective) from
//! the 128-byte id that uniqueId() returned on rank 0.
class RcclTransport : public ExchangeTransport
    {
    public:
    RcclTransport(const std::string& unique_id, unsigned int rank, unsigned int world);
    ~RcclTransport() override;
    RcclTransport(const RcclTransport&) = delete;
    RcclTransport& operator=(const RcclTransport&) = delete;
    //! a fresh ncclUniqueId (rank 0 calls this and distributes the bytes)
    static std::string uniqueId();
    //! which librccl the calls go to ("... (already loaded in this process)" or "... (opened by hoomd_amd)")
    static std::string libraryOrigin();
    void allToAll(uintptr_t d_send, const std::vector<unsigned int>& send_rows, uintptr_t d_recv, const std::vector<unsigned int>& recv_rows,
                  unsigned int row_bytes, uintptr_t stream) override;
    void allReduceMaxWord(uintptr_t d_word, uintptr_t stream) override;
    bool anyRank(bool local) override;
    std::vector<double> allReduceSumHost(const std::vector<double>& v) override;
    uint64_t n_exchanges = 0;
    //! send this rank's own rows through ncclSend / ncclRecv to itself instead of copying them (tests: the only way to
    //! run the send / receive calls on a machine with one GPU)
    bool self_through_rccl = false;

    private:
    unsigned int m_rank, m_world;
    void* m_comm = nullptr; // ncclComm_t
    DeviceArray<double> m_words;
    };

class CommunicatorGPU : public Communicator
    {
    public:
    //! grid: domains per direction, rank = (k * grid[1] + j) * grid[0] + i; glo / gL: the global box
    CommunicatorGPU(std::shared_ptr<SystemDefinition> sysdef, std::shared_ptr<ExchangeTransport> transport, unsigned int rank, unsigned int world,
                    std::array<unsigned int, 3> grid, std::array<double, 3> glo, std::array<double, 3> gL);
    void addNeighborList(std::shared_ptr<NeighborListGPUBinned> nl)
        {
        m_nlists.push_back(nl);
        }
    void communicate(uint64_t timestep) override;
    void forceMigrate() override
        {
        m_force_migrate = true;
        }
    Scalar getGhostLayerMaxWidth() const override;
    bool anyRank(bool local) override
        {
        return m_transport->anyRank(local);
        }
    uintptr_t beginRebuildVote(uintptr_t d_flag, unsigned int checkn, uintptr_t stream) override;
    bool endRebuildVote() override;
    void allReduceSum(double* v, int n) override
        {
        const std::vector<double> r = m_transport->allReduceSumHost(std::vector<double>(v, v + n));
        if ((int)r.size() != n)
            throw std::runtime_error("allReduceSumHost returned another number of values");
        std::copy(r.begin(), r.end(), v);
        }
    unsigned int getRank() const override
        {
        return m_rank;
        }

    //! hoomd/Communicator.cc migrateParticles; true when this rank's particle set changed
    bool migrateParticles();
    //! hoomd/Communicator.cc exchangeGhosts
    void exchangeGhosts(Scalar r_ghost);
    //! hoomd/Communicator.cc beginUpdateGhosts / finishUpdateGhosts
    void updateGhosts(hipStream_t stream);
    void setUseVoteRows(bool b)
        {
        m_use_vote_rows = b;
        }
    void setFixedMigrate(bool b)
        {
        m_fixed_migrate = b && m_all_peers;
        }
    //! ghost messages sized from the previous exchange instead of an exchange of counts (when all ranks are peers)
    void setUseGhostSlots(bool b)
        {
        m_use_ghost_slots = b;
        }
    //! room of a ghost message: rows of the previous exchange + rows / div (div 0: nothing) + pad
    void setGhostSlotRoom(unsigned int div, unsigned int pad)
        {
        m_ghost_room_div = div;
        m_ghost_room_pad = pad;
        }
    //! forget the sizes of the previous ghost exchange (the next one exchanges counts)
    void forgetGhostRows()
        {
        m_have_prev_rows = false;
        }
    void setMigrateSlotRows(unsigned int n)
        {
        m_mig_slot_rows = std::max(n, 1u);
        }
    //! Pack kernels read their sizes where the counting kernels left them (device memory) and the host looks at counts
    //! and headers together, once per half of the exchange (fixed-size messages only); off: every count is fetched
    //! before the kernel that needs it is launched
    void setUseDeviceCounts(bool b)
        {
        m_device_counts = b;
        }
    uint64_t n_device_count_exchanges = 0;
    // statistics
    uint64_t n_migrated_out = 0, n_migrated_in = 0, n_vote_exchanges = 0, n_count_exchanges = 0, n_ghost_slot_overflows = 0;
    bool usesFixedMigrate() const
        {
        return m_fixed_migrate;
        }
    unsigned int getMigrateSlotRows() const
        {
        return m_mig_slot_rows;
        }

    private:
    hmd_pdata_arrays_t pdataArrays() const;
    bool migrateParticlesDeviceCounts();
    bool exchangeGhostsDeviceCounts(Scalar r_ghost, hmd_ghost_send_list_args_t& a, hmd_pack_pos_args_t& pa, hmd_unpack_ghost_init_args_t& u, unsigned int& total);
    void sortKeys(DeviceArray<unsigned long long>& keys, unsigned int n, unsigned int key_bits);
    void updateGhostsWithVote(unsigned int* d_flag, hipStream_t stream);
    void buildVotePlan();
    unsigned int rankAt(int i, int j, int k) const;

    std::shared_ptr<SystemDefinition> m_sysdef;
    std::shared_ptr<ParticleData> m_pdata;
    std::shared_ptr<ExchangeTransport> m_transport;
    std::vector<std::weak_ptr<NeighborListGPUBinned>> m_nlists;
    unsigned int m_rank, m_world;
    std::array<unsigned int, 3> m_grid, m_coords;
    std::array<double, 3> m_glo, m_gL;
    std::array<bool, 3> m_decomposed;
    bool m_force_migrate = true, m_in_communicate = false, m_ghosts_fresh = false, m_use_vote_rows = true;
    bool m_all_peers = false, m_fixed_migrate = false, m_have_vote_plan = false, m_device_counts = true;
    DeviceArray<unsigned int> m_dir_dest_dev;
    unsigned int m_ghost_key_cap = 0;
    // one entry per direction towards a neighbor domain, sorted by destination rank
    std::vector<unsigned int> m_dir_dest;
    DeviceArray<unsigned int> m_dir_mask;
    DeviceArray<Scalar3> m_shift_table;
    // migration
    DeviceArray<int> m_mig_dest;
    DeviceArray<unsigned int> m_mig_counters, m_scratch;
    DeviceArray<unsigned long long> m_mig_keys, m_keys_sorted;
    DeviceArray<long long> m_mig_out, m_mig_in;
    DeviceArray<unsigned char> m_sort_ws;
    unsigned int m_mig_slot_rows = 1024;
    // ghost layer
    DeviceArray<double> m_r_ghost;
    double m_r_ghost_value = -1;
    DeviceArray<unsigned long long> m_keys;
    DeviceArray<unsigned int> m_send_counters, m_send_idx, m_count_out, m_count_in;
    DeviceArray<unsigned char> m_send_dir;
    DeviceArray<Scalar4> m_init_out, m_init_in, m_pack_buf, m_vote_out, m_vote_in;
    unsigned int m_n_send = 0;
    std::vector<unsigned int> m_send_rows, m_recv_rows, m_aug_send, m_aug_recv, m_prev_send_rows, m_prev_recv_rows;
    bool m_use_ghost_slots = true, m_have_prev_rows = false;
    unsigned int m_ghost_room_div = 8, m_ghost_room_pad = 64;
    DeviceArray<unsigned int> m_out_row, m_flag_rows;
    DeviceArray<int> m_in_map;
    unsigned int m_n_out = 0, m_n_in = 0;
    WordReadback m_readback;
    // the vote in flight
    unsigned int* m_vote_word = nullptr;
    unsigned int m_vote_checkn = 0;
    hipStream_t m_vote_stream = nullptr;
    };

// A plugin built outside the tree holds the layout of the classes above in its own objects (their inline accessors,
// the offsets of the members its templates touch).  Sizes of the classes a pair potential reaches into, folded into
// one number: the plugin computes it from the headers it was compiled with, the core module from its own, and the
// export functions refuse to register a plugin whose number differs (rebuild the plugin against this installation).
inline unsigned long long hostLayoutFingerprintOfTheseHeaders()
    {
    unsigned long long h = 1469598103934665603ull;
    const size_t sizes[] = {sizeof(ParticleData), sizeof(SystemDefinition), sizeof(Compute), sizeof(CellListGPU), sizeof(NeighborListGPUBinned),
                            sizeof(ForceCompute), sizeof(BoxDim), sizeof(KernelTimer), sizeof(hmd_pair_args_t), sizeof(hmd_tile_geom_t)};
    for (size_t v : sizes)
        h = (h ^ (unsigned long long)v) * 1099511628211ull;
    return h;
    }
//! the same number as the core module was compiled with (defined in particle_data.cc)
unsigned long long hostLayoutFingerprint();

    } // namespace hoomd_amd

#endif
