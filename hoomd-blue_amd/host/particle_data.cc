// particle_data.cc -- ParticleData, ParticleGroup, Compute and small utilities of the host layer.
#include "hoomd_host.h"

#include <chrono>
#include <cstdlib>

#include <algorithm>
#include <cmath>

namespace hoomd_amd
    {
void hipCheck(hipError_t e, const char* what)
    {
    if (e != hipSuccess)
        {
        std::ostringstream s;
        s << what << " failed: " << hipGetErrorString(e);
        throw std::runtime_error(s.str());
        }
    }

unsigned long long hostLayoutFingerprint()
    {
    return hostLayoutFingerprintOfTheseHeaders();
    }

void abiCheck(int code, const char* what)
    {
    if (code == 0)
        return;
    std::ostringstream s;
    if (code < 0)
        s << what << ": " << hmd_last_error() << " (code " << code << ")";
    else
        s << what << ": HIP error " << hipGetErrorString((hipError_t)code);
    throw std::runtime_error(s.str());
    }

PinnedWords::PinnedWords(size_t n)
    {
    hipCheck(hipHostMalloc((void**)&m_p, n * sizeof(unsigned int), hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc");
    std::memset(m_p, 0, n * sizeof(unsigned int));
    }

unsigned int* PinnedWords::deviceView()
    {
    if (!m_d)
        hipCheck(hipHostGetDevicePointer((void**)&m_d, m_p, 0), "hipHostGetDevicePointer");
    return m_d;
    }

void WordReadback::read(const unsigned int* d_words, unsigned int n, unsigned int* out, const char* what)
    {
    begin(d_words, n);
    finish(out, what);
    }

void WordReadback::begin(const unsigned int* d_words, unsigned int n)
    {
    if (n > 64)
        throw std::runtime_error("WordReadback: more than 64 words");
    m_n = n;
    if (n == 0)
        return;
    m_stamp++;
    abiCheck(hmd_publish_words(d_words, n, reinterpret_cast<unsigned long long*>(m_pinned.deviceView()), m_stamp, nullptr), "hmd_publish_words");
    }

unsigned long long* WordReadback::expect(unsigned int n, unsigned int& stamp)
    {
    if (n > 64)
        throw std::runtime_error("WordReadback: more than 64 words");
    m_n = n;
    stamp = ++m_stamp;
    return reinterpret_cast<unsigned long long*>(m_pinned.deviceView());
    }

void WordReadback::finish(unsigned int* out, const char* what)
    {
    volatile unsigned long long* h = reinterpret_cast<volatile unsigned long long*>(m_pinned.get());
    for (unsigned int k = 0; k < m_n; k++)
        out[k] = (unsigned int)pollStampedWord(h + k, m_stamp, what);
    }

// time the host has spent waiting for published words (a run that is bound by the device spends most of its time here)
static double g_poll_wait_s = 0.0;
double pollWaitSeconds(bool reset)
    {
    const double v = g_poll_wait_s;
    if (reset)
        g_poll_wait_s = 0.0;
    return v;
    }

unsigned long long pollStampedWord(volatile unsigned long long* h, unsigned int stamp, const char* what, hipStream_t other)
    {
    struct Stopwatch
        {
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~Stopwatch()
            {
            g_poll_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
        } stopwatch;
    static const double limit_s = []()
    {
        const char* e = std::getenv("HOOMD_AMD_POLL_TIMEOUT_S");
        const double v = e ? std::atof(e) : 0.0;
        return v > 0.0 ? v : 120.0;
    }();
    std::chrono::steady_clock::time_point t0;
    bool timing = false;
    for (uint64_t spins = 0;; spins++)
        {
        const unsigned long long v = *h;
        if ((unsigned int)(v >> 32) == stamp)
            return v;
        __builtin_ia32_pause();
        if ((spins & 0xfffff) == 0xfffff)
            {
            // surfaces a faulted kernel on either stream instead of spinning forever
            const hipError_t q0 = hipStreamQuery(0);
            if (q0 != hipSuccess && q0 != hipErrorNotReady)
                hipCheck(q0, what);
            hipError_t q1 = hipSuccess;
            if (other)
                {
                q1 = hipStreamQuery(other);
                if (q1 != hipSuccess && q1 != hipErrorNotReady)
                    hipCheck(q1, what);
                }
            const auto now = std::chrono::steady_clock::now();
            if (!timing)
                {
                t0 = now;
                timing = true;
                }
            else if (std::chrono::duration<double>(now - t0).count() > limit_s)
                {
                std::ostringstream oss;
                oss << what << ": word with stamp " << stamp << " did not arrive within " << limit_s << " s (last stamp seen "
                    << (unsigned int)(v >> 32) << "); default stream " << (q0 == hipSuccess ? "idle" : "busy");
                if (other)
                    oss << ", exchange stream " << (q1 == hipSuccess ? "idle" : "busy");
                throw std::runtime_error(oss.str());
                }
            }
        }
    }

PinnedWords::~PinnedWords()
    {
    if (m_p)
        (void)hipHostFree(m_p);
    }

void SnapshotParticleData::resize(unsigned int n)
    {
    size = n;
    pos.assign(n, {0, 0, 0});
    vel.assign(n, {0, 0, 0});
    accel.assign(n, {0, 0, 0});
    type.assign(n, 0);
    body.assign(n, HMD_NO_BODY);
    mass.assign(n, 1.0);
    charge.assign(n, 0.0);
    diameter.assign(n, 1.0);
    image.assign(n, {0, 0, 0});
    }

// hoomd/Compute.cc:42-76
bool Compute::shouldCompute(uint64_t timestep)
    {
    if (m_first_compute)
        {
        m_first_compute = false;
        m_last_computed = timestep;
        return true;
        }
    if (m_force_compute)
        {
        m_force_compute = false;
        return true;
        }
    if (m_last_computed != timestep)
        {
        m_last_computed = timestep;
        return true;
        }
    return false;
    }

// ---------------------------------------------------------------------------------------------
static double int_as_scalar(unsigned int a)
    {
    double v = 0.0;
    std::memcpy(&v, &a, sizeof(unsigned int));
    return v;
    }

static unsigned int scalar_as_int(double v)
    {
    unsigned int a;
    std::memcpy(&a, &v, sizeof(unsigned int));
    return a;
    }

ParticleData::ParticleData(const SnapshotParticleData& snap, const BoxDim& global_box, unsigned int n_global)
    : m_box(global_box), m_global_box(global_box)
    {
    m_type_mapping = snap.type_mapping;
    if (m_type_mapping.empty())
        throw std::runtime_error("snapshot defines no particle types");
    m_nglobal = n_global ? n_global : snap.size;
    initializeFromSnapshot(snap);
    }

unsigned int ParticleData::getTypeByName(const std::string& name) const
    {
    for (unsigned int i = 0; i < m_type_mapping.size(); i++)
        if (m_type_mapping[i] == name)
            return i;
    throw std::runtime_error("Type " + name + " not found!");
    }

void ParticleData::setBoxes(const BoxDim& global_box, const BoxDim& local_box)
    {
    m_global_box = global_box;
    m_box = local_box;
    m_box_signal.emit();
    }

void ParticleData::reallocate(unsigned int max_n)
    {
    m_maxn = max_n;
    m_pos.resize(max_n);
    m_vel.resize(max_n);
    m_accel.resize(max_n);
    m_image.resize(max_n);
    m_tag.resize(max_n);
    m_body.resize(max_n);
    m_charge.resize(max_n);
    m_diameter.resize(max_n);
    m_net_force.resize(max_n);
    // virial rows are pitch = max_n apart: contents are recomputed every step, no need to keep them
    m_net_virial.resize((size_t)6 * max_n, false);
    m_alt4.resize(max_n, false);
    m_alt3.resize(max_n, false);
    m_alti3.resize(max_n, false);
    m_altu.resize(max_n, false);
    m_alt1.resize(max_n, false);
    m_maxn_signal.emit();
    }

void ParticleData::resizeOwned(unsigned int n)
    {
    if (n + m_nghost > m_maxn)
        reallocate((unsigned int)((n + m_nghost) * 1.125) + 64);
    m_N = n;
    }

void ParticleData::setNGhosts(unsigned int n)
    {
    if (m_N + n > m_maxn)
        reallocate((unsigned int)((m_N + n) * 1.125) + 64);
    m_nghost = n;
    }

void ParticleData::initializeFromSnapshot(const SnapshotParticleData& snap)
    {
    const unsigned int n = snap.size;
    m_N = n;
    m_nghost = 0;
    if (n > m_maxn || m_maxn == 0)
        reallocate(std::max(n, 64u));
    m_rtag.resize(std::max(m_nglobal, n), false);

    std::vector<Scalar4> pos(n), vel(n);
    std::vector<Scalar3> accel(n);
    std::vector<int3> image(n);
    std::vector<unsigned int> tag(n), body(n);
    std::vector<Scalar> charge(n), diameter(n);
    for (unsigned int i = 0; i < n; i++)
        {
        if (snap.type[i] >= m_type_mapping.size())
            throw std::runtime_error("snapshot particle type id out of range");
        pos[i] = Scalar4 {snap.pos[i][0], snap.pos[i][1], snap.pos[i][2], int_as_scalar(snap.type[i])};
        vel[i] = Scalar4 {snap.vel[i][0], snap.vel[i][1], snap.vel[i][2], snap.mass[i]};
        accel[i] = Scalar3 {snap.accel[i][0], snap.accel[i][1], snap.accel[i][2]};
        image[i] = int3 {snap.image[i][0], snap.image[i][1], snap.image[i][2]};
        tag[i] = snap.tag.empty() ? i : snap.tag[i];
        if (tag[i] >= m_rtag.size())
            throw std::runtime_error("snapshot particle tag exceeds the global particle count");
        body[i] = snap.body[i];
        charge[i] = snap.charge[i];
        diameter[i] = snap.diameter[i];
        }
    m_pos.upload(pos.data(), n);
    m_vel.upload(vel.data(), n);
    m_accel.upload(accel.data(), n);
    m_image.upload(image.data(), n);
    m_tag.upload(tag.data(), n);
    abiCheck(hmd_build_rtag(m_rtag.get(), m_tag.get(), n, nullptr), "build rtag");
    m_rtag_stale = false;
    m_body.upload(body.data(), n);
    m_charge.upload(charge.data(), n);
    m_diameter.upload(diameter.data(), n);
    m_net_force.zero();
    m_accel_set = false;
    for (unsigned int i = 0; i < n && !m_accel_set; i++)
        m_accel_set = snap.accel[i][0] != 0 || snap.accel[i][1] != 0 || snap.accel[i][2] != 0;
    m_composition_generation++;
    notifyParticleSort();
    }

void ParticleData::takeSnapshot(SnapshotParticleData& snap) const
    {
    const unsigned int n = m_N;
    snap.resize(n);
    snap.type_mapping = m_type_mapping;
    auto pos = m_pos.toHost(n);
    auto vel = m_vel.toHost(n);
    auto accel = m_accel.toHost(n);
    auto image = m_image.toHost(n);
    auto tag = m_tag.toHost(n);
    auto body = m_body.toHost(n);
    auto charge = m_charge.toHost(n);
    auto diameter = m_diameter.toHost(n);
    // order by tag; under domain decomposition the local tags are a subset, keep their relative order
    std::vector<unsigned int> order(n);
    for (unsigned int i = 0; i < n; i++)
        order[i] = i;
    std::sort(order.begin(), order.end(), [&](unsigned int a, unsigned int b) { return tag[a] < tag[b]; });
    snap.body.assign(n, 0);
    snap.tag.assign(n, 0);
    for (unsigned int k = 0; k < n; k++)
        {
        const unsigned int i = order[k];
        snap.tag[k] = tag[i];
        snap.pos[k] = {pos[i].x, pos[i].y, pos[i].z};
        snap.type[k] = scalar_as_int(pos[i].w);
        snap.vel[k] = {vel[i].x, vel[i].y, vel[i].z};
        snap.mass[k] = vel[i].w;
        snap.accel[k] = {accel[i].x, accel[i].y, accel[i].z};
        snap.image[k] = {image[i].x, image[i].y, image[i].z};
        snap.body[k] = body[i];
        snap.charge[k] = charge[i];
        snap.diameter[k] = diameter[i];
        }
    }

void ParticleData::applyOrder(const unsigned int* d_order)
    {
    const unsigned int n = m_N;
    if (n == 0)
        return;
    abiCheck(hmd_gather_rows(m_alt4.get(), m_pos.get(), d_order, n, 32, nullptr), "gather pos");
    // ghosts (if any) keep their places behind the owned range
    if (m_nghost)
        hipCheck(hipMemcpyAsync(m_alt4.get() + n, m_pos.get() + n, sizeof(Scalar4) * m_nghost, hipMemcpyDeviceToDevice, 0), "copy ghosts");
    m_pos.swap(m_alt4);
    abiCheck(hmd_gather_rows(m_alt4.get(), m_vel.get(), d_order, n, 32, nullptr), "gather vel");
    m_vel.swap(m_alt4);
    abiCheck(hmd_gather_rows(m_alt3.get(), m_accel.get(), d_order, n, 24, nullptr), "gather accel");
    m_accel.swap(m_alt3);
    abiCheck(hmd_gather_rows(m_alti3.get(), m_image.get(), d_order, n, 12, nullptr), "gather image");
    m_image.swap(m_alti3);
    abiCheck(hmd_gather_rows(m_altu.get(), m_tag.get(), d_order, n, 4, nullptr), "gather tag");
    if (m_nghost)
        hipCheck(hipMemcpyAsync(m_altu.get() + n, m_tag.get() + n, sizeof(unsigned int) * m_nghost, hipMemcpyDeviceToDevice, 0), "copy ghost tags");
    m_tag.swap(m_altu);
    abiCheck(hmd_gather_rows(m_altu.get(), m_body.get(), d_order, n, 4, nullptr), "gather body");
    m_body.swap(m_altu);
    abiCheck(hmd_gather_rows(m_alt1.get(), m_charge.get(), d_order, n, 8, nullptr), "gather charge");
    m_charge.swap(m_alt1);
    abiCheck(hmd_gather_rows(m_alt1.get(), m_diameter.get(), d_order, n, 8, nullptr), "gather diameter");
    m_diameter.swap(m_alt1);
    if (!m_rtag_stale) // (a stale table is rebuilt whole by its next reader)
        abiCheck(hmd_build_rtag(m_rtag.get(), m_tag.get(), n, nullptr), "build rtag");
    notifyParticleSort();
    }

void ParticleData::refreshRtag() const
    {
    hipCheck(hipMemsetAsync(m_rtag.get(), 0xff, sizeof(unsigned int) * m_rtag.size(), nullptr), "clear rtag");
    abiCheck(hmd_build_rtag(m_rtag.get(), m_tag.get(), m_N + m_nghost, nullptr), "build rtag");
    m_rtag_stale = false;
    }

// ---------------------------------------------------------------------------------------------
ParticleGroup::ParticleGroup(std::shared_ptr<SystemDefinition> sysdef) : m_pdata(sysdef->getParticleData()), m_is_all(true) { }

ParticleGroup::ParticleGroup(std::shared_ptr<SystemDefinition> sysdef, const std::vector<unsigned int>& tags)
    : m_pdata(sysdef->getParticleData()), m_is_all(false), m_tags(tags)
    {
    std::sort(m_tags.begin(), m_tags.end());
    m_slot = m_pdata->getParticleSortSignal().connect([this]() { m_dirty = true; });
    }

ParticleGroup::~ParticleGroup()
    {
    if (m_slot >= 0)
        m_pdata->getParticleSortSignal().disconnect(m_slot);
    }

// hoomd/ParticleGroup.cc:430-500 (rebuild of the index list after a sort): host side, rare
void ParticleGroup::rebuild()
    {
    const unsigned int n = m_pdata->getN();
    std::vector<unsigned int> tag(n);
    if (n)
        hipCheck(hipMemcpy(tag.data(), m_pdata->tag(), sizeof(unsigned int) * n, hipMemcpyDeviceToHost), "download tags");
    std::vector<unsigned int> idx;
    for (unsigned int i = 0; i < n; i++)
        if (std::binary_search(m_tags.begin(), m_tags.end(), tag[i]))
            idx.push_back(i);
    m_n_local = (unsigned int)idx.size();
    m_idx.resize(std::max<size_t>(idx.size(), 1), false);
    m_idx.upload(idx.data(), idx.size());
    m_dirty = false;
    }

unsigned int ParticleGroup::getNumMembers()
    {
    if (m_is_all)
        return m_pdata->getN();
    if (m_dirty)
        rebuild();
    return m_n_local;
    }

const unsigned int* ParticleGroup::getIndexArray()
    {
    if (m_is_all)
        return nullptr;
    if (m_dirty)
        rebuild();
    return m_idx.get();
    }

// ---------------------------------------------------------------------------------------------
KernelTimer& KernelTimer::get()
    {
    static KernelTimer t;
    return t;
    }

void KernelTimer::enable(bool on)
    {
    m_on = on;
    }

void KernelTimer::begin(const char* name, hipStream_t stream)
    {
    Rec r;
    r.name = name;
    r.stream = stream;
    if (!m_pool.empty())
        {
        r.e0 = m_pool.back().first;
        r.e1 = m_pool.back().second;
        m_pool.pop_back();
        }
    else
        {
        hipCheck(hipEventCreate(&r.e0), "hipEventCreate");
        hipCheck(hipEventCreate(&r.e1), "hipEventCreate");
        }
    hipCheck(hipEventRecord(r.e0, stream), "hipEventRecord");
    m_open.push_back(r);
    }

void KernelTimer::end(const char* name)
    {
    for (size_t i = m_open.size(); i-- > 0;)
        if (m_open[i].name == name)
            {
            hipCheck(hipEventRecord(m_open[i].e1, m_open[i].stream), "hipEventRecord");
            m_done.push_back(m_open[i]);
            m_open.erase(m_open.begin() + i);
            break;
            }
    // keep the number of live events bounded
    if (m_done.size() > 4096)
        collect();
    }

std::vector<std::tuple<std::string, uint64_t, double>> KernelTimer::collect()
    {
    hipCheck(hipStreamSynchronize(0), "hipStreamSynchronize");
    std::vector<float> elapsed(m_done.size(), 0.0f);
    for (size_t k = 0; k < m_done.size(); k++)
        {
        auto& r = m_done[k];
        if (r.stream)
            hipCheck(hipEventSynchronize(r.e1), "hipEventSynchronize");
        hipCheck(hipEventElapsedTime(&elapsed[k], r.e0, r.e1), "hipEventElapsedTime");
        }
    // a negative limit is relative: that fraction of the median sample of the same name
    std::vector<std::pair<std::string, double>> limits;
    auto limit_of = [&](const std::string& name) -> double
    {
        if (m_min_ms >= 0)
            return m_min_ms;
        for (auto& l : limits)
            if (l.first == name)
                return l.second;
        std::vector<float> v;
        for (size_t k = 0; k < m_done.size(); k++)
            if (m_done[k].name == name)
                v.push_back(elapsed[k]);
        std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
        limits.emplace_back(name, -m_min_ms * (double)v[v.size() / 2]);
        return limits.back().second;
    };
    for (size_t k = 0; k < m_done.size(); k++)
        {
        auto& r = m_done[k];
        const float ms = elapsed[k];
        if (ms < limit_of(r.name))
            r.name += "_skipped";
        bool found = false;
        for (auto& a : m_acc)
            if (std::get<0>(a) == r.name)
                {
                std::get<1>(a) += 1;
                std::get<2>(a) += ms;
                found = true;
                }
        if (!found)
            m_acc.emplace_back(r.name, 1, (double)ms);
        m_pool.emplace_back(r.e0, r.e1);
        }
    m_done.clear();
    return m_acc;
    }

void KernelTimer::reset()
    {
    collect();
    m_acc.clear();
    }

    } // namespace hoomd_amd
