// communicator.cc -- CommunicatorGPU: particle migration, ghost-layer exchange, per-step ghost update and the rebuild
// vote of a domain-decomposed run, one process per GPU.
//
// Role of hoomd/Communicator.cc (communicate :1510-1569, migrateParticles, exchangeGhosts, beginUpdateGhosts /
// finishUpdateGhosts) and hoomd/CommunicatorGPU.cc:1620-2916 with the kernels of hoomd/CommunicatorGPU.cu.
//
// Designed for point-to-point links rather than staged sweeps along x, y, z: every (particle, direction) ghost image is
// one entry of a single send list sorted by destination rank, so a time step needs ONE pack kernel and ONE all-to-all
// that lands directly in the ghost tail pos[N, N + n_ghost) of the receiver; edges and corners are separate entries,
// not forwarded copies.  Migration (rebuild steps only) routes a leaving particle straight to its owner, however far
// away.  Ghost forces are never sent back: the neighbor list is full.  When every rank is a neighbor of every other
// (the grids of 2, 4 and 8 GPUs) migration messages have a fixed capacity and carry their own counts, and the rebuild
// vote rides in the halo messages, so those steps need no separate collective.
#include "hoomd_host.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <sstream>

namespace hoomd_amd
    {
namespace
    {
constexpr unsigned int ROW_WORDS = HMD_MIGRATE_ROW_WORDS; // 64-bit words of a migrating particle
constexpr unsigned int ROW_BYTES = ROW_WORDS * 8;

template<class T> void ensure(DeviceArray<T>& a, size_t n, double slack = 1.25, size_t pad = 64)
    {
    if (a.size() < n)
        a.resize((size_t)(n * slack) + pad, false);
    }

unsigned int bitsFor(unsigned int v)
    {
    unsigned int b = 1;
    while ((1ull << b) <= v)
        b++;
    return b;
    }
    } // namespace

CommunicatorGPU::CommunicatorGPU(std::shared_ptr<SystemDefinition> sysdef, std::shared_ptr<ExchangeTransport> transport, unsigned int rank,
                                 unsigned int world, std::array<unsigned int, 3> grid, std::array<double, 3> glo, std::array<double, 3> gL)
    : m_sysdef(sysdef), m_pdata(sysdef->getParticleData()), m_transport(transport), m_rank(rank), m_world(world), m_grid(grid), m_glo(glo), m_gL(gL)
    {
    if (!transport)
        throw std::invalid_argument("CommunicatorGPU needs a transport");
    if (grid[0] * grid[1] * grid[2] != world || rank >= world)
        throw std::invalid_argument("CommunicatorGPU: the grid does not match the number of ranks");
    m_coords = {rank % grid[0], (rank / grid[0]) % grid[1], rank / (grid[0] * grid[1])};
    for (int a = 0; a < 3; a++)
        m_decomposed[a] = grid[a] > 1;

    // one entry per direction towards a neighbor domain (plan bits as hoomd/CommunicatorGPU.cu:306-358: 1 << 2a for
    // the upper face of axis a, 2 << 2a for the lower one), with the periodic shift a ghost gets on the way
    struct Dir
        {
        unsigned int dest, mask;
        Scalar3 shift;
        };
    std::vector<Dir> dirs;
    for (int dx = -1; dx <= 1; dx++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dz = -1; dz <= 1; dz++)
                {
                const int d[3] = {dx, dy, dz};
                if (!dx && !dy && !dz)
                    continue;
                bool ok = true;
                for (int a = 0; a < 3; a++)
                    if (d[a] != 0 && !m_decomposed[a])
                        ok = false;
                if (!ok)
                    continue;
                unsigned int mask = 0;
                double shift[3] = {0, 0, 0};
                for (int a = 0; a < 3; a++)
                    {
                    if (d[a] == 1)
                        {
                        mask |= 1u << (2 * a);
                        if (m_coords[a] + 1 >= m_grid[a])
                            shift[a] = -m_gL[a];
                        }
                    else if (d[a] == -1)
                        {
                        mask |= 2u << (2 * a);
                        if (m_coords[a] == 0)
                            shift[a] = m_gL[a];
                        }
                    }
                dirs.push_back({rankAt((int)m_coords[0] + dx, (int)m_coords[1] + dy, (int)m_coords[2] + dz), mask, Scalar3 {shift[0], shift[1], shift[2]}});
                }
    std::sort(dirs.begin(), dirs.end(), [](const Dir& a, const Dir& b) { return a.dest != b.dest ? a.dest < b.dest : a.mask < b.mask; });
    std::vector<unsigned int> masks;
    std::vector<Scalar3> shifts;
    for (auto& d : dirs)
        {
        m_dir_dest.push_back(d.dest);
        masks.push_back(d.mask);
        shifts.push_back(d.shift);
        }
    m_dir_mask.resize(std::max<size_t>(masks.size(), 1), false);
    m_shift_table.resize(std::max<size_t>(shifts.size(), 1), false);
    m_dir_mask.upload(masks.data(), masks.size());
    m_shift_table.upload(shifts.data(), shifts.size());
    m_dir_dest_dev.resize(std::max<size_t>(m_dir_dest.size(), 1), false);
    m_dir_dest_dev.upload(m_dir_dest.data(), m_dir_dest.size());

    // is every other rank a halo partner?
    std::vector<unsigned int> peers(m_dir_dest);
    std::sort(peers.begin(), peers.end());
    peers.erase(std::unique(peers.begin(), peers.end()), peers.end());
    std::vector<unsigned int> others;
    for (unsigned int r = 0; r < world; r++)
        if (r != rank)
            others.push_back(r);
    m_all_peers = peers == others; // (one rank: no peers, no other ranks)
    m_fixed_migrate = m_all_peers && world <= 64;
    m_send_rows.assign(world, 0);
    m_recv_rows.assign(world, 0);
    m_mig_counters.resize(1 + world, false);
    m_mig_keys.resize(4096, false);
    m_send_counters.resize(1 + 26, false);
    m_count_out.resize(world, false);
    m_count_in.resize(2 * (size_t)world, false);
    }

unsigned int CommunicatorGPU::rankAt(int i, int j, int k) const
    {
    const int nx = (int)m_grid[0], ny = (int)m_grid[1], nz = (int)m_grid[2];
    return (unsigned int)((((k % nz + nz) % nz) * ny + ((j % ny + ny) % ny)) * nx + ((i % nx + nx) % nx));
    }

hmd_pdata_arrays_t CommunicatorGPU::pdataArrays() const
    {
    hmd_pdata_arrays_t p;
    p.d_pos = m_pdata->pos();
    p.d_vel = m_pdata->vel();
    p.d_accel = m_pdata->accel();
    p.d_image = m_pdata->image();
    p.d_tag = m_pdata->tag();
    p.d_body = m_pdata->body();
    p.d_charge = m_pdata->charge();
    p.d_diameter = m_pdata->diameter();
    return p;
    }

Scalar CommunicatorGPU::getGhostLayerMaxWidth() const
    {
    Scalar w = 0;
    for (auto& wnl : m_nlists)
        if (auto nl = wnl.lock())
            w = std::max(w, nl->getGhostLayerWidth());
    return w;
    }

// sorted keys end up in m_keys_sorted
void CommunicatorGPU::sortKeys(DeviceArray<unsigned long long>& keys, unsigned int n, unsigned int key_bits)
    {
    ensure(m_keys_sorted, n, 1.25, 1024);
    const size_t need = hmd_sort_keys_workspace_bytes(n);
    if (m_sort_ws.size() < need)
        m_sort_ws.resize(need + need / 4, false);
    abiCheck(hmd_sort_keys(keys.get(), m_keys_sorted.get(), n, key_bits, m_sort_ws.get(), m_sort_ws.size(), nullptr), "hmd_sort_keys");
    }

// hoomd/Communicator.cc:1510-1569
void CommunicatorGPU::communicate(uint64_t timestep)
    {
    bool migrate = m_force_migrate;
    m_ghosts_fresh = false;
    m_in_communicate = true;
    try
        {
        // the neighbor lists decide (distance check + vote); the vote refreshes the ghost positions on its way
        for (auto& wnl : m_nlists)
            if (auto nl = wnl.lock())
                if (nl->peekUpdate(timestep))
                    migrate = true;
        }
    catch (...)
        {
        m_in_communicate = false;
        throw;
        }
    m_in_communicate = false;
    if (migrate)
        {
        m_force_migrate = false;
        migrateParticles();
        // on every rank, whether or not its own particle set changed: the forced-update state of the neighbor list
        // must stay identical across ranks or their collective calls diverge
        m_pdata->notifyParticleSort();
        exchangeGhosts(getGhostLayerMaxWidth());
        }
    else if (!m_ghosts_fresh)
        updateGhosts(nullptr);
    }

// ---- migration ----------------------------------------------------------------------------------
// Migration with fixed-size messages and every size read on the device: owner decision, sort of the (padded) key array,
// pack, exchange, and only then one look of the host at the counts that stayed and the headers that arrived.
bool CommunicatorGPU::migrateParticlesDeviceCounts()
    {
    const unsigned int N = m_pdata->getN();
    m_pdata->setNGhosts(0);
    ensure(m_mig_dest, std::max(N, 1u), 1.125);
    hmd_migrate_dest_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.d_pos = m_pdata->pos();
    a.N = N;
    a.rank = m_rank;
    for (int d = 0; d < 3; d++)
        {
        a.global_lo[d] = m_glo[d];
        a.global_L[d] = m_gL[d];
        a.grid[d] = m_grid[d];
        a.decomposed[d] = m_decomposed[d] ? 1 : 0;
        }
    a.d_dest = m_mig_dest.get();
    a.d_n_leave = m_mig_counters.get();
    a.d_rank_count = m_mig_counters.get() + 1;
    a.n_ranks = m_world;
    hmd_migrate_pack_args_t pk;
    std::memset(&pk, 0, sizeof(pk));
    for (int d = 0; d < 3; d++)
        {
        pk.global_lo[d] = m_glo[d];
        pk.global_L[d] = m_gL[d];
        pk.decomposed[d] = m_decomposed[d] ? 1 : 0;
        }
    pk.n_ranks = m_world;
    const unsigned int n_words = 3 * m_world + 1;
    ensure(m_count_in, n_words, 1.0, 0);
    std::vector<unsigned int> words(n_words);
    unsigned int slot_rows = 0, n_leave = 0;
    while (true)
        {
        const unsigned int cap = (unsigned int)m_mig_keys.size();
        // (counters to zero; places past the last key hold a destination no rank has, so they sort behind every key)
        abiCheck(hmd_comm_prep(m_mig_counters.get(), 1 + m_world, m_mig_keys.get(), cap, nullptr), "hmd_comm_prep");
        a.d_keys = m_mig_keys.get();
        a.capacity = cap;
        abiCheck(hmd_migrate_dest(&a, nullptr), "hmd_migrate_dest");
        sortKeys(m_mig_keys, cap, 32 + bitsFor(m_world));
        pk.arrays = pdataArrays();
        pk.d_keys = m_keys_sorted.get();
        bool keys_overflowed = false;
        while (true)
            {
            slot_rows = m_mig_slot_rows;
            const size_t rows = (size_t)m_world * (slot_rows + 1);
            if (m_mig_out.size() < rows * ROW_WORDS)
                m_mig_out.resize(rows * ROW_WORDS, false);
            if (m_mig_in.size() < rows * ROW_WORDS)
                m_mig_in.resize(rows * ROW_WORDS, false);
            pk.slot_rows = slot_rows;
            pk.d_out = m_mig_out.get();
            abiCheck(hmd_migrate_pack_slots_dev(&pk, m_mig_counters.get(), m_mig_counters.get() + 1, cap, nullptr), "hmd_migrate_pack_slots_dev");
            const std::vector<unsigned int> each(m_world, slot_rows + 1);
            m_transport->allToAll((uintptr_t)m_mig_out.get(), each, (uintptr_t)m_mig_in.get(), each, ROW_BYTES, 0);
            unsigned int stamp = 0;
            unsigned long long* d_pub = m_readback.expect(n_words, stamp);
            abiCheck(hmd_migrate_slot_headers_and(m_mig_in.get(), slot_rows, m_world, m_mig_counters.get(), 1 + m_world, m_count_in.get(), d_pub, stamp, nullptr),
                     "hmd_migrate_slot_headers_and");
            m_readback.finish(words.data(), "migration headers and counts");
            n_device_count_exchanges++;
            unsigned int most = 0;
            for (unsigned int r = 0; r < m_world; r++)
                {
                if (words[2 * r + 1] == 0xffffffffu)
                    keys_overflowed = true; // some rank's list of leavers was incomplete: every rank sees it and starts over
                else
                    most = std::max(most, words[2 * r + 1]);
                }
            if (keys_overflowed || most <= slot_rows)
                break;
            m_mig_slot_rows = 2 * most; // the same decision on every rank: send again with room for all
            }
        n_leave = words[2 * m_world];
        if (!keys_overflowed)
            break;
        if (n_leave > cap)
            m_mig_keys.resize((size_t)(n_leave * 1.25) + 1024, false);
        }
    std::vector<unsigned int> recv_counts(m_world, 0);
    unsigned int n_recv = 0;
    for (unsigned int r = 0; r < m_world; r++)
        {
        recv_counts[r] = words[2 * r];
        n_recv += recv_counts[r];
        }
    n_migrated_out += n_leave;
    n_migrated_in += n_recv;
    if (n_leave == 0 && n_recv == 0)
        return false;
    const unsigned int n_keep = N - n_leave;
    if (n_leave)
        {
        ensure(m_scratch, 2 * (size_t)n_leave);
        hmd_migrate_fill_args_t f;
        std::memset(&f, 0, sizeof(f));
        f.arrays = pdataArrays();
        f.d_dest = m_mig_dest.get();
        f.rank = m_rank;
        f.N = N;
        f.d_keys = m_keys_sorted.get();
        f.n_leave = n_leave;
        f.d_scratch = m_scratch.get();
        abiCheck(hmd_migrate_fill(&f, nullptr), "hmd_migrate_fill");
        }
    m_pdata->resizeOwned(n_keep + n_recv);
    if (n_recv)
        {
        hmd_migrate_unpack_args_t u;
        std::memset(&u, 0, sizeof(u));
        u.arrays = pdataArrays();
        u.d_in = m_mig_in.get();
        u.first = n_keep;
        u.n = n_recv;
        u.slot_rows = slot_rows;
        u.n_ranks = m_world;
        unsigned int off = 0;
        for (unsigned int r = 0; r < m_world; r++)
            {
            u.recv_off[r] = off;
            off += recv_counts[r];
            }
        u.recv_off[m_world] = off;
        abiCheck(hmd_migrate_unpack(&u, nullptr), "hmd_migrate_unpack");
        }
    return true;
    }

bool CommunicatorGPU::migrateParticles()
    {
    ScopedKernelTimer timer("comm_migrate");
    if (m_fixed_migrate && m_device_counts && 3 * m_world + 1 <= 64)
        return migrateParticlesDeviceCounts();
    const unsigned int N = m_pdata->getN();
    m_pdata->setNGhosts(0); // ghosts are exchanged again afterwards
    ensure(m_mig_dest, std::max(N, 1u), 1.125);
    hmd_migrate_dest_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.d_pos = m_pdata->pos();
    a.N = N;
    a.rank = m_rank;
    for (int d = 0; d < 3; d++)
        {
        a.global_lo[d] = m_glo[d];
        a.global_L[d] = m_gL[d];
        a.grid[d] = m_grid[d];
        a.decomposed[d] = m_decomposed[d] ? 1 : 0;
        }
    a.d_dest = m_mig_dest.get();
    a.d_n_leave = m_mig_counters.get();
    a.d_rank_count = m_mig_counters.get() + 1;
    a.n_ranks = m_world;
    std::vector<unsigned int> counters(1 + m_world);
    while (true)
        {
        m_mig_counters.zero();
        a.d_keys = m_mig_keys.get();
        a.capacity = (unsigned int)m_mig_keys.size();
        abiCheck(hmd_migrate_dest(&a, nullptr), "hmd_migrate_dest");
        for (unsigned int o = 0; o < counters.size(); o += 64)
            m_readback.read(m_mig_counters.get() + o, std::min<unsigned int>(64, (unsigned int)counters.size() - o), counters.data() + o, "migration counts");
        if (counters[0] <= m_mig_keys.size())
            break;
        m_mig_keys.resize((size_t)(counters[0] * 1.25) + 1024, false);
        }
    const unsigned int n_leave = counters[0];
    std::vector<unsigned int> send_counts(counters.begin() + 1, counters.end()), recv_counts(m_world, 0);
    if (n_leave)
        sortKeys(m_mig_keys, n_leave, 32 + bitsFor(m_world));

    hmd_migrate_pack_args_t pk;
    std::memset(&pk, 0, sizeof(pk));
    pk.arrays = pdataArrays();
    pk.d_keys = n_leave ? m_keys_sorted.get() : nullptr;
    pk.n = n_leave;
    for (int d = 0; d < 3; d++)
        {
        pk.global_lo[d] = m_glo[d];
        pk.global_L[d] = m_gL[d];
        pk.decomposed[d] = m_decomposed[d] ? 1 : 0;
        }
    unsigned int slot_rows = 0, n_recv = 0;
    if (m_fixed_migrate)
        {
        // Every rank exchanges with every other rank: messages of a fixed number of rows, each headed by its particle
        // count and the sender's largest count.  All ranks see all headers, so all of them know every message size,
        // and whether any message was too small, without an exchange of counts.
        unsigned int off = 0;
        for (unsigned int r = 0; r < m_world; r++)
            {
            pk.send_off[r] = off;
            off += send_counts[r];
            }
        pk.send_off[m_world] = off;
        pk.n_ranks = m_world;
        std::vector<unsigned int> headers(2 * (size_t)m_world);
        ensure(m_count_in, 2 * (size_t)m_world, 1.0, 0);
        while (true)
            {
            slot_rows = m_mig_slot_rows;
            const size_t rows = (size_t)m_world * (slot_rows + 1);
            // (each on its own: the exchange with counts grows them independently, and the two forms can alternate)
            if (m_mig_out.size() < rows * ROW_WORDS)
                m_mig_out.resize(rows * ROW_WORDS, false);
            if (m_mig_in.size() < rows * ROW_WORDS)
                m_mig_in.resize(rows * ROW_WORDS, false);
            pk.slot_rows = slot_rows;
            pk.d_out = m_mig_out.get();
            abiCheck(hmd_migrate_pack(&pk, nullptr), "hmd_migrate_pack");
            const std::vector<unsigned int> each(m_world, slot_rows + 1);
            m_transport->allToAll((uintptr_t)m_mig_out.get(), each, (uintptr_t)m_mig_in.get(), each, ROW_BYTES, 0);
            // word 0 and 1 of the first row of every message
            abiCheck(hmd_migrate_slot_headers(m_mig_in.get(), slot_rows, m_world, m_count_in.get(), nullptr), "hmd_migrate_slot_headers");
            for (unsigned int o = 0; o < headers.size(); o += 64)
                m_readback.read(m_count_in.get() + o, std::min<unsigned int>(64, (unsigned int)headers.size() - o), headers.data() + o, "migration headers");
            unsigned int most = 0;
            for (unsigned int r = 0; r < m_world; r++)
                most = std::max(most, headers[2 * r + 1]);
            if (most <= slot_rows)
                break;
            m_mig_slot_rows = 2 * most; // the same decision on every rank: send again with room for all
            }
        for (unsigned int r = 0; r < m_world; r++)
            {
            recv_counts[r] = headers[2 * r];
            n_recv += recv_counts[r];
            }
        }
    else
        {
        if (n_leave)
            {
            ensure(m_mig_out, (size_t)n_leave * ROW_WORDS);
            pk.d_out = m_mig_out.get();
            abiCheck(hmd_migrate_pack(&pk, nullptr), "hmd_migrate_pack");
            }
        m_count_out.upload(send_counts.data(), m_world);
        const std::vector<unsigned int> one(m_world, 1);
        m_transport->allToAll((uintptr_t)m_count_out.get(), one, (uintptr_t)m_count_in.get(), one, 4, 0);
        m_count_in.download(recv_counts.data(), m_world);
        n_count_exchanges++;
        n_recv = std::accumulate(recv_counts.begin(), recv_counts.end(), 0u);
        ensure(m_mig_in, (size_t)std::max(n_recv, 1u) * ROW_WORDS);
        ensure(m_mig_out, ROW_WORDS);
        // every rank takes part in the collective, also with nothing to send or receive
        m_transport->allToAll((uintptr_t)m_mig_out.get(), send_counts, (uintptr_t)m_mig_in.get(), recv_counts, ROW_BYTES, 0);
        }
    n_migrated_out += n_leave;
    n_migrated_in += n_recv;
    if (n_leave == 0 && n_recv == 0)
        return false;
    const unsigned int n_keep = N - n_leave;
    if (n_leave)
        {
        ensure(m_scratch, 2 * (size_t)n_leave);
        hmd_migrate_fill_args_t f;
        std::memset(&f, 0, sizeof(f));
        f.arrays = pdataArrays();
        f.d_dest = m_mig_dest.get();
        f.rank = m_rank;
        f.N = N;
        f.d_keys = m_keys_sorted.get();
        f.n_leave = n_leave;
        f.d_scratch = m_scratch.get();
        abiCheck(hmd_migrate_fill(&f, nullptr), "hmd_migrate_fill");
        }
    m_pdata->resizeOwned(n_keep + n_recv); // (keeps the contents when it has to reallocate)
    if (n_recv)
        {
        hmd_migrate_unpack_args_t u;
        std::memset(&u, 0, sizeof(u));
        u.arrays = pdataArrays();
        u.d_in = m_mig_in.get();
        u.first = n_keep;
        u.n = n_recv;
        if (slot_rows)
            {
            u.slot_rows = slot_rows;
            u.n_ranks = m_world;
            unsigned int off = 0;
            for (unsigned int r = 0; r < m_world; r++)
                {
                u.recv_off[r] = off;
                off += recv_counts[r];
                }
            u.recv_off[m_world] = off;
            }
        abiCheck(hmd_migrate_unpack(&u, nullptr), "hmd_migrate_unpack");
        }
    return true;
    }

// The ghost layer with messages sized from the previous exchange and every size read on the device: send list, sort of
// the (padded) key array, pack, exchange, then one look of the host at the counts that stayed and the headers that
// arrived.  True when the ghosts are in place; false when a message was too small (the caller then exchanges counts
// first, with the send list this call leaves behind).
bool CommunicatorGPU::exchangeGhostsDeviceCounts(Scalar r_ghost, hmd_ghost_send_list_args_t& a, hmd_pack_pos_args_t& pa, hmd_unpack_ghost_init_args_t& u,
                                                 unsigned int& total)
    {
    const unsigned int ndir = (unsigned int)m_dir_dest.size();
    auto room = [this](unsigned int rows) { return rows + (m_ghost_room_div ? rows / m_ghost_room_div : 0u) + m_ghost_room_pad; };
    // the key array is sorted whole: keep it close to what the previous exchange needed
    unsigned int prev_total = 0;
    for (unsigned int r = 0; r < m_world; r++)
        prev_total += m_prev_send_rows[r];
    m_ghost_key_cap = std::max(m_ghost_key_cap, prev_total + prev_total / 8 + 1024);
    const unsigned int n_words = 2 * m_world + 1 + ndir;
    ensure(m_count_in, n_words, 1.0, 0);
    std::vector<unsigned int> words(n_words);
    hmd_ghost_slots_t so, si;
    while (true)
        {
        const unsigned int cap = m_ghost_key_cap;
        ensure(m_keys, cap, 1.0, 0);
        abiCheck(hmd_comm_prep(m_send_counters.get(), 1 + 26, m_keys.get(), cap, nullptr), "hmd_comm_prep");
        a.d_keys = m_keys.get();
        a.capacity = cap;
        abiCheck(hmd_ghost_send_list(&a, nullptr), "hmd_ghost_send_list");
        sortKeys(m_keys, cap, 32 + 5);
        ensure(m_send_idx, cap, 1.0, 0);
        ensure(m_send_dir, cap, 1.0, 0);
        std::memset(&so, 0, sizeof(so));
        std::memset(&si, 0, sizeof(si));
        so.n_ranks = si.n_ranks = m_world;
        std::vector<unsigned int> out_rows(m_world), in_rows(m_world);
        unsigned int out_total = 0, in_total = 0;
        for (unsigned int r = 0; r < m_world; r++)
            {
            so.cap[r] = room(m_prev_send_rows[r]);
            si.cap[r] = room(m_prev_recv_rows[r]);
            so.row0[r] = out_total;
            si.row0[r] = in_total;
            out_rows[r] = so.cap[r] + 1;
            in_rows[r] = si.cap[r] + 1;
            out_total += out_rows[r];
            in_total += in_rows[r];
            }
        ensure(m_init_out, out_total);
        ensure(m_init_in, in_total);
        pa.d_out = m_init_out.get();
        pa.d_pos = m_pdata->pos();
        pa.d_send_idx = m_send_idx.get();
        pa.d_send_shift_id = m_send_dir.get();
        pa.d_shift_table = m_shift_table.get();
        // (the send list -- particle index and direction of every row -- is split off the sorted keys by the pack itself)
        abiCheck(hmd_pack_ghost_init_slots_dev(&pa, m_pdata->tag(), &so, m_send_counters.get(), m_send_counters.get() + 1, m_dir_dest_dev.get(), ndir, cap,
                                               m_keys_sorted.get(), nullptr),
                 "hmd_pack_ghost_init_slots_dev");
        m_transport->allToAll((uintptr_t)m_init_out.get(), out_rows, (uintptr_t)m_init_in.get(), in_rows, sizeof(Scalar4), 0);
        unsigned int stamp = 0;
        unsigned long long* d_pub = m_readback.expect(n_words, stamp);
        abiCheck(hmd_ghost_slot_headers_and(m_init_in.get(), &si, m_send_counters.get(), 1 + ndir, m_count_in.get(), d_pub, stamp, nullptr),
                 "hmd_ghost_slot_headers_and");
        m_readback.finish(words.data(), "ghost message headers and counts");
        n_device_count_exchanges++;
        total = words[2 * m_world];
        bool list_overflowed = false;
        for (unsigned int r = 0; r < m_world; r++)
            list_overflowed = list_overflowed || (words[2 * r + 1] & 2u) != 0;
        if (!list_overflowed)
            break;
        // some rank's send list was incomplete: every rank sees it and starts over, that rank with a larger array
        if (total > cap)
            m_ghost_key_cap = (unsigned int)(total * 1.25) + 1024;
        }
    m_n_send = total;
    std::fill(m_send_rows.begin(), m_send_rows.end(), 0u);
    for (unsigned int k = 0; k < ndir; k++)
        m_send_rows[m_dir_dest[k]] += words[2 * m_world + 1 + k];
    bool too_small = false;
    for (unsigned int r = 0; r < m_world; r++)
        too_small = too_small || (words[2 * r + 1] & 1u) != 0;
    if (too_small)
        {
        n_ghost_slot_overflows++; // every rank falls back to the exchange with counts
        return false;
        }
    for (unsigned int r = 0; r < m_world; r++)
        {
        m_recv_rows[r] = words[2 * r];
        si.off[r + 1] = si.off[r] + m_recv_rows[r];
        }
    const unsigned int n_ghost = si.off[m_world];
    m_pdata->setNGhosts(n_ghost);
    u.d_in = m_init_in.get();
    u.n = n_ghost;
    u.arrays = pdataArrays();
    abiCheck(hmd_unpack_ghost_init_slots(&u, &si, nullptr), "hmd_unpack_ghost_init_slots");
    return true;
    }

// ---- ghost layer ----------------------------------------------------------------------------------
void CommunicatorGPU::exchangeGhosts(Scalar r_ghost)
    {
    ScopedKernelTimer timer("comm_ghosts");
    const unsigned int N = m_pdata->getN();
    m_pdata->setNGhosts(0);
    hmd_box_t lbox = m_pdata->getBox().b;
    for (int d = 0; d < 3; d++)
        if (m_decomposed[d] && (lbox.hi[d] - lbox.lo[d]) <= 2.0 * r_ghost)
            throw std::runtime_error("Simulation box too small for domain decomposition: a sub-box must be wider than twice the ghost layer");
    const unsigned int ndir = (unsigned int)m_dir_dest.size();
    const unsigned int ntypes = std::max(m_pdata->getNTypes(), 1u);
    if (m_r_ghost.size() != ntypes || m_r_ghost_value != r_ghost)
        {
        m_r_ghost.resize(ntypes, false);
        const std::vector<double> w(ntypes, r_ghost);
        m_r_ghost.upload(w.data(), ntypes);
        m_r_ghost_value = r_ghost;
        }
    if (m_keys.size() == 0)
        m_keys.resize(std::max<size_t>(N / 2, 1024), false);
    // plan, per-direction selection and counts in one kernel; one readback; one sort puts the keys
    // (direction, particle index) in the order of the send list
    const bool device_counts = m_device_counts && m_use_ghost_slots && m_all_peers && m_have_prev_rows && 2 * m_world + 27 <= 64;
    hmd_ghost_send_list_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.ghost_plan.d_pos = m_pdata->pos();
    a.ghost_plan.N = N;
    a.ghost_plan.box = lbox;
    a.ghost_plan.d_r_ghost = m_r_ghost.get();
    a.ghost_plan.ntypes = ntypes;
    for (int d = 0; d < 3; d++)
        if (m_decomposed[d])
            a.ghost_plan.mask |= 3u << (2 * d);
    a.d_dir_mask = m_dir_mask.get();
    a.n_dir = ndir;
    a.d_total = m_send_counters.get();
    a.d_dir_count = m_send_counters.get() + 1;
    hmd_pack_pos_args_t pa;
    std::memset(&pa, 0, sizeof(pa));
    hmd_unpack_ghost_init_args_t u;
    std::memset(&u, 0, sizeof(u));
    u.first = N;
    unsigned int total = 0;
    bool done = false;
    if (device_counts)
        done = exchangeGhostsDeviceCounts(r_ghost, a, pa, u, total);
    else
        {
        unsigned int counters[27];
        while (true)
            {
            m_send_counters.zero();
            a.d_keys = m_keys.get();
            a.capacity = (unsigned int)m_keys.size();
            abiCheck(hmd_ghost_send_list(&a, nullptr), "hmd_ghost_send_list");
            m_readback.read(m_send_counters.get(), 1 + ndir, counters, "ghost send counts");
            if (counters[0] <= m_keys.size())
                break;
            m_keys.resize((size_t)(counters[0] * 1.25) + 1024, false);
            }
        total = counters[0];
        sortKeys(m_keys, total, 32 + 5);
        ensure(m_send_idx, total, 1.25, 1024);
        ensure(m_send_dir, total, 1.25, 1024);
        abiCheck(hmd_ghost_split_keys(m_keys_sorted.get(), total, m_send_idx.get(), m_send_dir.get(), nullptr), "hmd_ghost_split_keys");
        m_n_send = total;
        std::fill(m_send_rows.begin(), m_send_rows.end(), 0u);
        for (unsigned int k = 0; k < ndir; k++)
            m_send_rows[m_dir_dest[k]] += counters[1 + k];
        }
    // one message per peer carries position, type and tag; the unpack kernel also sets the fields that are not
    // communicated
    pa.d_pos = m_pdata->pos();
    pa.d_send_idx = m_send_idx.get();
    pa.d_send_shift_id = m_send_dir.get();
    pa.d_shift_table = m_shift_table.get();
    pa.n_send = total;
    if (!done && !device_counts && m_use_ghost_slots && m_all_peers && m_have_prev_rows && 2 * m_world <= 64)
        {
        // Messages with room for somewhat more rows than the same pair of ranks exchanged last time -- a number both
        // ends know -- each headed by its row count and by whether any message of its sender was too small.  Every
        // rank hears from every other rank, so all of them learn all counts and reach the same verdict without an
        // exchange of counts before the messages.
        auto room = [this](unsigned int rows) { return rows + (m_ghost_room_div ? rows / m_ghost_room_div : 0u) + m_ghost_room_pad; };
        hmd_ghost_slots_t so, si;
        std::memset(&so, 0, sizeof(so));
        std::memset(&si, 0, sizeof(si));
        so.n_ranks = si.n_ranks = m_world;
        std::vector<unsigned int> out_rows(m_world), in_rows(m_world);
        unsigned int out_total = 0, in_total = 0;
        for (unsigned int r = 0; r < m_world; r++)
            {
            so.off[r + 1] = so.off[r] + m_send_rows[r];
            so.cap[r] = room(m_prev_send_rows[r]);
            si.cap[r] = room(m_prev_recv_rows[r]);
            so.row0[r] = out_total;
            si.row0[r] = in_total;
            out_rows[r] = so.cap[r] + 1;
            in_rows[r] = si.cap[r] + 1;
            out_total += out_rows[r];
            in_total += in_rows[r];
            if (m_send_rows[r] > so.cap[r])
                so.flag = 1;
            }
        ensure(m_init_out, out_total);
        ensure(m_init_in, in_total);
        ensure(m_count_in, 2 * (size_t)m_world, 1.0, 0);
        pa.d_out = m_init_out.get();
        abiCheck(hmd_pack_ghost_init_slots(&pa, m_pdata->tag(), &so, nullptr), "hmd_pack_ghost_init_slots");
        m_transport->allToAll((uintptr_t)m_init_out.get(), out_rows, (uintptr_t)m_init_in.get(), in_rows, sizeof(Scalar4), 0);
        abiCheck(hmd_ghost_slot_headers(m_init_in.get(), &si, m_count_in.get(), nullptr), "hmd_ghost_slot_headers");
        std::vector<unsigned int> words(2 * (size_t)m_world);
        m_readback.read(m_count_in.get(), 2 * m_world, words.data(), "ghost message headers");
        bool too_small = so.flag != 0;
        for (unsigned int r = 0; r < m_world; r++)
            too_small = too_small || words[2 * r + 1] != 0;
        if (!too_small)
            {
            for (unsigned int r = 0; r < m_world; r++)
                {
                m_recv_rows[r] = words[2 * r];
                si.off[r + 1] = si.off[r] + m_recv_rows[r];
                }
            const unsigned int n_ghost = si.off[m_world];
            m_pdata->setNGhosts(n_ghost);
            u.d_in = m_init_in.get();
            u.n = n_ghost;
            u.arrays = pdataArrays();
            abiCheck(hmd_unpack_ghost_init_slots(&u, &si, nullptr), "hmd_unpack_ghost_init_slots");
            done = true;
            }
        else
            n_ghost_slot_overflows++; // every rank falls through to the exchange with counts
        }
    if (!done)
        {
        // how many ghosts arrive from whom
        m_count_out.upload(m_send_rows.data(), m_world);
        const std::vector<unsigned int> one(m_world, 1);
        m_transport->allToAll((uintptr_t)m_count_out.get(), one, (uintptr_t)m_count_in.get(), one, 4, 0);
        m_count_in.download(m_recv_rows.data(), m_world);
        n_count_exchanges++;
        const unsigned int n_ghost = std::accumulate(m_recv_rows.begin(), m_recv_rows.end(), 0u);
        m_pdata->setNGhosts(n_ghost);
        ensure(m_init_out, std::max(total, 1u));
        ensure(m_init_in, std::max(n_ghost, 1u));
        pa.d_out = m_init_out.get();
        pa.d_pos = m_pdata->pos(); // (room for the ghosts may have moved the arrays)
        abiCheck(hmd_pack_ghost_init(&pa, m_pdata->tag(), nullptr), "hmd_pack_ghost_init");
        m_transport->allToAll((uintptr_t)m_init_out.get(), m_send_rows, (uintptr_t)m_init_in.get(), m_recv_rows, sizeof(Scalar4), 0);
        u.d_in = m_init_in.get();
        u.n = n_ghost;
        u.arrays = pdataArrays();
        abiCheck(hmd_unpack_ghost_init(&u, nullptr), "hmd_unpack_ghost_init");
        }
    m_prev_send_rows = m_send_rows;
    m_prev_recv_rows = m_recv_rows;
    m_have_prev_rows = true;
    buildVotePlan();
    m_pdata->markRtagStale(); // whoever reads the tag -> index table next (bond table, exclusions, snapshots) rebuilds it
    }

// Row maps for the halo exchange that also carries the rebuild vote: every message gets one extra row.  Only when this
// rank exchanges ghosts with every other rank (then the maximum over the received rows and the own flag is the
// maximum over all ranks).
void CommunicatorGPU::buildVotePlan()
    {
    m_have_vote_plan = false;
    if (!m_all_peers || m_world > HMD_VOTE_PLAN_MAX_RANKS)
        return;
    m_aug_send.assign(m_world, 0);
    m_aug_recv.assign(m_world, 0);
    hmd_vote_plan_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.n_ranks = m_world;
    a.rank = m_rank;
    unsigned int so = 0, ro = 0;
    m_n_out = m_n_in = 0;
    for (unsigned int r = 0; r < m_world; r++)
        {
        a.send_off[r] = so;
        a.recv_off[r] = ro;
        so += m_send_rows[r];
        ro += m_recv_rows[r];
        m_aug_send[r] = m_send_rows[r] + (r != m_rank ? 1 : 0);
        m_aug_recv[r] = m_recv_rows[r] + (r != m_rank ? 1 : 0);
        m_n_out += m_aug_send[r];
        m_n_in += m_aug_recv[r];
        }
    a.send_off[m_world] = so;
    a.recv_off[m_world] = ro;
    ensure(m_out_row, std::max(so, 1u));
    ensure(m_flag_rows, m_world, 1.0, 0);
    ensure(m_in_map, std::max(m_n_in, 1u));
    a.d_out_row = m_out_row.get();
    a.d_flag_rows = m_flag_rows.get();
    a.d_in_map = m_in_map.get();
    abiCheck(hmd_vote_plan(&a, nullptr), "hmd_vote_plan");
    ensure(m_vote_out, std::max(m_n_out, 1u));
    ensure(m_vote_in, std::max(m_n_in, 1u));
    m_have_vote_plan = true;
    }

// Ghost positions and the rebuild vote in one exchange; afterwards the word at d_flag holds the maximum of the flag
// words of all ranks.
void CommunicatorGPU::updateGhostsWithVote(unsigned int* d_flag, hipStream_t stream)
    {
    n_vote_exchanges++;
    hmd_pack_vote_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.pack.d_out = m_vote_out.get();
    a.pack.d_pos = m_pdata->pos();
    a.pack.d_send_idx = m_send_idx.get();
    a.pack.d_send_shift_id = m_send_dir.get();
    a.pack.d_shift_table = m_shift_table.get();
    a.pack.n_send = m_n_send;
    a.d_out_row = m_out_row.get();
    a.d_flag_word = d_flag;
    a.d_flag_rows = m_flag_rows.get();
    a.n_flag_rows = m_world - 1;
    abiCheck(hmd_pack_ghost_positions_vote(&a, stream), "hmd_pack_ghost_positions_vote");
    m_transport->allToAll((uintptr_t)m_vote_out.get(), m_aug_send, (uintptr_t)m_vote_in.get(), m_aug_recv, sizeof(Scalar4), (uintptr_t)stream);
    hmd_unpack_vote_args_t u;
    std::memset(&u, 0, sizeof(u));
    u.d_in = m_vote_in.get();
    u.d_map = m_in_map.get();
    u.n_rows = m_n_in;
    u.d_ghost_pos = m_pdata->pos() + m_pdata->getN();
    u.d_flag_word = d_flag;
    abiCheck(hmd_unpack_ghost_positions(&u, stream), "hmd_unpack_ghost_positions");
    }

// hoomd/Communicator.cc beginUpdateGhosts / finishUpdateGhosts
void CommunicatorGPU::updateGhosts(hipStream_t stream)
    {
    ensure(m_pack_buf, std::max(m_n_send, 1u));
    if (m_n_send)
        {
        hmd_pack_pos_args_t a;
        std::memset(&a, 0, sizeof(a));
        a.d_out = m_pack_buf.get();
        a.d_pos = m_pdata->pos();
        a.d_send_idx = m_send_idx.get();
        a.d_send_shift_id = m_send_dir.get();
        a.d_shift_table = m_shift_table.get();
        a.n_send = m_n_send;
        abiCheck(hmd_pack_ghost_positions(&a, stream), "hmd_pack_ghost_positions");
        }
    m_transport->allToAll((uintptr_t)m_pack_buf.get(), m_send_rows, (uintptr_t)(m_pdata->pos() + m_pdata->getN()), m_recv_rows, sizeof(Scalar4),
                          (uintptr_t)stream);
    }

// Device-side rebuild vote: the flag word of the distance check (the number of the last check that found the list
// stale, the same count on every rank) is MAX-reduced in place over all ranks, with no host round trip between the
// distance check and the reduction.  On the (common) steps where only ghost positions need refreshing the halo
// exchange is queued first, so one wait covers both and work predicated on the vote can be queued behind it.
uintptr_t CommunicatorGPU::beginRebuildVote(uintptr_t d_flag, unsigned int checkn, uintptr_t stream)
    {
    unsigned int* word = reinterpret_cast<unsigned int*>(d_flag);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    m_vote_word = word;
    m_vote_checkn = checkn;
    m_vote_stream = s;
    if (m_in_communicate && !m_force_migrate && !m_ghosts_fresh)
        {
        m_ghosts_fresh = true;
        if (m_have_vote_plan && m_use_vote_rows)
            {
            updateGhostsWithVote(word, s);
            return d_flag;
            }
        updateGhosts(s);
        }
    m_transport->allReduceMaxWord(d_flag, stream);
    return d_flag;
    }

bool CommunicatorGPU::endRebuildVote()
    {
    unsigned int v = 0;
    // the vote was queued on m_vote_stream, which may be a non-blocking stream the null stream does not order after
    hipCheck(hipMemcpyAsync(&v, m_vote_word, sizeof(v), hipMemcpyDeviceToHost, m_vote_stream), "vote D2H");
    hipCheck(hipStreamSynchronize(m_vote_stream), "vote D2H");
    return v == m_vote_checkn;
    }

    } // namespace hoomd_amd
