// bonds.cc -- BondData (hoomd/BondedGroupData.{h,cc}, bonds only) and the FENE + WCA bond force
// (hoomd/md/PotentialBond.h, hoomd/md/PotentialBondGPU.h).
//
// Under domain decomposition the reference migrates bonds with their particles and marks bond partners for the ghost
// exchange (hoomd/CommunicatorGPU.cc:718-1618).  Here the topology is static data: every rank holds the whole table
// keyed by tag (8 bytes per bond end: 128 MB for 8 million beads with two bonds each, out of 288 GB), and the table by
// local index is rebuilt from it with one kernel after every migration; a partner is found through rtag among the
// owned particles and the ghosts, which the ghost layer of the neighbor list (r_cut + buffer) already holds for
// every bond shorter than that.  Each rank evaluates a bond for the ends it owns, as it does for pairs.
#include "hoomd_host.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>

namespace hoomd_amd
    {
BondData::BondData(std::shared_ptr<ParticleData> pdata, const SnapshotBondData& snap) : m_pdata(pdata), m_snap(snap), m_flags(1)
    {
    const unsigned int n_tags = m_pdata->getNGlobal();
    if (m_snap.type_id.size() != m_snap.groups.size())
        throw std::runtime_error("bonds: typeid and group arrays differ in length");
    // table keyed by tag, bonds in the order they were defined (on both of their ends)
    std::vector<unsigned int> n(std::max(n_tags, 1u), 0);
    for (size_t b = 0; b < m_snap.groups.size(); b++)
        {
        const auto& g = m_snap.groups[b];
        if (g[0] >= n_tags || g[1] >= n_tags)
            throw std::runtime_error("Particle tag out of bounds in bond");
        if (g[0] == g[1])
            throw std::runtime_error("A bond cannot join a particle with itself");
        if (m_snap.type_id[b] >= m_snap.type_mapping.size())
            throw std::runtime_error("Bond type id out of bounds");
        n[g[0]]++;
        n[g[1]]++;
        }
    const unsigned int max_b = std::max(1u, *std::max_element(n.begin(), n.end()));
    m_tag_blist_idx = hmd_index2d_t {std::max(n_tags, 1u), max_b};
    std::vector<hmd_uint2> tab((size_t)m_tag_blist_idx.w * max_b, hmd_uint2 {0u, 0u});
    std::fill(n.begin(), n.end(), 0u);
    for (size_t b = 0; b < m_snap.groups.size(); b++)
        for (int e = 0; e < 2; e++)
            {
            const unsigned int me = m_snap.groups[b][e], other = m_snap.groups[b][1 - e];
            tab[(size_t)n[me] * m_tag_blist_idx.w + me] = hmd_uint2 {other, m_snap.type_id[b]};
            n[me]++;
            }
    m_tag_n_bonds.resize(n.size(), false);
    m_tag_blist.resize(tab.size(), false);
    hipCheck(hipMemcpy(m_tag_n_bonds.get(), n.data(), sizeof(unsigned int) * n.size(), hipMemcpyHostToDevice), "bond counts H2D");
    hipCheck(hipMemcpy(m_tag_blist.get(), tab.data(), sizeof(hmd_uint2) * tab.size(), hipMemcpyHostToDevice), "bond table H2D");
    m_sort_slot = m_pdata->getParticleSortSignal().connect([this]() { m_dirty = true; });
    m_maxn_slot = m_pdata->getMaxParticleNumberChangeSignal().connect([this]() { m_dirty = true; });
    }

BondData::~BondData()
    {
    m_pdata->getParticleSortSignal().disconnect(m_sort_slot);
    m_pdata->getMaxParticleNumberChangeSignal().disconnect(m_maxn_slot);
    }

void BondData::update()
    {
    if (!m_dirty)
        return;
    m_dirty = false;
    const unsigned int pitch = std::max(m_pdata->getMaxN(), 1u);
    m_blist_idx = hmd_index2d_t {pitch, m_tag_blist_idx.h};
    m_n_bonds.resize(pitch, false);
    m_blist.resize((size_t)pitch * m_blist_idx.h, false);
    m_flags.get()[0] = 0;
    hmd_bond_table_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.N = m_pdata->getN();
    a.n_total = m_pdata->getN() + m_pdata->getNGhosts();
    a.d_tag = m_pdata->tag();
    a.d_rtag = m_pdata->rtag();
    a.d_tag_n_bonds = m_tag_n_bonds.get();
    a.d_tag_blist = m_tag_blist.get();
    a.tag_blist_idx = m_tag_blist_idx;
    a.d_n_bonds = m_n_bonds.get();
    a.d_blist = m_blist.get();
    a.blist_idx = m_blist_idx;
    a.d_flags = m_flags.deviceView();
    abiCheck(hmd_update_bond_table(&a, nullptr), "hmd_update_bond_table");
    }

void BondData::checkComplete()
    {
    if (m_flags.get()[0])
        throw std::runtime_error("Error: a bond is incomplete: one of its particles is neither owned by this rank nor in its ghost layer (bonds longer "
                                 "than the ghost layer, r_cut + buffer of the neighbor list, cannot be followed)");
    }

// ---------------------------------------------------------------------------------------------
PotentialBondFENEGPU::PotentialBondFENEGPU(std::shared_ptr<SystemDefinition> sysdef)
    : ForceCompute(sysdef), m_bonds(sysdef->getBondData()), m_flags(1)
    {
    if (!m_bonds)
        throw std::runtime_error("bond.fene: the system defines no bonds or bond types");
    hmd_fene_params_t zero;
    std::memset(&zero, 0, sizeof(zero));
    m_params.assign(std::max(m_bonds->getNTypes(), 1u), zero);
    m_flags.get()[0] = 0;
    }

void PotentialBondFENEGPU::setParams(unsigned int type, Scalar k, Scalar r0, Scalar epsilon, Scalar sigma, Scalar delta)
    {
    if (type >= m_bonds->getNTypes())
        throw std::runtime_error("Invalid bond type specified");
    hmd_fene_params_t p;
    p.k = k;
    p.r_0 = r0;
    p.delta = delta;
    p.sigma_6 = sigma * sigma * sigma * sigma * sigma * sigma;
    p.epsilon_x_4 = Scalar(4.0) * epsilon;
    m_params[type] = p;
    m_params_dirty = true;
    }

std::array<Scalar, 5> PotentialBondFENEGPU::getParams(unsigned int type) const
    {
    if (type >= m_bonds->getNTypes())
        throw std::runtime_error("Invalid bond type specified");
    const auto& p = m_params[type];
    return {p.k, p.r_0, p.epsilon_x_4 / 4.0, std::pow(p.sigma_6, 1. / 6.), p.delta};
    }

void PotentialBondFENEGPU::computeForces(uint64_t timestep)
    {
    // a flag raised by an earlier launch is reported now; the last launches of a run by checkErrors()
    if (m_flags.get()[0])
        checkErrors();
    if (m_params_dirty)
        {
        m_d_params.resize(m_params.size(), false);
        hipCheck(hipMemcpy(m_d_params.get(), m_params.data(), sizeof(hmd_fene_params_t) * m_params.size(), hipMemcpyHostToDevice), "bond params H2D");
        m_params_dirty = false;
        }
    hmd_bond_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.d_blist = m_bonds->getGPUTable(); // brings the table up to date with the particle order
    a.d_n_bonds = m_bonds->getNBondsArray();
    a.blist_idx = m_bonds->getGPUTableIndexer();
    a.d_force = m_force.get();
    a.d_virial = m_virial.get();
    a.virial_pitch = m_virial_pitch;
    a.N = m_pdata->getN();
    a.d_pos = m_pdata->pos();
    a.box = m_pdata->getGlobalBox().b;
    a.n_bond_types = (unsigned int)m_params.size();
    a.d_flags = m_flags.deviceView();
    a.compute_virial = (m_pdata->getFlags() & (1u << pressure_tensor)) ? 1 : 0;
    ScopedKernelTimer t("bond_force");
    abiCheck(hmd_compute_bond_forces_fene(&a, m_d_params.get(), nullptr), "hmd_compute_bond_forces_fene");
    }

// hoomd/md/PotentialBondGPU.h:130-140
void PotentialBondFENEGPU::checkErrors()
    {
    m_bonds->checkComplete();
    if (m_flags.get()[0])
        {
        m_flags.get()[0] = 0;
        throw std::runtime_error("bond.fene: bond out of bounds\nError in bond calculation");
        }
    }
    } // namespace hoomd_amd
