// cell_list.cc -- CellListGPU host class: dimensions, memory, adjacency, overflow loop.
// Follows hoomd/CellList.cc:83-131 (dimensions), :133-200 (compute), :208-344 (memory),
// :346-402 (adjacency), :549-607 (conditions) and hoomd/CellListGPU.cc:39-232 (device path).
#include "hoomd_host.h"

#include <algorithm>
#include <cmath>

namespace hoomd_amd
    {
CellListGPU::CellListGPU(std::shared_ptr<SystemDefinition> sysdef) : Compute(sysdef), m_h_cond(4)
    {
    m_conditions.resize(1);
    m_sort_slot = m_pdata->getParticleSortSignal().connect([this]() { m_particles_sorted = true; });
    m_box_slot = m_pdata->getBoxChangeSignal().connect([this]() { m_box_changed = true; });
    }

CellListGPU::~CellListGPU()
    {
    m_pdata->getParticleSortSignal().disconnect(m_sort_slot);
    m_pdata->getBoxChangeSignal().disconnect(m_box_slot);
    }

// hoomd/CellList.cc:83-131
std::array<unsigned int, 3> CellListGPU::computeDimensions()
    {
    const BoxDim& box = m_pdata->getBox();
    const auto L = box.getNearestPlaneDistance();
    m_ghost_width = Scalar3 {0, 0, 0};
    if (m_sysdef->isDomainDecomposed())
        {
        const Scalar gw = m_sysdef->getCommunicator()->getGhostLayerMaxWidth();
        if (gw > 0)
            {
            if (!box.b.periodic[0])
                m_ghost_width.x = gw;
            if (!box.b.periodic[1])
                m_ghost_width.y = gw;
            if (m_sysdef->getNDimensions() == 3 && !box.b.periodic[2])
                m_ghost_width.z = gw;
            }
        }
    auto round_down = [this](unsigned int v) { return (v / m_multiple) * m_multiple; };
    std::array<unsigned int, 3> dim;
    dim[0] = round_down((unsigned int)((L[0] + 2.0 * m_ghost_width.x) / m_nominal_width));
    dim[1] = round_down((unsigned int)((L[1] + 2.0 * m_ghost_width.y) / m_nominal_width));
    dim[2] = (m_sysdef->getNDimensions() == 3) ? round_down((unsigned int)((L[2] + 2.0 * m_ghost_width.z) / m_nominal_width)) : 1;
    for (auto& d : dim)
        if (d == 0)
            d = 1;
    return dim;
    }

void CellListGPU::initializeAll()
    {
    initializeWidth();
    initializeMemory();
    }

void CellListGPU::initializeWidth()
    {
    m_dim = computeDimensions();
    }

// hoomd/CellList.cc:228-344
void CellListGPU::initializeMemory()
    {
    const unsigned int ncells = m_dim[0] * m_dim[1] * m_dim[2];
    if (m_Nmax == 0)
        {
        m_Nmax = (unsigned int)std::ceil(double(m_pdata->getN() + m_pdata->getNGhosts()) / double(ncells));
        if (m_Nmax == 0)
            m_Nmax = 1;
        }
    m_cell_size.resize(ncells, false);
    unsigned int u[3];
    for (int d = 0; d < 3; d++)
        u[d] = m_dim[d] > m_radius * 2 + 1 ? m_radius * 2 + 1 : m_dim[d];
    m_n_adj = (m_sysdef->getNDimensions() == 2) ? u[0] * u[1] : u[0] * u[1] * u[2];
    m_cell_adj.resize(m_compute_adj ? (size_t)m_n_adj * ncells : 0, false);
    m_xyzf.resize((size_t)m_Nmax * ncells, false);
    if (m_compute_type_body)
        m_type_body.resize((size_t)m_Nmax * ncells, false);
    else
        m_type_body.resize(0, false);
    if (m_compute_idx)
        m_idx.resize((size_t)m_Nmax * ncells, false);
    else
        m_idx.resize(0, false);
    if (m_compute_adj)
        initializeCellAdj();
    }

// hoomd/CellList.cc:346-402 : built on the host once per geometry change, then uploaded
void CellListGPU::initializeCellAdj()
    {
    const int mx = (int)m_dim[0], my = (int)m_dim[1], mz = (int)m_dim[2];
    const int r = (int)m_radius;
    const int rk = (m_sysdef->getNDimensions() == 2) ? 0 : r;
    std::vector<unsigned int> h_adj((size_t)m_n_adj * mx * my * mz);
    std::vector<unsigned int> adj;
    for (int k = 0; k < mz; k++)
        for (int j = 0; j < my; j++)
            for (int i = 0; i < mx; i++)
                {
                const unsigned int cur = (unsigned int)((k * my + j) * mx + i);
                adj.clear();
                for (int nk = k - rk; nk <= k + rk; nk++)
                    for (int nj = j - r; nj <= j + r; nj++)
                        for (int ni = i - r; ni <= i + r; ni++)
                            {
                            const int wi = ((ni % mx) + mx) % mx, wj = ((nj % my) + my) % my, wk = ((nk % mz) + mz) % mz;
                            adj.push_back((unsigned int)((wk * my + wj) * mx + wi));
                            }
                std::sort(adj.begin(), adj.end());
                adj.erase(std::unique(adj.begin(), adj.end()), adj.end());
                std::copy(adj.begin(), adj.end(), h_adj.begin() + (size_t)m_n_adj * cur);
                }
    m_cell_adj.upload(h_adj.data(), h_adj.size());
    }

void CellListGPU::computeCellList()
    {
    ScopedKernelTimer t("cell_list");
    hmd_cell_list_args_t a;
    std::memset(&a, 0, sizeof(a));
    a.d_cell_size = m_cell_size.get();
    a.d_xyzf = m_xyzf.get();
    a.d_type_body = m_compute_type_body ? m_type_body.get() : nullptr;
    a.d_cell_idx = m_compute_idx ? m_idx.get() : nullptr;
    a.d_conditions = m_conditions.get();
    a.d_pos = m_pdata->pos();
    a.d_charge = m_pdata->charge();
    a.d_diameter = m_pdata->diameter();
    a.d_body = m_pdata->body();
    a.N = m_pdata->getN();
    a.n_ghost = m_pdata->getNGhosts();
    a.Nmax = m_Nmax;
    a.flag_charge = m_flag_charge;
    a.flag_type = m_flag_type;
    a.box = m_pdata->getBox().b;
    a.ci = getCellIndexer();
    a.cli = getCellListIndexer();
    a.ghost_width = m_ghost_width;
    a.clear_cell_size = 1;
    abiCheck(hmd_compute_cell_list(&a, nullptr), "hmd_compute_cell_list");
    if (m_sort)
        {
        hmd_sort_cell_list_args_t s;
        std::memset(&s, 0, sizeof(s));
        s.d_cell_size = m_cell_size.get();
        s.d_xyzf = m_xyzf.get();
        s.d_type_body = a.d_type_body;
        s.d_cell_idx = a.d_cell_idx;
        s.ci = a.ci;
        s.cli = a.cli;
        s.flag_is_index = (!m_flag_charge && !m_flag_type) ? 1 : 0;
        if (s.flag_is_index || s.d_cell_idx)
            abiCheck(hmd_sort_cell_list(&s, nullptr), "hmd_sort_cell_list");
        }
    }

// hoomd/CellList.cc:549-607
bool CellListGPU::checkConditions()
    {
    unsigned int h[3];
    m_readback.read(reinterpret_cast<const unsigned int*>(m_conditions.get()), 3, h, "cell list conditions");
    return evaluateConditions(h);
    }

// The check that compute() left to the consumer (setDeferConditions): true when the list had overflowed.  It has then
// been computed again, with the check, so what the consumer built from the truncated list must be built again.
bool CellListGPU::finishConditions()
    {
    if (!m_conditions_pending)
        return false;
    m_conditions_pending = false;
    unsigned int h[3];
    m_readback.finish(h, "cell list conditions");
    bool overflowed = evaluateConditions(h);
    const bool any = overflowed;
    while (overflowed)
        {
        initializeAll();
        m_conditions.zero();
        computeCellList();
        overflowed = checkConditions();
        }
    return any;
    }

bool CellListGPU::evaluateConditions(const unsigned int (&h)[3])
    {
    bool result = false;
    if (h[0] > m_Nmax)
        {
        m_Nmax = h[0] + m_nmax_headroom;
        result = true;
        }
    if (h[1])
        {
        const unsigned int n = h[1] - 1;
        unsigned int tag = 0;
        hipCheck(hipMemcpy(&tag, m_pdata->tag() + n, sizeof(unsigned int), hipMemcpyDeviceToHost), "tag D2H");
        std::ostringstream s;
        s << "Particle with unique tag " << tag << " has NaN for its position.";
        throw std::runtime_error(s.str());
        }
    if (h[2])
        {
        const unsigned int n = h[2] - 1;
        unsigned int tag = 0;
        Scalar4 p;
        hipCheck(hipMemcpy(&tag, m_pdata->tag() + n, sizeof(unsigned int), hipMemcpyDeviceToHost), "tag D2H");
        hipCheck(hipMemcpy(&p, m_pdata->pos() + n, sizeof(Scalar4), hipMemcpyDeviceToHost), "pos D2H");
        const hmd_box_t& b = m_pdata->getBox().b;
        std::ostringstream s;
        s << "Particle with unique tag " << tag << " is no longer in the simulation box.\n\n"
          << "Cartesian coordinates: \nx: " << p.x << " y: " << p.y << " z: " << p.z << "\n"
          << "Local box lo: (" << b.lo[0] << ", " << b.lo[1] << ", " << b.lo[2] << ")\n"
          << "          hi: (" << b.hi[0] << ", " << b.hi[1] << ", " << b.hi[2] << ")";
        throw std::runtime_error(s.str());
        }
    return result;
    }

// hoomd/CellList.cc:133-200
void CellListGPU::compute(uint64_t timestep)
    {
    bool force = false;
    if (m_nominal_width == 0)
        throw std::runtime_error("Cell: cell width must be non-zero");
    if (m_params_changed)
        {
        initializeAll();
        m_params_changed = false;
        force = true;
        }
    if (m_box_changed)
        {
        const auto new_dim = computeDimensions();
        if (new_dim == m_dim)
            initializeWidth();
        else
            initializeAll();
        m_box_changed = false;
        force = true;
        }
    if (m_particles_sorted)
        {
        m_particles_sorted = false;
        force = true;
        }
    if (shouldCompute(timestep) || force)
        {
        if (m_defer_conditions)
            {
            if (m_conditions_pending) // (nobody asked about the previous computation)
                finishConditions();
            computeCellList();
            m_readback.begin(reinterpret_cast<const unsigned int*>(m_conditions.get()), 3);
            m_conditions_pending = true;
            return;
            }
        bool overflowed = false;
        do
            {
            computeCellList();
            overflowed = checkConditions();
            if (overflowed)
                {
                initializeAll();
                m_conditions.zero();
                }
            } while (overflowed);
        }
    }

    } // namespace hoomd_amd
