// rccl_transport.cc -- RcclTransport: the message-passing layer of CommunicatorGPU on RCCL itself.
//
// The reference's communicator issues its sends and receives from the host class (hoomd/CommunicatorGPU.cc:1924-2012
// migration, :2570-2814 ghost exchange, :3068-3160 ghost update: MPI_Isend / MPI_Irecv per neighbor, host-staged).
// Here the same calls are ncclSend / ncclRecv inside one ncclGroupStart / ncclGroupEnd per exchange, queued by this C++
// class on the stream the pack and unpack kernels run on: device buffers travel GPU to GPU over xGMI, no Python and no
// second stream in between.  The rebuild vote is a one-word ncclAllReduce(max) on the same stream.
//
// RCCL is not linked: the process already holds a copy (PyTorch loads its own librccl), and two copies in one process
// would each bring up their own transports.  The library that is already loaded is looked up by name (RTLD_NOLOAD);
// only when there is none is the system's librccl opened.  The communicator is created from a ncclUniqueId that rank 0
// makes and hands to the others over whatever channel the caller has (the existing process group).
#include "hoomd_host.h"

#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstring>
#include <sstream>

namespace hoomd_amd
    {
namespace
    {
struct RcclApi
    {
    void* handle = nullptr;
    std::string origin;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    };

template<class F> void resolve(RcclApi& api, F& fn, const char* name)
    {
    fn = reinterpret_cast<F>(dlsym(api.handle, name));
    if (!fn)
        throw std::runtime_error(std::string("RCCL: symbol ") + name + " not found in " + api.origin);
    }

RcclApi& rccl()
    {
    static RcclApi api;
    if (api.handle)
        return api;
    // the copy the process already holds first (PyTorch's), then the system's
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names)
        if (!api.handle && (api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD)))
            api.origin = std::string(n) + " (already loaded in this process)";
    for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if (!api.handle && (api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL)))
            api.origin = std::string(n) + " (opened by hoomd_amd)";
    if (!api.handle)
        throw std::runtime_error("RCCL: librccl.so is neither loaded in this process nor found by the dynamic loader");
    resolve(api, api.GetUniqueId, "ncclGetUniqueId");
    resolve(api, api.CommInitRank, "ncclCommInitRank");
    resolve(api, api.CommDestroy, "ncclCommDestroy");
    resolve(api, api.GroupStart, "ncclGroupStart");
    resolve(api, api.GroupEnd, "ncclGroupEnd");
    resolve(api, api.Send, "ncclSend");
    resolve(api, api.Recv, "ncclRecv");
    resolve(api, api.AllReduce, "ncclAllReduce");
    resolve(api, api.GetErrorString, "ncclGetErrorString");
    return api;
    }

void ncclCheck(ncclResult_t r, const char* what)
    {
    if (r != ncclSuccess)
        {
        std::ostringstream s;
        s << "RCCL: " << what << " failed: " << rccl().GetErrorString(r);
        throw std::runtime_error(s.str());
        }
    }
    } // namespace

std::string RcclTransport::uniqueId()
    {
    ncclUniqueId id;
    ncclCheck(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    return std::string(id.internal, NCCL_UNIQUE_ID_BYTES);
    }

std::string RcclTransport::libraryOrigin()
    {
    return rccl().origin;
    }

RcclTransport::RcclTransport(const std::string& unique_id, unsigned int rank, unsigned int world) : m_rank(rank), m_world(world)
    {
    if (unique_id.size() != NCCL_UNIQUE_ID_BYTES)
        throw std::invalid_argument("RcclTransport: the unique id must have 128 bytes (RcclTransport.unique_id() of rank 0)");
    if (world == 0 || rank >= world)
        throw std::invalid_argument("RcclTransport: rank must be below the number of ranks");
    ncclUniqueId id;
    std::memcpy(id.internal, unique_id.data(), NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    ncclCheck(rccl().CommInitRank(&comm, (int)world, id, (int)rank), "ncclCommInitRank");
    m_comm = comm;
    m_words.resize(64, false);
    }

RcclTransport::~RcclTransport()
    {
    if (m_comm)
        (void)rccl().CommDestroy(static_cast<ncclComm_t>(m_comm));
    }

// hoomd/CommunicatorGPU.cc:1986-2012 / 2570-2814 / 3068-3160: one send and one receive per peer, all of them in one
// group so that RCCL progresses them together; this rank's own rows (a sub-box that is its own periodic neighbor along
// an axis the grid does not split) are a device-to-device copy on the same stream.
void RcclTransport::allToAll(uintptr_t d_send, const std::vector<unsigned int>& send_rows, uintptr_t d_recv, const std::vector<unsigned int>& recv_rows,
                             unsigned int row_bytes, uintptr_t stream)
    {
    if (send_rows.size() != m_world || recv_rows.size() != m_world)
        throw std::invalid_argument("RcclTransport::allToAll: one row count per rank is required");
    if (send_rows[m_rank] != recv_rows[m_rank])
        throw std::invalid_argument("RcclTransport::allToAll: a rank sends itself as many rows as it receives from itself");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ncclComm_t comm = static_cast<ncclComm_t>(m_comm);
    const char* out = reinterpret_cast<const char*>(d_send);
    char* in = reinterpret_cast<char*>(d_recv);
    size_t so = 0, ro = 0, self_so = 0, self_ro = 0;
    bool any = false;
    for (unsigned int r = 0; r < m_world; r++)
        if ((r != m_rank || self_through_rccl) && (send_rows[r] || recv_rows[r]))
            any = true;
    if (any)
        ncclCheck(rccl().GroupStart(), "ncclGroupStart");
    for (unsigned int r = 0; r < m_world; r++)
        {
        const size_t sb = (size_t)send_rows[r] * row_bytes, rb = (size_t)recv_rows[r] * row_bytes;
        if (r == m_rank && !self_through_rccl)
            {
            self_so = so;
            self_ro = ro;
            }
        else
            {
            if (sb)
                ncclCheck(rccl().Send(out + so, sb, ncclUint8, (int)r, comm, s), "ncclSend");
            if (rb)
                ncclCheck(rccl().Recv(in + ro, rb, ncclUint8, (int)r, comm, s), "ncclRecv");
            }
        so += sb;
        ro += rb;
        }
    if (any)
        ncclCheck(rccl().GroupEnd(), "ncclGroupEnd");
    const size_t self_bytes = (size_t)send_rows[m_rank] * row_bytes;
    if (self_bytes && !self_through_rccl)
        hipCheck(hipMemcpyAsync(in + self_ro, out + self_so, self_bytes, hipMemcpyDeviceToDevice, s), "copy of the rank's own rows");
    n_exchanges++;
    }

// hoomd/md/NeighborListGPU.cc:84-98 (MPI_Allreduce of the check result), here on the device word itself
void RcclTransport::allReduceMaxWord(uintptr_t d_word, uintptr_t stream)
    {
    if (m_world == 1)
        return;
    void* w = reinterpret_cast<void*>(d_word);
    ncclCheck(rccl().AllReduce(w, w, 1, ncclUint32, ncclMax, static_cast<ncclComm_t>(m_comm), reinterpret_cast<hipStream_t>(stream)), "ncclAllReduce");
    }

bool RcclTransport::anyRank(bool local)
    {
    if (m_world == 1)
        return local;
    unsigned int v = local ? 1u : 0u;
    unsigned int* d = reinterpret_cast<unsigned int*>(m_words.get());
    hipCheck(hipMemcpyAsync(d, &v, sizeof(v), hipMemcpyHostToDevice, nullptr), "vote H2D");
    ncclCheck(rccl().AllReduce(d, d, 1, ncclUint32, ncclMax, static_cast<ncclComm_t>(m_comm), nullptr), "ncclAllReduce");
    hipCheck(hipMemcpyAsync(&v, d, sizeof(v), hipMemcpyDeviceToHost, nullptr), "vote D2H");
    hipCheck(hipStreamSynchronize(nullptr), "vote");
    return v != 0u;
    }

// hoomd/ForceCompute.cc:218 (MPI_Allreduce of the energy), hoomd/md/PotentialPair.h:430 (of the type counts)
std::vector<double> RcclTransport::allReduceSumHost(const std::vector<double>& v)
    {
    if (m_world == 1 || v.empty())
        return v;
    if (m_words.size() < v.size())
        m_words.resize(v.size(), false);
    double* d = m_words.get();
    std::vector<double> out(v.size());
    hipCheck(hipMemcpyAsync(d, v.data(), sizeof(double) * v.size(), hipMemcpyHostToDevice, nullptr), "sum H2D");
    ncclCheck(rccl().AllReduce(d, d, v.size(), ncclFloat64, ncclSum, static_cast<ncclComm_t>(m_comm), nullptr), "ncclAllReduce");
    hipCheck(hipMemcpyAsync(out.data(), d, sizeof(double) * v.size(), hipMemcpyDeviceToHost, nullptr), "sum D2H");
    hipCheck(hipStreamSynchronize(nullptr), "sum");
    return out;
    }

    } // namespace hoomd_amd
