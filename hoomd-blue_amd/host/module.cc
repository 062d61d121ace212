// module.cc -- pybind11 module `_hoomd_amd`: the surface the Python wrappers (hoomd_amd/*.py) call.
// Mirrors the exports the reference's wrappers rely on: hoomd/md/NeighborList.cc:1990-2021,
// hoomd/md/NeighborListGPUBinned.cc:184-197, hoomd/md/PotentialPair.h:1017-1036,
// hoomd/ForceCompute.cc:628-643, hoomd/md/TwoStepConstantVolume.cc:318-330,
// hoomd/md/IntegratorTwoStep.cc:421-440, example_plugins/pair_plugin/module.cc:20-26.
#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "hoomd_host.h"

#include "hoomd/md/EvaluatorPairLJ.h"
#include "hoomd/md/PotentialPairGPU.h"

namespace py = pybind11;
using namespace hoomd_amd;

namespace
    {
template<class T, size_t K> std::vector<std::array<T, K>> rows(py::array_t<T, py::array::c_style | py::array::forcecast> a, size_t n, const char* what)
    {
    if (a.ndim() != 2 || (size_t)a.shape(0) != n || (size_t)a.shape(1) != K)
        throw std::invalid_argument(std::string(what) + " has the wrong shape");
    std::vector<std::array<T, K>> out(n);
    std::memcpy(out.data(), a.data(), sizeof(T) * K * n);
    return out;
    }

template<class T> std::vector<T> col(py::array_t<T, py::array::c_style | py::array::forcecast> a, size_t n, const char* what)
    {
    if (a.ndim() != 1 || (size_t)a.shape(0) != n)
        throw std::invalid_argument(std::string(what) + " has the wrong shape");
    return std::vector<T>(a.data(), a.data() + n);
    }

SnapshotParticleData make_snapshot(py::dict d)
    {
    SnapshotParticleData s;
    const unsigned int n = d["N"].cast<unsigned int>();
    s.resize(n);
    s.type_mapping = d["types"].cast<std::vector<std::string>>();
    if (d.contains("position"))
        s.pos = rows<double, 3>(d["position"].cast<py::array_t<double, py::array::c_style | py::array::forcecast>>(), n, "position");
    if (d.contains("velocity"))
        s.vel = rows<double, 3>(d["velocity"].cast<py::array_t<double, py::array::c_style | py::array::forcecast>>(), n, "velocity");
    if (d.contains("acceleration"))
        s.accel = rows<double, 3>(d["acceleration"].cast<py::array_t<double, py::array::c_style | py::array::forcecast>>(), n, "acceleration");
    if (d.contains("image"))
        s.image = rows<int, 3>(d["image"].cast<py::array_t<int, py::array::c_style | py::array::forcecast>>(), n, "image");
    if (d.contains("typeid"))
        s.type = col<unsigned int>(d["typeid"].cast<py::array_t<unsigned int, py::array::c_style | py::array::forcecast>>(), n, "typeid");
    if (d.contains("body"))
        s.body = col<unsigned int>(d["body"].cast<py::array_t<unsigned int, py::array::c_style | py::array::forcecast>>(), n, "body");
    if (d.contains("mass"))
        s.mass = col<double>(d["mass"].cast<py::array_t<double, py::array::c_style | py::array::forcecast>>(), n, "mass");
    if (d.contains("charge"))
        s.charge = col<double>(d["charge"].cast<py::array_t<double, py::array::c_style | py::array::forcecast>>(), n, "charge");
    if (d.contains("diameter"))
        s.diameter = col<double>(d["diameter"].cast<py::array_t<double, py::array::c_style | py::array::forcecast>>(), n, "diameter");
    if (d.contains("tag"))
        s.tag = col<unsigned int>(d["tag"].cast<py::array_t<unsigned int, py::array::c_style | py::array::forcecast>>(), n, "tag");
    return s;
    }

template<class T, size_t K> py::array_t<T> to_np(const std::vector<std::array<T, K>>& v)
    {
    py::array_t<T> a({v.size(), K});
    if (!v.empty())
        std::memcpy(a.mutable_data(), v.data(), sizeof(T) * K * v.size());
    return a;
    }

template<class T> py::array_t<T> to_np1(const std::vector<T>& v)
    {
    py::array_t<T> a(v.size());
    if (!v.empty())
        std::memcpy(a.mutable_data(), v.data(), sizeof(T) * v.size());
    return a;
    }

py::dict take_snapshot(const ParticleData& pdata)
    {
    SnapshotParticleData s;
    pdata.takeSnapshot(s);
    py::dict d;
    d["N"] = s.size;
    d["types"] = s.type_mapping;
    d["position"] = to_np(s.pos);
    d["velocity"] = to_np(s.vel);
    d["acceleration"] = to_np(s.accel);
    d["image"] = to_np(s.image);
    d["typeid"] = to_np1(s.type);
    d["body"] = to_np1(s.body);
    d["mass"] = to_np1(s.mass);
    d["charge"] = to_np1(s.charge);
    d["diameter"] = to_np1(s.diameter);
    d["tag"] = to_np1(s.tag);
    return d;
    }

//! trampoline: the message-passing layer lives in Python (torch.distributed: RCCL in production, gloo in tests)
class PyExchangeTransport : public ExchangeTransport
    {
    public:
    using ExchangeTransport::ExchangeTransport;
    void allToAll(uintptr_t d_send, const std::vector<unsigned int>& send_rows, uintptr_t d_recv, const std::vector<unsigned int>& recv_rows,
                  unsigned int row_bytes, uintptr_t stream) override
        {
        PYBIND11_OVERRIDE_PURE(void, ExchangeTransport, allToAll, d_send, send_rows, d_recv, recv_rows, row_bytes, stream);
        }
    void allReduceMaxWord(uintptr_t d_word, uintptr_t stream) override
        {
        PYBIND11_OVERRIDE_PURE(void, ExchangeTransport, allReduceMaxWord, d_word, stream);
        }
    bool anyRank(bool local) override
        {
        PYBIND11_OVERRIDE_PURE(bool, ExchangeTransport, anyRank, local);
        }
    std::vector<double> allReduceSumHost(const std::vector<double>& v) override
        {
        PYBIND11_OVERRIDE(std::vector<double>, ExchangeTransport, allReduceSumHost, v);
        }
    };
    } // namespace

PYBIND11_MODULE(_hoomd_amd, m)
    {
    m.doc() = "C++ host layer of the MI355X-native HOOMD-blue short-range MD backend";

    m.def("device_count", []()
          {
          int n = 0;
          if (hipGetDeviceCount(&n) != hipSuccess)
              n = 0;
          return n;
          });
    m.def("set_device", [](int d) { hipCheck(hipSetDevice(d), "hipSetDevice"); });
    m.def("device_name", [](int d)
          {
          hipDeviceProp_t p;
          hipCheck(hipGetDeviceProperties(&p, d), "hipGetDeviceProperties");
          return std::string(p.name) + " (" + p.gcnArchName + ")";
          });
    m.def("synchronize", []() { hipCheck(hipDeviceSynchronize(), "hipDeviceSynchronize"); });
    m.def("poll_wait_seconds", &pollWaitSeconds, py::arg("reset") = false);
    m.def("abi_version", []() { return hmd_abi_version(); });

    m.def("enable_kernel_timing", [](bool on, const std::string& only, double min_us, unsigned int every)
          {
          KernelTimer::get().setFilter(only);
          KernelTimer::get().setMinMs(min_us * 1e-3);
          KernelTimer::get().setEvery(every);
          KernelTimer::get().enable(on);
          },
          py::arg("on"), py::arg("only") = "", py::arg("min_us") = 0.0, py::arg("every") = 1);
    m.def("reset_kernel_timing", []() { KernelTimer::get().reset(); });
    m.def("kernel_timing", []()
          {
          py::dict d;
          for (auto& r : KernelTimer::get().collect())
              d[py::str(std::get<0>(r))] = py::make_tuple(std::get<1>(r), std::get<2>(r));
          return d;
          });

    py::class_<BoxDim>(m, "BoxDim")
        .def(py::init<Scalar, Scalar, Scalar, Scalar, Scalar, Scalar>(), py::arg("Lx"), py::arg("Ly"), py::arg("Lz"), py::arg("xy") = 0.0,
             py::arg("xz") = 0.0, py::arg("yz") = 0.0)
        .def("setLoHi", [](BoxDim& b, std::array<Scalar, 3> lo, std::array<Scalar, 3> hi) { b.setLoHi(lo.data(), hi.data()); })
        .def("setPeriodic", &BoxDim::setPeriodic)
        .def_property_readonly("L", [](const BoxDim& b) { return std::array<Scalar, 3> {b.b.L[0], b.b.L[1], b.b.L[2]}; })
        .def_property_readonly("lo", [](const BoxDim& b) { return std::array<Scalar, 3> {b.b.lo[0], b.b.lo[1], b.b.lo[2]}; })
        .def_property_readonly("hi", [](const BoxDim& b) { return std::array<Scalar, 3> {b.b.hi[0], b.b.hi[1], b.b.hi[2]}; })
        .def_property_readonly("tilt", [](const BoxDim& b) { return std::array<Scalar, 3> {b.b.xy, b.b.xz, b.b.yz}; })
        .def_property_readonly("periodic", [](const BoxDim& b) { return std::array<bool, 3> {(bool)b.b.periodic[0], (bool)b.b.periodic[1], (bool)b.b.periodic[2]}; })
        .def("getNearestPlaneDistance", &BoxDim::getNearestPlaneDistance)
        .def("getVolume", &BoxDim::getVolume, py::arg("twod") = false)
        .def("address", [](BoxDim& b) { return (uintptr_t)&b.b; });

    py::class_<ParticleData, std::shared_ptr<ParticleData>>(m, "ParticleData")
        .def_property_readonly("N", &ParticleData::getN)
        .def_property_readonly("n_ghosts", &ParticleData::getNGhosts)
        .def_property("N_global", &ParticleData::getNGlobal, &ParticleData::setNGlobal)
        .def_property_readonly("max_n", &ParticleData::getMaxN)
        .def_property_readonly("types", &ParticleData::getTypeMapping)
        .def("getTypeByName", &ParticleData::getTypeByName)
        .def("getBox", &ParticleData::getBox)
        .def("getGlobalBox", &ParticleData::getGlobalBox)
        .def("setBoxes", &ParticleData::setBoxes)
        .def("resizeOwned", &ParticleData::resizeOwned)
        .def("setNGhosts", &ParticleData::setNGhosts)
        .def("takeSnapshot", &take_snapshot)
        .def("initializeFromSnapshot", [](ParticleData& p, py::dict d) { p.initializeFromSnapshot(make_snapshot(d)); })
        .def("applyOrder", [](ParticleData& p, uintptr_t d_order) { p.applyOrder((const unsigned int*)d_order); })
        .def("notifyParticleSort", &ParticleData::notifyParticleSort)
        .def("getFlags", &ParticleData::getFlags)
        .def("setFlags", &ParticleData::setFlags)
        .def_property_readonly("sort_count", &ParticleData::getSortCount)
        // raw device addresses for the Python-level halo exchange (wrapped as torch tensors there)
        .def_property_readonly("pos_ptr", [](ParticleData& p) { return (uintptr_t)p.pos(); })
        .def_property_readonly("vel_ptr", [](ParticleData& p) { return (uintptr_t)p.vel(); })
        .def_property_readonly("accel_ptr", [](ParticleData& p) { return (uintptr_t)p.accel(); })
        .def_property_readonly("image_ptr", [](ParticleData& p) { return (uintptr_t)p.image(); })
        .def_property_readonly("tag_ptr", [](ParticleData& p) { return (uintptr_t)p.tag(); })
        .def_property_readonly("rtag_ptr", [](ParticleData& p) { return (uintptr_t)p.rtag(); })
        .def_property_readonly("body_ptr", [](ParticleData& p) { return (uintptr_t)p.body(); })
        .def_property_readonly("charge_ptr", [](ParticleData& p) { return (uintptr_t)p.charge(); })
        .def_property_readonly("diameter_ptr", [](ParticleData& p) { return (uintptr_t)p.diameter(); })
        .def_property_readonly("net_force_ptr", [](ParticleData& p) { return (uintptr_t)p.netForceForIntegration(); })
        .def("rebuildRTag", [](ParticleData& p) { abiCheck(hmd_build_rtag(p.rtag(), p.tag(), p.getN(), nullptr), "hmd_build_rtag"); });

    py::class_<Communicator, std::shared_ptr<Communicator>>(m, "Communicator")
        .def("communicate", &Communicator::communicate)
        .def("forceMigrate", &Communicator::forceMigrate)
        .def("getGhostLayerMaxWidth", &Communicator::getGhostLayerMaxWidth)
        .def("anyRank", &Communicator::anyRank)
        .def("beginRebuildVote", &Communicator::beginRebuildVote)
        .def("endRebuildVote", &Communicator::endRebuildVote);
    py::class_<ExchangeTransport, PyExchangeTransport, std::shared_ptr<ExchangeTransport>>(m, "ExchangeTransport").def(py::init<>());
    py::class_<RcclTransport, ExchangeTransport, std::shared_ptr<RcclTransport>>(m, "RcclTransport", py::dynamic_attr())
        .def(py::init([](py::bytes id, unsigned int rank, unsigned int world) { return std::make_shared<RcclTransport>(std::string(id), rank, world); }))
        .def_static("unique_id", []() { return py::bytes(RcclTransport::uniqueId()); })
        .def_static("library_origin", &RcclTransport::libraryOrigin)
        .def("allToAll", &RcclTransport::allToAll)
        .def("allReduceMaxWord", &RcclTransport::allReduceMaxWord)
        .def("anyRank", &RcclTransport::anyRank)
        .def("allReduceSumHost", &RcclTransport::allReduceSumHost)
        .def_readonly("n_exchanges", &RcclTransport::n_exchanges)
        .def_readwrite("self_through_rccl", &RcclTransport::self_through_rccl);
    py::class_<CommunicatorGPU, Communicator, std::shared_ptr<CommunicatorGPU>>(m, "CommunicatorGPU")
        .def(py::init<std::shared_ptr<SystemDefinition>, std::shared_ptr<ExchangeTransport>, unsigned int, unsigned int, std::array<unsigned int, 3>,
                      std::array<double, 3>, std::array<double, 3>>())
        .def("addNeighborList", &CommunicatorGPU::addNeighborList)
        .def("migrateParticles", &CommunicatorGPU::migrateParticles)
        .def("exchangeGhosts", &CommunicatorGPU::exchangeGhosts)
        .def("setUseVoteRows", &CommunicatorGPU::setUseVoteRows)
        .def("setFixedMigrate", &CommunicatorGPU::setFixedMigrate)
        .def("setMigrateSlotRows", &CommunicatorGPU::setMigrateSlotRows)
        .def("setUseDeviceCounts", &CommunicatorGPU::setUseDeviceCounts)
        .def_readonly("n_device_count_exchanges", &CommunicatorGPU::n_device_count_exchanges)
        .def("setUseGhostSlots", &CommunicatorGPU::setUseGhostSlots)
        .def("forgetGhostRows", &CommunicatorGPU::forgetGhostRows)
        .def("setGhostSlotRoom", &CommunicatorGPU::setGhostSlotRoom)
        .def_readonly("n_ghost_slot_overflows", &CommunicatorGPU::n_ghost_slot_overflows)
        .def_property_readonly("fixed_migrate", &CommunicatorGPU::usesFixedMigrate)
        .def_property_readonly("migrate_slot_rows", &CommunicatorGPU::getMigrateSlotRows)
        .def_readonly("n_migrated_out", &CommunicatorGPU::n_migrated_out)
        .def_readonly("n_migrated_in", &CommunicatorGPU::n_migrated_in)
        .def_readonly("n_vote_exchanges", &CommunicatorGPU::n_vote_exchanges)
        .def_readonly("n_count_exchanges", &CommunicatorGPU::n_count_exchanges);

    py::class_<SystemDefinition, std::shared_ptr<SystemDefinition>>(m, "SystemDefinition")
        .def(py::init([](py::dict snap, const BoxDim& box, unsigned int ndim, unsigned int seed, unsigned int n_global)
                      {
                      auto sd = std::make_shared<SystemDefinition>(make_snapshot(snap), box, ndim, seed, n_global);
                      return sd;
                      }),
             py::arg("snapshot"), py::arg("box"), py::arg("ndim") = 3, py::arg("seed") = 0, py::arg("n_global") = 0)
        .def("getParticleData", &SystemDefinition::getParticleData)
        .def("getNDimensions", &SystemDefinition::getNDimensions)
        .def("getSeed", &SystemDefinition::getSeed)
        .def("isDomainDecomposed", &SystemDefinition::isDomainDecomposed)
        .def("setCommunicator", &SystemDefinition::setCommunicator)
        .def("getCommunicator", &SystemDefinition::getCommunicator)
        .def("setBonds", [](SystemDefinition& sd, py::array_t<unsigned int, py::array::c_style | py::array::forcecast> group,
                            py::array_t<unsigned int, py::array::c_style | py::array::forcecast> typeid_, std::vector<std::string> types)
             {
             SnapshotBondData b;
             const size_t n = (size_t)typeid_.size();
             if ((size_t)group.size() != 2 * n)
                 throw std::runtime_error("bonds.group must have shape (N, 2)");
             b.groups.resize(n);
             b.type_id.resize(n);
             for (size_t i = 0; i < n; i++)
                 {
                 b.groups[i] = {group.data()[2 * i], group.data()[2 * i + 1]};
                 b.type_id[i] = typeid_.data()[i];
                 }
             b.type_mapping = types;
             sd.setBondData(b);
             })
        .def("getBonds", [](SystemDefinition& sd)
             {
             py::dict d;
             auto bd = sd.getBondData();
             const size_t n = bd ? bd->getNGlobal() : 0;
             py::array_t<unsigned int> group({n, (size_t)2});
             py::array_t<unsigned int> tid(n);
             for (size_t i = 0; i < n; i++)
                 {
                 group.mutable_data()[2 * i] = bd->getSnapshot().groups[i][0];
                 group.mutable_data()[2 * i + 1] = bd->getSnapshot().groups[i][1];
                 tid.mutable_data()[i] = bd->getSnapshot().type_id[i];
                 }
             d["N"] = n;
             d["group"] = group;
             d["typeid"] = tid;
             d["types"] = bd ? bd->getSnapshot().type_mapping : std::vector<std::string>();
             return d;
             });

    py::class_<ParticleGroup, std::shared_ptr<ParticleGroup>>(m, "ParticleGroup")
        .def(py::init<std::shared_ptr<SystemDefinition>>())
        .def(py::init<std::shared_ptr<SystemDefinition>, const std::vector<unsigned int>&>())
        .def("getNumMembersGlobal", &ParticleGroup::getNumMembersGlobal)
        .def("getNumMembers", &ParticleGroup::getNumMembers)
        .def("getTranslationalDOF", &ParticleGroup::getTranslationalDOF)
        .def("setTranslationalDOF", &ParticleGroup::setTranslationalDOF);

    py::class_<CellListGPU, std::shared_ptr<CellListGPU>>(m, "CellListGPU")
        .def(py::init<std::shared_ptr<SystemDefinition>>())
        .def("setNominalWidth", &CellListGPU::setNominalWidth)
        .def("setRadius", &CellListGPU::setRadius)
        .def("setComputeTypeBody", &CellListGPU::setComputeTypeBody)
        .def("setComputeIdx", &CellListGPU::setComputeIdx)
        .def("setFlagIndex", &CellListGPU::setFlagIndex)
        .def("setFlagCharge", &CellListGPU::setFlagCharge)
        .def("setFlagType", &CellListGPU::setFlagType)
        .def("setSortCellList", &CellListGPU::setSortCellList)
        .def("compute", &CellListGPU::compute)
        .def("getDim", &CellListGPU::getDim)
        .def("getNmax", &CellListGPU::getNmax)
        .def("getCellSizes", [](CellListGPU& c) { return to_np1(c.getCellSizeArray().toHost()); })
        .def("getXYZF", [](CellListGPU& c)
             {
             auto v = c.getXYZFArray().toHost();
             py::array_t<double> a({v.size(), (size_t)4});
             if (!v.empty())
                 std::memcpy(a.mutable_data(), v.data(), sizeof(Scalar4) * v.size());
             return a;
             })
        .def("getCellAdj", [](CellListGPU& c) { return to_np1(c.getCellAdjArray().toHost()); });

    py::class_<NeighborListGPUBinned, std::shared_ptr<NeighborListGPUBinned>> nl(m, "NeighborListGPUBinned");
    py::enum_<NeighborListGPUBinned::storageMode>(nl, "storageMode")
        .value("half", NeighborListGPUBinned::half)
        .value("full", NeighborListGPUBinned::full)
        .export_values();
    nl.def(py::init<std::shared_ptr<SystemDefinition>, Scalar>())
        .def_property("buffer", &NeighborListGPUBinned::getRBuff, &NeighborListGPUBinned::setRBuff)
        .def_property("rebuild_check_delay", &NeighborListGPUBinned::getRebuildCheckDelay, &NeighborListGPUBinned::setRebuildCheckDelay)
        .def_property("check_dist", &NeighborListGPUBinned::getDistCheck, &NeighborListGPUBinned::setDistCheck)
        .def_property("deterministic", &NeighborListGPUBinned::getDeterministic, &NeighborListGPUBinned::setDeterministic)
        .def_property_readonly("num_builds", &NeighborListGPUBinned::getNumUpdates)
        .def_property("tiles", &NeighborListGPUBinned::getTilesEnabled, &NeighborListGPUBinned::setTilesEnabled)
        .def_property("tile_min_occupancy", &NeighborListGPUBinned::getTileMinOccupancy, &NeighborListGPUBinned::setTileMinOccupancy)
        .def_property_readonly("tiles_active", &NeighborListGPUBinned::tilesActive)
        .def_property_readonly("tile_capacity", &NeighborListGPUBinned::getTileCapacity)
        .def_property("overlap", &NeighborListGPUBinned::getOverlapEnabled, &NeighborListGPUBinned::setOverlapEnabled)
        .def_property_readonly("num_overlapped", &NeighborListGPUBinned::getNumOverlapped)
        .def_property("comm_stream_high_priority", &NeighborListGPUBinned::getCommStreamHighPriority, &NeighborListGPUBinned::setCommStreamHighPriority)
        .def_property("boundary_on_comm_stream", &NeighborListGPUBinned::getBoundaryOnCommStream, &NeighborListGPUBinned::setBoundaryOnCommStream)
        .def_property_readonly("comm_stream_probe", &NeighborListGPUBinned::getCommStreamProbe)
        .def_property("ranks_share_gpu", &NeighborListGPUBinned::getRanksShareGPU, &NeighborListGPUBinned::setRanksShareGPU)
        .def("setStorageMode", &NeighborListGPUBinned::setStorageMode)
        .def("getStorageMode", &NeighborListGPUBinned::getStorageMode)
        .def("setFilterBody", &NeighborListGPUBinned::setFilterBody)
        .def("getMaxRCut", &NeighborListGPUBinned::getMaxRCut)
        .def("getMinRCut", &NeighborListGPUBinned::getMinRCut)
        .def("getMaxRList", &NeighborListGPUBinned::getMaxRList)
        .def("getMinRList", &NeighborListGPUBinned::getMinRList)
        .def("getGhostLayerWidth", &NeighborListGPUBinned::getGhostLayerWidth)
        .def("forceUpdate", &NeighborListGPUBinned::forceUpdate)
        .def("getSmallestRebuild", &NeighborListGPUBinned::getSmallestRebuild)
        .def("getNumUpdates", &NeighborListGPUBinned::getNumUpdates)
        .def("getNumDangerousUpdates", &NeighborListGPUBinned::getNumDangerousUpdates)
        .def("getNumExclusions", &NeighborListGPUBinned::getNumExclusions)
        .def("resetStats", &NeighborListGPUBinned::resetStats)
        .def("addExclusion", &NeighborListGPUBinned::addExclusion)
        .def("addExclusionsFromBonds", &NeighborListGPUBinned::addExclusionsFromBonds)
        .def("clearExclusions", &NeighborListGPUBinned::clearExclusions)
        .def("compute", &NeighborListGPUBinned::compute)
        .def("peekUpdate", &NeighborListGPUBinned::peekUpdate)
        .def("getDim", &NeighborListGPUBinned::getDim)
        .def("getNmax", &NeighborListGPUBinned::getNmax)
        .def("getNmaxPerType", &NeighborListGPUBinned::getNmaxPerType)
        .def("getCellList", &NeighborListGPUBinned::getCellList)
        .def("getLocalPairList", [](NeighborListGPUBinned& n)
             {
             auto v = n.getLocalPairList();
             py::array_t<unsigned int> a({v.size(), (size_t)2});
             auto r = a.mutable_unchecked<2>();
             for (size_t i = 0; i < v.size(); i++)
                 {
                 r(i, 0) = v[i].first;
                 r(i, 1) = v[i].second;
                 }
             return a;
             })
        .def("getPairList", [](NeighborListGPUBinned& n)
             {
             auto v = n.getPairList();
             py::array_t<unsigned int> a({v.size(), (size_t)2});
             auto r = a.mutable_unchecked<2>();
             for (size_t i = 0; i < v.size(); i++)
                 {
                 r(i, 0) = v[i].first;
                 r(i, 1) = v[i].second;
                 }
             return a;
             })
        // LocalNeighborListData role (hoomd/md/NeighborList.h:662-739): host copies of the raw arrays
        .def("getNNeighArray", [](NeighborListGPUBinned& n, unsigned int N) { return to_np1(n.getNNeighArray().toHost(N)); })
        .def("getHeadListArray", [](NeighborListGPUBinned& n, unsigned int N)
             {
             auto v = n.getHeadList().toHost(N);
             return to_np1(std::vector<uint64_t>(v.begin(), v.end()));
             })
        .def("getNListArray", [](NeighborListGPUBinned& n) { return to_np1(n.getNListArray().toHost()); })
        .def_property_readonly("n_neigh_ptr", [](NeighborListGPUBinned& n) { return (uintptr_t)n.getNNeighArray().get(); })
        .def_property_readonly("nlist_ptr", [](NeighborListGPUBinned& n) { return (uintptr_t)n.getNListArray().get(); })
        .def_property_readonly("nlist_size", [](NeighborListGPUBinned& n) { return n.getNListArray().size(); })
        .def_property_readonly("head_list_ptr", [](NeighborListGPUBinned& n) { return (uintptr_t)n.getHeadList().get(); });

    py::class_<ForceCompute, std::shared_ptr<ForceCompute>>(m, "ForceCompute")
        .def("compute", &ForceCompute::compute)
        .def("getForcesAndEnergies", [](ForceCompute& f) { return to_np(f.getForcesAndEnergies()); })
        .def("getVirials", [](ForceCompute& f) { return to_np(f.getVirials()); })
        .def("calcEnergySum", &ForceCompute::calcEnergySum)
        .def("getExternalEnergy", &ForceCompute::getExternalEnergy)
        .def("getExternalVirial", &ForceCompute::getExternalVirial)
        .def_property_readonly("force_ptr", [](ForceCompute& f) { return (uintptr_t)f.getForceArray(); })
        .def("checkErrors", [](ForceCompute& f)
             {
             hipCheck(hipStreamSynchronize(0), "hipStreamSynchronize");
             f.checkErrors();
             });

    py::class_<PotentialBondFENEGPU, ForceCompute, std::shared_ptr<PotentialBondFENEGPU>>(m, "PotentialBondFENEGPU")
        .def(py::init<std::shared_ptr<SystemDefinition>>())
        .def("setParams", [](PotentialBondFENEGPU& self, unsigned int type, py::dict v)
             {
             self.setParams(type, v["k"].cast<Scalar>(), v["r0"].cast<Scalar>(), v["epsilon"].cast<Scalar>(), v["sigma"].cast<Scalar>(),
                            v["delta"].cast<Scalar>());
             })
        .def("getParams", [](PotentialBondFENEGPU& self, unsigned int type)
             {
             const auto p = self.getParams(type);
             py::dict v;
             v["k"] = p[0];
             v["r0"] = p[1];
             v["epsilon"] = p[2];
             v["sigma"] = p[3];
             v["delta"] = p[4];
             return v;
             });

    // pair potentials: one export line per class, as a plugin's module has (example_plugins/pair_plugin/module.cc:20-26)
    hoomd::md::detail::export_PotentialPair<hoomd::md::EvaluatorPairLJ>(m, "PotentialPairLJ");
    hoomd::md::detail::export_PotentialPairGPU<hoomd::md::EvaluatorPairLJ>(m, "PotentialPairLJGPU");

    py::class_<ComputeThermoGPU, std::shared_ptr<ComputeThermoGPU>>(m, "ComputeThermoGPU")
        .def(py::init<std::shared_ptr<SystemDefinition>, std::shared_ptr<ParticleGroup>>())
        .def("compute", &ComputeThermoGPU::compute)
        .def("forceCompute", &ComputeThermoGPU::forceCompute)
        .def_property_readonly("kinetic_energy", &ComputeThermoGPU::getKineticEnergy)
        .def_property_readonly("potential_energy", &ComputeThermoGPU::getPotentialEnergy)
        .def_property_readonly("kinetic_temperature", &ComputeThermoGPU::getTranslationalTemperature)
        .def_property_readonly("pressure", &ComputeThermoGPU::getPressure)
        .def_property_readonly("momentum", &ComputeThermoGPU::getMomentum)
        .def_property_readonly("degrees_of_freedom", &ComputeThermoGPU::getTranslationalDOF)
        .def("setAllreduce", [](ComputeThermoGPU& t, py::object fn)
             {
             if (fn.is_none())
                 {
                 t.m_allreduce_sum = nullptr;
                 return;
                 }
             t.m_allreduce_sum = [fn](double* v, int n)
             {
                 py::gil_scoped_acquire gil;
                 py::list l;
                 for (int i = 0; i < n; i++)
                     l.append(v[i]);
                 py::list r = fn(l);
                 for (int i = 0; i < n; i++)
                     v[i] = r[i].cast<double>();
             };
             });

    py::class_<Thermostat, std::shared_ptr<Thermostat>>(m, "Thermostat").def_property("kT", &Thermostat::getT, &Thermostat::setT);
    py::class_<BerendsenThermostat, Thermostat, std::shared_ptr<BerendsenThermostat>>(m, "BerendsenThermostat")
        .def(py::init<Scalar, std::shared_ptr<ParticleGroup>, std::shared_ptr<ComputeThermoGPU>, std::shared_ptr<SystemDefinition>, Scalar>());
    py::class_<BussiThermostat, Thermostat, std::shared_ptr<BussiThermostat>>(m, "BussiThermostat")
        .def(py::init<Scalar, std::shared_ptr<ParticleGroup>, std::shared_ptr<ComputeThermoGPU>, std::shared_ptr<SystemDefinition>>());

    py::class_<IntegrationMethodTwoStep, std::shared_ptr<IntegrationMethodTwoStep>>(m, "IntegrationMethodTwoStep");
    py::class_<TwoStepConstantVolumeGPU, IntegrationMethodTwoStep, std::shared_ptr<TwoStepConstantVolumeGPU>>(m, "TwoStepConstantVolumeGPU")
        .def(py::init<std::shared_ptr<SystemDefinition>, std::shared_ptr<ParticleGroup>, std::shared_ptr<Thermostat>>())
        .def("setThermostat", &TwoStepConstantVolumeGPU::setThermostat)
        .def_property_readonly("num_steps_with_sums", &TwoStepConstantVolumeGPU::getNumStepsWithSums)
        .def_property_readonly("num_deferred_factors", &TwoStepConstantVolumeGPU::getNumDeferredFactors)
        .def_property("maximum_displacement", &TwoStepConstantVolumeGPU::getLimit, &TwoStepConstantVolumeGPU::setLimit)
        .def("integrateStepOne", &TwoStepConstantVolumeGPU::integrateStepOne)
        .def("integrateStepTwo", &TwoStepConstantVolumeGPU::integrateStepTwo)
        .def("setDeltaT", &TwoStepConstantVolumeGPU::setDeltaT);

    py::class_<IntegratorTwoStep, std::shared_ptr<IntegratorTwoStep>>(m, "IntegratorTwoStep")
        .def(py::init<std::shared_ptr<SystemDefinition>, Scalar>())
        .def_property("dt", &IntegratorTwoStep::getDeltaT, &IntegratorTwoStep::setDeltaT)
        .def("setForces", [](IntegratorTwoStep& i, std::vector<std::shared_ptr<ForceCompute>> f)
             {
             i.getForces() = f;
             i.notifyForcesChanged();
             })
        .def("setMethods", [](IntegratorTwoStep& i, std::vector<std::shared_ptr<IntegrationMethodTwoStep>> meth)
             {
             i.getIntegrationMethods() = meth;
             i.notifyForcesChanged();
             })
        .def("update", &IntegratorTwoStep::update)
        .def_property("step_rider", &IntegratorTwoStep::getStepRider, &IntegratorTwoStep::setStepRider)
        .def_property_readonly("num_rider_steps", &IntegratorTwoStep::getNumRiderSteps)
        .def("prepRun", &IntegratorTwoStep::prepRun)
        .def("computeNetForce", &IntegratorTwoStep::computeNetForceGPU)
        .def("computeLinearMomentum", &IntegratorTwoStep::computeLinearMomentum)
        .def("updateGroupDOF", &IntegratorTwoStep::updateGroupDOF)
        .def("getTranslationalDOF", &IntegratorTwoStep::getTranslationalDOF);

    py::class_<ParticleSorter, std::shared_ptr<ParticleSorter>>(m, "ParticleSorter")
        .def(py::init<std::shared_ptr<SystemDefinition>, uint64_t>())
        .def_property("period", &ParticleSorter::getPeriod, &ParticleSorter::setPeriod)
        .def("setWidth", &ParticleSorter::setWidth)
        .def("update", &ParticleSorter::update)
        .def("sortNow", &ParticleSorter::sortNow);

    py::class_<System, std::shared_ptr<System>>(m, "System")
        .def(py::init<std::shared_ptr<SystemDefinition>, uint64_t>())
        .def("setIntegrator", &System::setIntegrator)
        .def("setSorter", &System::setSorter)
        .def("setPressureFlag", &System::setPressureFlag)
        .def("run", &System::run, py::arg("nsteps"), py::arg("prepare") = true, py::call_guard<py::gil_scoped_release>())
        .def("getCurrentTimeStep", &System::getCurrentTimeStep)
        .def("getLastTPS", &System::getLastTPS)
        .def("getLastWalltime", &System::getLastWalltime);
    }
