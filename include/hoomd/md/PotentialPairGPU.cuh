// hoomd/md/PotentialPairGPU.cuh -- the pair-force kernel driver as a template over the evaluator
// (role of hoomd/md/PotentialPairGPU.cuh:140-590 of the reference).
//
// An evaluator's kernels are instantiated by one translation unit of the HIP compiler that includes the evaluator's
// header and this file and names the instantiation, exactly as the reference's plugins do
// (example_plugins/pair_plugin/PotentialPairExampleGPUKernel.cu:13-15):
//
//     template __attribute__((visibility("default"))) hipError_t
//     gpu_compute_pair_forces<EvaluatorPairExample>(const pair_args_t&, const EvaluatorPairExample::param_type*);
//
// The host class PotentialPairGPU<evaluator> (hoomd/md/PotentialPairGPU.h) calls that symbol.  One instantiation holds
// both kernel families: the row kernel (one wave section per particle over its neighbor-list row) and the tile kernel
// (2x2x2-cell blocks staged in LDS, 16-bit rows), for the three shift modes with and without the virial.
#ifndef HOOMD_AMD_POTENTIAL_PAIR_GPU_CUH
#define HOOMD_AMD_POTENTIAL_PAIR_GPU_CUH

#include "hoomd/HOOMDMath.h"
#include "hoomd_amd.h"
#include <hip/hip_runtime_api.h>

namespace hoomd
    {
namespace md
    {
namespace kernel
    {
// Everything a launch needs besides the per-type-pair parameters.  Built by PotentialPairGPU::launchKernel; evaluator
// code never looks inside.
struct pair_args_t
    {
    hmd_pair_args_t args;                //!< arrays, box, tables, shift mode, predication (include/hoomd_amd.h)
    const hmd_tile_geom_t* tile_geom;    //!< non-null: use the tile kernel with these rows
    const unsigned short* d_tile_nlist;
    hipStream_t stream;
    const hmd_fused_step_args_t* step;   //!< non-null (tile kernel only): the integration pass rides on the kernel
    };

// hipSuccess, or hipErrorInvalidValue with the reason in hmd_last_error()
template<class evaluator> hipError_t gpu_compute_pair_forces(const pair_args_t& pair_args, const typename evaluator::param_type* d_params);
    } // namespace kernel
    } // namespace md
    } // namespace hoomd

#ifdef __HIPCC__
#include "pair_force.cuh"
#include "pair_force_tiles.cuh"

namespace hoomd
    {
namespace md
    {
namespace kernel
    {
template<class evaluator> hipError_t gpu_compute_pair_forces(const pair_args_t& pair_args, const typename evaluator::param_type* d_params)
    {
    static_assert(std::is_trivially_copyable<typename evaluator::param_type>::value || __is_trivially_copyable(typename evaluator::param_type),
                  "param_type is copied into LDS byte by byte");
    const int rc = pair_args.tile_geom
                       ? hmd::launch_pair_forces_tiles<evaluator>(&pair_args.args, pair_args.tile_geom, pair_args.d_tile_nlist, d_params, pair_args.stream, pair_args.step)
                       : hmd::launch_pair_forces<evaluator>(&pair_args.args, d_params, pair_args.stream, "pair");
    return rc == HMD_OK ? hipSuccess : hipErrorInvalidValue;
    }
    } // namespace kernel
    } // namespace md
    } // namespace hoomd
#endif

#endif
