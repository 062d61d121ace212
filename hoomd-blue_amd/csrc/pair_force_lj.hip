// pair_force_lj.hip -- the Lennard-Jones instantiation of the pair-force driver and its C entry points.
// Replaces the reference's generated per-evaluator unit (hoomd/md/CMakeLists.txt:498-547 with
// hoomd/md/EvaluatorPairLJ.h).
#include "hoomd/md/EvaluatorPairLJ.h"
#include "hoomd/md/PotentialPairGPU.cuh"

namespace hoomd
    {
namespace md
    {
namespace kernel
    {
template __attribute__((visibility("default"))) hipError_t
gpu_compute_pair_forces<EvaluatorPairLJ>(const pair_args_t& pair_args, const EvaluatorPairLJ::param_type* d_params);
    }
    }
    }

using hoomd::md::EvaluatorPairLJ;
static_assert(sizeof(EvaluatorPairLJ::param_type) == sizeof(hmd_lj_params_t), "the evaluator's parameter block is the C structure");

extern "C" int hmd_compute_pair_forces_lj(const hmd_pair_args_t* args, const hmd_lj_params_t* d_params, void* stream)
    {
    return hmd::launch_pair_forces<EvaluatorPairLJ>(args, static_cast<const EvaluatorPairLJ::param_type*>(d_params), stream, "lj");
    }

extern "C" int hmd_compute_pair_forces_lj_tiles(const hmd_pair_args_t* args, const hmd_tile_geom_t* geom, const unsigned short* d_tile_nlist,
                                              const hmd_lj_params_t* d_params, void* stream)
    {
    return hmd::launch_pair_forces_tiles<EvaluatorPairLJ>(args, geom, d_tile_nlist, static_cast<const EvaluatorPairLJ::param_type*>(d_params), stream);
    }

extern "C" int hmd_compute_pair_forces_lj_tiles_step(const hmd_pair_args_t* args, const hmd_tile_geom_t* geom, const unsigned short* d_tile_nlist,
                                                   const hmd_lj_params_t* d_params, const hmd_fused_step_args_t* step, void* stream)
    {
    HMD_REQUIRE(step, HMD_ERR_INVALID_ARGUMENT, "pair forces (tiles) with the integration pass: step is required");
    return hmd::launch_pair_forces_tiles<EvaluatorPairLJ>(args, geom, d_tile_nlist, static_cast<const EvaluatorPairLJ::param_type*>(d_params), stream, step);
    }

extern "C" int hmd_pair_forces_tiles_step_supported(const hmd_pair_args_t* args, const hmd_fused_step_args_t* step)
    {
    HMD_REQUIRE(args && step, HMD_ERR_INVALID_ARGUMENT, "hmd_pair_forces_tiles_step_supported: args and step are required");
    const char* why = nullptr;
    if (hmd::pft_step_supported(*args, *step, &why))
        return HMD_OK;
    hmd_set_error(why);
    return HMD_ERR_UNSUPPORTED;
    }
