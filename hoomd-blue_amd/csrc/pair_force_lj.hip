// pair_force_lj.hip -- instantiates the pair-force driver for the Lennard-Jones evaluator.
// Replaces the reference's generated per-evaluator unit (hoomd/md/CMakeLists.txt:498-547 with
// hoomd/md/EvaluatorPairLJ.h).
#include "evaluators.cuh"
#include "pair_force.cuh"
#include "pair_force_tiles.cuh"

extern "C" int hmd_compute_pair_forces_lj(const hmd_pair_args_t* args, const hmd_lj_params_t* d_params, void* stream)
    {
    return hmd::launch_pair_forces<hmd::EvaluatorPairLJ>(args, d_params, stream, "lj");
    }

extern "C" int hmd_compute_pair_forces_lj_tiles(const hmd_pair_args_t* args, const hmd_tile_geom_t* geom, const unsigned short* d_tile_nlist,
                                              const hmd_lj_params_t* d_params, void* stream)
    {
    return hmd::launch_pair_forces_tiles<hmd::EvaluatorPairLJ>(args, geom, d_tile_nlist, d_params, stream);
    }
