// pair_force_example.hip -- what a pair-potential plugin ships: one translation unit that
// instantiates the driver for its evaluator, as
// example_plugins/pair_plugin/PotentialPairExampleGPUKernel.cu:13-15 does in the reference.
#include "evaluators.cuh"
#include "pair_force.cuh"
#include "pair_force_tiles.cuh"

extern "C" int hmd_compute_pair_forces_example(const hmd_pair_args_t* args, const hmd_example_params_t* d_params, void* stream)
    {
    return hmd::launch_pair_forces<hmd::EvaluatorPairExample>(args, d_params, stream, "example_pair");
    }

extern "C" int hmd_compute_pair_forces_example_tiles(const hmd_pair_args_t* args, const hmd_tile_geom_t* geom, const unsigned short* d_tile_nlist,
                                              const hmd_example_params_t* d_params, void* stream)
    {
    return hmd::launch_pair_forces_tiles<hmd::EvaluatorPairExample>(args, geom, d_tile_nlist, d_params, stream);
    }
