// pair_force_example.hip -- C entry points for the example plugin's evaluator (the harmonic pair of
// example_plugins/pair_plugin/EvaluatorPairExample.h), so that the kernel templates can be exercised through the C
// boundary with an evaluator that was written against the plugin contract alone.  The plugin itself does not use
// these: it names its own instantiation (example_plugins/pair_plugin/PotentialPairExampleGPUKernel.hip).
#include "EvaluatorPairExample.h"
#include "hoomd/md/PotentialPairGPU.cuh"

using hoomd::md::EvaluatorPairExample;
static_assert(sizeof(EvaluatorPairExample::param_type) == sizeof(hmd_example_params_t), "the evaluator's parameter block is the C structure");

extern "C" int hmd_compute_pair_forces_example(const hmd_pair_args_t* args, const hmd_example_params_t* d_params, void* stream)
    {
    return hmd::launch_pair_forces<EvaluatorPairExample>(args, reinterpret_cast<const EvaluatorPairExample::param_type*>(d_params), stream, "example_pair");
    }

extern "C" int hmd_compute_pair_forces_example_tiles(const hmd_pair_args_t* args, const hmd_tile_geom_t* geom, const unsigned short* d_tile_nlist,
                                              const hmd_example_params_t* d_params, void* stream)
    {
    return hmd::launch_pair_forces_tiles<EvaluatorPairExample>(args, geom, d_tile_nlist, reinterpret_cast<const EvaluatorPairExample::param_type*>(d_params),
                                                               stream);
    }
