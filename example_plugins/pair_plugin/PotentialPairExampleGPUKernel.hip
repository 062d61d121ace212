// The one translation unit of the HIP compiler a pair-potential plugin has: it names the instantiation of the
// kernel driver for its evaluator (example_plugins/pair_plugin/PotentialPairExampleGPUKernel.cu:13-15 of the reference).
#include "EvaluatorPairExample.h"
#include "hoomd/md/PotentialPairGPU.cuh"

namespace hoomd
    {
namespace md
    {
namespace kernel
    {
template __attribute__((visibility("default"))) hipError_t
gpu_compute_pair_forces<EvaluatorPairExample>(const pair_args_t& pair_args, const EvaluatorPairExample::param_type* d_params);
    }
    }
    }
