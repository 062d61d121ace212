// The Python module of the plugin: one export line per class (example_plugins/pair_plugin/module.cc:20-26 of the
// reference).  PotentialPair<evaluator> holds the per-type-pair tables, the tail corrections and the host-side pair
// energies; PotentialPairGPU<evaluator> launches the kernels instantiated in PotentialPairExampleGPUKernel.hip.
#include "EvaluatorPairExample.h"

#include "hoomd/md/PotentialPair.h"
#include <pybind11/pybind11.h>
#ifdef ENABLE_HIP
#include "hoomd/md/PotentialPairGPU.h"
#endif

namespace hoomd
    {
namespace md
    {
// the module name is the Makefile's NAME with an underscore in front
PYBIND11_MODULE(_pair_plugin, m)
    {
    detail::export_PotentialPair<EvaluatorPairExample>(m, "PotentialPairExample");
#ifdef ENABLE_HIP
    detail::export_PotentialPairGPU<EvaluatorPairExample>(m, "PotentialPairExampleGPU");
#endif
    }
    }
    }
