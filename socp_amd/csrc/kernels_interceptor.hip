// kernels_interceptor.hip -- the interceptor model's kernels (models_interceptor.hpp), reference operation order:
// MUST be compiled with -ffp-contract=off.  The model sits behind a launch table exactly like an out-of-tree
// plugin (plugin_impl.hpp); capi.cpp binds the table to SOCP_MODEL_INTERCEPTOR.  Its own translation unit
// keeps the two-chart right-hand side (~10 transcendental calls per evaluation) out of the other models'
// build and lets make compile it in parallel.
#include "models_interceptor.hpp"
#include "plugin_impl.hpp"

namespace socp {

const ModelLaunchers *interceptor_launchers()
{
    // interceptor.cpp:34-58 constructor defaults in SOCP_INTERCEPTOR_* order; ModelInt uses data->stepNbr = 50
    static const ModelLaunchers t = plugin::table<InterceptorModel>(
        18, 50,
        {0.00075, 7500, 0.00005, 0.442, 200, 200, 10, 1500, 3.14159265358979323846 / 6, 1, 1500, 1, 0, 1, 0,
         6378145, 3.986e14, 0.1});
    return &t;
}

}  // namespace socp
