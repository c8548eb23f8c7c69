// kernels_interceptor_fast.hip -- throughput flavour of the interceptor model (InterceptorT<true>,
// models_interceptor.hpp): compiled WITH FMA contraction.  Same launch-table mechanism as
// kernels_interceptor.hip; capi.cpp uses this table when the context's variant is SOCP_VARIANT_LANE_FAST, for
// both integrators (the table carries its own adaptive instantiations).
#include "models_interceptor.hpp"
#include "plugin_impl.hpp"

namespace socp {

const ModelLaunchers *interceptor_launchers_fast()
{
    static const ModelLaunchers t = plugin::table<InterceptorFast>(
        18, 50,
        {0.00075, 7500, 0.00005, 0.442, 200, 200, 10, 1500, 3.14159265358979323846 / 6, 1, 1500, 1, 0, 1, 0,
         6378145, 3.986e14, 0.1});
    return &t;
}

}  // namespace socp
