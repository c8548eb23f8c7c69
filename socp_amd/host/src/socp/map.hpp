// map.hpp -- abstract penalty-map interface kept for source compatibility (reference map.hpp:15-47).
// Maps are host-side polymorphic objects with file I/O; models that depend on one have no device
// dynamics (out of scope).
#ifndef SOCP_AMD_MAP_HPP_
#define SOCP_AMD_MAP_HPP_

#include "commonType.hpp"

#include <vector>

class map
{
public:
    typedef std::vector<real> mstate;
    map() {}
    virtual ~map() {}
    virtual real Function(mstate const &X) const = 0;
    virtual mstate Gradient(mstate const &X) const = 0;
};

#endif
