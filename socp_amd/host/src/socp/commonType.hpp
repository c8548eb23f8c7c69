// commonType.hpp -- scalar type of the SOCP host API (mirror of the reference header of the same
// name, commonType.hpp:8-19).  The device path is double precision only.
#ifndef SOCP_AMD_COMMONTYPE_HPP_
#define SOCP_AMD_COMMONTYPE_HPP_

#if defined(floatType)
#error "socp_amd: real = float is not built (the gfx950 kernels are FP64; SURVEY 2, row 12)"
#endif

#ifndef real
typedef double real;
#endif
#ifndef __cminpack_double__
#define __cminpack_double__
#endif

#endif
