/*
 * cminpack.h -- the two CMinPack entry points SOCP binds, exported by libsocp_hip.so.
 *
 * The reference's shooting.cpp includes <cminpack.h> (shooting.cpp:16) and its only undefined
 * link symbols are `hybrd` and `hybrj` (call sites shooting.cpp:803-826 and :830-851, names
 * spelled through __cminpack_func__(), precision selected by commonType.hpp:8-19 which defines
 * __cminpack_double__).  CMinPack itself is an un-pinned ExternalProject of the reference
 * (src/socp/CMakeLists.txt:11-24) and is not vendored; this header + libsocp_hip.so replace it
 * for those two symbols with the same C signatures, argument meaning, `info` codes and
 * callback conventions (MINPACK user guide: hybrd / hybrj).
 *
 * Differences a caller can observe: `nprint` is accepted and ignored (the reference passes 0,
 * shooting.cpp:101); only real = double is built.
 */
#ifndef SOCP_CMINPACK_H_
#define SOCP_CMINPACK_H_

#ifdef __cminpack_float__
#error "libsocp_hip.so provides the double-precision MINPACK entry points only"
#endif

#ifndef __cminpack_real__
#define __cminpack_real__ double
#endif
#ifndef __cminpack_func__
#define __cminpack_func__(name) name
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* fcn(p, n, x, fvec, iflag): iflag = 1 function value, 2 = one forward-difference column.
 * Return value < 0 aborts the solve and becomes `info` (shooting.cpp:873). */
typedef int (*cminpack_func_nn)(void *p, int n, const double *x, double *fvec, int iflag);

/* fcn(p, n, x, fvec, fjac, ldfjac, iflag): iflag = 1 -> fvec only, 2 -> fjac only
 * (column-major, leading dimension ldfjac; shooting.cpp:883-910). */
typedef int (*cminpack_funcder_nn)(void *p, int n, const double *x, double *fvec, double *fjac,
                                   int ldfjac, int iflag);

/* info: 0 improper input, 1 converged (relative error between two iterates <= xtol),
 * 2 maxfev reached, 3 xtol too small, 4 / 5 not making good progress (5 Jacobians / 10 iterates),
 * < 0 user abort. */
int hybrd(cminpack_func_nn fcn, void *p, int n, double *x, double *fvec, double xtol, int maxfev,
          int ml, int mu, double epsfcn, double *diag, int mode, double factor, int nprint,
          int *nfev, double *fjac, int ldfjac, double *r, int lr, double *qtf,
          double *wa1, double *wa2, double *wa3, double *wa4);

int hybrj(cminpack_funcder_nn fcn, void *p, int n, double *x, double *fvec, double *fjac, int ldfjac,
          double xtol, int maxfev, double *diag, int mode, double factor, int nprint,
          int *nfev, int *njev, double *r, int lr, double *qtf,
          double *wa1, double *wa2, double *wa3, double *wa4);

#ifdef __cplusplus
}
#endif
#endif /* SOCP_CMINPACK_H_ */
