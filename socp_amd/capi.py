"""ctypes bindings over the C-ABI of libsocp_hip.so (include/socp_hip.h, cminpack.h, socp_solver.h).

Thin plumbing only: every call goes straight to the shared library, which has no CPU path.
If the library is missing, importing the loader raises -- nothing here falls back to NumPy.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SOCP_LIB_PATH") or os.path.join(_HERE, "_build", "libsocp_hip.so")      # (SOCP_LIB_PATH: A/B builds of the library)

OK, ERR_ARG, ERR_HIP, ERR_NO_DEVICE, ERR_UNSUPPORTED = 0, -1, -2, -3, -4
MODEL_GODDARD, MODEL_DOUBLE_INTEGRATOR, MODEL_COVID19, MODEL_INTERCEPTOR = 1, 2, 3, 4
INTERCEPTOR_PARAM_NAMES = ["c0", "hr", "d0", "eta", "propellant_mass", "empty_mass", "q", "ve", "alpha_max", "u_max",
                           "a_max", "mu_gft", "muT", "muV", "muC", "R_Earth", "mu0", "chartLimit"]
FIXED, FREE, CONTINUOUS = 0, 1, 2
VARIANT_AUTO, VARIANT_LANE_EXACT, VARIANT_LANE_FAST = 0, 1, 2
EVAL_RHS, EVAL_CONTROL, EVAL_HAMILTONIAN = 0, 1, 2
REQ_DONE, REQ_FVEC, REQ_JAC = 0, 1, 2
INT_RK4, INT_DOPRI5 = 0, 1
GODDARD_PARAM_NAMES = ["C", "b", "KD", "kr", "u_max", "mu1", "mu2", "singularControl"]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p

FCN = C.CFUNCTYPE(C.c_int, _vp, C.c_int, _dp, _dp, C.c_int)
FCNDER = C.CFUNCTYPE(C.c_int, _vp, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int)
FDJAC = C.CFUNCTYPE(C.c_int, _vp, C.c_int, _dp, _dp, C.c_double, _dp, C.c_int)

_lib = None


CHAIN_PLAIN, CHAIN_PARAM, CHAIN_DATA = 0, 1, 2
SOLVER_AUTO, SOLVER_HOST, SOLVER_DEVICE, SOLVER_DEVICE_FAST = 0, 1, 2, 3
FACTOR_EXACT, FACTOR_FAST = 0, 1


class ChainOptions(C.Structure):
    """socp_chain_options (include/socp_solver.h)."""
    _fields_ = [("kind", C.c_int), ("param_index", C.c_int), ("step", C.c_double), ("step_min", C.c_double),
                ("xtol", C.c_double), ("maxfev", C.c_int), ("epsfcn", C.c_double), ("factor", C.c_double),
                ("dedup", C.c_int), ("speculate", C.c_int), ("max_rounds", C.c_int), ("analytic_jac", C.c_int),
                ("solver", C.c_int)]


class ChainStats(C.Structure):
    _fields_ = [("rounds", C.c_longlong), ("jacobians_launched", C.c_longlong), ("jacobians_from_cache", C.c_longlong),
                ("speculative_rounds", C.c_longlong), ("restarts", C.c_longlong), ("wall_ms", C.c_double)]


class SocpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libsocp_hip error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Load libsocp_hip.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            # a source-only checkout: compile the HIP library now (hipcc cross-compiles gfx950 anywhere)
            import subprocess
            try:
                subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])
            except Exception as exc:
                raise ImportError("%s is not built and building it failed (%s): run "
                                  "`python -c 'import __graft_entry__ as g; g.build()'`" % (LIB_PATH, exc))
        # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64 (same soname as
        # /opt/rocm's).  Importing torch FIRST makes the loader bind this library to that copy, so
        # torch tensors/streams and our kernels share one runtime; the other order gives two runtimes
        # and torch then reports "No HIP GPUs are available".  Without torch, /opt/rocm's is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.socp_last_error.restype = C.c_char_p
        L.socp_last_error.argtypes = [_vp]
        L.socp_ctx_create.argtypes = [C.POINTER(_vp), C.c_int, C.c_int]
        for name in ("socp_ctx_destroy", "socp_ctx_synchronize", "socp_ctx_warm_up"):
            getattr(L, name).argtypes = [_vp]
        L.socp_ctx_set_params.argtypes = [_vp, _dp, C.c_int]
        L.socp_ctx_get_params.argtypes = [_vp, _dp, C.c_int]
        L.socp_ctx_set_step_number.argtypes = [_vp, C.c_int]
        L.socp_ctx_set_integrator.argtypes = [_vp, C.c_int, C.c_double]
        L.socp_ctx_set_switching_times.argtypes = [_vp, _dp, C.c_int]
        L.socp_ctx_set_variant.argtypes = [_vp, C.c_int]
        L.socp_ctx_set_stream.argtypes = [_vp, _vp, C.c_int]
        L.socp_ctx_aux_stream.argtypes = [_vp, C.POINTER(C.c_void_p)]
        L.socp_ctx_dims.argtypes = [_vp, _ip, _ip, _ip]
        L.socp_ctx_control_dim.argtypes = [_vp]
        L.socp_ctx_device.argtypes = [_vp]
        L.socp_ctx_num_params.argtypes = [_vp]
        L.socp_ctx_model_id.argtypes = [_vp]
        L.socp_ctx_has_variational.argtypes = [_vp]
        L.socp_ctx_counters.argtypes = [_vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        L.socp_integrate_batch.argtypes = [_vp, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int]
        L.socp_integrate_batch_dev.argtypes = [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int]
        L.socp_integrate_dense.argtypes = [_vp, C.c_double, C.c_double, _dp, _dp, _dp, _dp, C.c_int, _ip]
        L.socp_integrate_dense_aux.argtypes = [_vp, C.c_double, C.c_double, _dp, _dp, _dp, _dp, _dp, C.c_int, _ip]
        L.socp_eval_batch.argtypes = [_vp, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, _dp, C.c_int]
        L.socp_problem_set.argtypes = [_vp, C.c_int, _ip, _ip, _dp, _dp]
        L.socp_problem_num_param.argtypes = [_vp]
        L.socp_timeline.argtypes = [_vp, _dp, _dp]
        L.socp_residual_batch.argtypes = [_vp, C.c_int, _dp, _dp]
        L.socp_residual_batch_dev.argtypes = [_vp, C.c_int, _vp, _vp]
        L.socp_fd_jacobian.argtypes = [_vp, _dp, _dp, C.c_double, _dp, C.c_int]
        L.socp_fd_jacobian_dev.argtypes = [_vp, _vp, _vp, C.c_double, _vp, C.c_int]
        L.socp_var_jacobian.argtypes = [_vp, _dp, _dp]
        L.socp_fd_jacobian_multi_dev.argtypes = [_vp, C.c_int, _vp, _vp, C.c_double, _vp, C.c_int]
        L.socp_fd_rows_dev.argtypes = [_vp, C.c_int, _vp, C.c_double, _vp]
        L.socp_fd_rows.argtypes = [_vp, C.c_int, _dp, C.c_double, _dp]
        L.socp_fd_diff_dev.argtypes = [_vp, C.c_int, _vp, C.c_double, _vp, _vp]
        L.socp_plugin_load.argtypes = [C.c_char_p]
        L.socp_problem_set_blocks_dev.argtypes = [_vp, _vp, C.c_int, _vp, _vp]
        L.socp_residual_batch_blocks.argtypes = [_vp, C.c_int, _dp, _dp, C.c_int, _dp, _dp, _dp]
        L.socp_problem_num_nodes.argtypes = [_vp]
        L.socp_ctx_get_switching_times.argtypes = [_vp, _dp]
        L.socp_chains_solve.argtypes = [_vp, C.c_int, C.POINTER(ChainOptions), _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _ip,
                                        _ip, _dp, _dp, _dp, C.POINTER(ChainStats)]
        L.socp_chains_solve_ex.argtypes = [_vp, C.c_int, C.POINTER(ChainOptions), _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _ip,
                                           _ip, _ip, _dp, _dp, _dp, C.POINTER(ChainStats)]
        L.socp_var_jacobian_multi_dev.argtypes = [_vp, C.c_int, _vp, _vp]
        L.socp_multistart_solve.argtypes = [_vp, C.c_int, _dp, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int,
                                            _dp, _ip, _ip, _dp, C.POINTER(C.c_longlong)]
        L.hybrd.argtypes = [FCN, _vp, C.c_int, _dp, _dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double,
                            _dp, C.c_int, C.c_double, C.c_int, _ip, _dp, C.c_int, _dp, C.c_int, _dp,
                            _dp, _dp, _dp, _dp]
        L.socp_hybrd_batched.argtypes = [FCN, FDJAC] + L.hybrd.argtypes[1:]
        L.hybrj.argtypes = [FCNDER, _vp, C.c_int, _dp, _dp, _dp, C.c_int, C.c_double, C.c_int, _dp, C.c_int,
                            C.c_double, C.c_int, _ip, _ip, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.socp_hybr_create.restype = _vp
        L.socp_hybr_create.argtypes = [C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int]
        L.socp_hybr_destroy.argtypes = [_vp]
        L.socp_hybr_start.argtypes = [_vp, _dp, _dp]
        L.socp_hybr_advance.argtypes = [_vp, C.c_int, C.POINTER(_dp), C.POINTER(_dp)]
        for name in ("socp_hybr_info", "socp_hybr_nfev", "socp_hybr_njev"):
            getattr(L, name).argtypes = [_vp]
        for name in ("socp_hybr_x", "socp_hybr_fvec"):
            getattr(L, name).argtypes = [_vp]
            getattr(L, name).restype = _dp
        L.socp_hybr_trust_region.argtypes = [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.socp_hybr_trust_region.restype = None
        L.socp_workspace_release.argtypes = [C.c_int]
        L.socp_workspace_release.restype = C.c_double
        L.socp_workspace_cached_bytes.argtypes = [C.c_int]
        L.socp_workspace_cached_bytes.restype = C.c_double
        L.socp_qr_factor_batch.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _ip, C.POINTER(C.c_double)]
        L.socp_ctx_get_variant.argtypes = [_vp]
        _lib = L
    return _lib


def workspace_release(device=-1):
    """socp_workspace_release (include/socp_solver.h): frees the device engine's kept arenas of `device` (< 0: all); bytes freed."""
    return lib().socp_workspace_release(int(device))


def workspace_cached_bytes(device=-1):
    """socp_workspace_cached_bytes: device + pinned bytes the device engine keeps for its next call on `device` (< 0: all)."""
    return lib().socp_workspace_cached_bytes(int(device))


def qr_factor_batch(J, b, flavour=FACTOR_FAST, reps=1, device=-1, outputs=True):
    """socp_qr_factor_batch (include/socp_solver.h): J[count][n][n] as matrices J[k][i][j] = J_k(i, j), b[count][n].
    Returns dict(Q[count][n][n], R[count][n][n] (upper triangular, unpacked), qtb, rdiag, acnorm, sing, kernel_ms)."""
    L = lib()
    J = _f64(J)
    count, n = J.shape[0], J.shape[1]
    Jcm = np.ascontiguousarray(np.transpose(J, (0, 2, 1)))                   # column-major per problem
    b = _f64(b).reshape(count, n)
    ms = C.c_double(0)
    if not outputs:
        rc = L.socp_qr_factor_batch(int(device), n, count, _d(Jcm), _d(b), int(flavour), int(reps), None, None, None, None, None, None, C.byref(ms))
        if rc != OK:
            raise RuntimeError("socp_qr_factor_batch: %d" % rc)
        return {"kernel_ms": ms.value}
    Q = np.empty((count, n, n))
    Rp = np.empty((count, n * (n + 1) // 2))
    qtb, rdiag, acnorm = np.empty((count, n)), np.empty((count, n)), np.empty((count, n))
    sing = np.zeros(count, dtype=np.int32)
    rc = L.socp_qr_factor_batch(int(device), n, count, _d(Jcm), _d(b), int(flavour), int(reps), _d(Q), _d(Rp), _d(qtb), _d(rdiag), _d(acnorm),
                                sing.ctypes.data_as(_ip), C.byref(ms))
    if rc != OK:
        raise RuntimeError("socp_qr_factor_batch: %d" % rc)
    R = np.zeros((count, n, n))
    iu = np.triu_indices(n)
    R[:, iu[0], iu[1]] = Rp                                                  # packed by rows = row-major order of the upper triangle
    return {"Q": Q, "R": R, "qtb": qtb, "rdiag": rdiag, "acnorm": acnorm, "sing": sing, "kernel_ms": ms.value}


def plugin_load(path):
    """Load an out-of-tree device model (include/socp_plugin.h); afterwards Context(model_id) works."""
    rc = lib().socp_plugin_load(os.fsencode(path))
    if rc != OK:
        raise SocpError(rc, lib().socp_last_error(None).decode())


def _d(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Context:
    """One device context = one model object with its packed parameters (socp_ctx)."""

    def __init__(self, model_id, device=-1, nparams=None):
        self.L = lib()
        self.h = _vp()
        rc = self.L.socp_ctx_create(C.byref(self.h), int(model_id), int(device))
        if rc != OK:
            raise SocpError(rc, self.L.socp_last_error(None).decode())
        dim, s, sj = C.c_int(), C.c_int(), C.c_int()
        self.L.socp_ctx_dims(self.h, C.byref(dim), C.byref(s), C.byref(sj))
        self.model_id, self.dim, self.s, self.s_jac = model_id, dim.value, s.value, sj.value
        self.nu = self.L.socp_ctx_control_dim(self.h)
        self.n = None
        self.nparams = nparams
        if nparams is None and model_id == MODEL_INTERCEPTOR:
            self.nparams = len(INTERCEPTOR_PARAM_NAMES)

    def close(self):
        if self.h:
            self.L.socp_ctx_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            raise SocpError(rc, self.L.socp_last_error(self.h).decode())

    # -- configuration
    def aux_stream(self):
        """The context's second stream (socp_ctx_aux_stream): created on the first call (~6 ms), then kept."""
        st = C.c_void_p()
        self._chk(self.L.socp_ctx_aux_stream(self.h, C.byref(st)))
        return st.value

    def warm_up(self):
        """socp_ctx_warm_up: the process's one-time costs now (second stream, copy-engine start-up)."""
        self._chk(self.L.socp_ctx_warm_up(self.h))

    def has_variational(self):
        return self.L.socp_ctx_has_variational(self.h) == 1

    def set_params(self, params):
        p = _f64(params)
        self._chk(self.L.socp_ctx_set_params(self.h, _d(p), len(p)))

    def get_params(self):
        n = self.nparams if self.nparams is not None else (3 if self.model_id == MODEL_DOUBLE_INTEGRATOR else 8)
        p = np.empty(n)
        self._chk(self.L.socp_ctx_get_params(self.h, _d(p), n))
        return p

    def set_param(self, name, value):
        p = self.get_params()
        names = INTERCEPTOR_PARAM_NAMES if self.model_id == MODEL_INTERCEPTOR else GODDARD_PARAM_NAMES
        p[names.index(name)] = value
        self.set_params(p)

    def set_step_number(self, n):
        self._chk(self.L.socp_ctx_set_step_number(self.h, int(n)))

    def set_integrator(self, kind, tol=1e-8):
        """kind: 0 = fixed-step RK4, 1 = adaptive Dormand-Prince 5(4) with abs = rel tolerance tol."""
        self._chk(self.L.socp_ctx_set_integrator(self.h, int(kind), float(tol)))

    def set_switching_times(self, sw):
        sw = _f64(sw)
        self._chk(self.L.socp_ctx_set_switching_times(self.h, _d(sw), len(sw)))

    def set_variant(self, v):
        self._chk(self.L.socp_ctx_set_variant(self.h, int(v)))

    def get_variant(self):
        return int(self.L.socp_ctx_get_variant(self.h))

    def set_stream(self, stream_ptr, use_own=False):
        """stream_ptr: hipStream_t as int (0 = the default stream); use_own=True: context's own stream."""
        self._chk(self.L.socp_ctx_set_stream(self.h, _vp(stream_ptr), int(bool(use_own))))

    def synchronize(self):
        self._chk(self.L.socp_ctx_synchronize(self.h))

    def counters(self):
        a, b = C.c_longlong(), C.c_longlong()
        self._chk(self.L.socp_ctx_counters(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # -- trajectories (host buffers)
    def integrate_batch(self, t0, tf, X0, sw=None, is_jac=0):
        X0 = _f64(X0)
        B = X0.shape[0]
        t0 = _f64(np.broadcast_to(t0, (B,)))
        tf = _f64(np.broadcast_to(tf, (B,)))
        Xf = np.empty_like(X0)
        swp = None
        if sw is not None:
            sw = _f64(sw)
            swp = _d(sw)
        self._chk(self.L.socp_integrate_batch(self.h, B, _d(t0), _d(tf), swp, _d(X0), _d(Xf), int(is_jac)))
        return Xf

    def integrate_batch_dev(self, B, d_t0, d_tf, d_sw, d_X0, d_Xf, is_jac=0):
        """Device pointers (ints); enqueue only."""
        self._chk(self.L.socp_integrate_batch_dev(self.h, int(B), _vp(d_t0), _vp(d_tf), _vp(d_sw), _vp(d_X0),
                                                  _vp(d_Xf), int(is_jac)))

    def integrate_dense(self, t0, tf, X0, sw=None, cap=None):
        """One trajectory, state after every step: returns (times[rows], X[rows][s])."""
        X0 = _f64(X0)
        cap = cap or 20000
        dense = np.empty((cap, len(X0)))
        times = np.empty(cap)
        rows = C.c_int(0)
        swp = _d(_f64(sw)) if sw is not None else None
        self._chk(self.L.socp_integrate_dense(self.h, float(t0), float(tf), swp, _d(X0), _d(dense), _d(times), cap,
                                              C.byref(rows)))
        k = min(rows.value, cap)
        return times[:k].copy(), dense[:k].copy()

    def integrate_dense_aux(self, t0, tf, X0, sw=None, cap=None):
        """As integrate_dense, plus each row's two auxiliary scalars: (times, X[rows][s], aux[rows][2])."""
        X0 = _f64(X0)
        cap = cap or 20000
        dense = np.empty((cap, len(X0)))
        times = np.empty(cap)
        aux = np.empty((cap, 2))
        rows = C.c_int(0)
        swp = _d(_f64(sw)) if sw is not None else None
        self._chk(self.L.socp_integrate_dense_aux(self.h, float(t0), float(tf), swp, _d(X0), _d(dense), _d(times),
                                                  _d(aux), cap, C.byref(rows)))
        k = min(rows.value, cap)
        return times[:k].copy(), dense[:k].copy(), aux[:k].copy()

    def eval_batch(self, what, t, X, sw=None, is_jac=0):
        X = _f64(X)
        B = X.shape[0]
        t = _f64(np.broadcast_to(t, (B,)))
        out_len = {EVAL_RHS: X.shape[1], EVAL_CONTROL: self.nu, EVAL_HAMILTONIAN: (self.s + 1) if is_jac else 1}[what]
        out = np.empty((B, out_len))
        swp = None
        if sw is not None:
            sw = _f64(sw)
            swp = _d(sw)
        self._chk(self.L.socp_eval_batch(self.h, what, B, _d(t), swp, _d(X), X.shape[1], _d(out), int(is_jac)))
        return out

    def var_jacobian(self, z):
        """Analytic (variational) shooting Jacobian J[row, col] (hybrj path)."""
        z = _f64(z)
        Jcm = np.empty((self.n, self.n))
        self._chk(self.L.socp_var_jacobian(self.h, _d(z), _d(Jcm)))
        return Jcm.T.copy()

    # -- shooting problem
    def problem_set(self, mode_t, mode_x, time, xnode):
        mode_t = np.ascontiguousarray(mode_t, dtype=np.int32)
        M = len(mode_t) - 1
        mode_x = np.ascontiguousarray(mode_x, dtype=np.int32).reshape(M + 1, self.dim)
        time = _f64(time)
        xnode = _f64(xnode).reshape(M + 1, self.s)
        self._chk(self.L.socp_problem_set(self.h, M, mode_t.ctypes.data_as(_ip), mode_x.ctypes.data_as(_ip),
                                          _d(time), _d(xnode)))
        self.M = M
        self.n = self.L.socp_problem_num_param(self.h)
        return self.n

    def timeline(self, z):
        z = _f64(z)
        tl = np.empty(self.M + 1)
        self._chk(self.L.socp_timeline(self.h, _d(z), _d(tl)))
        return tl

    def residual_batch(self, Z):
        Z = _f64(Z)
        assert Z.ndim == 2 and Z.shape[1] == self.n
        F = np.empty_like(Z)
        self._chk(self.L.socp_residual_batch(self.h, Z.shape[0], _d(Z), _d(F)))
        return F

    def residual(self, z):
        return self.residual_batch(_f64(z)[None, :])[0]

    def residual_batch_dev(self, B, d_Z, d_F):
        self._chk(self.L.socp_residual_batch_dev(self.h, int(B), _vp(d_Z), _vp(d_F)))

    def fd_jacobian(self, z, fvec, epsfcn=1e-15, dedup=False):
        """Returns J[row, col] (the C side is column-major)."""
        z, fvec = _f64(z), _f64(fvec)
        Jcm = np.empty((self.n, self.n))
        self._chk(self.L.socp_fd_jacobian(self.h, _d(z), _d(fvec), float(epsfcn), _d(Jcm), int(bool(dedup))))
        return Jcm.T.copy()

    def fd_rows(self, Z, epsfcn=1e-15):
        """Rows[np][n+1][n]: F(z) and the n forward-difference residuals, one launch."""
        Z = _f64(Z).reshape(-1, self.n)
        rows = np.empty((Z.shape[0], self.n + 1, self.n))
        self._chk(self.L.socp_fd_rows(self.h, Z.shape[0], _d(Z), float(epsfcn), _d(rows)))
        return rows

    def fd_rows_dev(self, np_, d_Z, epsfcn, d_rows):
        self._chk(self.L.socp_fd_rows_dev(self.h, int(np_), _vp(d_Z), float(epsfcn), _vp(d_rows)))

    def fd_diff_dev(self, np_, d_Z, epsfcn, d_rows, d_fjac):
        self._chk(self.L.socp_fd_diff_dev(self.h, int(np_), _vp(d_Z), float(epsfcn), _vp(d_rows), _vp(d_fjac)))

    def fd_jacobian_multi_dev(self, np_, d_Z, d_Fvec, epsfcn, d_Fjac, dedup=False):
        self._chk(self.L.socp_fd_jacobian_multi_dev(self.h, int(np_), _vp(d_Z), _vp(d_Fvec), float(epsfcn),
                                                    _vp(d_Fjac), int(bool(dedup))))

    def multistart_solve(self, Z0, xtol=1e-8, maxfev=10000, epsfcn=1e-15, factor=1.0, dedup=True):
        """Lock-step hybrd solves of the rows of Z0.  Returns dict(z, info, nfev, fnorm, rounds)."""
        Z0 = _f64(Z0).reshape(-1, self.n)
        P = Z0.shape[0]
        Z = np.empty_like(Z0)
        info = np.zeros(P, dtype=np.int32)
        nfev = np.zeros(P, dtype=np.int32)
        fnorm = np.zeros(P)
        rounds = C.c_longlong(0)
        self._chk(self.L.socp_multistart_solve(self.h, P, _d(Z0), float(xtol), int(maxfev), float(epsfcn), float(factor),
                                               int(bool(dedup)), _d(Z), info.ctypes.data_as(_ip),
                                               nfev.ctypes.data_as(_ip), _d(fnorm), C.byref(rounds)))
        return dict(z=Z, info=info, nfev=nfev, fnorm=fnorm, rounds=rounds.value)

    def residual_batch_blocks(self, Z, params=None, time=None, xnode=None):
        """Residual of every row of Z with that row's OWN parameter block [B][nparams + 2] (parameters, sw0, sw1) and / or
        boundary tables time [B][M+1], xnode [B][(M+1)*2d] (None: the shared ones)."""
        Z = _f64(Z).reshape(-1, self.n)
        B = Z.shape[0]
        F = np.empty_like(Z)
        pp = _f64(params).reshape(B, -1) if params is not None else None
        tt = _f64(time).reshape(B, -1) if time is not None else None
        xx = _f64(xnode).reshape(B, -1) if xnode is not None else None
        self._chk(self.L.socp_residual_batch_blocks(self.h, B, _d(Z), _d(pp) if pp is not None else None,
                                                    pp.shape[1] if pp is not None else 0, _d(tt) if tt is not None else None,
                                                    _d(xx) if xx is not None else None, _d(F)))
        return F

    def chains_solve(self, Z0, kind=CHAIN_PLAIN, param_index=0, step=1.0, step_min=1e-12, goal=None, params=None,
                     time_prev=None, x_prev=None, time_goal=None, x_goal=None, xtol=1e-8, maxfev=10000, epsfcn=1e-15,
                     factor=1.0, dedup=True, speculate=-1, max_rounds=0, analytic_jac=False, solver=SOLVER_AUTO):
        """Lock-step continuation chains (socp_chains_solve).  Returns a dict of per-chain arrays + 'stats'."""
        Z0 = _f64(Z0).reshape(-1, self.n)
        P = Z0.shape[0]
        opt = ChainOptions(int(kind), int(param_index), float(step), float(step_min), float(xtol), int(maxfev), float(epsfcn),
                           float(factor), int(bool(dedup)), int(speculate), int(max_rounds), int(bool(analytic_jac)), int(solver))
        keep = []

        def arr(a, width=None):
            if a is None:
                return None
            a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (P,) if width is None else (P, width)))
            keep.append(a)
            return _d(a)
        nodes = self.L.socp_problem_num_nodes(self.h)
        Z = np.empty_like(Z0)
        info = np.zeros(P, dtype=np.int32)
        nfev_last = np.zeros(P, dtype=np.int32)
        nfev_total = np.zeros(P, dtype=np.int32)
        solves = np.zeros(P, dtype=np.int32)
        njev = np.zeros(P, dtype=np.int32)
        b = np.zeros(P)
        pf = np.zeros(P)
        fn = np.zeros(P)
        st = ChainStats()
        ip = lambda a: a.ctypes.data_as(_ip)  # noqa: E731
        self._chk(self.L.socp_chains_solve_ex(self.h, P, C.byref(opt), _d(Z0), arr(params, len(self.get_params())), arr(goal),
                                              arr(time_prev, nodes), arr(x_prev, nodes * self.s), arr(time_goal, nodes),
                                              arr(x_goal, nodes * self.s), _d(Z), ip(info), ip(nfev_last), ip(nfev_total), ip(njev),
                                              ip(solves), _d(b), _d(pf), _d(fn), C.byref(st)))
        return dict(z=Z, info=info, nfev=nfev_last, nfev_total=nfev_total, njev=njev, solves=solves, b_reached=b, param_final=pf, fnorm=fn,
                    stats={k: getattr(st, k) for k, _ in ChainStats._fields_})

    def fd_jacobian_dev(self, d_z, d_fvec, epsfcn, d_fjac, dedup=False):
        self._chk(self.L.socp_fd_jacobian_dev(self.h, _vp(d_z), _vp(d_fvec), float(epsfcn), _vp(d_fjac),
                                              int(bool(dedup))))


# ---------------------------------------------------------------------------------------------
# MINPACK entry points (host code inside the same library)
# ---------------------------------------------------------------------------------------------

def _workspace(n):
    return dict(fvec=np.zeros(n), diag=np.ones(n), fjac=np.zeros((n, n)), r=np.zeros(n * (n + 1) // 2),
                qtf=np.zeros(n), wa1=np.zeros(n), wa2=np.zeros(n), wa3=np.zeros(n), wa4=np.zeros(n))


def hybrd(func, x0, xtol=1e-8, maxfev=10000, ml=None, mu=None, epsfcn=1e-15, mode=1, factor=1.0,
          diag=None, fdjac=None):
    """Drive the library's hybrd (or socp_hybrd_batched when `fdjac` is given) with Python callables.

    func(x) -> F(x) (or None to abort);  fdjac(x, fvec, epsfcn) -> J[row, col].
    Returns dict(x, fvec, info, nfev, fjac (Q, column-major as MINPACK leaves it), r, qtf, diag).
    """
    L = lib()
    x = _f64(np.array(x0, dtype=np.float64))
    n = len(x)
    ws = _workspace(n)
    if diag is not None:
        ws["diag"][:] = diag
    ml = n - 1 if ml is None else ml
    mu = n - 1 if mu is None else mu
    nfev = C.c_int(0)

    def _fcn(p, nn, xp, fp, iflag):
        out = func(np.ctypeslib.as_array(xp, shape=(nn,)).copy())
        if out is None:
            return -1
        np.ctypeslib.as_array(fp, shape=(nn,))[:] = out
        return 0

    def _jac(p, nn, xp, fp, eps, jp, ld):
        J = fdjac(np.ctypeslib.as_array(xp, shape=(nn,)).copy(), np.ctypeslib.as_array(fp, shape=(nn,)).copy(), eps)
        if J is None:
            return -1
        np.ctypeslib.as_array(jp, shape=(nn, ld))[:, :nn] = np.asarray(J).T   # column-major
        return 0

    cb = FCN(_fcn)
    args = [None, n, _d(x), _d(ws["fvec"]), xtol, maxfev, ml, mu, epsfcn, _d(ws["diag"]), mode, factor, 0,
            C.byref(nfev), _d(ws["fjac"]), n, _d(ws["r"]), len(ws["r"]), _d(ws["qtf"]),
            _d(ws["wa1"]), _d(ws["wa2"]), _d(ws["wa3"]), _d(ws["wa4"])]
    if fdjac is None:
        info = L.hybrd(cb, *args)
    else:
        info = L.socp_hybrd_batched(cb, FDJAC(_jac), *args)
    return dict(x=x, fvec=ws["fvec"], info=info, nfev=nfev.value, fjac=ws["fjac"], r=ws["r"], qtf=ws["qtf"],
                diag=ws["diag"])


def hybrj(func, jac, x0, xtol=1e-8, maxfev=10000, mode=1, factor=1.0, diag=None):
    """Drive the library's hybrj.  func(x) -> F, jac(x) -> J[row, col]."""
    L = lib()
    x = _f64(np.array(x0, dtype=np.float64))
    n = len(x)
    ws = _workspace(n)
    if diag is not None:
        ws["diag"][:] = diag
    nfev, njev = C.c_int(0), C.c_int(0)

    def _fcn(p, nn, xp, fp, jp, ld, iflag):
        xx = np.ctypeslib.as_array(xp, shape=(nn,)).copy()
        if iflag == 1:
            out = func(xx)
            if out is None:
                return -1
            np.ctypeslib.as_array(fp, shape=(nn,))[:] = out
        else:
            J = jac(xx)
            if J is None:
                return -1
            np.ctypeslib.as_array(jp, shape=(nn, ld))[:, :nn] = np.asarray(J).T
        return 0

    cb = FCNDER(_fcn)
    info = L.hybrj(cb, None, n, _d(x), _d(ws["fvec"]), _d(ws["fjac"]), n, xtol, maxfev, _d(ws["diag"]), mode,
                   factor, 0, C.byref(nfev), C.byref(njev), _d(ws["r"]), len(ws["r"]), _d(ws["qtf"]),
                   _d(ws["wa1"]), _d(ws["wa2"]), _d(ws["wa3"]), _d(ws["wa4"]))
    return dict(x=x, fvec=ws["fvec"], info=info, nfev=nfev.value, njev=njev.value, fjac=ws["fjac"], r=ws["r"],
                qtf=ws["qtf"], diag=ws["diag"])


class HybrSolver:
    """Resumable solver object (socp_hybr_*)."""

    def __init__(self, n, xtol=1e-8, maxfev=10000, epsfcn=1e-15, mode=1, factor=1.0, analytic_jac=False):
        self.L = lib()
        self.n = n
        self.h = _vp(self.L.socp_hybr_create(n, xtol, maxfev, epsfcn, mode, factor, int(analytic_jac)))
        self._xe = _dp()
        self._out = _dp()

    def __del__(self):
        try:
            self.L.socp_hybr_destroy(self.h)
        except Exception:
            pass

    def start(self, x0, diag=None):
        x0 = _f64(x0)
        self.L.socp_hybr_start(self.h, _d(x0), _d(_f64(diag)) if diag is not None else None)

    def advance(self, flag=0):
        """Returns (request, x_eval view, out view)."""
        req = self.L.socp_hybr_advance(self.h, int(flag), C.byref(self._xe), C.byref(self._out))
        if req == REQ_DONE:
            return req, None, None
        xe = np.ctypeslib.as_array(self._xe, shape=(self.n,))
        out = np.ctypeslib.as_array(self._out, shape=(self.n,) if req == REQ_FVEC else (self.n * self.n,))
        return req, xe, out

    @property
    def info(self):
        return self.L.socp_hybr_info(self.h)

    @property
    def nfev(self):
        return self.L.socp_hybr_nfev(self.h)

    @property
    def njev(self):
        return self.L.socp_hybr_njev(self.h)

    @property
    def trust_region(self):
        """(delta, |diag x|, |F|) of the current iterate."""
        d, xn, fn = C.c_double(0), C.c_double(0), C.c_double(0)
        self.L.socp_hybr_trust_region(self.h, C.byref(d), C.byref(xn), C.byref(fn))
        return d.value, xn.value, fn.value

    @property
    def x(self):
        return np.ctypeslib.as_array(self.L.socp_hybr_x(self.h), shape=(self.n,)).copy()

    @property
    def fvec(self):
        return np.ctypeslib.as_array(self.L.socp_hybr_fvec(self.h), shape=(self.n,)).copy()
