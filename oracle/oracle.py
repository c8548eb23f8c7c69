"""ctypes bindings for the CPU checker -- TEST INFRASTRUCTURE, not product.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module (the product package ``socp_amd`` never does).

* ``Oracle``  -> oracle/_build/libsocp_oracle.so : plain-C restatement (oracle/socp_oracle.c)
* ``Ref``     -> oracle/_ref/libsocp_ref.so      : the reference's own objects behind a C shim
                 (oracle/ref_driver.cpp); present when built in the authoring container, travels
                 to the GPU box as a prebuilt file.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "libsocp_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libsocp_ref.so")

MODEL_GODDARD, MODEL_DINT, MODEL_COVID, MODEL_INTERCEPTOR = 1, 2, 3, 4
FIXED, FREE, CONTINUOUS = 0, 1, 2
GODDARD_PARAM_NAMES = ["C", "b", "KD", "kr", "u_max", "mu1", "mu2", "singularControl"]
INTERCEPTOR_PARAM_NAMES = ["c0", "hr", "d0", "eta", "propellant_mass", "empty_mass", "q", "ve", "alpha_max", "u_max",
                           "a_max", "mu_gft", "muT", "muV", "muC", "R_Earth", "mu0", "chartLimit"]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _d(a):
    return a.ctypes.data_as(_dp)


def build(ref=True):
    """(Re)build the checker libraries. Building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


class _OrcModel(C.Structure):
    _fields_ = [("model_id", C.c_int), ("dim", C.c_int), ("step_nbr", C.c_int),
                ("p", C.c_double * 24), ("nsw", C.c_int), ("sw", C.c_double * 64),
                ("chart", C.c_int), ("stage", C.c_int), ("integrator", C.c_int), ("tol", C.c_double)]


class _OrcProblem(C.Structure):
    _fields_ = [("dim", C.c_int), ("num_multi", C.c_int), ("mode_t", _ip), ("mode_x", _ip),
                ("time", _dp), ("xnode", _dp)]


class Problem:
    """Shooting problem data as `shooting::data_struct` holds it (shooting.cpp:21-54)."""

    def __init__(self, dim, mode_t, mode_x, time, xnode):
        self.dim = int(dim)
        self.M = len(mode_t) - 1
        self.mode_t = np.ascontiguousarray(mode_t, dtype=np.int32)
        self.mode_x = np.ascontiguousarray(mode_x, dtype=np.int32).reshape(self.M + 1, self.dim)
        self.time = np.ascontiguousarray(time, dtype=np.float64)
        self.xnode = np.ascontiguousarray(xnode, dtype=np.float64).reshape(self.M + 1, 2 * self.dim)
        self.n = 2 * self.dim * self.M + int(np.sum(self.mode_t == FREE))

    def c_struct(self):
        return _OrcProblem(self.dim, self.M, self.mode_t.ctypes.data_as(_ip),
                           self.mode_x.ctypes.data_as(_ip), _d(self.time), _d(self.xnode))


class Oracle:
    def __init__(self, model_id, step_nbr=None, params=None):
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        self.lib = C.CDLL(ORACLE_SO)
        self.lib.orc_goddard_singular_control.restype = C.c_double
        self.lib.orc_integrate.restype = C.c_long
        self.lib.orc_model_int.restype = C.c_long
        self.m = _OrcModel()
        self.lib.orc_model_init(C.byref(self.m), model_id)
        if step_nbr is not None:
            self.m.step_nbr = int(step_nbr)
        if params is not None:
            self.set_params(params)

    # -- parameters
    def set_params(self, params):
        for i, v in enumerate(params):
            self.m.p[i] = float(v)

    def set_param(self, name, v):
        names = INTERCEPTOR_PARAM_NAMES if self.m.model_id == MODEL_INTERCEPTOR else GODDARD_PARAM_NAMES
        self.m.p[names.index(name)] = float(v)

    def params(self):
        return np.array(self.m.p[:], dtype=np.float64)

    def set_integrator(self, kind, tol=1e-8):
        """0: fixed-step RK4; 1: adaptive Dormand-Prince for every trajectory the residual / FD Jacobian integrates."""
        self.m.integrator = int(kind)
        self.m.tol = float(tol)

    def set_switching(self, sw):
        self.m.nsw = len(sw)
        for i, v in enumerate(sw):
            self.m.sw[i] = float(v)

    @property
    def s(self):
        return 2 * self.m.dim

    def state_len(self, is_jac):
        return self.lib.orc_state_len(C.byref(self.m), int(is_jac))

    # -- model layer
    def rhs(self, t, X, is_jac=0):
        X = np.ascontiguousarray(X, dtype=np.float64)
        out = np.empty(len(X) if self.m.model_id == MODEL_DINT or not is_jac else self.s)
        if self.m.model_id == MODEL_GODDARD:
            out = np.empty(self.s)
        self.lib.orc_rhs(C.byref(self.m), C.c_double(t), _d(X), int(is_jac), _d(out))
        return out

    def control(self, t, X):
        X = np.ascontiguousarray(X, dtype=np.float64)
        u = np.empty(3)
        self.lib.orc_control(C.byref(self.m), C.c_double(t), _d(X), _d(u))
        return u[:self.lib.orc_control_dim(C.byref(self.m))].copy()

    def hamiltonian(self, t, X, is_jac=0):
        X = np.ascontiguousarray(X, dtype=np.float64)
        out = np.empty(self.s + 1 if is_jac else 1)
        self.lib.orc_hamiltonian(C.byref(self.m), C.c_double(t), _d(X), int(is_jac), _d(out))
        return out

    def singular_control(self, t, X):
        X = np.ascontiguousarray(X, dtype=np.float64)
        return self.lib.orc_goddard_singular_control(C.byref(self.m), C.c_double(t), _d(X))

    # -- ODE layer
    def rk4_step(self, t, X, step, is_jac=0):
        X = np.array(X, dtype=np.float64)
        self.lib.orc_rk4_step(C.byref(self.m), C.c_double(t), _d(X), C.c_double(step), int(is_jac))
        return X

    def traj(self, t0, X0, tf, is_jac=0):
        """model::ComputeTraj (for the interceptor: both stages + chart handling; sets chart/stage)."""
        X0 = np.ascontiguousarray(X0, dtype=np.float64)
        Xf = np.empty_like(X0)
        self.lib.orc_compute_traj(C.byref(self.m), C.c_double(t0), _d(X0), C.c_double(tf), int(is_jac), _d(Xf))
        return Xf

    # -- interceptor only
    def set_flags(self, chart, stage):
        self.m.chart, self.m.stage = int(chart), int(stage)

    def flags(self):
        return self.m.chart, self.m.stage

    def hamiltonian_at(self, t, X, u_beta):
        """interceptor H with the control held at (u, beta) instead of recomputed from X."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        ub = np.ascontiguousarray(u_beta, dtype=np.float64)
        self.lib.orc_interceptor_hamiltonian_at.restype = C.c_double
        return self.lib.orc_interceptor_hamiltonian_at(C.byref(self.m), C.c_double(t), _d(X), _d(ub))

    def chart12(self, X1):
        X1 = np.ascontiguousarray(X1, dtype=np.float64)
        X2 = np.empty(12)
        self.lib.orc_interceptor_chart12(C.byref(self.m), _d(X1), _d(X2))
        return X2

    def chart21(self, X2):
        X2 = np.ascontiguousarray(X2, dtype=np.float64)
        X1 = np.empty(12)
        self.lib.orc_interceptor_chart21(C.byref(self.m), _d(X2), _d(X1))
        return X1

    def lu6_solve(self, A, b):
        A = np.ascontiguousarray(A, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.empty(6)
        self.lib.orc_lu6_solve(_d(A), _d(b), _d(x))
        return x

    def traj_trace(self, t0, X0, tf):
        """interceptor ComputeTraj with the trace observer: (Xf, rows[k] = (t, X[12], chart, stage))."""
        rows = []
        OBS = C.CFUNCTYPE(None, C.c_void_p, C.c_double, _dp, C.c_int, C.c_int)

        def cb(_ctx, t, X, chart, stage):
            rows.append((t, np.array([X[i] for i in range(12)]), chart, stage))

        X0 = np.ascontiguousarray(X0, dtype=np.float64)
        Xf = np.empty(12)
        self.lib.orc_interceptor_compute_traj_obs(C.byref(self.m), C.c_double(t0), _d(X0), C.c_double(tf), _d(Xf), OBS(cb), None)
        return Xf, rows

    def init_analytical(self, ti, Xi, tf, Xf):
        Xi = np.array(Xi, dtype=np.float64)
        Xf = np.ascontiguousarray(Xf, dtype=np.float64)
        self.lib.orc_interceptor_init_analytical(C.byref(self.m), C.c_double(ti), _d(Xi), C.c_double(tf), _d(Xf))
        return Xi

    def traj_dopri5(self, t0, X0, tf, tol):
        """Adaptive Dormand-Prince segment, initial step (tf - t0)/stepNbr. Returns (Xf, accepted, rejected)."""
        X = np.array(X0, dtype=np.float64)
        rej = C.c_long(0)
        self.lib.orc_integrate_dopri5.restype = C.c_long
        n = self.lib.orc_integrate_dopri5(C.byref(self.m), _d(X), C.c_double(t0), C.c_double(tf),
                                          C.c_double((tf - t0) / self.m.step_nbr), C.c_double(tol), C.byref(rej))
        return X, n, rej.value

    def integrate_batch(self, t0, tf, X0, aux_sw=None, is_jac=0):
        X0 = np.ascontiguousarray(X0, dtype=np.float64)
        B = X0.shape[0]
        t0 = np.ascontiguousarray(np.broadcast_to(t0, (B,)), dtype=np.float64)
        tf = np.ascontiguousarray(np.broadcast_to(tf, (B,)), dtype=np.float64)
        Xf = np.empty_like(X0)
        aux = None
        if aux_sw is not None:
            aux_sw = np.ascontiguousarray(aux_sw, dtype=np.float64)
            aux = _d(aux_sw)
        self.lib.orc_integrate_batch(C.byref(self.m), B, _d(t0), _d(tf), aux, _d(X0), _d(Xf), int(is_jac))
        return Xf

    # -- shooting layer
    def timeline(self, prob, z):
        z = np.ascontiguousarray(z, dtype=np.float64)
        tl = np.empty(prob.M + 1)
        ps = prob.c_struct()
        self.lib.orc_compute_timeline(C.byref(self.m), C.byref(ps), _d(z), _d(tl))
        return tl

    def residual_block(self, which, t, X, other, mode, is_jac=0):
        """One default residual block of model.hpp:90-328 (BLOCK_* below), value or Jacobian form."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        other = np.ascontiguousarray(other, dtype=np.float64)
        mode = np.ascontiguousarray(mode, dtype=np.int32)
        out = np.empty(1024)
        k = self.lib.orc_residual_block(C.byref(self.m), int(which), C.c_double(t), _d(X), _d(other),
                                        mode.ctypes.data_as(C.POINTER(C.c_int)), int(is_jac), _d(out))
        return out[:k].copy()

    def residual(self, prob, z):
        z = np.ascontiguousarray(z, dtype=np.float64)
        assert z.shape == (prob.n,)
        F = np.empty(prob.n)
        ps = prob.c_struct()
        self.lib.orc_shooting_function(C.byref(self.m), C.byref(ps), _d(z), _d(F))
        return F

    def residual_batch(self, prob, Z):
        Z = np.ascontiguousarray(Z, dtype=np.float64)
        F = np.empty_like(Z)
        ps = prob.c_struct()
        self.lib.orc_residual_batch(C.byref(self.m), C.byref(ps), Z.shape[0], _d(Z), _d(F))
        return F

    def jacobian(self, prob, z):
        """Variational (hybrj) Jacobian, returned as J[row, col]."""
        z = np.ascontiguousarray(z, dtype=np.float64)
        J = np.empty((prob.n, prob.n))
        ps = prob.c_struct()
        self.lib.orc_shooting_jacobian(C.byref(self.m), C.byref(ps), _d(z), _d(J))
        return J

    def fdjac(self, prob, z, fvec=None, epsfcn=1e-15):
        """MINPACK fdjac1 Jacobian, returned as J[row, col]."""
        z = np.ascontiguousarray(z, dtype=np.float64)
        if fvec is None:
            fvec = self.residual(prob, z)
        fvec = np.ascontiguousarray(fvec, dtype=np.float64)
        Jcm = np.empty((prob.n, prob.n))
        ps = prob.c_struct()
        self.lib.orc_fdjac1(C.byref(self.m), C.byref(ps), _d(z), _d(fvec), C.c_double(epsfcn), _d(Jcm))
        return Jcm.T.copy()


BLOCK_INITIAL, BLOCK_INITIAL_H, BLOCK_FINAL, BLOCK_FINAL_H, BLOCK_SWITCHING_TIMES = 0, 1, 2, 3, 4


def have_ref():
    return os.path.exists(REF_SO)


class Ref:
    """The reference's own model objects (goddard / doubleIntegrator) behind oracle/ref_driver.cpp."""

    def __init__(self, model_id, step_nbr=10, model_order=0):
        self.lib = C.CDLL(REF_SO)
        L = self.lib
        L.ref_goddard_new.restype = C.c_void_p
        L.ref_dint_new.restype = C.c_void_p
        L.ref_goddard_traj_batch.restype = C.c_double
        self.model_id = model_id
        L.ref_covid_new.restype = C.c_void_p
        if model_id == MODEL_GODDARD:
            self.h = C.c_void_p(L.ref_goddard_new(int(step_nbr)))
        elif model_id == MODEL_COVID:
            self.h = C.c_void_p(L.ref_covid_new())
        else:
            self.h = C.c_void_p(L.ref_dint_new(int(model_order)))
        self.dim = L.ref_model_dim(self.h)
        self.s = 2 * self.dim

    def __del__(self):
        try:
            self.lib.ref_model_free(self.h)
        except Exception:
            pass

    def set_param(self, name, v):
        assert self.lib.ref_goddard_set(self.h, name.encode(), C.c_double(v)) == 0

    def set_params(self, params):
        if self.model_id == MODEL_GODDARD:
            for nme, v in zip(GODDARD_PARAM_NAMES, params):
                self.set_param(nme, v)
        elif self.model_id == MODEL_COVID:
            p = np.ascontiguousarray(params, dtype=np.float64)
            assert self.lib.ref_covid_set(self.h, _d(p)) == 0
        else:
            self.lib.ref_dint_set(self.h, C.c_double(params[0]), C.c_double(params[1]), C.c_double(params[2]))

    def set_switching(self, sw):
        sw = np.ascontiguousarray(sw, dtype=np.float64)
        self.lib.ref_model_switching_update(self.h, _d(sw), len(sw))

    def rhs(self, t, X, is_jac=0):
        X = np.ascontiguousarray(X, dtype=np.float64)
        out = np.empty(256)
        k = self.lib.ref_model_rhs(self.h, C.c_double(t), _d(X), len(X), int(is_jac), _d(out), 256)
        return out[:k].copy()

    def control(self, t, X):
        X = np.ascontiguousarray(X, dtype=np.float64)
        out = np.empty(8)
        k = self.lib.ref_model_control(self.h, C.c_double(t), _d(X), len(X), _d(out), 8)
        return out[:k].copy()

    def hamiltonian(self, t, X, is_jac=0):
        X = np.ascontiguousarray(X, dtype=np.float64)
        out = np.empty(64)
        k = self.lib.ref_model_hamiltonian(self.h, C.c_double(t), _d(X), len(X), int(is_jac), _d(out), 64)
        return out[:k].copy()

    def residual_block(self, which, t, X, other, mode, is_jac=0):
        """model::Initial[H]Function / Final[H]Function / SwitchingTimesFunction of the reference object."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        other = np.ascontiguousarray(other, dtype=np.float64)
        mode = np.ascontiguousarray(mode, dtype=np.int32)
        out = np.empty(1024)
        k = self.lib.ref_model_block(self.h, int(which), C.c_double(t), _d(X), len(X), _d(other), len(other),
                                     mode.ctypes.data_as(C.POINTER(C.c_int)), int(is_jac), _d(out), 1024)
        assert k >= 0
        return out[:k].copy()

    def rk4_step(self, t, X, step, is_jac=0):
        X = np.array(X, dtype=np.float64)
        self.lib.ref_rk4_step(self.h, C.c_double(t), _d(X), len(X), C.c_double(step), int(is_jac))
        return X

    def traj(self, t0, X0, tf, is_jac=0):
        X0 = np.ascontiguousarray(X0, dtype=np.float64)
        Xf = np.empty_like(X0)
        self.lib.ref_model_traj(self.h, C.c_double(t0), _d(X0), len(X0), C.c_double(tf), int(is_jac), _d(Xf))
        return Xf

    def goddard_traj_batch(self, threads, step_nbr, params, t0, tf, X0, cpus=None):
        """CPU baseline B1 (reference objects, one per thread; thread k pinned to logical CPU cpus[k] when given).
        Returns (Xf, seconds)."""
        X0 = np.ascontiguousarray(X0, dtype=np.float64)
        B = X0.shape[0]
        t0 = np.ascontiguousarray(np.broadcast_to(t0, (B,)), dtype=np.float64)
        tf = np.ascontiguousarray(np.broadcast_to(tf, (B,)), dtype=np.float64)
        params = np.ascontiguousarray(params, dtype=np.float64)
        Xf = np.empty_like(X0)
        if cpus is not None:
            cp = np.ascontiguousarray(cpus, dtype=np.int32)
            assert len(cp) >= threads
            self.lib.ref_goddard_traj_batch_pinned.restype = C.c_double
            sec = self.lib.ref_goddard_traj_batch_pinned(int(threads), cp.ctypes.data_as(_ip), int(step_nbr), _d(params), B, _d(t0), _d(tf),
                                                         _d(X0), _d(Xf))
        else:
            sec = self.lib.ref_goddard_traj_batch(int(threads), int(step_nbr), _d(params), B, _d(t0), _d(tf), _d(X0), _d(Xf))
        return Xf, sec
