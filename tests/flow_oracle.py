"""Test infrastructure: the reference's host flow (shooting.cpp set-up, continuation, solution
read-back) restated in Python over the CPU oracle, with the Newton solve done by SciPy's MINPACK
(`solver="scipy"`) or by the library's own hybrd (`solver="socp"`).  Used to generate and check the
golden solutions of the reference's test programs.  Not product code.

Citations: shooting.cpp:165-291 (SetMode/InitShooting), :568-595 (SolveShooting), :695-778
(parameter continuation), :598-692 (data continuation), :383-437 (Move(tf)), :1462-1508
(UpdateSolution).
"""
import numpy as np
from scipy.optimize import _minpack

from oracle.oracle import Oracle, Problem, FIXED, FREE, CONTINUOUS, MODEL_GODDARD, MODEL_DINT


class OracleShooting:
    def __init__(self, orc, M, solver="scipy", use_jac=False):
        self.o, self.M, self.d = orc, M, orc.m.dim
        self.s = 2 * self.d
        self.solver, self.use_jac = solver, use_jac
        self.xtol, self.maxfev, self.epsfcn, self.factor = 1e-8, 10000, 1e-15, 1.0
        self.step_min = 1e-12
        self.nfev = self.njev = 0
        self.total_fev = 0

    # -- set-up
    def set_mode_final(self, mode_tf, mode_xf):
        M, d = self.M, self.d
        self.mode_t = [FIXED] + [CONTINUOUS] * (M - 1) + [mode_tf]
        self.mode_x = np.full((M + 1, d), CONTINUOUS, dtype=np.int32)
        self.mode_x[0] = FIXED
        self.mode_x[M] = mode_xf
        self._resize()

    def set_mode(self, mode_t, mode_x):
        self.mode_t = list(mode_t)
        self.mode_x = np.array(mode_x, dtype=np.int32)
        self._resize()

    def _resize(self):
        self.n = self.s * self.M + sum(1 for m in self.mode_t if m == FREE)

    def _pack(self):
        z = np.empty(self.n)
        z[:self.s * self.M] = self.X[:self.M].ravel()
        z[self.s * self.M:] = [self.time[j] for j in range(self.M + 1) if self.mode_t[j] == FREE]
        self.z = z

    def init_uniform(self, ti, Xi, tf, Xf):
        M = self.M
        self.time = np.array([ti + i * (tf - ti) / M for i in range(M + 1)])
        self.X = np.zeros((M + 1, self.s))
        self.X[0], self.X[M] = Xi, Xf
        for i in range(1, M):
            self.X[i] = self.o.traj(ti, Xi, self.time[i])
        self.timed, self.time_prec = self.time.copy(), self.time.copy()
        self.Xd, self.X_prec = self.X.copy(), self.X.copy()
        self._pack()

    def init_nodes(self, vt, vX):
        self.time = np.array(vt, dtype=float)
        self.X = np.array(vX, dtype=float)
        self.timed, self.time_prec = self.time.copy(), self.time.copy()
        self.Xd, self.X_prec = self.X.copy(), self.X.copy()
        self._pack()

    def set_desired(self, vt, vX):
        self.timed = np.array(vt, dtype=float)
        self.Xd = np.array(vX, dtype=float)

    # -- residual / solve
    def problem(self):
        return Problem(self.d, self.mode_t, self.mode_x, self.time, self.X)

    def _solve(self, z0):
        prob = self.problem()
        f = lambda z: self.o.residual(prob, z)
        if self.solver == "scipy":
            if self.use_jac:
                r = _minpack._hybrj(f, lambda z: self.o.jacobian(prob, z), z0.copy(), (), 1, 0, self.xtol, self.maxfev,
                                    self.factor, None)
                self.njev = r[1]["njev"]
            else:
                r = _minpack._hybrd(f, z0.copy(), (), 1, self.xtol, self.maxfev, -10, -10, self.epsfcn, self.factor, None)
            z, info = r[0], r[2]
            self.nfev = r[1]["nfev"]
        else:
            from socp_amd import capi
            if self.use_jac:
                out = capi.hybrj(f, lambda z: self.o.jacobian(prob, z), z0, xtol=self.xtol, maxfev=self.maxfev, factor=self.factor)
                self.njev = out["njev"]
            else:
                out = capi.hybrd(f, z0, xtol=self.xtol, maxfev=self.maxfev, epsfcn=self.epsfcn, factor=self.factor)
            z, info, self.nfev = out["x"], out["info"], out["nfev"]
        self.total_fev += self.nfev
        return z, info

    def solve(self):
        """SolveOCP(0.0)"""
        self.time = self.timed.copy()
        self.X[:, :self.d] = self.Xd[:, :self.d]
        z, info = self._solve(self.z)
        if info == 1:
            self.z = z
        return info

    def solve_param(self, step, set_param, start, goal):
        """SolveOCP(step, Rdata, Rgoal): homotopy on one model parameter with step bisection."""
        step = 1.0 if step <= 0 else step
        b, b_prec = min(step, 1.0), 0.0
        set_param((1 - b) * start + b * goal)
        self.time = self.timed.copy()
        self.X[:, :self.d] = self.Xd[:, :self.d]
        zt = self.z.copy()
        self.stage_fev = 0
        while True:
            zt, info = self._solve(zt)
            self.stage_fev += self.nfev
            if info != 1:
                stop = abs(b - b_prec) < self.step_min
                b = b_prec + (b - b_prec) / 2
                zt = self.z.copy()
                set_param((1 - b) * start + b * goal)
                if stop:
                    return info
            elif b == 1:
                self.z = zt
                return 1
            else:
                b_prec, b = b, min(b + step, 1.0)
                self.z = zt.copy()
                set_param((1 - b) * start + b * goal)

    def solve_data(self, step):
        """SolveOCP(step > 0): homotopy on the boundary data."""
        b, b_prec = min(step, 1.0), 0.0

        def blend(bb):
            self.time = (1 - bb) * self.time_prec + bb * self.timed
            self.X[:, :self.d] = (1 - bb) * self.X_prec[:, :self.d] + bb * self.Xd[:, :self.d]
        blend(b)
        zt = self.z.copy()
        while True:
            zt, info = self._solve(zt)
            if info != 1:
                stop = abs(b - b_prec) < self.step_min
                b = b_prec + (b - b_prec) / 2
                zt = self.z.copy()
                blend(b)
                if stop:
                    return info
            elif b == 1:
                self.z = zt
                self.time_prec = self.timed.copy()
                self.X_prec[:, :self.d] = self.Xd[:, :self.d]
                return 1
            else:
                b_prec, b = b, min(b + step, 1.0)
                self.z = zt.copy()
                blend(b)

    # -- read-back
    def timeline(self):
        return self.o.timeline(self.problem(), self.z)

    def move(self, tf):
        """Move(tf): state on the stored trajectory at time tf."""
        M, s = self.M, self.s
        t0 = self.time[0] if self.mode_t[0] == FIXED else self.z[s * M]
        t_end = self.time[M] if self.mode_t[M] == FIXED else self.z[self.n - 1]
        target = tf if (t0 <= tf <= t_end) else t_end
        tl = self.timeline()
        seg = 0
        while tl[seg + 1] < target:
            seg += 1
        node = seg if 0 < seg < M else 0
        return self.o.traj(tl[seg], self.z[s * node:s * (node + 1)], target)

    def get_solution(self):
        M, s = self.M, self.s
        tl = self.timeline()
        t_end = self.time[M] if self.mode_t[M] == FIXED else self.z[self.n - 1]
        X1 = self.z[:s].copy()
        for i in range(M + 1):
            self.time[i] = tl[i]
            self.X[i] = X1
            if i < M - 1:
                X1 = self.z[s * (i + 1):s * (i + 2)].copy()
            elif i == M - 1:
                X1 = self.move(t_end)
        return self.time.copy(), self.X.copy()


def goddard_test_flow(solver="scipy", step_nbr=10, M=6):
    """tests/testGoddard.cpp:24-156 on the oracle.  Returns a list of stage dicts."""
    o = Oracle(MODEL_GODDARD, step_nbr=step_nbr)
    o.set_param("mu2", 1.0)
    sh = OracleShooting(o, M, solver)
    sh.xtol = 1e-6
    mode_xf = np.zeros(7, dtype=np.int32)
    mode_xf[3:7] = FREE
    sh.set_mode_final(FREE, mode_xf)
    Xi = np.array([0.999949994, 1e-4, 0.01, 1e-10, 1e-10, 1e-10, 1.0] + [0.1] * 7)
    Xf = np.zeros(14)
    Xf[0] = 1.01
    sh.init_uniform(0.0, Xi, 0.1, Xf)
    stages = []
    o.set_param("KD", 0.0)
    info = sh.solve()
    stages.append(dict(stage="no_drag", info=info, nfev=sh.nfev, z=sh.z.copy()))
    info = sh.solve_param(1.0, lambda v: o.set_param("KD", v), 0.0, 310.0)
    stages.append(dict(stage="drag_continuation", info=info, nfev=sh.stage_fev, z=sh.z.copy()))
    info = sh.solve_param(1.0, lambda v: o.set_param("mu2", v), 1.0, 0.2)
    stages.append(dict(stage="mu2_continuation", info=info, nfev=sh.stage_fev, z=sh.z.copy()))
    if M != 6:
        return stages
    vt, vX = sh.get_solution()
    tf = vt[M]
    s1, s2 = 0.0227, 0.08
    vt = np.array([0.0, s1 / 2, s1, (s2 + s1) / 2, s2, (s2 + tf) / 2, tf])
    vX = np.stack([sh.move(t) for t in vt])
    mode_t = [FIXED, CONTINUOUS, FREE, CONTINUOUS, FREE, CONTINUOUS, FREE]
    mode_x = np.full((M + 1, 7), CONTINUOUS, dtype=np.int32)
    mode_x[0] = FIXED
    mode_x[M] = mode_xf
    sh.set_mode(mode_t, mode_x)
    sh.init_nodes(vt, vX)
    o.set_param("mu2", 0.0)
    o.set_param("singularControl", -1.0)
    info = sh.solve()
    stages.append(dict(stage="singular_arc", info=info, nfev=sh.nfev, z=sh.z.copy()))
    return stages


def goddard_initial_guess(step_nbr=10, M=6):
    """The unknown vector testGoddard.cpp holds after InitShooting (interior nodes seeded with KD = 310)."""
    o = Oracle(MODEL_GODDARD, step_nbr=step_nbr)
    o.set_param("mu2", 1.0)
    sh = OracleShooting(o, M, "scipy")
    mode_xf = np.zeros(7, dtype=np.int32)
    mode_xf[3:7] = FREE
    sh.set_mode_final(FREE, mode_xf)
    Xi = np.array([0.999949994, 1e-4, 0.01, 1e-10, 1e-10, 1e-10, 1.0] + [0.1] * 7)
    Xf = np.zeros(14)
    Xf[0] = 1.01
    sh.init_uniform(0.0, Xi, 0.1, Xf)
    return sh.z.copy()


def goddard_single_stage(k, z_init, xtol, solver="scipy", step_nbr=10, backend=None):
    """One solve of the testGoddard flow (k = 1..4) started from the unknowns z_init -- the oracle-side
    twin of `goddard_flow stage k ...` (tests/cpp/goddard_flow.cpp).  `backend`: optional object with the
    Oracle interface (set_param / traj / residual / timeline ...) to run the same flow on another engine."""
    M = 6
    o = backend if backend is not None else Oracle(MODEL_GODDARD, step_nbr=step_nbr)
    o.set_param("mu2", 1.0)
    o.set_param("KD", 310.0)
    sh = OracleShooting(o, M, solver)
    sh.xtol = xtol
    mode_xf = np.zeros(7, dtype=np.int32)
    mode_xf[3:7] = FREE
    sh.set_mode_final(FREE, mode_xf)
    Xf = np.zeros(14)
    Xf[0] = 1.01
    tf = z_init[-1]
    vt = np.array([0.0 + i * (tf - 0.0) / M for i in range(M + 1)])
    sh.init_nodes(vt, np.vstack([np.asarray(z_init[:84]).reshape(M, 14), Xf]))
    if k == 1:
        o.set_param("KD", 0.0)
        info = sh.solve()
        nfev = sh.nfev
    elif k == 2:
        o.set_param("KD", 0.0)
        info = sh.solve_param(1.0, lambda v: o.set_param("KD", v), 0.0, 310.0)
        nfev = sh.stage_fev
    elif k == 3:
        info = sh.solve_param(1.0, lambda v: o.set_param("mu2", v), 1.0, 0.2)
        nfev = sh.stage_fev
    else:
        o.set_param("mu2", 0.2)
        vt, vX = sh.get_solution()
        tf = vt[M]
        s1, s2 = 0.0227, 0.08
        vt = np.array([0.0, s1 / 2, s1, (s2 + s1) / 2, s2, (s2 + tf) / 2, tf])
        vX = np.stack([sh.move(t) for t in vt])
        mode_x = np.full((M + 1, 7), CONTINUOUS, dtype=np.int32)
        mode_x[0] = FIXED
        mode_x[M] = mode_xf
        sh.set_mode([FIXED, CONTINUOUS, FREE, CONTINUOUS, FREE, CONTINUOUS, FREE], mode_x)
        sh.init_nodes(vt, vX)
        o.set_param("mu2", 0.0)
        o.set_param("singularControl", -1.0)
        info = sh.solve()
        nfev = sh.nfev
    return dict(info=int(info), nfev=int(nfev), z=sh.z.copy())


def dint_basic_flow(solver="scipy", model_order=1, xtol=1e-8):
    """tests/testDoubleIntegrator.cpp:24-145 on the oracle (hybrj when model_order == 1)."""
    o = Oracle(MODEL_DINT)
    sh = OracleShooting(o, 1, solver, use_jac=bool(model_order))
    sh.xtol = xtol
    sh.set_mode_final(FREE, np.zeros(6, dtype=np.int32))
    Xi = np.zeros(12)
    Xi[6:] = 0.01
    Xf = np.zeros(12)
    Xf[0], Xf[1] = 10.0, 15.0
    sh.init_uniform(0.0, Xi, 10.0, Xf)
    out = []
    info = sh.solve()
    out.append(dict(stage="solve", info=int(info), nfev=int(sh.nfev), njev=int(sh.njev), z=sh.z.copy()))
    Xf2 = Xf.copy()
    Xf2[1] = 20.0
    sh.timed = np.array([0.0, 10.0])
    sh.Xd = np.vstack([Xi, Xf2])
    info = sh.solve_data(1.0)
    out.append(dict(stage="data_continuation", info=int(info), nfev=int(sh.nfev), njev=int(sh.njev), z=sh.z.copy()))
    if info == 1:
        info = sh.solve_param(1.0, lambda v: o.m.p.__setitem__(2, v), 0.01, 0.02)
    out.append(dict(stage="muT_continuation", info=int(info), nfev=int(sh.nfev), njev=int(sh.njev), z=sh.z.copy()))
    return out


def dint_wp_flow(solver="scipy", model_order=1, xtol=1e-8, M=2):
    """tests/testDoubleIntegrator_WP.cpp:26-152 on the oracle."""
    o = Oracle(MODEL_DINT)
    sh = OracleShooting(o, M, solver, use_jac=bool(model_order))
    sh.xtol = xtol
    mode_t = [FIXED] + [FREE] * M
    mode_x = np.zeros((M + 1, 6), dtype=np.int32)
    mode_x[1:M, 3:6] = CONTINUOUS
    sh.set_mode(mode_t, mode_x)
    vt = np.array([60.0 * i / M for i in range(M + 1)])
    vX = np.zeros((M + 1, 12))
    for i in range(M + 1):
        vX[i, 0] = 20.0 * i / M
        if i < M:
            vX[i, 6:] = 0.001
    sh.init_nodes(vt, vX)
    out = []
    info = sh.solve()
    out.append(dict(stage="solve", info=int(info), nfev=int(sh.nfev), njev=int(sh.njev), z=sh.z.copy()))
    if M == 2:
        vX2 = vX.copy()
        vX2[1, 1], vX2[2, 1], vX2[2, 2] = 15.0, 5.0, 10.0
        sh.set_desired(vt, vX2)
        info = sh.solve_data(1.0)
        out.append(dict(stage="data_continuation", info=int(info), nfev=int(sh.nfev), njev=int(sh.njev), z=sh.z.copy()))
        if info == 1:
            info = sh.solve_param(1.0, lambda v: o.m.p.__setitem__(2, v), 0.01, 0.02)
        out.append(dict(stage="muT_continuation", info=int(info), nfev=int(sh.nfev), njev=int(sh.njev), z=sh.z.copy()))
    return out


def covid_flow(solver="scipy", xtol=1e-8, stages=3, step_nbr=1000):
    """tests/testCovid19.cpp:30-108 on the oracle: M = 20, fixed tf, S/E/I free at tf, R pinned; solve,
    then data continuation of the target R(tf) 0.6 -> 0.7 in steps of 0.1 and of tf 30 -> 365 in steps of
    0.01 (shooting.cpp:598-692)."""
    from oracle.oracle import MODEL_COVID
    o = Oracle(MODEL_COVID, step_nbr=step_nbr)
    o.m.p[0], o.m.p[1], o.m.p[2] = 3.4, 14.0, 5.0          # R0, Tinf, Tinc (testCovid19.cpp:41-43)
    M = 20
    sh = OracleShooting(o, M, solver)
    sh.xtol = xtol
    sh.set_mode_final(FIXED, np.array([FREE, FREE, FREE, FIXED], dtype=np.int32))
    Xi = np.array([0.93, 0.003, 0.01, 0.057, -0.001, 0.001, 0.0, 0.0])
    Xf = np.zeros(8)
    Xf[3] = 0.6
    sh.init_uniform(0.0, Xi, 30.0, Xf)
    out = []
    info = sh.solve()
    out.append(dict(stage="solve", info=int(info), nfev=int(sh.nfev), z=sh.z.copy()))
    if stages > 1:
        Xf2 = Xf.copy()
        Xf2[3] = 0.7
        sh.timed[0], sh.timed[M] = 0.0, 30.0
        sh.Xd[0], sh.Xd[M] = Xi, Xf2
        info = sh.solve_data(0.1)
        out.append(dict(stage="target_continuation", info=int(info), nfev=int(sh.nfev), z=sh.z.copy()))
    if stages > 2:
        sh.timed[0], sh.timed[M] = 0.0, 365.0
        sh.Xd[0], sh.Xd[M] = Xi, Xf2
        info = sh.solve_data(0.01)
        out.append(dict(stage="horizon_continuation", info=int(info), nfev=int(sh.nfev), z=sh.z.copy()))
    return out


def interceptor_flow(solver="scipy", xtol=1e-8, scenario=1):
    """tests/testInterceptor.cpp on the oracle (PARITY UNPINNED, oracle/interceptor_oracle.c): analytical guess
    at mu_gft = 0 -> Newton solve -> continuation of mu_gft to 1 (step 0.1) -> continuation of the boundary data
    to the scenario (step 0.1).  M = 1, final time and final velocity free: n = 13."""
    from oracle.oracle import MODEL_INTERCEPTOR
    RE = 6378145.0
    d = 6
    mode_xf = np.zeros(d, dtype=np.int32)
    mode_xf[1] = FREE
    out = []
    # initState(): testInterceptor.cpp:165-218
    o = Oracle(MODEL_INTERCEPTOR)
    sh = OracleShooting(o, 1, solver)
    sh.xtol = xtol
    sh.set_mode_final(FREE, mode_xf)
    X0 = np.zeros(12)
    X0[:6] = [1000, 1000, np.pi / 4, 0.0, 5454661 / RE, 46086 / RE]
    X1 = np.zeros(12)
    X1[:6] = [6000, 1000, 0.01 * np.pi, 0.01 * np.pi, (5454661 + 27829.0) / RE, 46086 / RE]
    o.set_param("mu_gft", 0.0)
    X0 = o.init_analytical(0.0, X0, 10.0, X1)
    sh.init_uniform(0.0, X0, 10.0, X1)
    info = sh.solve()
    out.append(dict(stage="analytical_guess", info=int(info), nfev=int(sh.nfev), z=sh.z.copy()))
    if info == 1:
        info = sh.solve_param(0.1, lambda v: o.set_param("mu_gft", v), 0.0, 1.0)
        out.append(dict(stage="mu_gft_continuation", info=int(info), nfev=int(sh.nfev), z=sh.z.copy()))
    if info != 1:
        return out
    vt, vX = sh.get_solution()
    # solve(): testInterceptor.cpp:117-160
    Xi = np.zeros(12)
    Xf = np.zeros(12)
    Xi[:6] = [3000, 1000, 0.0, 0.0, 5454661 / RE, 46086 / RE]
    Xf[1] = 1000
    if scenario == 1:
        Xi[2] = -np.pi / 6
        Xf[0], Xf[2], Xf[3], Xf[4], Xf[5] = 12000, 0.0, np.pi / 8, 5475000 / RE, 42000 / RE
    elif scenario == 2:
        Xi[2] = np.pi / 4
        Xf[0], Xf[2], Xf[3], Xf[4], Xf[5] = 12000, -np.pi / 4, -np.pi / 2, 5485000 / RE, 36178 / RE
    else:
        Xi[2] = 0.0
        Xf[0], Xf[2], Xf[3], Xf[4], Xf[5] = 3000, 0.0, 0.0, 5485000 / RE, 46086 / RE
    o2 = Oracle(MODEL_INTERCEPTOR)
    sh2 = OracleShooting(o2, 1, solver)
    sh2.xtol = xtol
    sh2.set_mode_final(FREE, mode_xf)
    sh2.init_uniform(vt[0], vX[0], vt[1], vX[1])
    sh2.set_desired([0.0, 20.0], [Xi, Xf])
    info = sh2.solve_data(0.1)
    out.append(dict(stage="scenario_continuation", info=int(info), nfev=int(sh2.nfev), z=sh2.z.copy()))
    return out
