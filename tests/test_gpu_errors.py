"""GPU: error behaviour of the C-ABI -- bad arguments and unsupported combinations are reported through
status codes and socp_last_error, never by crashing or by silently computing something else."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_status_codes_and_messages():
    from socp_amd import capi
    L = capi.lib()
    with pytest.raises(capi.SocpError) as e:
        capi.Context(99)
    assert e.value.code == capi.ERR_UNSUPPORTED
    ctx = capi.Context(capi.MODEL_GODDARD)
    with pytest.raises(capi.SocpError) as e:
        ctx.set_params([1.0, 2.0])                       # wrong parameter count
    assert e.value.code == capi.ERR_ARG and "parameter count" in str(e.value)
    with pytest.raises(capi.SocpError):
        ctx.set_step_number(0)
    buf = np.zeros(14)
    dp = buf.ctypes.data_as(C.POINTER(C.c_double))
    assert L.socp_residual_batch(ctx.h, 1, dp, dp) == capi.ERR_ARG          # no problem set yet
    assert b"no problem" in L.socp_last_error(ctx.h)
    # CONTINUOUS end node, bad mode value; a FREE interior state is accepted since round 4 (the model's SwitchingStateFunction hook,
    # tests/test_gpu_plugin.py)
    d = 7
    with pytest.raises(capi.SocpError):
        ctx.problem_set([capi.FIXED, capi.CONTINUOUS], np.zeros((2, d), dtype=np.int32), [0.0, 1.0], np.zeros((2, 14)))
    mx = np.zeros((3, d), dtype=np.int32)
    mx[1, 0] = capi.FREE
    assert ctx.problem_set([capi.FIXED, capi.CONTINUOUS, capi.FIXED], mx, [0.0, 0.5, 1.0], np.zeros((3, 14))) == 28
    mx[1, 0] = 7
    with pytest.raises(capi.SocpError):
        ctx.problem_set([capi.FIXED, capi.CONTINUOUS, capi.FIXED], mx, [0.0, 0.5, 1.0], np.zeros((3, 14)))
    # goddard has no variational equations (modelOrder 0)
    with pytest.raises(capi.SocpError) as e:
        ctx.integrate_batch(0.0, 0.1, np.zeros((1, 14)), is_jac=1)
    assert e.value.code == capi.ERR_UNSUPPORTED
    with pytest.raises(capi.SocpError):
        ctx.set_integrator(capi.INT_DOPRI5, 0.0)
    with pytest.raises(capi.SocpError):
        ctx.set_variant(17)
    with pytest.raises(capi.SocpError):
        ctx.set_variant(3)                               # the former "wave" value: not a variant of the state-only path
    # null pointers
    assert L.socp_integrate_batch(ctx.h, 4, None, None, None, None, None, 0) == capi.ERR_ARG
    assert L.socp_ctx_destroy(None) == capi.OK
    ctx.close()


def test_nan_and_empty_inputs_do_not_hang():
    """NaN states propagate (the reference does the same, SURVEY 8b 'Error conventions'); an empty batch
    is a no-op; the adaptive integrator poisons a row it cannot step instead of looping."""
    from socp_amd import capi
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_param("mu2", 1.0)
    X0 = np.full((3, 14), np.nan)
    X0[1] = [0.999949994, 1e-4, 0.01, 1e-10, 1e-10, 1e-10, 1.0, -8.12, 7.8e-3, 0.78, -0.48, 5.7e-4, 5.7e-2, 0.0996]
    Xf = ctx.integrate_batch(0.0, 0.1, X0)
    assert np.all(np.isnan(Xf[0])) and np.all(np.isfinite(Xf[1])) and np.all(np.isnan(Xf[2]))
    assert ctx.integrate_batch(0.0, 0.1, np.empty((0, 14))).shape == (0, 14)
    ctx.set_integrator(capi.INT_DOPRI5, 1e-8)
    Xa = ctx.integrate_batch(0.0, 0.1, X0)
    assert np.all(np.isnan(Xa[0])) and np.all(np.isfinite(Xa[1]))
    ctx.close()


def test_stalled_time_loop_terminates():
    """dt below the spacing of t: `t += dt` no longer advances and the reference's loop would spin forever
    (odeTools.cpp:136-145).  On the device the step counter stops it; the launch must return."""
    from socp_amd import capi
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_param("mu2", 1.0)
    ctx.set_step_number(10)
    X0 = np.array([[0.999949994, 1e-4, 0.01, 1e-10, 1e-10, 1e-10, 1.0, -8.12, 7.8e-3, 0.78, -0.48, 5.7e-4, 5.7e-2, 0.0996]])
    t0 = np.array([1.0e10])
    tf = t0 + 1.0e-7                      # spacing of doubles at 1e10 is 1.9e-6 > dt = 1e-8
    Xf = ctx.integrate_batch(t0, tf, X0)
    assert Xf.shape == (1, 14)
    ctx.close()


def test_selected_integrator_is_the_one_that_runs():
    """an entry point must never return fixed-step numbers with SOCP_OK while the adaptive integrator is selected.
    Dense output follows the selected integrator since round 3 (under the adaptive one its rows are the accepted steps,
    tests/test_gpu_parity.py); the variational Jacobian since round 4 (tests/test_gpu_variational.py) -- round 3 refused it."""
    from socp_amd import capi
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_param("mu2", 1.0)
    X0 = np.array([0.999949994, 1e-4, 0.01, 1e-10, 1e-10, 1e-10, 1.0, -8.12, 7.8e-3, 0.78, -0.48, 5.7e-4, 5.7e-2, 0.0996])
    rows, _ = ctx.integrate_dense(0.0, 0.01, X0)
    assert len(rows) == 11
    ctx.set_integrator(capi.INT_DOPRI5, 1e-8)
    t_ad, x_ad = ctx.integrate_dense(0.0, 0.01, X0)
    assert t_ad[0] == 0.0 and t_ad[-1] == 0.01 and len(t_ad) != 11 and np.all(np.isfinite(x_ad))      # not the fixed-step rows
    ctx.close()
    di = capi.Context(capi.MODEL_DOUBLE_INTEGRATOR)
    mx = np.zeros((2, 6), dtype=np.int32)
    n = di.problem_set([capi.FIXED, capi.FIXED], mx, [0.0, 1.0], np.zeros((2, 12)))
    z = np.full(n, 0.01)
    J = di.var_jacobian(z)
    assert J.shape == (n, n) and np.isfinite(J).all()
    di.set_integrator(capi.INT_DOPRI5, 1e-8)
    Ja = di.var_jacobian(z)
    assert np.isfinite(Ja).all() and not np.array_equal(Ja, J) and np.max(np.abs(Ja - J)) <= 1e-4 * np.max(np.abs(J))
    di.close()


def test_multistart_from_a_fresh_thread_uses_the_context_device():
    """socp_multistart_solve allocates staging buffers and a second stream; it must do so on the CONTEXT's device
    whatever the calling thread's current device is.  One GPU here, so what can be checked is that a call from a new
    thread (which starts on device 0 with no prior hipSetDevice) works and leaves the caller's device alone; the context
    reports the device it lives on."""
    import threading
    from socp_amd import capi, sweep
    ctx = capi.Context(capi.MODEL_GODDARD, device=0)
    assert capi.lib().socp_ctx_device(ctx.h) == 0
    ctx.set_params(sweep.GODDARD_PARAMS)
    ctx.set_step_number(50)
    sweep.goddard_single_shooting_problem(ctx)
    Z0 = sweep.goddard_starts(8, 1e-3)
    ref = ctx.multistart_solve(Z0, xtol=1e-8)
    got = {}
    th = threading.Thread(target=lambda: got.update(ctx.multistart_solve(Z0, xtol=1e-8)))
    th.start(); th.join()
    assert np.array_equal(got["z"], ref["z"]) and np.array_equal(got["info"], ref["info"])
    ctx.close()


def test_chain_engine_out_of_memory_exits_cleanly():
    """The engine's error exits: with the card's memory taken, the device-solver engine cannot place its 2 GB of
    solver state -> SOCP_ERR_HIP, nothing leaked, the context's stream as it was and the context usable; the host-solver engine,
    asked for speculative FD rows that no longer fit, drops them, retries and solves (no iterate depends on speculation)."""
    import ctypes as C
    import torch
    from socp_amd import capi, sweep
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_params(sweep.GODDARD_PARAMS)
    ctx.set_step_number(10)
    ctx.set_variant(capi.VARIANT_LANE_FAST)
    assert sweep.goddard_multiple_shooting_problem(ctx, 6) == 85
    P = 20000
    Z0 = sweep.goddard_multiple_shooting_starts(ctx, sweep.goddard_starts(P, 0.05), 6)
    L = capi.lib()
    L.socp_ctx_get_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    before = C.c_void_p()
    assert L.socp_ctx_get_stream(ctx.h, C.byref(before)) == 0
    capi.workspace_release()                       # (the engine keeps its arenas between calls: this test is about a call that finds no memory)
    free0, total = torch.cuda.mem_get_info()
    #  the test owns the card: it sizes its reservation from what is free NOW, so it is meaningless -- and would fail for
    # reasons that are not the code's -- while another process holds a large part of the memory; the GPU suite runs in one process
    if free0 < 0.8 * total:
        pytest.skip("another process is using the card's memory (%.0f of %.0f GB free)" % (free0 / 2**30, total / 2**30))
    # what the host engine needs without its optional buffers: six rows and a state machine per chain on the device side plus
    # ONE chunk of Jacobians (it shrinks the chunk until it fits); leave 1.2 x that, at least 0.5 GB
    n = 85
    host_floor = P * 8 * n * 8 + 64 * n * n * 8
    leave = max(500 << 20, int(1.2 * host_floor))
    hog = torch.empty(free0 - leave, dtype=torch.uint8, device="cuda")
    with pytest.raises(capi.SocpError) as e:
        ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-8, solver=capi.SOLVER_DEVICE)     # needs 20000 x 100 KB of solver state
    assert e.value.code == capi.ERR_HIP
    # AUTO would have taken the device engine for this sweep (P n^2 = 1.4e8); with the memory gone it takes the host engine instead
    # of failing (its estimate is the device engine's own allocation plan; were the estimate to pass and the allocation to fail
    # after all, AUTO falls through to the host engine as well -- batchsolve.cpp)
    auto = ctx.chains_solve(Z0[:4000], kind=capi.CHAIN_PLAIN, xtol=1e-8)
    assert np.all(auto["info"] == 1)
    after = C.c_void_p()
    assert L.socp_ctx_get_stream(ctx.h, C.byref(after)) == 0 and after.value == before.value
    assert np.all(np.isfinite(ctx.residual(Z0[0])))                                 # the context still works
    # host solvers: the speculation cache (2 x 1.2 GB) does not fit either; the engine retries without it
    r = ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-8, solver=capi.SOLVER_HOST, speculate=1)
    assert np.all(r["info"] == 1) and r["stats"]["speculative_rounds"] == 0 and r["stats"]["jacobians_from_cache"] == 0
    del hog
    torch.cuda.empty_cache()
    capi.workspace_release()                       # (what the engine calls that DID run have kept)
    free1, _ = torch.cuda.mem_get_info()
    assert free1 >= free0 - (64 << 20)                                              # nothing of the failed call is still allocated
    ok = ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-8, solver=capi.SOLVER_DEVICE)
    assert np.array_equal(ok["z"], r["z"]) and np.array_equal(ok["nfev"], r["nfev"])
    ctx.close()


def test_context_second_stream_is_created_once_and_left_alone():
    """socp_ctx_aux_stream: the same stream on every call, not the context's launch stream; the lock-step engines use it while they
    run and put the context back on its own stream; a cloned context has its own."""
    from socp_amd import capi, sweep
    ctx = capi.Context(capi.MODEL_GODDARD)
    L = capi.lib()
    L.socp_ctx_get_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    main = C.c_void_p()
    assert L.socp_ctx_get_stream(ctx.h, C.byref(main)) == 0
    a, b = ctx.aux_stream(), ctx.aux_stream()
    assert a == b and a is not None and a != main.value
    ctx.set_params(sweep.GODDARD_PARAMS)
    ctx.set_step_number(20)
    ctx.set_variant(capi.VARIANT_LANE_FAST)
    sweep.goddard_single_shooting_problem(ctx)
    Z0 = sweep.goddard_starts(16, 1e-3)
    for solver in (capi.SOLVER_HOST, capi.SOLVER_DEVICE):
        r = ctx.chains_solve(Z0, kind=capi.CHAIN_PLAIN, xtol=1e-8, solver=solver)
        assert np.all(r["info"] == 1)
        now = C.c_void_p()
        assert L.socp_ctx_get_stream(ctx.h, C.byref(now)) == 0 and now.value == main.value
        assert ctx.aux_stream() == a
    L.socp_ctx_clone.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    twin = C.c_void_p()
    assert L.socp_ctx_clone(ctx.h, 0, C.byref(twin)) == 0
    st = C.c_void_p()
    assert L.socp_ctx_aux_stream(twin, C.byref(st)) == 0 and st.value not in (None, a)
    assert L.socp_ctx_destroy(twin) == 0
    ctx.close()
