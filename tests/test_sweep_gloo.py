"""CPU, world_size 2, gloo: the multi-start sweep's sharding and gather (socp_amd/sweep.py).  The local
solve is played by the CPU oracle + the library's hybrd here (tests may use the oracle; the product
path calls socp_multistart_solve on the GPU) -- what is under test is the N > 1 plumbing: contiguous
shards, no exchange until the end, one all_gather of fixed-size records, start order preserved."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_solver(step_nbr):
    from oracle.oracle import Oracle, Problem, MODEL_GODDARD, FIXED, FREE
    from socp_amd import capi, sweep
    o = Oracle(MODEL_GODDARD, step_nbr=step_nbr, params=sweep.GODDARD_PARAMS)
    mode_x = np.zeros((2, 7), dtype=np.int32)
    mode_x[1, 3:7] = FREE
    X = np.zeros((2, 14))
    X[0, :7] = sweep.X0_STATE
    X[1, 0] = 1.01
    prob = Problem(7, [FIXED, FIXED], mode_x, np.array([0.0, sweep.TF]), X)

    def solve(Zb):
        z, info, nfev, fn = [], [], [], []
        for z0 in Zb:
            r = capi.hybrd(lambda v: o.residual(prob, v), z0, xtol=1e-8, epsfcn=1e-15)
            z.append(r["x"]); info.append(r["info"]); nfev.append(r["nfev"]); fn.append(np.linalg.norm(r["fvec"]))
        return dict(z=np.array(z).reshape(-1, 14), info=np.array(info, dtype=np.int32), nfev=np.array(nfev, dtype=np.int32),
                    fnorm=np.array(fn), rounds=0)
    return solve


def _worker(rank, world, port, P, out_dir, gather="tensor"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if gather == "list":
        os.environ["SOCP_SWEEP_GATHER"] = "list"
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from socp_amd import sweep
    Z0 = sweep.goddard_starts(P, 1e-3)
    table, local = sweep.run_sweep(Z0, _oracle_solver(20), dist)
    np.save(os.path.join(out_dir, "table_%d.npy" % rank), table)
    np.save(os.path.join(out_dir, "count_%d.npy" % rank), np.array([len(local["info"])]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_is_a_partition():
    from socp_amd.sweep import shard
    for P in (0, 1, 7, 8, 4096, 4099):
        for W in (1, 2, 3, 8):
            blocks = [shard(P, r, W) for r in range(W)]
            assert blocks[0][0] == 0 and blocks[-1][1] == P
            assert all(blocks[r][1] == blocks[r + 1][0] for r in range(W - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_c_abi_shard_is_the_same_partition():
    """socp_sweep_shard (what the C++ multi-GPU entry points use) == socp_amd.sweep.shard."""
    import ctypes as C
    from socp_amd import capi
    from socp_amd.sweep import shard
    L = capi.lib()
    L.socp_sweep_shard.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.socp_sweep_shard.restype = None
    for P in (0, 1, 7, 8, 301, 4096, 65536):
        for W in (1, 2, 3, 8, 16):
            for r in range(W):
                lo, hi = C.c_int(-1), C.c_int(-1)
                L.socp_sweep_shard(P, r, W, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == shard(P, r, W), (P, W, r)


def test_two_rank_sweep_matches_single_process(tmp_path, built):
    P, world = 7, 2                      # odd: the shards differ in size (4 + 3)
    mp.spawn(_worker, args=(world, 29613, P, str(tmp_path)), nprocs=world, join=True)
    t0 = np.load(tmp_path / "table_0.npy")
    t1 = np.load(tmp_path / "table_1.npy")
    assert np.array_equal(t0, t1) and t0.shape == (P, 17)             # every rank holds the full table
    assert [int(np.load(tmp_path / ("count_%d.npy" % r))[0]) for r in range(world)] == [4, 3]
    from socp_amd import sweep
    single, _ = sweep.run_sweep(sweep.goddard_starts(P, 1e-3), _oracle_solver(20), None)
    assert np.array_equal(single, t0)                                 # same records, in start order
    assert np.all(t0[:, -2] == 1)                                     # eps = 1e-3 is inside the basin


@pytest.mark.parametrize("gather", ["tensor", "list"])
def test_eight_rank_sweep_with_an_odd_start_count(tmp_path, built, gather):
    """the shape of the driver's 8-GPU job, rehearsed where eight ranks can run -- on the CPU (a one-GPU box admits six
    processes on its card): world_size 8, an odd total of 11 starts (blocks of 2, 2, 2, 1, 1, 1, 1, 1: shorter than the padded
    gather record), one collective, every rank ends with the same table in start order, equal to a single process's.  Both forms of
    the gather (the form is chosen BEFORE communicating, from facts every rank shares -- here forced through the
    environment to cover the list form too)."""
    P, world = 11, 8
    mp.spawn(_worker, args=(world, 29641 if gather == "tensor" else 29643, P, str(tmp_path), gather), nprocs=world, join=True)
    tables = [np.load(tmp_path / ("table_%d.npy" % r)) for r in range(world)]
    assert all(np.array_equal(t, tables[0]) for t in tables) and tables[0].shape == (P, 17)
    assert [int(np.load(tmp_path / ("count_%d.npy" % r))[0]) for r in range(world)] == [2, 2, 2, 1, 1, 1, 1, 1]
    from socp_amd import sweep
    single, _ = sweep.run_sweep(sweep.goddard_starts(P, 1e-3), _oracle_solver(20), None)
    assert np.array_equal(single, tables[0])


def test_the_gather_form_is_decided_without_communicating():
    """The single-tensor / list decision of run_sweep reads the torch build, the backend name and the environment only."""
    from socp_amd import sweep

    class FakeDist:
        def __init__(self, backend, has):
            self._b = backend
            if has:
                self.all_gather_into_tensor = lambda *a: None
        def get_backend(self):
            return self._b
    assert sweep.single_tensor_gather(FakeDist("nccl", True)) and sweep.single_tensor_gather(FakeDist("gloo", True))
    assert not sweep.single_tensor_gather(FakeDist("nccl", False)) and not sweep.single_tensor_gather(FakeDist("mpi", True))
    # gloo: CPU buffers only, and only on a torch build whose ProcessGroupGloo implements the single-tensor form
    assert sweep.single_tensor_gather(FakeDist("nccl", True), "cuda") and not sweep.single_tensor_gather(FakeDist("gloo", True), "cuda")
    real = sweep.torch_at_least
    try:
        sweep.torch_at_least = lambda major, minor: False                 # an old build that HAS the attribute
        assert not sweep.single_tensor_gather(FakeDist("gloo", True)) and sweep.single_tensor_gather(FakeDist("nccl", True))
    finally:
        sweep.torch_at_least = real
    assert sweep.torch_at_least(2, 1) and not sweep.torch_at_least(99, 0)
    os.environ["SOCP_SWEEP_GATHER"] = "list"
    try:
        assert not sweep.single_tensor_gather(FakeDist("nccl", True))
        os.environ["SOCP_SWEEP_GATHER"] = "tensor"
        assert sweep.single_tensor_gather(FakeDist("mpi", True)) and not sweep.single_tensor_gather(FakeDist("mpi", False))
    finally:
        del os.environ["SOCP_SWEEP_GATHER"]
