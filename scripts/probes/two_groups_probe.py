"""Probe: does a throughput-bound sweep finish sooner as TWO concurrent engine calls on one GPU (two host threads, two contexts:
the host logic, state read-back and solver kernels of one half overlapping the trajectory launches of the other)?
   python scripts/probes/two_groups_probe.py [starts]"""
import json
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from socp_amd import capi, sweep  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576


def make_ctx():
    ctx = capi.Context(capi.MODEL_GODDARD)
    ctx.set_params(sweep.GODDARD_PARAMS)
    ctx.set_step_number(10000)
    ctx.set_variant(capi.VARIANT_LANE_FAST)
    sweep.goddard_single_shooting_problem(ctx)
    ctx.warm_up()
    return ctx


base = sweep.goddard_starts(65536, 1e-3)
Z0 = np.concatenate([base * np.concatenate([np.ones(7), np.full(7, 1.0 + 1e-7 * b)])[None, :] for b in range(-(-P // 65536))])[:P]
ctxs = [make_ctx() for _ in range(4)]
kw = dict(kind=capi.CHAIN_PLAIN, xtol=1e-8, max_rounds=40)
for c in ctxs:
    c.chains_solve(Z0[:64], **kw)
out = {}
for groups in (1, 2, 4, 1, 2):
    res = [None] * groups
    blocks = np.array_split(np.arange(P), groups)

    def work(k):
        res[k] = ctxs[k].chains_solve(Z0[blocks[k]], **kw)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(groups)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    conv = sum(int(np.sum(r["info"] == 1)) for r in res)
    out.setdefault(str(groups), []).append(round(wall, 4))
    print(json.dumps({"starts": P, "groups": groups, "wall_s": wall, "converged": conv}), flush=True)
print(json.dumps({"starts": P, "walls_by_groups": out}))
