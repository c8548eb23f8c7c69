#!/usr/bin/env python3
"""profiles/sweep_curve_latest.json from a bench line: the one-GPU sweep curve an N > 1 run of bench.py states its expectation from
(bench.py: recorded_sweep_curve).      python scripts/sweep_curve.py profiles/r06_bench_final.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
d = json.loads(open(src).read().strip().split("\n")[-1])
c = d["sweep_curve_one_gpu"]
out = {"curve": c["curve"], "curve_config5": c.get("curve_config5", []), "rk4_steps": c["rk4_steps"], "max_rounds": c["max_rounds"],
       "source": "%s (python bench.py, N = 1, one MI355X)" % os.path.relpath(os.path.abspath(src), ROOT)}
json.dump(out, open(os.path.join(ROOT, "profiles", "sweep_curve_latest.json"), "w"), indent=1)
print(json.dumps(out))
