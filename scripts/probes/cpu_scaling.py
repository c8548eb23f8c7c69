"""B1 against the thread count on this host (pinned, reference model::ComputeTraj, 1e4 RK4 steps): where does it stop scaling?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from oracle import oracle as orc
model, cores, quota = bench.host_cpus()
print(model, "quota", quota, "cores", cores)
ref = orc.Ref(orc.MODEL_GODDARD, step_nbr=10000)
Z = bench.make_starts(64, seed=1)
for pin in (True, False):
    for T in (1, 4, 8, 12, 15, 16):
        X0 = bench.fd_rows_inputs(Z, 12 * T)
        best = 0
        for _ in range(2):
            _, s = ref.goddard_traj_batch(T, 10000, bench.GODDARD_PARAMS, 0.0, bench.TF, X0, cpus=cores[:T] if pin else None)
            best = max(best, len(X0) / s)
        print("pinned" if pin else "free  ", T, round(best, 1), round(best / T, 1))
print(open("/sys/fs/cgroup/cpu.stat").read())
