#!/bin/bash
# Round-4 evidence run (on the GPU box, from the repo root, through gpurun):  bash scripts/profile_r04.sh
# Everything lands under gpurun_out/profiles_r04/ (copied into profiles/ afterwards).
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles_r04
mkdir -p $OUT
export SOCP_MULTISTART_TRACE=1
sweep() { tag=$1; shift; timeout -k 5 200 python3 -m socp_amd.sweep "$@" > $OUT/r04_$tag.json 2> $OUT/r04_$tag.trace || echo "FAILED $tag"; grep -h "set-up" $OUT/r04_$tag.trace | tail -1 | cut -c1-260 > $OUT/r04_$tag.engine.txt; rm -f $OUT/r04_$tag.trace; echo "$tag $(python3 -c "import json; r=json.load(open('$OUT/r04_$tag.json')); print(round(r['wall_s'],4), r['converged'], r.get('rounds_rank0'))")"; }
# ---- sweeps: host solvers vs device solvers -------------------------------------------------------------------------------
for s in host device device_fast; do
  sweep sweep_4096_M6_$s --starts 4096 --segments 6 --rk4-steps 10000 --solver $s
  sweep sweep_4096_M9_$s --starts 4096 --segments 9 --rk4-steps 10000 --solver $s
  sweep sweep_interceptor_2048_$s --model interceptor --starts 2048 --solver $s
  sweep sweep_interceptor_256_$s --model interceptor --starts 256 --solver $s
  sweep chains_kd_4096_N10_$s --starts 4096 --continuation kd --rk4-steps 10 --solver $s
  sweep chains_kd_4096_N10000_$s --starts 4096 --continuation kd --rk4-steps 10000 --solver $s
done
sweep sweep_4096_fast --starts 4096 --rk4-steps 10000
sweep sweep_4096_fast_maxrounds40 --starts 4096 --rk4-steps 10000 --max-rounds 40
sweep sweep_65536_fast_maxrounds40 --starts 65536 --rk4-steps 10000 --max-rounds 40
sweep sweep_4096_exact --starts 4096 --rk4-steps 10000 --variant exact
# ---- kernel traces of the engine kernels (solver kernels, chains, variational, interceptor) -------------------------------
trace() { tag=$1; shift; rm -rf $OUT/tmp_$tag; timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp_$tag -o t -- python3 "$@" > /dev/null 2>&1; f=$(find $OUT/tmp_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r04_${tag}_kernel_stats.csv; rm -rf $OUT/tmp_$tag; echo "traced $tag: $(wc -l < $OUT/r04_${tag}_kernel_stats.csv 2>/dev/null) rows"; }
trace devsolver_interceptor_2048 -m socp_amd.sweep --model interceptor --starts 2048 --solver device
trace devsolver_fast_interceptor_2048 -m socp_amd.sweep --model interceptor --starts 2048 --solver device_fast
trace devsolver_chains_kd_N10 -m socp_amd.sweep --starts 4096 --continuation kd --rk4-steps 10 --solver device
trace devsolver_sweep_M9 -m socp_amd.sweep --starts 4096 --segments 9 --rk4-steps 10000 --solver device
trace hostsolver_chains_kd_N10000 -m socp_amd.sweep --starts 4096 --continuation kd --rk4-steps 10000 --solver host
trace variational_dint_M64 scripts/measure_configs.py
# ---- the bench workload: kernel stats + PMC (separate passes) for both flavours --------------------------------------------
bash scripts/profile_bench.sh r04 > $OUT/profile_bench.log 2>&1; cp gpurun_out/profiles_r04/* $OUT/ 2>/dev/null; tail -3 $OUT/profile_bench.log
# ---- the matrix-core factorisation on its own: kernel stats, PMC traffic, phase clocks -------------------------------------------
bash scripts/profile_factor.sh r04 253 2048 > $OUT/profile_factor.log 2>&1; cp gpurun_out/r04_factor_* $OUT/ 2>/dev/null; tail -2 $OUT/profile_factor.log | cut -c1-300
touch socp_amd/csrc/kernels_factor_fast.hip; make -s -C socp_amd/csrc > /dev/null 2>&1      # (back to the production build of the kernel)
# ---- the default bench line, and the one-GPU sweep curve the N > 1 runs state their expectation from -------------------------------
timeout -k 5 600 python3 bench.py > $OUT/r04_bench.json 2> $OUT/r04_bench.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/profiles_r04/r04_bench.json"))
print("bench", d["value"], d["roofline"]["frac"], d["exact"]["value"], d["cpu_baseline"]["value"], d["cpu_baseline"]["p1"]["value"],
      {k: round(d[k]["wall_s"], 3) for k in ("sweep", "sweep_large", "sweep_xl", "sweep_config5")}, d["north_star_128"]["fast"]["jacobian_ms"])
c = d["sweep_curve_one_gpu"]
json.dump({"curve": c["curve"], "curve_config5": c.get("curve_config5", []), "rk4_steps": c["rk4_steps"], "max_rounds": c["max_rounds"],
           "source": "profiles/r04_bench_final.json (python bench.py, N = 1, one MI355X)"},
          open("gpurun_out/profiles_r04/sweep_curve_latest.json", "w"), indent=1)
PY
python3 scripts/kernel_meta.py --all --json $OUT/r04_kernel_meta.json > $OUT/r04_kernel_meta.txt 2>&1
ls $OUT | wc -l
