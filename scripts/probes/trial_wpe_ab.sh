# A/B of the trial launches of the device solver (on the GPU box): wavefronts per SIMD they are built for x the workgroup-size rule.
# Builds (here, before the call): socp_amd/_build = the product (SOCP_SOLVER_TRIAL_WPE as committed), _build_t3 / _build_t4 =
#   cp -a socp_amd/_build socp_amd/_build_tW; rm socp_amd/_build_tW/{kernels_solver.o,libsocp_hip.so}; make -C socp_amd/csrc OUT=$PWD/socp_amd/_build_tW SOLVER_DEFS="-DSOCP_SOLVER_TRIAL_WPE(M)=W"
# SOCP_SOLVER_FIT=1: round 4's rule (trial launches shrink their workgroups until every problem is resident); 0: a thread per column.
python3 -m socp_amd.sweep --model interceptor --starts 2048 --solver device_fast > /dev/null 2>&1      # (the box's first large allocation)
for B in _build _build_t3 _build_t4; do
  [ -f socp_amd/$B/libsocp_hip.so ] || continue
  export SOCP_LIB_PATH=$PWD/socp_amd/$B/libsocp_hip.so
  for FIT in 0 1; do
  export SOCP_SOLVER_FIT=$FIT
  for w in "--model interceptor --starts 2048" "--model interceptor --starts 16384" "--starts 4096 --continuation kd --rk4-steps 10" "--starts 4096 --segments 9 --rk4-steps 10" "--starts 4096 --segments 6 --rk4-steps 10"; do
    echo "$B fit=$FIT | $w | $(for rep in 1 2 3; do python3 -m socp_amd.sweep $w --solver ${SOLVER:-device_fast} 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['wall_s'],4), d['converged'], d.get('rounds_rank0'), end='  ')"; done)"
  done
  done
done
