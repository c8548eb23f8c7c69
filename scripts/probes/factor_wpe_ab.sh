# A/B: wavefronts per SIMD the small-strip instantiations of the matrix-core factor kernel are budgeted for.  Builds (here, before the call):
#   for W in 22 32 33: cp -a socp_amd/_build socp_amd/_build_w$W; rm socp_amd/_build_w$W/{kernels_factor_fast.o,libsocp_hip.so};
#                      make -C socp_amd/csrc OUT=$PWD/socp_amd/_build_w$W FACTOR_DEFS="-DSOCP_FACTOR_WPE_SMALL=<first digit> -DSOCP_FACTOR_WPE_MID=<second>"
# ("43" = the product build, socp_amd/_build)
for W in 43 33 32 22; do for cfg in "48 4096" "64 4096" "85 4096" "96 4096" "127 4096"; do
  LIB=$PWD/socp_amd/_build_w$W/libsocp_hip.so; [ $W = 43 ] && LIB=$PWD/socp_amd/_build/libsocp_hip.so
  echo "w$W $cfg $(SOCP_MEASURE_ONLY=fast SOCP_LIB_PATH=$LIB python3 scripts/measure_factor.py $cfg 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['fast']['kernel_ms'],3))")"
done; done
