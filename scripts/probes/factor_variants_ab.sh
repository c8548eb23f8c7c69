# A/B of build variants of the matrix-core refresh against the product build.  Build a variant (here, before the call) with
#   tag=blk4; mkdir -p socp_amd/_build_$tag; cp socp_amd/_build/*.o socp_amd/_build_$tag/; rm socp_amd/_build_$tag/kernels_factor_fast.o
#   make -C socp_amd/csrc OUT=$PWD/socp_amd/_build_$tag FACTOR_DEFS="-DSOCP_FACTOR_BLK=4"      (or -DSOCP_FACTOR_WPE_MID=3, ...)
# and name the tags in VARIANTS (default: every socp_amd/_build_* that holds a library)
VARIANTS=${VARIANTS:-$(ls -d socp_amd/_build_* 2>/dev/null | sed 's|socp_amd/_build||')}
for rep in 1 2; do for T in "" $VARIANTS; do for cfg in "253 2048" "200 2048" "127 4096" "85 4096"; do
  [ -f socp_amd/_build$T/libsocp_hip.so ] || continue
  echo "build${T:-_product} | $cfg | $(SOCP_MEASURE_ONLY=fast SOCP_LIB_PATH=$PWD/socp_amd/_build$T/libsocp_hip.so python3 scripts/measure_factor.py $cfg 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['fast']['kernel_ms'],3))")"
done; done; done
