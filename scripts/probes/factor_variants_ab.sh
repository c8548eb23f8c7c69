# A/B of build variants of the matrix-core refresh (socp_amd/_build_<tag>: make -C socp_amd/csrc OUT=... FACTOR_DEFS="-DSOCP_FACTOR_BLK=2" etc.)
for rep in 1 2; do for T in "" _nopipe; do for cfg in "253 2048" "200 2048" "127 4096" "85 4096"; do
  [ -f socp_amd/_build$T/libsocp_hip.so ] || continue
  echo "build${T:-_product} | $cfg | $(SOCP_MEASURE_ONLY=fast SOCP_LIB_PATH=$PWD/socp_amd/_build$T/libsocp_hip.so python3 scripts/measure_factor.py $cfg 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['fast']['kernel_ms'],3))")"
done; done; done
