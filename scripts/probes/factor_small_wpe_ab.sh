for rep in 1 2; do for T in "" _small2 _small3; do for cfg in "40 4096" "48 4096" "56 4096" "64 4096"; do
  echo "build${T:-_product} | $cfg | $(SOCP_MEASURE_ONLY=fast SOCP_LIB_PATH=$PWD/socp_amd/_build$T/libsocp_hip.so python3 scripts/measure_factor.py $cfg 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['fast']['kernel_ms'],3))")"
done; done; done
