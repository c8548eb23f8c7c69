# A/B: the pair's strips through the panel wavefronts' registers (SOCP_FACTOR_STAGED=0) or through the LDS tiles, loaded and stored by all wavefronts (1)
for R in 0 1 0 1; do for cfg in "253 2048" "200 2048" "127 4096" "85 4096" "48 4096"; do
echo "staged=$R | $cfg | $(SOCP_FACTOR_STAGED=$R SOCP_MEASURE_ONLY=fast python3 scripts/measure_factor.py $cfg 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['fast']['kernel_ms'],3))")"
done; done
