# the matrix-core refresh alone at the sizes of the A/Bs (product build), twice
for R in 1 2; do for cfg in "253 2048" "200 2048" "127 4096" "85 4096"; do
echo "$cfg | $(SOCP_MEASURE_ONLY=fast python3 scripts/measure_factor.py $cfg 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['fast']['kernel_ms'],3))")"
done; done
