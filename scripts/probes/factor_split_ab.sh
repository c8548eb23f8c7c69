# A/B: the matrix-core refresh as one launch (SOCP_FACTOR_SPLIT=0) or as a qrfac launch + a qform launch (1)
for R in 0 1 0 1; do for cfg in "253 2048" "200 2048" "127 4096" "85 4096" "48 4096"; do
echo "split=$R | $cfg | $(SOCP_FACTOR_SPLIT=$R SOCP_MEASURE_ONLY=fast python3 scripts/measure_factor.py $cfg 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['fast']['kernel_ms'],3))")"
done; done
