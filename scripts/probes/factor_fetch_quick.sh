# FETCH_SIZE / WRITE_SIZE of the refresh's launches (2048 x n = 253), one pass each, quick
export TMPDIR=/tmp SOCP_MEASURE_ONLY=fast
M="python3 scripts/measure_factor.py 253 2048 3"
rm -rf gpurun_out/fq_*
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fq_fetch -- $M > /dev/null 2>&1
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/fq_write -- $M > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for name in ("fetch", "write"):
    for f in glob.glob("gpurun_out/fq_%s/**/*counter_collection.csv" % name, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "factor_fast_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 64 * 2048]
        by = {}
        for r in rows: by.setdefault(r["Kernel_Name"][:60], []).append(float(r["Counter_Value"]))
        for k, v in by.items(): print(name, k, [round(x) for x in v])
PY
$M
