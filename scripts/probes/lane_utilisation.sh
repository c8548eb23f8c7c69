#!/bin/bash
# How many of the VALU instructions of the bench kernels run with their lanes switched on?  (thread-cycles / (64 x instruction-cycles))
export TMPDIR=/tmp
for v in fast exact; do
  rm -rf gpurun_out/lane_$v
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d gpurun_out/lane_$v -- python3 bench.py --variant $v --steps 2 --warmup 1 --cpu-seconds 0 --lean > /dev/null 2>&1
  python3 - "$v" <<'PY'
import csv, glob, sys, collections
v = sys.argv[1]
f = glob.glob('gpurun_out/lane_%s/**/*counter_collection.csv' % v, recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:70]][r['Counter_Name']] += float(r['Counter_Value'])
for k, c in acc.items():
    if c['SQ_INSTS_VALU'] < 1e9:
        continue
    w = c['SQ_WAVES']
    print(v, k, 'VALU/wave %.0f  lane utilisation %.3f' % (c['SQ_INSTS_VALU'] / w, c['SQ_THREAD_CYCLES_VALU'] / (64.0 * c['SQ_ACTIVE_INST_VALU'])))
PY
done
