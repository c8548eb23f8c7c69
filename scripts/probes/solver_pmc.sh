#!/bin/bash
# Instruction counts of the device solver's launches (PMC pass, no tracing): bash scripts/probes/solver_pmc.sh
export TMPDIR=/tmp
OUT=gpurun_out/solver_pmc; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/kd -- python3 -m socp_amd.sweep --starts 4096 --continuation kd --rk4-steps 10 --solver device --warmup 0 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/solver_pmc/kd/**/*counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# one row per (dispatch, counter)
disp = collections.OrderedDict()
for r in rows:
    k = (r['Dispatch_Id'], r['Kernel_Name'][:60], r.get('Grid_Size'), r.get('Workgroup_Size'))
    disp.setdefault(k, {})[r['Counter_Name']] = float(r['Counter_Value'])
n = 0
for k, c in disp.items():
    if 'advance_kernel' not in k[1]:
        continue
    w = c.get('SQ_WAVES', 0) or 1
    print(k[0], k[1][-22:], 'grid', k[2], 'wg', k[3], 'waves %d' % w, ' per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f  wave-cycles %.0f' % (
        c.get('SQ_INSTS_VALU', 0) / w, c.get('SQ_INSTS_SALU', 0) / w, c.get('SQ_INSTS_LDS', 0) / w, c.get('SQ_INSTS_VMEM_RD', 0) / w, c.get('SQ_INSTS_VMEM_WR', 0) / w,
        4 * c.get('SQ_WAVE_CYCLES', 0) / w))
    n += 1
    if n > 14:
        break
PY
