export TMPDIR=/tmp
set -e
for v in fast exact; do
  st=3; [ $v = exact ] && st=2
  B="python3 bench.py --variant $v --steps $st --warmup 1 --cpu-seconds 0"
  rm -rf gpurun_out/p_${v}_*
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_${v}_stats -- $B > gpurun_out/p_${v}_bench.json 2>gpurun_out/p_${v}.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_${v}_fetch -- $B > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p_${v}_write -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/p_${v}_sq -- $B > /dev/null 2>&1
  echo done $v
done
