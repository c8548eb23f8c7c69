# kernel-level A/B of the KD-chain sweep (n = 85): round-4 library against the product build (rocprofv3 --kernel-trace --stats)
export TMPDIR=/tmp
for B in ${BUILDS:-_build_r04 _build}; do
  export SOCP_LIB_PATH=$PWD/socp_amd/$B/libsocp_hip.so
  rm -rf gpurun_out/kd_$B
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kd_$B -- python3 -m socp_amd.sweep --starts 4096 --continuation kd --rk4-steps 10 --solver ${SOLVER:-device} > /dev/null 2>&1
  f=$(find gpurun_out/kd_$B -name "*kernel_stats.csv" | head -1)
  echo "== $B"; python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:12]: print("%-70s calls %5s total %8.3f ms avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
  rm -rf gpurun_out/kd_$B
done
