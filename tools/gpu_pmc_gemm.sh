#!/bin/bash
# PMC passes of a short classifier run (tools/stage_times.py), for the GEMM kernel; summaries into gpurun_out/pmc_gemm_<tag>.json
set -o pipefail
TAG=${1:-x}; shift || true
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_gemm_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/tools/stage_times.py --reps 3 $*"
pass() { local n=$1; shift; local rc=0
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$n" -- $CMD > "$OUT/$n.log" 2>&1 || rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $n timed out"; exit $rc; fi
  [ $rc -ne 0 ] && { echo "pass $n rc $rc"; tail -3 "$OUT/$n.log"; }
}
pass p1 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES
pass p3 TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass p4 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass p5 FETCH_SIZE
pass p6 WRITE_SIZE
cd "$REPO"
python3 tools/pmc_summary.py counters "$OUT/p1" "$OUT/p2" "$OUT/p3" "$OUT/p4" "$OUT/p5" "$OUT/p6" > "$OUT/summary.json"
rm -rf "$OUT"/p[1-6]
echo done
