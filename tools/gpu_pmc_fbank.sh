#!/bin/bash
# PMC passes of tools/fbank_loop.py (cfg-3 shaped feature launches); summary into gpurun_out/pmc_fbank_<tag>/summary.json
set -o pipefail
TAG=${1:-x}; shift || true
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_fbank_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/tools/fbank_loop.py $*"
pass() { local n=$1; shift; local rc=0
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$n" -- $CMD > "$OUT/$n.log" 2>&1 || rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $n timed out"; exit $rc; fi
  [ $rc -ne 0 ] && { echo "pass $n rc $rc"; tail -3 "$OUT/$n.log"; }
}
pass p1 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
pass p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES
pass p3 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES
pass p4 SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SMEM
cd "$REPO"
python3 tools/pmc_summary.py counters "$OUT/p1" "$OUT/p2" "$OUT/p3" "$OUT/p4" > "$OUT/summary.json"
rm -rf "$OUT"/p[1-4]
echo done
