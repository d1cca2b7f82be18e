#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats (and, with "pmc", LDS / MFMA counter passes) of tools/run_sincnet.py.
#   bash tools/gpu_profile_sincnet.sh <tag> [pmc]
set -eo pipefail
TAG=${1:-r05}; MODE=${2:-stats}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_sinc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/tools/run_sincnet.py --reps 20 --check 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/stats"
if [ "$MODE" = "pmc" ]; then
    run_pass() {
        local name=$1; shift
        local rc=0
        timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- $CMD > "$OUT/$name.log" 2>&1 || rc=$?
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[profile] pass $name timed out: stopping"; exit $rc; fi
    }
    run_pass p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
    run_pass p2 SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
    run_pass p3 SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES
    run_pass p4 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES
    run_pass fetch FETCH_SIZE
    run_pass write WRITE_SIZE
    cd "$REPO"
    python3 tools/pmc_summary.py counters "$OUT/p1" "$OUT/p2" "$OUT/p3" "$OUT/p4" > "$OUT/counters.json"
    python3 tools/pmc_summary.py traffic "$OUT/fetch" "$OUT/write" > "$OUT/hbm_traffic.json"
    rm -rf "$OUT/p1" "$OUT/p2" "$OUT/p3" "$OUT/p4" "$OUT/fetch" "$OUT/write"
fi
echo "[profile] done: $OUT"
