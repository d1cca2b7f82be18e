#!/bin/bash
# Run on the GPU box (through gpurun): kernel stats + PMC passes of the bench command, summaries into gpurun_out/.
#   bash tools/gpu_profile.sh <tag> [extra bench args]
# Passes are separate (rocprofv3 --pmc must not be combined with other tracing on this pool) and sequential.
set -eo pipefail
TAG=${1:-r02}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 3 --warmup 1 --settle 0 --no-cpu-baseline --no-sequential --no-sincnet --no-reference-shape --in-flight 1 $*"
run_pass() {   # name, counters...
    local name=$1; shift
    echo "[profile] pass $name: $*"
    local rc=0
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/$name.log" 2>&1 || rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[profile] pass $name timed out: stopping (no further GPU step)"; exit $rc; fi
    if [ $rc -ne 0 ]; then echo "[profile] pass $name failed with rc $rc (unknown counter?): see $OUT/$name.log; continuing"; tail -3 "$OUT/$name.log"; fi
}
echo "[profile] kernel stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sincnet --no-reference-shape "$@" > "$OUT/stats.log" 2>&1
echo "[profile] kernel stats of the step submitted alone (throughput recurrence)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/seqstats" -- python3 $REPO/bench.py --steps 5 --warmup 2 --settle 0 --in-flight 1 --rec-tile 16 --no-cpu-baseline --no-sincnet --no-sequential --no-reference-shape > "$OUT/seqstats.log" 2>&1
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
run_pass p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
run_pass p2 SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
run_pass p3 SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES
run_pass p4 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES
cd "$REPO"
python3 tools/pmc_summary.py traffic "$OUT/fetch" "$OUT/write" > "$OUT/hbm_traffic.json"
python3 tools/pmc_summary.py counters "$OUT/p1" "$OUT/p2" "$OUT/p3" "$OUT/p4" > "$OUT/mfma_busy.json"
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/seqstats" -name "*kernel_stats.csv" -exec cp {} "$OUT/sequential_kernel_stats.csv" \;
# raw traces are large: keep only the summaries
rm -rf "$OUT/fetch" "$OUT/write" "$OUT/p1" "$OUT/p2" "$OUT/p3" "$OUT/p4" "$OUT/stats" "$OUT/seqstats"
echo "[profile] done: $OUT"
