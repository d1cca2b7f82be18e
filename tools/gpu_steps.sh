#!/bin/bash
# Run a list of GPU steps on the box, each under its own timeout; a step that times out (or is killed) ends the call --
# nothing else is started on a GPU that may be wedged.  A step that merely FAILS (a red test) does not stop the list.
#   bash tools/gpu_steps.sh "<seconds> <name> <command...>" ...
# stdout/stderr of each step go to gpurun_out/<name>.log
mkdir -p gpurun_out
overall=0
for step in "$@"; do
    secs=${step%% *}; rest=${step#* }; name=${rest%% *}; cmd=${rest#* }
    echo "[steps] $name (limit ${secs}s): $cmd"
    start=$(date +%s)
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "[steps] $name: rc $rc after $(( $(date +%s) - start )) s"
    tail -n 6 "gpurun_out/$name.log" | sed 's/^/    /'
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[steps] $name hit its limit: stopping"; exit $rc; fi
    [ $rc -ne 0 ] && overall=$rc
done
exit $overall
