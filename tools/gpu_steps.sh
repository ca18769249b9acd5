#!/bin/bash
# Run several GPU steps in ONE gpurun call: each argument is "name::seconds::command"; a step's output goes to
# gpurun_out/$OUT/name.log.  An ordinary failure (a failing test) does not stop the later steps; a step that was killed at its
# time limit (rc 124 / 137) does -- after a timeout nothing else is started on the GPU in the same call.
#   OUT=r04a tools/gpu_steps.sh "ops::600::python -m pytest tests/test_ops_gpu.py -m gpu -x -q" "bench::300::python bench.py"
OUT=${OUT:-steps}
mkdir -p "gpurun_out/$OUT"
worst=0
for step in "$@"; do
    name=${step%%::*}; rest=${step#*::}; secs=${rest%%::*}; cmd=${rest#*::}
    log="gpurun_out/$OUT/$name.log"
    echo "== $name (limit ${secs}s): $cmd"
    start=$(date +%s)
    timeout -k 10 "$secs" bash -c "$cmd" > "$log" 2>&1
    rc=$?
    echo "== $name rc=$rc after $(( $(date +%s) - start ))s"
    tail -n "${TAIL:-6}" "$log"
    [ $rc -ne 0 ] && worst=$rc
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "== $name hit its time limit: no further GPU step in this call"
        break
    fi
done
exit $worst
