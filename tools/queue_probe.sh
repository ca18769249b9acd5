#!/bin/bash
# Does the number of HIP hardware queues (GPU_MAX_HW_QUEUES, default 4) limit how many decode streams overlap?
#   tools/queue_probe.sh "<queue counts>" "<stream counts>" [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 ${3:-1}); do
  for Q in ${1:-default 8}; do
    for S in ${2:-3 4 5}; do
      if [ $Q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$Q; fi
      echo -n "queues=$Q streams=$S: "; python3 $ROOT/bench.py --no-cpu-baseline --streams $S --steps 40 2>&1 >/dev/null | tail -1 | cut -c1-60
    done
  done
done
