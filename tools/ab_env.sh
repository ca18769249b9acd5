#!/bin/bash
# A/B timing of a run-time switch on ONE box:  tools/ab_env.sh VAR [rounds]   (VAR=0 against VAR unset)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 ${2:-2}); do
  for S in 4 1; do
    echo -n "on   streams=$S: "; python3 $ROOT/bench.py --no-cpu-baseline --streams $S 2>&1 >/dev/null | tail -1 | cut -c1-90
    echo -n "off  streams=$S: "; env $1=0 python3 $ROOT/bench.py --no-cpu-baseline --streams $S 2>&1 >/dev/null | tail -1 | cut -c1-90
  done
done
