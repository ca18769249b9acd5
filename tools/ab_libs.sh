#!/bin/bash
# A/B(/C) timing of several builds of the library on ONE box, alternating:  tools/ab_libs.sh <rounds> <lib> [<lib> ...]
# (paths relative to the repository; 4 streams and 1 stream per round).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=$1; shift
for i in $(seq 1 $ROUNDS); do
  for S in 4 1; do
    for L in "$@"; do
      echo -n "$L streams=$S: "; OVC_LIBRARY=$ROOT/$L python3 $ROOT/bench.py --no-cpu-baseline --also-precision none --streams $S 2>&1 >/dev/null | grep "\[bench\] gpu:" | cut -c1-100
    done
  done
done
