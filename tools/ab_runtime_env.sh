#!/bin/bash
# A/B of a HIP-RUNTIME environment variable (not a library hook) on ONE box with the shipped library:
#   tools/ab_runtime_env.sh VAR=VALUE [rounds] [streams...]       e.g.  tools/ab_runtime_env.sh HIP_FORCE_DEV_KERNARG=1 3 4 1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
SETTING=$1; ROUNDS=${2:-2}; shift; shift
STREAMS=${@:-4 1}
for i in $(seq 1 $ROUNDS); do
  for S in $STREAMS; do
    echo -n "unset streams=$S: "; python3 $ROOT/bench.py --no-cpu-baseline --also-precision none --streams $S 2>&1 >/dev/null | grep "\[bench\] gpu:" | cut -c1-100
    echo -n "$SETTING streams=$S: "; env $SETTING python3 $ROOT/bench.py --no-cpu-baseline --also-precision none --streams $S 2>&1 >/dev/null | grep "\[bench\] gpu:" | cut -c1-100
  done
done
