#!/bin/bash
# A/B of one environment setting on ONE box:  tools/ab_envval.sh "VAR=value" [rounds] [streams...]   (set against unset)
# Loads the MEASUREMENT build (tools/libovc_hooks.so, see ab_env.sh); lines are marked invalid for credit.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export OVC_LIBRARY=${OVC_LIBRARY:-$ROOT/tools/libovc_hooks.so}
SET=$1; ROUNDS=${2:-2}; shift; shift
STREAMS=${@:-4 1}
for i in $(seq 1 $ROUNDS); do
  for S in $STREAMS; do
    echo -n "unset      streams=$S: "; python3 $ROOT/bench.py --allow-measurement-hooks --no-cpu-baseline --also-precision none --streams $S $AB_ARGS 2>&1 >/dev/null | grep "\[bench\] gpu:" | cut -c1-100
    echo -n "$SET streams=$S: "; env $SET python3 $ROOT/bench.py --allow-measurement-hooks --no-cpu-baseline --also-precision none --streams $S $AB_ARGS 2>&1 >/dev/null | grep "\[bench\] gpu:" | cut -c1-100
  done
done
