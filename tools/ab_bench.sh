#!/bin/bash
# A/B timing on ONE box: the in-tree libovc.so against another build of the same ABI (tools/libovc_base.bin),
# alternating, 3 streams and 1 stream.   tools/ab_bench.sh [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PREV=$ROOT/tools/libovc_base.bin
for i in $(seq 1 ${1:-2}); do
  for S in 4 1; do
    echo -n "new  streams=$S: "; python3 $ROOT/bench.py --no-cpu-baseline --also-precision none --streams $S 2>&1 >/dev/null | grep "\[bench\] gpu:" | cut -c1-100
    echo -n "prev streams=$S: "; OVC_LIBRARY=$PREV python3 $ROOT/bench.py --no-cpu-baseline --also-precision none --streams $S 2>&1 >/dev/null | grep "\[bench\] gpu:" | cut -c1-100
  done
done
