#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PREV=$ROOT/tools/libovc_base.bin
run() { python3 $ROOT/bench.py --no-cpu-baseline --also-precision none --streams $1 2>&1 >/dev/null | grep "\[bench\] gpu:" | cut -c1-100; }
for i in 1 2; do
  for S in 4 1; do
    echo -n "new lazy    streams=$S: "; run $S
    echo -n "new eager   streams=$S: "; OVC_EAGER_LAYER_NORM=1 run $S
    echo -n "wave0 eager streams=$S: "; OVC_EAGER_LAYER_NORM=1 OVC_LIBRARY=$PREV run $S
    echo -n "wave0 lazy  streams=$S: "; OVC_LIBRARY=$PREV run $S
  done
done
