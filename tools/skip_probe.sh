#!/bin/bash
# What does each class of non-GEMM decode launch cost with 4 batches in flight / on one stream?  (OVC_DEBUG_SKIP, timing only:
# the results are garbage.)  Needs the MEASUREMENT build: python -m openviic_amd.csrc.build --hooks -> tools/libovc_hooks.so.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export OVC_LIBRARY=${OVC_LIBRARY:-$ROOT/tools/libovc_hooks.so}
for R in 1 2; do
for MASK in 0 1 2 4 8 15; do
  for S in 4 1; do
    echo -n "skip=$MASK streams=$S: "; env OVC_DEBUG_SKIP=$MASK python3 $ROOT/bench.py --allow-measurement-hooks --no-cpu-baseline --also-precision none --streams $S 2>&1 >/dev/null | grep "\[bench\] gpu:" | cut -c1-70
  done
done
done
