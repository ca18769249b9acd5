#!/bin/bash
# A/B timing of a run-time switch on ONE box:  tools/ab_env.sh VAR [rounds] [streams...]   (VAR=0 against VAR unset)
# Prints the fp32 headline line of each run ("[bench] gpu: ... captions/s"), alternating on / off.
# Both legs load the MEASUREMENT build (python -m openviic_amd.csrc.build --hooks -> tools/libovc_hooks.so): the shipped library
# does not read OVC_DEBUG_* / OVC_KSPLIT_* / the kernel A/B switches at all, and bench.py refuses them without the flag below.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export OVC_LIBRARY=${OVC_LIBRARY:-$ROOT/tools/libovc_hooks.so}
VAR=$1; ROUNDS=${2:-2}; shift; shift
STREAMS=${@:-4 1}
for i in $(seq 1 $ROUNDS); do
  for S in $STREAMS; do
    echo -n "unset streams=$S: "; python3 $ROOT/bench.py --allow-measurement-hooks --no-cpu-baseline --also-precision none --streams $S $AB_ARGS 2>&1 >/dev/null | grep "\[bench\] gpu:" | cut -c1-100
    echo -n "$VAR=0 streams=$S: "; env $VAR=0 python3 $ROOT/bench.py --allow-measurement-hooks --no-cpu-baseline --also-precision none --streams $S $AB_ARGS 2>&1 >/dev/null | grep "\[bench\] gpu:" | cut -c1-100
  done
done
