#!/bin/bash
# SQ counters of one kernel (substring match) during a short single-stream bench:  tools/pmc_kernel.sh <substr> "<counters>"
#   (PMC_ARGS="--precision f16x3": further bench.py arguments)
SUB=$1; CNT=${2:-"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmck; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --also-precision none --streams 1 $PMC_ARGS > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - "$OUT" "$SUB" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"] and int(r["Grid_Size"]) // int(r["Workgroup_Size"]) >= 512:
        c = agg[r["Counter_Name"]]; c[0] += 1; c[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print("%-24s launches %4d  mean per launch %14.0f" % (k, n, v / n))
PY
