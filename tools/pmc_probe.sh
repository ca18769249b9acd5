#!/bin/bash
# One PMC pass over a short single-stream bench run:  tools/pmc_probe.sh <tag> "<counters>"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 1 > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    n[(k, r["Counter_Name"])] += 1
for k, c in sorted(agg.items()):
    print(k, {name: round(v / n[(k, name)]) for name, v in c.items()})
PY
