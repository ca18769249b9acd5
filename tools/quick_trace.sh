#!/bin/bash
# Per-kernel median durations of the single-stream decode for one or more builds of the library (same box A/B):
#   tools/quick_trace.sh <tag> [lib ...]      (default: the in-tree build); output gpurun_out/<tag>_<i>.txt
TAG=${1:-qt}; shift
LIBS=${@:-openviic_amd/csrc/libovc.so}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for L in $LIBS; do
  OUT=$ROOT/gpurun_out/${TAG}_$i
  rm -rf $OUT; mkdir -p $OUT
  OVC_LIBRARY=$ROOT/$L rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1 ${QT_ARGS} > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
  python3 - "$OUT" "$L" <<'PY' | tee $ROOT/gpurun_out/${TAG}_$i.txt
import csv, glob, sys, collections
out, lib = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/*/*_kernel_trace.csv")[0]
per = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    wgs = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    per[(name, wgs)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in per.values())
print("== %s: %.2f ms of kernel time in the trace" % (lib, tot / 1e3))
for (name, wgs), d in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:16]:
    d.sort()
    print("%-52s wg=%5d n=%5d med %8.2f us  sum %8.2f ms" % (name[:52], wgs, len(d), d[len(d) // 2], sum(d) / 1e3))
PY
  i=$((i+1))
done
