#!/bin/bash
# Per-kernel median durations of tools/resident_loop.py at batch size B:  tools/trace_loop.sh <tag> <B> [repeats]
TAG=${1:-tl}; B=${2:-1}; REPS=${3:-8}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/resident_loop.py $B 30 > $OUT/untraced.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/tools/resident_loop.py $B $REPS > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - "$OUT" "$REPS" <<'PY' | tee $OUT/per_kernel.txt
import csv, glob, sys, collections
out, reps = sys.argv[1], int(sys.argv[2])
print(open(out + "/untraced.txt").read().strip().splitlines()[-1])
f = glob.glob(out + "/trace/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) * 4 // (reps + 4):]            # drop the warm-up batches
per = collections.defaultdict(list)
gaps = []
for prev, r in zip(rows, rows[1:]):
    gaps.append((int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])) / 1e3)
for r in rows:
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    wgs = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    per[(name, wgs)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in per.values())
gaps.sort()
print("%d launches per batch, %.3f ms of kernel time per batch; gap between consecutive kernels: median %.2f us, mean %.2f us"
      % (len(rows) // reps, tot / reps / 1e3, gaps[len(gaps) // 2], sum(g for g in gaps if g < 50) / max(1, sum(1 for g in gaps if g < 50))))
for (name, wgs), d in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:24]:
    d.sort()
    print("%-56s wg=%5d n/batch=%5.1f med %7.2f us  sum/batch %7.3f ms" % (name[:56], wgs, len(d) / reps, d[len(d) // 2], sum(d) / reps / 1e3))
PY
