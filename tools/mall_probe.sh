#!/bin/bash
# does the decode cross-attention's K/V (3 layers x 52 MB per batch of 256, re-read on each of the 20 steps) hit
# the 256 MB Infinity Cache if the batches in flight are sized for it?   tools/mall_probe.sh <tag>
#   1. captions/s for (images per stream) x (streams) at a constant number of images per timed region;
#   2. per-kernel medians of the single-stream decode at each batch size (rocprofv3 kernel trace): cross-attention us per launch
#      and the bytes it streams -> effective TB/s (HBM alone delivers ~5 on this part; more = Infinity Cache hits).
# The TCC counters rocprofv3 offers on gfx950 (TCC_EA0_RDREQ_*) count requests LEAVING the L2 -- Infinity Cache hits included
# -- so a hit share cannot be read from them; the duration per byte is the observable.
TAG=${1:-mall}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "# captions/s: images per stream x streams (bench.py --batch B --streams S, fp32, 5120 images per timed region)" | tee $OUT/table.txt
for CFG in "256 4" "256 2" "256 1" "128 4" "128 8" "128 2" "512 2" "512 1" "64 4" "64 8" "1024 1"; do
  set -- $CFG; B=$1; S=$2; STEPS=$((5120 / B)); [ $STEPS -lt 8 ] && STEPS=8
  LINE=$(python3 $ROOT/bench.py --batch $B --streams $S --steps $STEPS --warmup 4 --no-cpu-baseline --also-precision none 2>&1 >/dev/null | grep "\[bench\] gpu:" | cut -c1-110)
  echo "B=$B streams=$S steps=$STEPS: $LINE" | tee -a $OUT/table.txt
done
echo "# single-stream per-kernel medians by batch size (rocprofv3 --kernel-trace; K/V bytes per cross-attention launch = B x 50 x 512 x 4 x 2)" | tee -a $OUT/table.txt
for B in 32 64 128 256 512; do
  rm -rf $OUT/trace_$B
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$B -- python3 $ROOT/tools/resident_loop.py $B 6 > $OUT/trace_$B.log 2>&1 || { tail -3 $OUT/trace_$B.log; exit 1; }
  python3 - "$OUT/trace_$B" "$B" <<'PY' | tee -a $OUT/table.txt
import csv, glob, sys, collections
out, B = sys.argv[1], int(sys.argv[2])
f = glob.glob(out + "/*/*_kernel_trace.csv")[0]
per = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
    per[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
def med(name):
    d = sorted(per.get(name, [0.0])); return d[len(d) // 2]
cross = med("decode_cross_attention_mfma_kernel")
mb = B * 50 * 512 * 4 * 2 / 1e6
print("B=%4d: cross-attention %6.2f us for %6.1f MB = %5.2f TB/s | self-attention %6.2f us | layer_norm %5.2f us | fused update %5.2f us"
      % (B, cross, mb, mb / cross, med("decode_self_attention_mfma_kernel"), med("layer_norm_rows"), med("beam_fused_update_kernel")))
PY
done
