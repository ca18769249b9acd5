#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel stats + PMC passes, outputs under gpurun_out/<tag>/.
#   tools/profile_round.sh <tag> [config]        (EXTRA="--precision f16x3": further bench.py arguments for every run)
set -o pipefail
TAG=${1:-r01}
CFG=${2:-standard_transformer}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export OVC_TUNE_CACHE=$OUT/tune_cache.json
cd /tmp && export TMPDIR=/tmp
# 1. plain bench (fills the GEMM tuning cache so that the profiled runs contain no tuning launches)
python3 $ROOT/bench.py --steps 20 --warmup 5 --config $CFG $EXTRA ${NOCPU:+--no-cpu-baseline} > $OUT/bench.json 2> $OUT/bench.err || exit 1
python3 $ROOT/bench.py --steps 20 --warmup 5 --config $CFG $EXTRA --streams 1 --no-cpu-baseline --also-precision none > $OUT/bench_1stream.json 2>> $OUT/bench.err || exit 1
# 2. kernel trace + stats, one stream
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --also-precision none --streams 1 --config $CFG $EXTRA > $OUT/trace.log 2>&1 || exit 1
# 2b. the same for the headline mode (4 streams): kernels of different batches overlap, so durations do not add up to wall time
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace4 -- python3 $ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --also-precision none --config $CFG $EXTRA > $OUT/trace4.log 2>&1 || exit 1
# 3. PMC passes (separate runs; FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950)
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"; do
  NAME=$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$NAME -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --also-precision none --streams 1 --config $CFG $EXTRA > $OUT/pmc_$NAME.log 2>&1 || exit 1
done
echo done > $OUT/DONE
tail -1 $OUT/bench.json | cut -c1-400
