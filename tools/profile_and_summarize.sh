#!/bin/bash
# tools/profile_round.sh for one or more configurations, each condensed ON the box (the raw traces of three configurations exceed
# what gpurun copies back): summaries land in gpurun_out/<round>_profiles/, the raw trace directories are deleted.
#   tools/profile_and_summarize.sh r04z [config ...]        (no config = the standard transformer)
ROUND=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export OVC_PROFILE_DST=$ROOT/gpurun_out/${ROUND}_profiles
mkdir -p $OVC_PROFILE_DST
for CFG in ${@:-standard_transformer}; do
  TAG=$ROUND; [ "$CFG" != "standard_transformer" ] && TAG=${ROUND}_$CFG
  NOCPU=1 timeout -k 10 400 $ROOT/tools/profile_round.sh $TAG $CFG > $OVC_PROFILE_DST/$TAG.log 2>&1 || { tail -5 $OVC_PROFILE_DST/$TAG.log; exit 1; }
  python3 $ROOT/tools/summarize_profile.py $TAG | head -3
  rm -rf $ROOT/gpurun_out/$TAG/trace $ROOT/gpurun_out/$TAG/trace4 $ROOT/gpurun_out/$TAG/pmc_*
done
