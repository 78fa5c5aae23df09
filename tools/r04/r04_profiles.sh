#!/bin/bash
# session script: the round's profiles for every single-GPU BASELINE configuration
cd $GRAFT_REPO_ROOT
for cfg in c4 c3 c2 c5; do
  bash tools/collect_profiles.sh $cfg r04 > gpurun_out/collect_$cfg.log 2>&1
  tail -2 gpurun_out/collect_$cfg.log
done
