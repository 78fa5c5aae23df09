cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh c4 r02 && bash tools/collect_profiles.sh c3 r02 && bash tools/collect_profiles.sh c2 r02 && bash tools/collect_profiles.sh c5 r02 && bash tools/pmc_valu.sh c4 > gpurun_out/pmc_valu_c4.log 2>&1
tail -3 gpurun_out/pmc_valu_c4.log | cut -c1-200
