cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g14
python -m pytest tests -m gpu -x -q > gpurun_out/g14/pytest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/g14/pytest.log
bash tools/collect_profiles.sh c4 r02 && bash tools/collect_profiles.sh c3 r02
