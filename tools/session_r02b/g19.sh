cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g19
python -m pytest tests -m gpu -x -q > gpurun_out/g19/pytest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/g19/pytest.log
bash tools/collect_profiles.sh c5 r02
python bench.py > gpurun_out/g19/bench_default.json 2> gpurun_out/g19/bench_default.err; cut -c1-600 gpurun_out/g19/bench_default.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
