cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g18
python tools/trace_bench.py > gpurun_out/g18/trace_bench.log 2>&1
grep -v "^trace:" gpurun_out/g18/trace_bench.log | tail -8
grep "^trace:" gpurun_out/g18/trace_bench.log | awk 'NR%5==0' 
python -m pytest tests/test_trace.py -x -q > gpurun_out/g18/pytest.log 2>&1; tail -2 gpurun_out/g18/pytest.log
