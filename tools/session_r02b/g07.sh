cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g07
python -m pytest tests/test_gpu_properties.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/g07/pytest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/g07/pytest.log
for c in c4 c2 c5 c3; do
  for rep in 1 2; do
  python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events > gpurun_out/g07/${c}_ahead_noev_$rep.json 2> gpurun_out/g07/$c.err
  python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events --no-cull-ahead > gpurun_out/g07/${c}_noahead_noev_$rep.json 2> gpurun_out/g07/${c}_noahead.err
  done
  python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/g07/${c}_ahead_ev.json 2> gpurun_out/g07/$c.err
  python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-cull-ahead > gpurun_out/g07/${c}_noahead_ev.json 2> gpurun_out/g07/$c.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/g07/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print('%-20s'%f.split('/')[-1][:-5], 'ms/step %.4f'%d['ms_per_step'], {k:round(v,4) for k,v in d.get('wall_breakdown_ms_per_pass').items()})
PY
