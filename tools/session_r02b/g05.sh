cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g05
python -m pytest tests -m gpu -x -q > gpurun_out/g05/pytest.log 2>&1; echo "pytest rc $?"
tail -5 gpurun_out/g05/pytest.log
for c in c4 c3 c2 c5; do
  python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/g05/$c.json 2> gpurun_out/g05/$c.err
  python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-cull-ahead > gpurun_out/g05/${c}_noahead.json 2> gpurun_out/g05/${c}_noahead.err
done
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob('gpurun_out/g05/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'ERR', e); continue
    r=d.get('roofline',{})
    print('%-12s'%f.split('/')[-1][:-5], 'ms/step %.4f'%d['ms_per_step'], 'frac %.4f'%r.get('frac'), {k:round(v,4) for k,v in r.get('kernel_ms_per_pass').items()}, d['check'].get('hits'), {k:round(v,4) for k,v in d.get('wall_breakdown_ms_per_pass').items()})
PY
