cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g04
python -m pytest tests -m gpu -x -q > gpurun_out/g04/pytest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/g04/pytest.log
for c in c4 c3 c2 c5; do
  python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/g04/$c.json 2> gpurun_out/g04/$c.err
done
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/g04/prof_c2 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config c2 --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/g04/prof_c2.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob('gpurun_out/g04/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'ERR', e); continue
    r=d.get('roofline',{})
    print('%-12s'%f.split('/')[-1][:-5], 'ms/step %.4f'%d['ms_per_step'], 'frac %.4f'%r.get('frac'), {k:round(v,4) for k,v in r.get('kernel_ms_per_pass').items()}, d['check'].get('hits'), d.get('wall_breakdown_ms_per_pass'))
for f in glob.glob('gpurun_out/g04/prof_c2/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ftkx' in r['Name'] or 'rocprim' in r['Name'] or 'anonymous' in r['Name'] or 'rocclr' in r['Name']:
            print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
