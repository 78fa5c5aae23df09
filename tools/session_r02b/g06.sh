cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g06
for c in c4 c2 c5 c3; do
  for rep in 1 2; do
  python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events > gpurun_out/g06/${c}_ahead_$rep.json 2> gpurun_out/g06/$c.err
  python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events --no-cull-ahead > gpurun_out/g06/${c}_noahead_$rep.json 2> gpurun_out/g06/${c}_noahead.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/g06/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print('%-16s'%f.split('/')[-1][:-5], 'ms/step %.4f'%d['ms_per_step'], {k:round(v,4) for k,v in d.get('wall_breakdown_ms_per_pass').items()})
PY
