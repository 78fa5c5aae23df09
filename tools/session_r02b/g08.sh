cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g08
for spec in "c5 64" "c5 32" "c2 128" "c2 32"; do
  set -- $spec
  for rep in 1 2; do
  python bench.py --config $1 --timesteps $2 --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events > gpurun_out/g08/$1_$2_ahead_$rep.json 2> gpurun_out/g08/err.txt
  python bench.py --config $1 --timesteps $2 --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events --no-cull-ahead > gpurun_out/g08/$1_$2_noahead_$rep.json 2> gpurun_out/g08/err.txt
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/g08/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print('%-22s'%f.split('/')[-1][:-5], 'ms/step %.4f'%d['ms_per_step'], {k:round(v,4) for k,v in d.get('wall_breakdown_ms_per_pass').items()})
PY
