cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g21
for v in 4 8 16; do
  for c in c2 c5 c3; do
    FTKX_EXACT_WG_PER_CU=$v python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/g21/${c}_wg$v.json 2> gpurun_out/g21/err.txt
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/g21/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-12s'%f.split('/')[-1][:-5], 'ms/step %.4f'%d['ms_per_step'], {k:round(v,4) for k,v in r['kernel_ms_per_pass'].items()})
PY
