cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g22
python bench.py --config c3 --exact-only --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/g22/c3_exact_only.json 2> gpurun_out/g22/err1.txt
python bench.py --config c3o --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/g22/c3o.json 2> gpurun_out/g22/err2.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/g22/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-16s'%f.split('/')[-1][:-5], 'ms/step %.3f'%d['ms_per_step'], 'value %.3e'%d['value'], {k:round(v,3) for k,v in r['kernel_ms_per_pass'].items()}, d['check'])
PY
tail -3 gpurun_out/g22/err2.txt
