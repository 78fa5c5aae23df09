cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g01
python -m pytest tests -m gpu -x -q > gpurun_out/g01/pytest.log 2>&1; echo "pytest rc $?" 
tail -3 gpurun_out/g01/pytest.log
python bench.py --steps 5 --warmup 1 > gpurun_out/g01/c4.json 2> gpurun_out/g01/c4.err && \
for c in c2 c3 c5; do
  python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/g01/$c.json 2> gpurun_out/g01/$c.err
  python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events > gpurun_out/g01/${c}_noev.json 2> gpurun_out/g01/${c}_noev.err
done
python bench.py --steps 5 --warmup 1 --no-kernel-events > gpurun_out/g01/c4_noev.json 2>&1
python tools/c2_breakdown.py > gpurun_out/g01/c2_breakdown.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/g01/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'ERR', e); continue
    r=d.get('roofline',{})
    print(f.split('/')[-1], 'ms/step %.4f'%d['ms_per_step'], 'frac', r.get('frac'), 'kernels', r.get('kernel_ms_per_pass'), d.get('wall_breakdown_ms_per_pass'))
PY
cat gpurun_out/g01/c2_breakdown.log
