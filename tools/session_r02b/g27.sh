cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g27
run() { n=$1; c=$2; shift; shift; env "$@" python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/g27/$n.json 2> gpurun_out/g27/err.txt; }
run c4_base c4 A=1
run c4_zc8 c4 FTKX_CULL_ZC=8
run c4_zc8_sc2 c4 FTKX_CULL_ZC=8 FTKX_CULL_STEP_CHUNK=2
run c4_zc2 c4 FTKX_CULL_ZC=2
run c4_sc2 c4 FTKX_CULL_STEP_CHUNK=2
run c4_sc8 c4 FTKX_CULL_STEP_CHUNK=8
run c3_base c3 A=1
run c3_zc8 c3 FTKX_CULL_ZC=8
run c3_sc2 c3 FTKX_CULL_STEP_CHUNK=2
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/g27/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    r=d['roofline']
    print('%-12s'%f.split('/')[-1][:-5], 'ms/step %.4f'%d['ms_per_step'], {k:round(v,4) for k,v in r['kernel_ms_per_pass'].items()}, {k:round(v,4) for k,v in d['wall_breakdown_ms_per_pass'].items()})
PY
