cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g03
python -m pytest tests -m gpu -x -q > gpurun_out/g03/pytest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/g03/pytest.log
python tools/ab_mask.py c4 4 -- "PD=2" "PD=3" > gpurun_out/g03/ab_c4.log 2>&1; cat gpurun_out/g03/ab_c4.log
run() { # name config env...
  n=$1; c=$2; shift; shift
  env "$@" python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/g03/$n.json 2> gpurun_out/g03/$n.err
}
run c4_base c4 A=1
run c4_sc4 c4 FTKX_CULL_STEP_CHUNK=4
run c4_sc8 c4 FTKX_CULL_STEP_CHUNK=8
run c4_zc2 c4 FTKX_CULL_ZC=2
run c4_zc2sc4 c4 FTKX_CULL_ZC=2 FTKX_CULL_STEP_CHUNK=4
run c3_base c3 A=1
run c3_sc4 c3 FTKX_CULL_STEP_CHUNK=4
run c3_zc2sc4 c3 FTKX_CULL_ZC=2 FTKX_CULL_STEP_CHUNK=4
run c2_base c2 A=1
run c2_sc2 c2 FTKX_CULL_STEP_CHUNK=2
run c2_sc4 c2 FTKX_CULL_STEP_CHUNK=4
run c2_sc8 c2 FTKX_CULL_STEP_CHUNK=8
run c5_base c5 A=1
run c5_sc4 c5 FTKX_CULL_STEP_CHUNK=4
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/g03/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'ERR', e); continue
    r=d.get('roofline',{})
    print('%-12s'%f.split('/')[-1][:-5], 'ms/step %.4f'%d['ms_per_step'], 'frac %.4f'%r.get('frac'), {k:round(v,4) for k,v in r.get('kernel_ms_per_pass').items()}, d['check'].get('hits'))
PY
