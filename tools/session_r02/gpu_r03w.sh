cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export FTKX_MASK_V=6 FTKX_MASK_TILE=3 FTKX_MASK_PD=3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v6 -- python3 tools/ab_mask.py c4 2 -- "V=6" > gpurun_out/prof_v6.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/prof_v6/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ftkx' in r['Name']: print(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3)
PY
