cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp ftk_amd/libftkx.so /tmp/libftkx_orig.so
for lib in orig EXP_USMALL EXP_UQUARTER; do
if [ $lib = orig ]; then cp /tmp/libftkx_orig.so ftk_amd/libftkx.so; else cp tools/probe/variants/libftkx_$lib.so ftk_amd/libftkx.so; fi
echo "=== $lib"
FTKX_MASK_V=6 FTKX_MASK_TILE=3 FTKX_MASK_PD=3 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$lib -- python3 tools/ab_mask.py c4 2 -- "V=6" > gpurun_out/prof_$lib.log 2>&1
python3 - $lib <<'PY'
import csv, glob, sys
for f in glob.glob('gpurun_out/prof_%s/**/*kernel_stats.csv' % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        if 'march' in r['Name']: print(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3)
PY
done
cp /tmp/libftkx_orig.so ftk_amd/libftkx.so
