cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 tools/probe/bw_probe m | tail -12
rm -rf gpurun_out/probe_pmc; rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/probe_pmc/f -- tools/probe/bw_probe m > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/probe_pmc/w -- tools/probe/bw_probe m > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag in ('f','w'):
    for f in glob.glob('gpurun_out/probe_pmc/%s/*/*counter_collection.csv' % tag):
        acc = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][-44:]
            acc.setdefault(k, []).append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(tag, k, 'GB=%.2f' % (sum(v) / len(v) * 1024 * (2 if tag == 'f' else 1) / 1e9), len(v))
PY
