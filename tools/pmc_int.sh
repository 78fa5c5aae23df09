#!/bin/bash
# Kernel trace and VALU counters of the integer regime ("int-VALU utilisation alongside"): separate rocprofv3 runs of
# tools/int_regime.py -- one --kernel-trace --stats run, then one --pmc run per counter group (no trace domains mixed in).
#   bash tools/pmc_int.sh <c3_exact_only|c3o> <label>    -> gpurun_out/r06_<label>_kernel_stats.csv, gpurun_out/r06_<label>_valu_summary.json  (copy into profiles/)
NAME=${1:-c3_exact_only}; LABEL=${2:-c3x}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_int_$LABEL; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/int_regime.py $NAME > $OUT/trace.log 2>&1
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) gpurun_out/r06_${LABEL}_kernel_stats.csv
grep "^{\"workload\"" $OUT/trace.log | tail -1 > gpurun_out/r06_${LABEL}_bench.json
for c in "SQ_INSTS_VALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE" VALUBusy; do
  d=$OUT/$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 tools/int_regime.py $NAME > $d.log 2>&1
done
python3 - "$LABEL" <<'PY'
import csv, glob, collections, json, subprocess, sys
label = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'gpurun_out/pmc_int_{label}/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'ftkx::' in r['Kernel_Name']:
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    d = dict(m)
    d['dispatches'] = max(len(v) for v in cs.values())
    if 'SQ_ACTIVE_INST_VALU' in m and m.get('SQ_WAVE_CYCLES'):
        d['valu_share_of_wave_cycles'] = m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES']
    if 'SQ_INSTS_VALU' in m and m.get('SQ_WAVES'):
        d['valu_instructions_per_wave'] = m['SQ_INSTS_VALU'] / m['SQ_WAVES']
    if 'VALUBusy' in m:
        d['valu_busy'] = m['VALUBusy'] / 100.0
    out[k] = d
head = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip() or None
json.dump({'config': label, 'kernel_sources_at': head,
           'note': 'average per dispatch; rocprofv3 --pmc, one counter group per run of tools/int_regime.py; VALUBusy is rocprofv3\'s derived metric (gfx94x formula: no gfx950 section in ROCm 7.2)', 'kernels': out},
          open(f'gpurun_out/r06_{label}_valu_summary.json', 'w'), indent=1)
for k, d in out.items():
    if 'tile_kernel' in k:
        print(k, {c: (round(v, 4) if v < 1000 else int(v)) for c, v in d.items()})
PY
