#!/bin/bash
# Integer / FP VALU utilisation next to the HBM figure: separate PMC passes over one bench run each (derived metrics need
# several raw counters; no trace domains mixed in).  usage: bash tools/pmc_valu.sh <config> [extra bench args]   -> gpurun_out/r04_<config>_valu_summary.json (copy into profiles/)
CFG=${1:-c4}; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_valu_$CFG; rm -rf $OUT; mkdir -p $OUT
for c in "SQ_INSTS_VALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE" VALUBusy; do
  d=$OUT/$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-streaming-tracker "$@" > $d.log 2>&1
done
python3 - "$CFG" <<'PY'
import csv, glob, collections, json, os, sys
cfg = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'gpurun_out/pmc_valu_{cfg}/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'ftkx::' in r['Kernel_Name']:
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    d = dict(m)
    # SQ_* cycle counters count quad-cycles summed over the SIMDs that ran the kernel's waves; SQ_BUSY_CYCLES is per SE-quad-cycle
    if 'SQ_ACTIVE_INST_VALU' in m and 'SQ_WAVE_CYCLES' in m and m['SQ_WAVE_CYCLES']:
        d['valu_share_of_wave_cycles'] = m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES']
    if 'SQ_WAIT_ANY' in m and 'SQ_WAVE_CYCLES' in m and m['SQ_WAVE_CYCLES']:
        d['waiting_share_of_wave_cycles'] = m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']
    if 'SQ_INSTS_VALU' in m and 'SQ_WAVES' in m and m['SQ_WAVES']:
        d['valu_instructions_per_wave'] = m['SQ_INSTS_VALU'] / m['SQ_WAVES']
    out[k] = d
os.makedirs('profiles', exist_ok=True)
json.dump({'config': cfg, 'note': 'average per dispatch; rocprofv3 --pmc, one counter group per run of bench.py --steps 1; derived VALUBusy uses the gfx94x formula (no gfx950 section in ROCm 7.2)', 'kernels': out},
          open(f'gpurun_out/r04_{cfg}_valu_summary.json', 'w'), indent=1)
for k, d in out.items():
    print(k, {c: (round(v, 4) if v < 1000 else int(v)) for c, v in d.items()})
PY
