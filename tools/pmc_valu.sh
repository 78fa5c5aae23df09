#!/bin/bash
# VALU utilisation of the mask kernel: separate PMC passes (derived metrics need several raw counters each)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_valu; rm -rf $OUT; mkdir -p $OUT
for c in VALUBusy SALUBusy MemUnitStalled "SQ_INSTS_VALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" MemUnitBusy FetchSize; do
  d=$OUT/$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline > $d.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/pmc_valu/*/')):
    for f in glob.glob(d + '*/*counter_collection.csv'):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'mask_march2_kernel<3, true, false>' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(os.path.basename(d.rstrip('/')), k, sum(v) / len(v), len(v))
PY
