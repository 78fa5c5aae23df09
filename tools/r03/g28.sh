#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g28; rm -rf $O; mkdir -p $O
export FTKX_TILE_FAN=${1:-2}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc1 -- python3 bench.py --config c3 --exact-only --steps 1 --warmup 0 --no-cpu-baseline > $O/b1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM --output-format csv -d $O/pmc2 -- python3 bench.py --config c3 --exact-only --steps 1 --warmup 0 --no-cpu-baseline > $O/b2.log 2>&1
python3 - <<'P'
import csv, glob, collections
for d in ("pmc1", "pmc2"):
    for f in glob.glob("gpurun_out/r03_g28/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "tile_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in acc: print(d, k, acc[k], n[k])
P
