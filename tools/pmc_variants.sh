#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the mask kernel under placement / load-policy variants (one rocprofv3 --pmc pass each)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_variants; rm -rf $OUT; mkdir -p $OUT
i=0
while read -r name envs; do
  i=$((i+1))
  for c in FETCH_SIZE; do
    d=$OUT/${name}_$c
    ( export $envs; rocprofv3 --pmc $c --output-format csv -d $d -- python3 bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline > $d.log 2>&1 )
  done
  ( export $envs; python3 bench.py --config c4 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$name', 'mask_ms=%.3f' % j['roofline']['kernel_ms_per_pass']['mask_kernel'])" )
done <<'LIST'
default FTKX_DUMMY=0
swz1 FTKX_MASK_SWIZZLE=1
grp1 FTKX_MASK_SWIZZLE=8
grp4 FTKX_MASK_SWIZZLE=8 FTKX_MASK_YG=4
grp32 FTKX_MASK_SWIZZLE=8 FTKX_MASK_YG=32
ntpriv FTKX_MASK_SWIZZLE=16
ntall FTKX_MASK_SWIZZLE=4
wpb8 FTKX_MASK_WPB=8
wpb8grp4 FTKX_MASK_WPB=8 FTKX_MASK_SWIZZLE=8 FTKX_MASK_YG=4
z128 FTKX_MASK_ZCHUNK=128
LIST
python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/pmc_variants/*_FETCH_SIZE/')):
    for f in glob.glob(d + '*/*counter_collection.csv'):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'mask_march' in r['Kernel_Name'] and ', true>' not in r['Kernel_Name'] and 'true, 1>' not in r['Kernel_Name']:
                acc[(r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(os.path.basename(d.rstrip('/')), k, 'GB=%.2f' % (sum(v) / len(v) * 1024 * 2 / 1e9), len(v))
PY
