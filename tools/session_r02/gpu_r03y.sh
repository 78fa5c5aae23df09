cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export FTKX_MASK_V=6 FTKX_MASK_TILE=3 FTKX_MASK_PD=3 FTKX_MASK_SWIZZLE=40
mkdir -p gpurun_out/pmc6
cp ftk_amd/libftkx.so /tmp/libftkx_orig.so
for lib in orig NO_STORES; do
if [ $lib = orig ]; then cp /tmp/libftkx_orig.so ftk_amd/libftkx.so; else cp tools/probe/variants/libftkx_$lib.so ftk_amd/libftkx.so; fi
rm -rf gpurun_out/pmc6; mkdir -p gpurun_out/pmc6
echo "=== $lib"
for c in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_WRITE_sum" "TCC_READ_sum TCC_HIT_sum" "TCC_MISS_sum TCC_WRITEBACK_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCC_TOO_MANY_EA_WRREQS_STALL_sum SQ_VMEM_WR_TA_DATA_FIFO_FULL" "GRBM_GUI_ACTIVE"; do
  d=gpurun_out/pmc6/$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 tools/ab_mask.py c4 1 -- "V=6" > $d.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc6/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'march' in r['Kernel_Name']:
            acc[r['Kernel_Name'].split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()): print('   %-34s %.4g' % (c, sum(v)/len(v)))
PY
done
cp /tmp/libftkx_orig.so ftk_amd/libftkx.so
