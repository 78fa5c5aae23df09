cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 1 3; do
FTKX_MASK_SWIZZLE=$v rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcx_$v -- python3 bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
grep mask_march2 gpurun_out/pmcx_$v/*/*counter_collection.csv | awk -F, -v v=$v '{print "swz="v, "FETCH_KB", $(NF-2)}' | tail -1
done
FTKX_MASK_ZCHUNK=512 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcx_z512 -- python3 bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
grep mask_march2 gpurun_out/pmcx_z512/*/*counter_collection.csv | awk -F, '{print "z512 FETCH_KB", $(NF-2)}' | tail -1
FTKX_MASK_SWIZZLE=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcx_0 -- python3 bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
grep mask_march2 gpurun_out/pmcx_0/*/*counter_collection.csv | awk -F, '{print "noswz FETCH_KB", $(NF-2)}' | tail -1
