cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
pm() { tag=$1; rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcy_$tag -- python3 bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; f=$(ls -t gpurun_out/pmcy_$tag/*/*counter_collection.csv | head -1); grep "mask_march2_kernel<3, true, false>" $f | awk -F, -v t=$tag '{print t, "FETCH_GB_x2", $(NF-2)*2*1024/1e9}' | tail -1; python3 bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('   mask ms', round(j['roofline']['kernel_ms_per_pass']['mask_kernel'],3))"; }
pm default
FTKX_MASK_ZCHUNK=16 pm z16
FTKX_MASK_ZCHUNK=64 pm z64
FTKX_MASK_WPB=2 pm wpb2
FTKX_MASK_WPB=2 FTKX_MASK_ZCHUNK=64 pm wpb2z64
FTKX_MASK_SWIZZLE=0 pm noswz
