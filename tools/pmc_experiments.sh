cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
pm() { tag=$1; rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcx_$tag -- python3 bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; grep mask_march2 gpurun_out/pmcx_$tag/*/*counter_collection.csv | awk -F, -v t=$tag '{print t, "FETCH_GB_x2", $(NF-2)*2*1024/1e9}' | tail -1; }
pm default
FTKX_MASK_WPB=12 pm wpb12
FTKX_MASK_ZCHUNK=128 pm z128
FTKX_MASK_WPB=12 FTKX_MASK_ZCHUNK=512 pm wpb12z512
FTKX_MASK_EDGE=0 pm halo
