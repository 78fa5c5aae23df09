bash tools/collect_profiles.sh c4 r02 2>&1 | tail -2
FTKX_MASK_YG=1 bash -c 'cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/profiles_r02_c4/pmc_fetch_yg1 -- python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1'
FTKX_MASK_YG=16 bash -c 'cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/profiles_r02_c4/pmc_fetch_yg16 -- python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1'
FTKX_MASK_TILE=1 bash -c 'cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/profiles_r02_c4/pmc_fetch_tile1 -- python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1'
for c in c3 c2 c5; do bash tools/collect_profiles.sh $c r02 2>&1 | tail -1; done
