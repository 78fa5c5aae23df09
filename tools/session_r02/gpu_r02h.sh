cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
one() { name=$1; shift
  env "$@" rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02h/pmc_$name -- python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  env "$@" python3 bench.py --config c4 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$name', 'ms/pass %.3f' % j['ms_per_step'], 'mask %.3f' % j['roofline']['kernel_ms_per_pass']['mask_kernel'], j['roofline']['kernel'])
" | tee -a gpurun_out/r02h/variants.txt
}
one t0yg1 FTKX_MASK_YG=1
one t0yg2 FTKX_MASK_YG=2
one t0yg4 FTKX_MASK_YG=4
one t0yg8 FTKX_MASK_YG=8
one t0yg16 FTKX_MASK_YG=16
one t1yg4 FTKX_MASK_TILE=1 FTKX_MASK_YG=4
one t1yg8 FTKX_MASK_TILE=1 FTKX_MASK_YG=8
one t6yg8 FTKX_MASK_TILE=6 FTKX_MASK_YG=8
one t6yg8pd3 FTKX_MASK_TILE=6 FTKX_MASK_YG=8 FTKX_MASK_PD=3
one t2yg8 FTKX_MASK_TILE=2 FTKX_MASK_YG=8
