cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g28
python bench.py --gpus 1 --config c3 --timesteps 64 --steps 2 --warmup 1 --no-cpu-baseline --dump-merged gpurun_out/g28/one.npz > gpurun_out/g28/one.json 2> gpurun_out/g28/one.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 4 --config c3 --steps 2 --warmup 1 --no-cpu-baseline --backend gloo --single-device --dump-merged gpurun_out/g28/many.npz > gpurun_out/g28/many.json 2> gpurun_out/g28/many.err
python - <<'PY'
import numpy as np, json
a=np.load('gpurun_out/g28/one.npz'); b=np.load('gpurun_out/g28/many.npz')
print('records', len(a['records']), len(b['records']), 'identical', a['records'].tobytes()==b['records'].tobytes(), 'curves equal', all(np.array_equal(a[k],b[k]) for k in ('curve_offsets','curve_indices','curve_loop')))
for f in ('one','many'):
    d=json.loads([l for l in open(f'gpurun_out/g28/{f}.json') if l.startswith('{')][-1])
    print(f, d['n_gpus'], d['scaling'], 'ms/step %.3f'%d['ms_per_step'], d['config']['workload'][:70], d['halo_exchange'] and {k:d['halo_exchange'][k] for k in ('in_timed_region','ms')}, d['check'])
PY
tail -2 gpurun_out/g28/many.err
