mkdir -p gpurun_out/r02r
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --config c4 --steps 2 --warmup 1 --no-cpu-baseline --backend gloo --single-device --compact-halo 2>gpurun_out/r02r/c4x2_compact.err > gpurun_out/r02r/c4x2_compact.json; tail -2 gpurun_out/r02r/c4x2_compact.err; python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r02r/c4x2_compact.json") if l.startswith("{")][-1])
print("compact:", j["ms_per_step"], j["halo_exchange"], j["check"])
PY
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --config c4 --steps 2 --warmup 1 --no-cpu-baseline --backend gloo --single-device 2>/dev/null > gpurun_out/r02r/c4x2_full.json; python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r02r/c4x2_full.json") if l.startswith("{")][-1])
print("full:", j["ms_per_step"], j["halo_exchange"], j["other_halo_convention"], j["check"])
PY
