for c in c4 c3 c2 c5; do bash tools/collect_profiles.sh $c r02 2>&1 | tail -1 | cut -c1-200; done
bash tools/pmc_valu.sh c4 2>&1 | tail -8 | cut -c1-400
python bench.py --config c3 --exact-only --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_c3_exact_only_bench.json 2>/dev/null; tail -1 gpurun_out/r02_c3_exact_only_bench.json | cut -c1-300
python bench.py --config c3o --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_c3o_bench.json 2>/dev/null; tail -1 gpurun_out/r02_c3o_bench.json | cut -c1-300
