bash tools/pmc_valu.sh c4 2>&1 | tail -8
bash tools/pmc_valu.sh c3 --exact-only 2>&1 | grep -i tile | tail -3
