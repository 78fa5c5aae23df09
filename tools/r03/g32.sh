#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g32; rm -rf $O; mkdir -p $O
for fan in 0 1 2; do
FTKX_TILE_FAN=$fan timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -k "exact_only" > $O/parity_$fan.log 2>&1; echo "fan $fan rc=$?"; grep -E "^FAILED|passed|failed" $O/parity_$fan.log | cut -c1-200 | head -20
done
timeout -k 10 300 python3 -m pytest tests/test_gpu_properties.py -q -k "tile_kernel_forms" > $O/forms.log 2>&1; echo "forms rc=$?"; grep -E "^FAILED|passed|failed|Error" $O/forms.log | cut -c1-200 | head
