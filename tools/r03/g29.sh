#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g33; rm -rf $O; mkdir -p $O
for fan in 2 1; do
FTKX_TILE_FAN=$fan python3 - <<P > $O/stamps_$fan.txt 2>&1
import ctypes, subprocess, sys, os
sys.argv = ["bench.py", "--config", "c3", "--exact-only", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
import runpy
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
import ftk_amd._lib as L
lib = ctypes.CDLL(os.path.join(os.path.dirname(L.__file__), "libftkx.so"))
out = (ctypes.c_ulonglong * 8)()
lib.ftkx_debug_tile_stamps(out, 1)
v = list(out); n = max(v[7], 1)
print("waves", v[7], "cycles per wave (100 MHz ticks?) by phase:", [round(x / n, 1) for x in v[:6]])
P
tail -2 $O/stamps_$fan.txt | cut -c1-200
done
