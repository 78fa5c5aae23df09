#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g14; rm -rf $O; mkdir -p $O
run() { # label, env...
  lbl=$1; shift
  env "$@" python3 bench.py --config c3 --steps 40 --warmup 3 --no-cpu-baseline --no-other-configs --no-streaming-tracker > $O/$lbl.json 2>$O/$lbl.err
  tail -1 $O/$lbl.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$lbl', round(j['ms_per_step'],4), 'mask', round(j['roofline']['avg_launch_ms'],4), round(j['roofline']['frac'],3), 'e2e', round(j['roofline_end_to_end']['frac'],3))"
}
run default A=1
run zc16 FTKX_MASK_ZCHUNK=16
run zc64 FTKX_MASK_ZCHUNK=64
run tail2_8 FTKX_MASK_TAIL=2,8
run tail4_8 FTKX_MASK_TAIL=4,8
run tail1_16 FTKX_MASK_TAIL=1,16
run tail2_16 FTKX_MASK_TAIL=2,16
run tail0 FTKX_MASK_TAIL=0,32
run yg8 FTKX_MASK_YG=8
run yg4 FTKX_MASK_YG=4
run default2 A=1
