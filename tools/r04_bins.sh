#!/bin/bash
cd $GRAFT_REPO_ROOT
L="--steps 24 --warmup 5 --no-cpu-baseline --no-other-configs --no-streaming-tracker"
for cfg in c2 c5; do
  echo -n "$cfg: "; python3 bench.py --config $cfg $L 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],4), j['ms_per_step_samples'])"
  echo -n "$cfg no copy kernel: "; FTKX_SERIES_COPY=0 python3 bench.py --config $cfg $L 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],4), j['ms_per_step_samples'])"
done
