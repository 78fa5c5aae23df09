#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 -c "
import torch; print('torch sees', torch.cuda.is_available(), torch.cuda.device_count())
import __graft_entry__ as g
g.build(); print('built')
g.smoke(); print('smoke ok')
" 2>&1 | tail -5
