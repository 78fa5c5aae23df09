#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 -c "
import __graft_entry__ as g
g.smoke(); print('smoke ok (no build, no torch first)')
" 2>&1 | tail -2
python3 -c "
import __graft_entry__ as g
g.build(); g.smoke(); print('smoke ok (build first)')
" 2>&1 | tail -2
