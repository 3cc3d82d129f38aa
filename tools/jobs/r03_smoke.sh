#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python -c "
import __graft_entry__ as g, time
t=time.time(); g.smoke(); print('smoke ok', round(time.time()-t,1))
"
