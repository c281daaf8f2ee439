#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s15
mkdir -p $O
cd $R
timeout 900 python3 tools/ln_shapes.py > $O/ln_shapes.log 2>&1; cat $O/ln_shapes.log | cut -c1-260
