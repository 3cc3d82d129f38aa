#!/bin/bash
# last check of the committed state: GPU suite, smoke(), the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r03_final_tests.log 2>&1; echo "pytest rc=$?"; tail -2 $O/r03_final_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 400 python bench.py > $O/r03_final_bench.json 2> $O/r03_final_bench.err; echo "bench rc=$?"; head -c 220 $O/r03_final_bench.json; echo
