#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $O/clk_*
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/clk_x3 -- python3 $R/bench.py --dtype fp32x3 --steps 3 --warmup 2 --no-cpu-baseline --stream-steps 0 > $O/clk_x3.log 2>&1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/clk_fp32 -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-x3 --stream-steps 0 > $O/clk_fp32.log 2>&1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/clk_bf16 -- python3 $R/bench.py --dtype bf16 --steps 3 --warmup 2 --no-cpu-baseline --stream-steps 0 > $O/clk_bf16.log 2>&1
cd $R
python3 tools/pmc_clock_by_name.py $O/clk_x3 > $O/clock_x3.txt 2>&1; head -14 $O/clock_x3.txt
python3 tools/pmc_clock_by_name.py $O/clk_fp32 > $O/clock_fp32.txt 2>&1; head -12 $O/clock_fp32.txt
python3 tools/pmc_clock_by_name.py $O/clk_bf16 > $O/clock_bf16.txt 2>&1; head -10 $O/clock_bf16.txt
rm -rf $O/clk_x3 $O/clk_fp32 $O/clk_bf16
