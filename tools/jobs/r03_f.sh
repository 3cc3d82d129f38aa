#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 300 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "pconv or patch or configs3" > $O/r03_tests_g.log 2>&1; rc=$?; tail -3 $O/r03_tests_g.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/kbench_pconv.py --skip-old > $O/r03_kbench_pconv_g.txt 2>&1; cat $O/r03_kbench_pconv_g.txt
cd /tmp; export TMPDIR=/tmp
rm -rf $O/r03_pm_*
timeout -k 10 300 rocprofv3 --pmc MfmaUtil --kernel-trace -d $O/r03_pm_mfma -- python3 $R/bench.py --dtype bf16 --batch 512 --size 448 --steps 2 --warmup 1 --no-cpu-baseline --stream-steps 0 > $O/r03_pm_mfma.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/r03_pm_clk -- python3 $R/bench.py --dtype bf16 --batch 512 --size 448 --steps 2 --warmup 1 --no-cpu-baseline --stream-steps 0 > $O/r03_pm_clk.log 2>&1 || exit 1
cd $R
python3 tools/pmc_by_name.py $O/r03_pm_mfma MfmaUtil > $O/r03_step_mfma_util_bf16_448.txt; head -24 $O/r03_step_mfma_util_bf16_448.txt
python3 tools/pmc_clock_by_name.py $O/r03_pm_clk > $O/r03_clock_bf16_448.txt; head -24 $O/r03_clock_bf16_448.txt
rm -rf $O/r03_pm_mfma $O/r03_pm_clk
