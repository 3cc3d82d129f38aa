#!/bin/bash
# pconv correctness + PMC counters of the patch kernels (LDS conflicts, LDS busy, waits, MFMA busy)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 400 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "pconv" > $O/r03_pconv_tests.log 2>&1; echo rc=$? >> $O/r03_pconv_tests.log; tail -12 $O/r03_pconv_tests.log
cd /tmp; export TMPDIR=/tmp
rm -rf $O/pmc_pconv
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $O/pmc_pconv -- python3 $R/tools/kbench_pconv.py --iters 2 --skip-old > $O/pmc_pconv.log 2>&1
tail -3 $O/pmc_pconv.log
cd $R
python3 tools/pmc_multi_by_name.py $O/pmc_pconv pconv > $O/r03_pconv_pmc_a.txt 2>&1; cat $O/r03_pconv_pmc_a.txt
rm -rf $O/pmc_pconv
