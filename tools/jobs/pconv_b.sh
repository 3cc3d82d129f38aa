#!/bin/bash
# HBM-side traffic and L2 hit rate of the patch kernels (separate --pmc passes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $O/pmc_pc1 $O/pmc_pc2
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_pc1 -- python3 $R/tools/kbench_pconv.py --iters 1 --skip-old > $O/pmc_pc1.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum WRITE_SIZE -d $O/pmc_pc2 -- python3 $R/tools/kbench_pconv.py --iters 1 --skip-old > $O/pmc_pc2.log 2>&1
cd $R
python3 tools/pmc_multi_by_name.py $O/pmc_pc1 pconv > $O/r03_pconv_pmc_fetch.txt 2>&1; cat $O/r03_pconv_pmc_fetch.txt
python3 tools/pmc_multi_by_name.py $O/pmc_pc2 pconv > $O/r03_pconv_pmc_l2.txt 2>&1; cat $O/r03_pconv_pmc_l2.txt
tail -2 $O/pmc_pc2.log
rm -rf $O/pmc_pc1 $O/pmc_pc2
