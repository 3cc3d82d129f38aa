#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $O/r03_pmt_*
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/r03_pmt_a -- python3 $R/tools/kbench_tall.py --iters 3 > $O/r03_pmt_a.log 2>&1 || { tail -5 $O/r03_pmt_a.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU -d $O/r03_pmt_b -- python3 $R/tools/kbench_tall.py --iters 3 > $O/r03_pmt_b.log 2>&1 || { tail -5 $O/r03_pmt_b.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_TCP_TA_DATA_STALL_CYCLES TCP_PENDING_STALL_CYCLES TA_BUSY_avr -d $O/r03_pmt_c -- python3 $R/tools/kbench_tall.py --iters 3 > $O/r03_pmt_c.log 2>&1 || tail -3 $O/r03_pmt_c.log
cd $R
for x in a b c; do python3 tools/pmc_multi_by_name.py $O/r03_pmt_$x gemm_tall; done > $O/r03_tall_pmc.txt 2>&1
cat $O/r03_tall_pmc.txt
rm -rf $O/r03_pmt_a $O/r03_pmt_b $O/r03_pmt_c
