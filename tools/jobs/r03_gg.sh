#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
( timeout -k 10 100 python tools/kbench_tall.py --iters 8; VQA_LIB=build_var/libvqa_talldiag.so timeout -k 10 100 python tools/kbench_tall.py --iters 8 --dbg 0,1,16,8,0 ) 2>&1 | grep -v amdgpu.ids
cd /tmp; export TMPDIR=/tmp; rm -rf $O/r03_pmt_c
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_TCP_TA_DATA_STALL_CYCLES TCP_PENDING_STALL_CYCLES TA_BUSY_avr -d $O/r03_pmt_c -- python3 $R/tools/kbench_tall.py --iters 3 > $O/r03_pmt_c.log 2>&1
cd $R; python3 tools/pmc_multi_by_name.py $O/r03_pmt_c gemm_tall; rm -rf $O/r03_pmt_c
