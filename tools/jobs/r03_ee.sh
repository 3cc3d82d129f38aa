#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -E "^\s*(Name|counter)|SQ_LDS|SQ_WAIT|SQ_ACTIVE_INST|SQ_INSTS_|SQ_WAVE_CYCLES|SQ_BUSY_CY|TCP_|TA_BUSY|SQ_INST_CYCLES|LDSBankConflict|MemUnit|Stall" | head -80 > $O/r03_counters.txt 2>&1
wc -l $O/r03_counters.txt; head -60 $O/r03_counters.txt
