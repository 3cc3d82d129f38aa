#!/bin/bash
# per-layer conv traffic (PMC FETCH_SIZE / WRITE_SIZE, separate passes) at the two bench shapes -> the json bench.py reads
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
T="timeout -k 10"
rm -rf $O/r03_pmcc_*
for c in FETCH_SIZE WRITE_SIZE; do
  $T 200 rocprofv3 --pmc $c -d $O/r03_pmcc_fp32_$c -- python3 $R/tools/pmc_conv_run.py > $O/r03_pmcc_fp32_$c.log 2>&1 || exit 1
  $T 200 rocprofv3 --pmc $c -d $O/r03_pmcc_bf16_$c -- python3 $R/tools/pmc_conv_run.py --dtype bf16 --batch 512 --size 448 > $O/r03_pmcc_bf16_$c.log 2>&1 || exit 1
done
cd $R
python3 tools/pmc_traffic_summary.py $O/r03_pmcc_fp32_FETCH_SIZE $O/r03_pmcc_fp32_WRITE_SIZE > $O/r03_conv_traffic_fp32_224_256.json
python3 tools/pmc_traffic_summary.py $O/r03_pmcc_bf16_FETCH_SIZE $O/r03_pmcc_bf16_WRITE_SIZE > $O/r03_conv_traffic_bf16_448_512.json
rm -rf $O/r03_pmcc_*_SIZE
cat $O/r03_conv_traffic_fp32_224_256.json $O/r03_conv_traffic_bf16_448_512.json
