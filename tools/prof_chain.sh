#!/bin/bash
# GPU box: kernel statistics of stage 2 of the default mode (10 M short reads; 2 000 long reads) + the SQ counters of pmc_chain.sh.
#   gpurun -- 'bash tools/prof_chain.sh'   ->  gpurun_out/pch/
set -o pipefail
R=$(pwd)
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pch
cd /tmp
CHAIN_STOPS=0 KMAHIP_CHAIN_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pch/short -o run -- python3 $R/tools/chain_stage_time.py 10000000 > $R/gpurun_out/pch/short.log 2>&1
CHAIN_STOPS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pch/long -o run -- python3 $R/tools/chain_long_time.py 20000 > $R/gpurun_out/pch/long.log 2>&1
find $R/gpurun_out/pch -name '*_kernel_trace.csv' -delete
cd $R
grep "fast route," gpurun_out/pch/short.log | tail -1
bash tools/pmc_chain.sh > gpurun_out/pch/counters.txt 2>&1
tail -3 gpurun_out/pch/counters.txt
