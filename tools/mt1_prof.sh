#!/bin/bash
# rocprofv3 kernel statistics of the C4-shaped run (tools/mt1_time.py): gpurun -- 'bash tools/mt1_prof.sh [reads]'
set -e -o pipefail
REPO=$(pwd)
OUT=$REPO/gpurun_out/mt1_prof
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $REPO/tools/mt1_time.py ${1:-20000} 10000 5000000 0 > $OUT/out.txt 2> $OUT/err.txt
cat $OUT/out.txt | tail -5
F=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
head -25 $F | cut -d, -f1-8
find $OUT -name '*_kernel_trace.csv' -size +20M -delete
