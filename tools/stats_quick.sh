#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel statistics of a short bench run (no CPU legs), the kernels of this library only.
#   gpurun --timeout 600 -- 'bash tools/stats_quick.sh tag [bench args]'
set -e -o pipefail
TAG=${1:-q}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o runc -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu "$@" > $OUT/stats.json 2> $OUT/stats.err
find $OUT -name '*_kernel_trace.csv' -delete
grep -h "anonymous namespace\|kmahip\|_kernel" $OUT/stats/*kernel_stats.csv $OUT/stats/*/*kernel_stats.csv 2>/dev/null | grep -v "at::native\|rocprim" | cut -d, -f1-4 | cut -c1-160 | head -30
