#!/bin/bash
# Run ON THE GPU BOX (via gpurun): bench line + rocprofv3 kernel stats + separate FETCH/WRITE PMC passes.
#   gpurun --timeout 1100 -- 'bash tools/profile_gpu.sh r1f'
# then here:  python tools/collect_profiles.py gpurun_out/r1f r1
set -e -o pipefail
TAG=${1:-prof}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
PCMD="python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu"
echo "python3 bench.py --steps 3 --warmup 1 --no-cpu  (PMC passes: --steps 1 --warmup 0)" > $OUT/cmd.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o runc -- $PCMD > $OUT/stats.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o runc -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o runc -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu > $OUT/write.json 2> $OUT/write.err
# keep the merge-back small: traces are large, stats/counters are what we need
find $OUT -name '*_kernel_trace.csv' -size +20M -delete
tail -1 $OUT/bench.json
