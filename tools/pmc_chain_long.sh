#!/bin/bash
# GPU box: SQ counters of chain_kernel (the lane-per-read route of stage 2) on 2 000 long reads.   gpurun -- 'bash tools/pmc_chain_long.sh'
set -o pipefail
R=$(pwd)
export TMPDIR=/tmp
export CHAIN_STOPS=${CHAIN_STOPS:-0}
cd /tmp
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_FLAT" "SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pcl/$tag -o run -- python3 $R/tools/chain_long_time.py > $R/gpurun_out/pcl_$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $R/gpurun_out/pcl_$tag.log; }
  find $R/gpurun_out/pcl/$tag -name '*_kernel_trace.csv' -delete
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pcl/*/**/*_counter_collection.csv', recursive=True)):
    tot = collections.defaultdict(float); disp = set()
    for row in csv.DictReader(open(f)):
        if 'chain_kernel' in row['Kernel_Name']:
            tot[row['Counter_Name']] += float(row['Counter_Value']); disp.add(row['Dispatch_Id'])
    n = max(1, len(disp))
    print('chain_kernel', {k: round(v / n / 1e6, 3) for k, v in sorted(tot.items())}, 'x', n)
PY
