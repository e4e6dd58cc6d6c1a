#!/bin/bash
# GPU box: SQ counters of the default mode's stage-2 kernels (2 M reads).   gpurun -- 'bash tools/pmc_chain.sh'
set -o pipefail
R=$(pwd)
export TMPDIR=/tmp
cd /tmp
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pc/$tag -o run -- python3 $R/tools/chain_stage_time.py 2000000 > $R/gpurun_out/pc_$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $R/gpurun_out/pc_$tag.log; }
  find $R/gpurun_out/pc/$tag -name '*_kernel_trace.csv' -delete
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pc/*/**/*_counter_collection.csv', recursive=True)):
    for kern in ('chain_fast_kernel', 'chain_anchor_kernel'):
        tot = collections.defaultdict(float); disp = set()
        for row in csv.DictReader(open(f)):
            if kern in row['Kernel_Name']:
                tot[row['Counter_Name']] += float(row['Counter_Value']); disp.add(row['Dispatch_Id'])
        n = max(1, len(disp))
        print(kern, {k: round(v / n / 1e6, 2) for k, v in sorted(tot.items())}, 'x', n)
PY
