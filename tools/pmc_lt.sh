#!/bin/bash
# GPU box: SQ counters of the long-read traceback kernels (39 588 reads of 10 kb = one pass, two calls).   gpurun -- 'bash tools/pmc_lt.sh'
set -o pipefail
R=$(pwd)
export TMPDIR=/tmp
cd /tmp
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/plt/$tag -o run -- python3 $R/tools/mt1_time.py 39588 10000 5000000 0 > $R/gpurun_out/plt_$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $R/gpurun_out/plt_$tag.log; }
  find $R/gpurun_out/plt/$tag -name '*_kernel_trace.csv' -delete
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/plt/*/**/*_counter_collection.csv', recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        kn = row['Kernel_Name']
        if 'lt_' not in kn: continue
        short = kn.split('(')[0].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
        if 'lt_lane' in short: short += ' grid ' + row.get('Grid_Size', '?') + ' lds ' + row.get('LDS_Block_Size', '?')
        per[short][row['Counter_Name']] += float(row['Counter_Value']) / 2     # two calls of the run
    for kern, tot in sorted(per.items()):
        print(kern, {k: round(v / 1e6, 2) for k, v in sorted(tot.items())})
PY
