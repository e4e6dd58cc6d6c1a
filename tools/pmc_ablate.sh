#!/bin/bash
# GPU box: SQ instruction counters of the scan kernel per ablation variant (diagnostic build).
#   gpurun -- 'bash tools/pmc_ablate.sh "0 1 3 7 8 16"'
set -o pipefail
R=$(pwd)
export TMPDIR=/tmp KMAHIP_LIB=$R/kma_amd/libkmahip_diag.so
cd /tmp
for v in $1; do
  export KMAHIP_ABLATE_SCAN=$v
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pa/v$v -o run -- python3 $R/tools/scan_only.py 2000000 2 both > $R/gpurun_out/pa_v$v.log 2>&1 || { echo "variant $v failed"; tail -3 $R/gpurun_out/pa_v$v.log; }
  find $R/gpurun_out/pa/v$v -name '*_kernel_trace.csv' -delete
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/pa/v*'), key=lambda x: int(x.rsplit('v',1)[1])):
    f = glob.glob(d + '/**/*_counter_collection.csv', recursive=True)
    if not f: continue
    for kern in ('scan_prefilter_kernel<false', 'scan_se_kernel<false', 'seed_tasks_kernel', 'align_tasks_kernel<false'):
        tot = collections.defaultdict(float); disp = set()
        for row in csv.DictReader(open(f[0])):
            if kern in row['Kernel_Name']:
                tot[row['Counter_Name']] += float(row['Counter_Value']); disp.add(row['Dispatch_Id'])
        n = max(1, len(disp))
        print(os.path.basename(d), kern.split('<')[0], {k: round(v / n / 1e6, 1) for k, v in sorted(tot.items())}, 'x', n)
PY
