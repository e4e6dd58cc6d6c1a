#!/bin/bash
# rocprofv3 kernel statistics of the whole device pipeline (stage 2 ... consensus) on the GPU box: a 10 M-read FASTQ against the 5k-gene
# index through examples/kmahip_map. usage: tools/prof_pipeline.sh [reads] ; summaries land in gpurun_out/prof_pipeline/
set -e
N=${1:-10000000}
OUT=gpurun_out/prof_pipeline
mkdir -p $OUT
export TMPDIR=/tmp
W=$(mktemp -d /tmp/profpipe.XXXX)
python3 - "$N" "$W" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
from kma_amd import formats, synth
n, w = int(sys.argv[1]), sys.argv[2]
names, seqs = synth.make_gene_db(1000, 5, 600, 1500, 0.04, seed=12345)
formats.write_index(os.path.join(w, "db5k"), names, seqs)
with open(os.path.join(w, "reads.fq"), "wb") as f:
    for a in range(0, n, 2_000_000):
        m = min(2_000_000, n - a)
        codes, _, _, _ = synth.make_reads(seqs, m, seed=1000 + a)
        bench.write_fastq_fixed(os.path.join(w, "part.fq"), codes)
        f.write(open(os.path.join(w, "part.fq"), "rb").read())
PY
make -C examples >/dev/null
export KMAHIP_MAP_TEARDOWN=1
ROOT=$(pwd)
cd /tmp
CMD="$ROOT/examples/kmahip_map -i $W/reads.fq -t_db $W/db5k -o $W/out -1t1"
rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -o pipe -- $CMD > $ROOT/$OUT/run.log 2>&1
# HBM traffic counters in passes of their own (never together with the traces gpurun refuses)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $W/fetch -o pipe -- $CMD > $ROOT/$OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $W/write -o pipe -- $CMD > $ROOT/$OUT/write.log 2>&1
cd $ROOT
echo "kmahip_map -i reads.fq ($N reads x 150 bp) -t_db db5k -o out -1t1" > $OUT/cmd.txt
grep "kmahip_map:" $OUT/run.log | tail -1 > $OUT/stages.txt
python3 tools/collect_pipeline_profile.py $W $OUT
rm -rf $W
