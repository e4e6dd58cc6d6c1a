#!/bin/bash
# rocprofv3 kernel statistics of the whole device pipeline (stage 2 ... consensus) on the GPU box: a 10 M-read FASTQ against the 5k-gene
# index through examples/kmahip_map. usage: tools/prof_pipeline.sh [reads [mode [name]]] ; mode = -1t1 (single end), -chain (the default
# mode) or -ipe (reads / 2 pairs); PROF_DB=50k: against the 50 k-gene database of C5; summaries land in gpurun_out/<name> (prof_pipeline)
set -e
N=${1:-10000000}
MODE=${2:--1t1}
OUT=gpurun_out/${3:-prof_pipeline}
mkdir -p $OUT
export TMPDIR=/tmp
W=$(mktemp -d /tmp/profpipe.XXXX)
python3 - "$N" "$W" "$MODE" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
from kma_amd import formats, synth
n, w, mode = int(sys.argv[1]), sys.argv[2], sys.argv[3]
if os.environ.get("PROF_DB") == "50k":      # the database of config C5: 5 000 families x 10 variants, indexed by examples/kmahip_index
    import subprocess
    names, seqs = synth.make_gene_db(5000, 10, 600, 1500, 0.04, seed=4321)
    synth.write_fasta(os.path.join(w, "db.fsa"), names, seqs)
    subprocess.check_call(["make", "-C", "examples"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["examples/kmahip_index", "-i", os.path.join(w, "db.fsa"), "-o", os.path.join(w, "db5k")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
else:
    names, seqs = synth.make_gene_db(1000, 5, 600, 1500, 0.04, seed=12345)
    formats.write_index(os.path.join(w, "db5k"), names, seqs)
if mode == "-ipe":
    with open(os.path.join(w, "r1.fq"), "wb") as f1, open(os.path.join(w, "r2.fq"), "wb") as f2:
        for a in range(0, n // 2, 250_000):
            m1, m2, _ = synth.make_pairs(seqs, min(250_000, n // 2 - a), seed=500 + a)
            for f, mm in ((f1, m1), (f2, m2)):
                bench.write_fastq_fixed(os.path.join(w, "part.fq"), mm)
                f.write(open(os.path.join(w, "part.fq"), "rb").read())
else:
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
if [ "$MODE" = "-ipe" ]; then
	CMD="$ROOT/examples/kmahip_map -ipe $W/r1.fq $W/r2.fq -t_db $W/db5k -o $W/out -1t1"
	SAY="kmahip_map -ipe r1.fq r2.fq ($((N / 2)) pairs of 2 x 150 bp) -t_db db5k -o out -1t1"
else
	CMD="$ROOT/examples/kmahip_map -i $W/reads.fq -t_db $W/db5k -o $W/out $MODE"
	SAY="kmahip_map -i reads.fq ($N reads x 150 bp) -t_db db5k -o out $MODE"
fi
if [ "$PROF_DB" = "50k" ]; then SAY="$SAY   (db5k here = the 50 k-gene database of config C5)"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -o pipe -- $CMD > $ROOT/$OUT/run.log 2>&1
# HBM traffic counters in passes of their own (never together with the traces gpurun refuses)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $W/fetch -o pipe -- $CMD > $ROOT/$OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $W/write -o pipe -- $CMD > $ROOT/$OUT/write.log 2>&1
if [ -n "$PROF_SQ" ]; then      # instruction mix and L2 behaviour per kernel (PROF_SQ=1), again in passes of their own
	rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $W/sq -o pipe -- $CMD > $ROOT/$OUT/sq.log 2>&1
	rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $W/tcc -o pipe -- $CMD > $ROOT/$OUT/tcc.log 2>&1
	rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $W/sqc -o pipe -- $CMD > $ROOT/$OUT/sqc.log 2>&1 || true
	python3 - $W $ROOT/$OUT/sq_counters.txt <<'PY'
import csv, glob, re, sys
tot = {}
for d in ("sq", "tcc", "sqc"):
    for path in glob.glob(f"{sys.argv[1]}/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            m = re.search(r"(\w+_kernel)(<[^>]*>)?", row["Kernel_Name"])
            k = (m.group(1) + (m.group(2) or "")) if m else row["Kernel_Name"][:50]
            tot.setdefault(k, {}).setdefault(row["Counter_Name"], 0.0)
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
with open(sys.argv[2], "w") as f:
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
        f.write(k + ": " + ", ".join(f"{c} {x:.4g}" for c, x in sorted(v.items())) + "\n")
PY
fi
cd $ROOT
echo "$SAY" > $OUT/cmd.txt
grep "kmahip_map:" $OUT/run.log | tail -1 > $OUT/stages.txt
python3 tools/collect_pipeline_profile.py $W $OUT
rm -rf $W
