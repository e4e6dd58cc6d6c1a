#!/bin/bash
# the file-to-file run with every KMAHIP_DEBUG_TIMING line (10 M reads by default): gpurun -- 'tools/map_debug.sh [reads [mode]]', mode = -1t1 | -chain
set -e
N=${1:-10000000}
MODE=${2:--1t1}
W=$(mktemp -d /tmp/mapdbg.XXXX)
python3 - "$N" "$W" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
from kma_amd import formats, synth
n, w = int(sys.argv[1]), sys.argv[2]
if os.environ.get("PROF_DB") == "50k":      # the database of config C5: 5 000 families x 10 variants, indexed by examples/kmahip_index
    import subprocess
    names, seqs = synth.make_gene_db(5000, 10, 600, 1500, 0.04, seed=4321)
    synth.write_fasta(os.path.join(w, "db.fsa"), names, seqs)
    subprocess.check_call(["make", "-C", "examples"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["examples/kmahip_index", "-i", os.path.join(w, "db.fsa"), "-o", os.path.join(w, "db5k")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
else:
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
examples/kmahip_map -i $W/reads.fq -t_db $W/db5k -o $W/out $MODE 2>&1 | tail -1; KMAHIP_DEBUG_TIMING=1 examples/kmahip_map -i $W/reads.fq -t_db $W/db5k -o $W/out $MODE 2>&1 | grep -v "^\[kmahip\] lt"
rm -rf $W
