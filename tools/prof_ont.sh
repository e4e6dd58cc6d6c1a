set -e
export TMPDIR=/tmp
cd /root/repo
python3 - <<'PY'
import os, sys, subprocess
sys.path.insert(0, os.getcwd())
import numpy as np
from kma_amd import synth
rng = np.random.default_rng(21)
names, seqs = synth.make_gene_db(1000, 5, 600, 1500, 0.04, seed=12345)
os.makedirs("/tmp/ont", exist_ok=True)
synth.write_fasta("/tmp/ont/db.fsa", names, seqs)
subprocess.run(["oracle/_ref/kma", "index", "-i", "/tmp/ont/db.fsa", "-o", "/tmp/ont/db"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
reads = []
for i in range(5000):
    parts = [rng.integers(0, 4, int(rng.integers(1000, 6000)), dtype=np.uint8)]
    for _ in range(int(rng.integers(1, 3))):
        g = seqs[int(rng.integers(0, len(seqs)))]
        parts.append(synth.revcomp_codes(g) if rng.random() < 0.5 else g)
        parts.append(rng.integers(0, 4, int(rng.integers(1000, 5000)), dtype=np.uint8))
    r = np.concatenate(parts)
    reads.append(synth.make_long_reads(r, 1, read_len=len(r), seed=100 + i)[0])
synth.write_fastq("/tmp/ont/ont.fq", reads, prefix="r", qual=b"5")
PY
make -C examples > /dev/null
export KMAHIP_MAP_TEARDOWN=1
ROOT=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ont/stats -o p -- $ROOT/examples/kmahip_map -i /tmp/ont/ont.fq -t_db /tmp/ont/db -o /tmp/ont/out -chain -bcNano > /tmp/ont/run.log 2>&1
head -14 $(find /tmp/ont/stats -name "*kernel_stats.csv") | cut -c1-160
