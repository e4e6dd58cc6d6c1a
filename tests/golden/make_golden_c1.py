#!/usr/bin/env python3
"""Config C1 of BASELINE.json at full size as an end-to-end fixture: 100 000 x 150 bp single-end reads against the
500-gene database (100 families x 5 variants), `kma -1t1 -t 1` of the compiled reference (oracle/_ref/kma). Inputs are
regenerated from seeds by the test (kma_amd.synth + kma_amd.formats.write_index), so only the reference's outputs are
stored: out.res and the consensus FASTA (out.fsa.gz). Only data is stored."""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from kma_amd import formats, synth  # noqa: E402

KMA = os.path.join(ROOT, "oracle", "_ref", "kma")
N_READS, FAMILIES = 100_000, 100


def inputs(tmp):
    names, seqs = synth.make_gene_db(FAMILIES, 5, 600, 1500, 0.04, seed=12345)
    prefix = os.path.join(tmp, "db")
    formats.write_index(prefix, names, seqs)
    reads, _, _, _ = synth.make_reads(seqs, N_READS, seed=1)
    return prefix, names, reads


if __name__ == "__main__":
    if not os.path.exists(KMA):
        sys.exit("oracle/_ref/kma missing: run `make -C oracle ref` first")
    out = os.path.join(HERE, "c1")
    os.makedirs(out, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        prefix, names, reads = inputs(tmp)
        fq = os.path.join(tmp, "reads.fq")
        synth.write_fastq(fq, reads)
        subprocess.run([KMA, "-i", fq, "-o", os.path.join(tmp, "out"), "-t_db", prefix, "-1t1", "-t", "1"], check=True,
                       stderr=subprocess.DEVNULL)
        shutil.copy(os.path.join(tmp, "out.res"), os.path.join(out, "out.res"))
        with open(os.path.join(tmp, "out.fsa"), "rb") as f, gzip.GzipFile(os.path.join(out, "out.fsa.gz"), "wb", mtime=0) as g:
            g.write(f.read())
    print("c1 fixture written:", sum(1 for _ in open(os.path.join(out, "out.res"))) - 1, "rows")
