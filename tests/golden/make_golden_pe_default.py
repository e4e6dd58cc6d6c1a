#!/usr/bin/env python3
"""Expected files of the committed paired fixture (tests/golden/pe: r1.fq.gz, r2.fq.gz and the index) in the reference's DEFAULT mode --
`kma -ipe r1 r2 -t_db db -o out -t 1`, no -1t1: couples by union pairing, the records that lost their mate through the chain finder --
written next to the `-1t1 -apm p` ones: out_default.res, out_default.fsa.gz, out_default.frag.gz, s2_default.bin.gz (the `-s2` tap),
s2_default_p.bin.gz (the `-s2` tap with `-apm p`), s2_force.bin.gz (the `-s2` tap of `-apm f -1t1`: stage 2 of forced pairing).

    python3 tests/golden/make_golden_pe_default.py        (needs oracle/_ref/kma: `make -C oracle ref`)
"""
import gzip
import lzma
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
KMA = os.path.join(ROOT, "oracle", "_ref", "kma")
PE = os.path.join(HERE, "pe")


def main():
    if not os.path.exists(KMA):
        sys.exit("oracle/_ref/kma missing: run `make -C oracle ref` first")
    with tempfile.TemporaryDirectory() as tmp:
        db = os.path.join(tmp, "db")
        with lzma.open(os.path.join(PE, "db.comp.b.xz"), "rb") as f, open(db + ".comp.b", "wb") as g:
            shutil.copyfileobj(f, g)
        for ext in (".length.b", ".seq.b", ".name"):
            shutil.copy(os.path.join(PE, "db" + ext), db + ext)
        base = [KMA, "-ipe", os.path.join(PE, "r1.fq.gz"), os.path.join(PE, "r2.fq.gz"), "-o", os.path.join(tmp, "out"), "-t_db", db, "-t", "1"]
        subprocess.run(base, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        s2 = subprocess.run(base + ["-s2"], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        # (the same stream with the pairing penalty, -apm p: what oracle/scan.c's paired half restates; for tests/test_oracle_golden.py)
        s2p = subprocess.run(base + ["-apm", "p", "-s2"], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        with gzip.GzipFile(os.path.join(PE, "s2_default_p.bin.gz"), "wb", mtime=0) as g:
            g.write(s2p)
        # (stage 2 of forced pairing, -apm f with -1t1: save_kmers_forcePair -- what kmahip_scan_pe restates with apm = 2)
        s2f = subprocess.run(base + ["-apm", "f", "-1t1", "-s2"], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        with gzip.GzipFile(os.path.join(PE, "s2_force.bin.gz"), "wb", mtime=0) as g:
            g.write(s2f)
        shutil.copy(os.path.join(tmp, "out.res"), os.path.join(PE, "out_default.res"))
        for name, data in (("out_default.fsa.gz", open(os.path.join(tmp, "out.fsa"), "rb").read()),
                           ("out_default.frag.gz", gzip.open(os.path.join(tmp, "out.frag.gz")).read()), ("s2_default.bin.gz", s2)):
            with gzip.GzipFile(os.path.join(PE, name), "wb", mtime=0) as g:
                g.write(data)
        print(open(os.path.join(tmp, "out.res")).read().count("\n") - 1, "templates,", gzip.open(os.path.join(tmp, "out.frag.gz")).read().count(b"\n"), "fragment rows,",
              len(s2), "bytes of S2 records")


if __name__ == "__main__":
    main()
