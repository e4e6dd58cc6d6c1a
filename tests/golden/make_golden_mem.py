#!/usr/bin/env python3
"""Expected files of the committed single-end fixture (tests/golden/se) under `-mem_mode` -- `kma -i reads.fq.gz -t_db db -o out -1t1
-mem_mode -t 1`: runKMA_MEM, ConClave on the template finder's own scores -- written next to the plain ones as mem.res, mem.fsa.gz and
mem.frag.gz.

    python3 tests/golden/make_golden_mem.py        (needs oracle/_ref/kma: `make -C oracle ref`)
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
SE = os.path.join(HERE, "se")


def main():
    if not os.path.exists(KMA):
        sys.exit("oracle/_ref/kma missing: run `make -C oracle ref` first")
    with tempfile.TemporaryDirectory() as tmp:
        db = os.path.join(tmp, "db")
        with lzma.open(os.path.join(SE, "db.comp.b.xz"), "rb") as f, open(db + ".comp.b", "wb") as g:
            shutil.copyfileobj(f, g)
        for ext in (".length.b", ".seq.b", ".name"):
            shutil.copy(os.path.join(SE, "db" + ext), db + ext)
        out = os.path.join(tmp, "out")
        subprocess.run([KMA, "-i", os.path.join(SE, "reads.fq.gz"), "-o", out, "-t_db", db, "-1t1", "-mem_mode", "-t", "1"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        shutil.copy(out + ".res", os.path.join(SE, "mem.res"))
        for name, data in (("mem.fsa.gz", open(out + ".fsa", "rb").read()), ("mem.frag.gz", gzip.open(out + ".frag.gz").read())):
            with gzip.GzipFile(os.path.join(SE, name), "wb", mtime=0) as g:
                g.write(data)
        print(open(out + ".res").read().count("\n") - 1, "templates,", gzip.open(out + ".frag.gz").read().count(b"\n"), "fragment rows")


if __name__ == "__main__":
    main()
