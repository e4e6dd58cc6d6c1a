"""Pin the CPU oracle against the reference's own stream taps (not gpu)."""
import numpy as np

import golden_util
import oracle
from kma_amd import formats


def test_s1_roundtrip_pack(golden_se):
    # our packer reproduces the reference's 2-bit words + N lists (compdna.c:99-127)
    b = golden_se["batch"]
    for i, r in enumerate(golden_se["s1"]):
        w = (r["seqlen"] + 31) // 32
        assert np.array_equal(b.seq[b.seq_off[i]:b.seq_off[i] + w], r["seq"])
        assert np.array_equal(b.N[b.N_off[i]:b.N_off[i + 1]], r["N"])


def test_oracle_scan_matches_s2_tap(golden_se):
    db = oracle.OracleDB(golden_se["prefix"])
    rc_flag, flag, T_off, T = db.scan_se(golden_se["batch"])
    n = golden_util.check_scan_against_s2(golden_se["s1"], golden_se["s2"], rc_flag, flag, T_off, T)
    assert n > 800


def test_oracle_scan_matches_s2_tap_exhaustive(golden_se):
    db = oracle.OracleDB(golden_se["prefix"])
    rc_flag, flag, T_off, T = db.scan_se(golden_se["batch"], exhaustive=1)
    golden_util.check_scan_against_s2(golden_se["s1"], golden_se["s2_ex"], rc_flag, flag, T_off, T)


def test_index_writer_semantics(golden_se, tmp_path):
    # our .comp.b writer yields the same k-mer -> template-set map and the same
    # equal-set partition as the reference's `kma index`
    import gzip
    names, seqs, cur = [], [], None
    lut = np.full(256, 255, np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
    with gzip.open(golden_se["dir"] + "/db.fsa.gz", "rb") as f:
        for line in f:
            line = line.strip()
            if line.startswith(b">"):
                names.append(line[1:].decode())
            else:
                seqs.append(lut[np.frombuffer(line, np.uint8)])
    formats.write_index(str(tmp_path / "mine"), names, seqs, k=16)
    ref = formats.read_comp_b(golden_se["prefix"] + ".comp.b")
    mine = formats.read_comp_b(str(tmp_path / "mine.comp.b"))
    assert (ref.DB_size, ref.n, ref.kmersize, ref.v_index) == (mine.DB_size, mine.n, mine.kmersize, mine.v_index)
    m_ref, m_mine = formats.comp_db_mapping(ref), formats.comp_db_mapping(mine)
    assert m_ref == m_mine
    # partition equality: value_index equal <=> set equal, in both
    assert len(set(ref.value_index.tolist())) == len(set(mine.value_index.tolist())) == len(set(m_ref.values()))
    # and the oracle gives identical stage-2 output on our index
    db = oracle.OracleDB(str(tmp_path / "mine"))
    rc_flag, flag, T_off, T = db.scan_se(golden_se["batch"])
    golden_util.check_scan_against_s2(golden_se["s1"], golden_se["s2"], rc_flag, flag, T_off, T)
    for ext in (".length.b", ".seq.b"):
        assert open(golden_se["prefix"] + ext, "rb").read() == open(str(tmp_path / "mine") + ext, "rb").read()
