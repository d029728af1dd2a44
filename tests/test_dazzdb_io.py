"""DAZZ_DB side formats written/read by damapper_amd.dazzdb: FASTA import, mask tracks (-m)."""
import os

import numpy as np

from damapper_amd import dazzdb


def test_fasta_import_roundtrip(tmp_path):
    d = str(tmp_path)
    rng = np.random.default_rng(5)
    a = "".join("ACGT"[x] for x in rng.integers(0, 4, 700))
    b = "".join("acgt"[x] for x in rng.integers(0, 4, 300))
    with open(os.path.join(d, "x.fa"), "w") as f:
        f.write(">chr1 some text\n")
        for i in range(0, len(a), 60):
            f.write(a[i:i + 60] + "\n")
        f.write(">scaf\n" + b[:100] + "NNNNNNNNNN" + b[100:] + "\n>tiny\nACG\n")
    seqs, names = dazzdb.read_fasta(os.path.join(d, "x.fa"), min_len=20)
    assert names == ["chr1", "scaf/1", "scaf/2"]
    assert [len(s) for s in seqs] == [700, 100, 200]
    assert "".join("ACGT"[x] for x in seqs[0]) == a
    assert dazzdb._main([os.path.join(d, "X.db"), os.path.join(d, "x.fa"), "-s", "0.0005", "-x", "20"]) == 0
    nb, _ = dazzdb.read_stub(os.path.join(d, "X.db"))
    assert nb == 2                                        # 700 | 100+200
    blk = dazzdb.load_db(os.path.join(d, "X.db"))
    assert blk.nreads == 3 and np.array_equal(blk.seq(2), seqs[2])
    try:
        dazzdb.read_fasta(os.path.join(d, "x.fa"), split_n=False)
        assert False
    except ValueError:
        pass


def test_mask_track_files_and_union(tmp_path):
    d = str(tmp_path)
    rng = np.random.default_rng(6)
    g = dazzdb.random_genome(rng, [5000, 4000, 3000, 2000])
    dazzdb.write_db(d, "G", g, block_size=9000)           # blocks: reads {0,1} and {2,3}
    m1 = [[(10, 50), (400, 900)], [], [(0, 100)], [(1990, 2000)]]
    m2 = [[(40, 60), (2000, 2100)], [(5, 6)], [(100, 150), (300, 301)], []]
    dazzdb.write_mask_track(d, "G", "a", m1)
    dazzdb.write_mask_track(d, "G", "b", m2)
    used = set()
    blk = dazzdb.load_db(os.path.join(d, "G.db"), part=2, masks=["a", "zz"], used=used)
    assert used == {"a"} and blk.mask_intervals() == m1[2:]
    assert blk.track.size == 8 and blk.track.name == b"mask"
    blk = dazzdb.load_db(os.path.join(d, "G.db"), masks=["a", "b"])
    assert blk.track.size == 4 and blk.track.name == b"merge"
    assert blk.mask_intervals() == [[(10, 60), (400, 900), (2000, 2100)], [(5, 6)], [(0, 150), (300, 301)], [(1990, 2000)]]
    # brute-force union per base
    for r, iv in enumerate(blk.mask_intervals()):
        cov = np.zeros(len(g[r]) + 1, dtype=bool)
        for t in (m1, m2):
            for b, e in t[r]:
                cov[b:e] = True
        got = np.zeros_like(cov)
        for b, e in iv:
            got[b:e] = True
        assert np.array_equal(cov, got)
    # the in-place complement mirrors the intervals and is an involution
    before = blk.mask_intervals()
    blk.complement_inplace()
    assert blk.mask_intervals()[0] == [(5000 - 2100, 5000 - 2000), (5000 - 900, 5000 - 400), (5000 - 60, 5000 - 10)]
    blk.complement_inplace()
    assert blk.mask_intervals() == before
    assert dazzdb.load_db(os.path.join(d, "G.db"), masks=["zz"]).track is None


def test_trimmed_view(tmp_path):
    """DBsplit -x<cutoff> without -a: the loader applies Trim_DB's rule and the stub's trimmed firsts."""
    d = str(tmp_path)
    rng = np.random.default_rng(9)
    seqs = dazzdb.random_genome(rng, [3000, 100, 2500, 50, 4000, 2000, 2600])
    best = [True, True, True, True, False, True, True]
    firsts = dazzdb.write_db(d, "T", seqs, block_size=6000, cutoff=500, all_reads=False, best=best)
    assert firsts == [0, 5, 7]                 # survivors 3000+2500 | 2000+2600 (4000 is not best)
    nb, st = dazzdb.read_stub(os.path.join(d, "T.db"))
    assert nb == 2 and st == [(0, 0), (5, 2), (7, 4)]
    assert dazzdb.read_stub_params(os.path.join(d, "T.db")) == (6000, 500, False)
    b2 = dazzdb.load_db(os.path.join(d, "T.db"), part=2)
    assert b2.nreads == 2 and b2.db.tfirst == 2 and [len(b2.seq(i)) for i in range(2)] == [2000, 2600]
    whole = dazzdb.load_db(os.path.join(d, "T.db"))
    assert [len(whole.seq(i)) for i in range(whole.nreads)] == [3000, 2500, 2000, 2600]
    # a mask over the UNTRIMMED reads follows the survivors
    dazzdb.write_mask_track(d, "T", "m", [[(1, 2)], [(3, 4)], [(5, 6)], [], [(7, 8)], [(9, 10)], []])
    b1 = dazzdb.load_db(os.path.join(d, "T.db"), part=1, masks=["m"])
    assert b1.mask_intervals() == [[(1, 2)], [(5, 6)]]


def test_packed_block_carries_the_bps_slice(tmp_path):
    """load_db(packed=True) -- the host half of the device-side Load_All_Reads: no bases, the same read records
    as the decoded block, and a slice + offsets from which every read of the (trimmed) view decodes to the
    same bases (the device half is tests/test_gpu_las.py::test_device_side_block_decode_equals_load_all_reads)."""
    rng = np.random.default_rng(11)
    lens = [1, 2, 3, 4, 5, 9, 4096, 4097] + [int(x) for x in rng.integers(1, 3000, 120)]
    seqs = [rng.integers(0, 4, n).astype(np.uint8) for n in lens]
    best = [i % 4 != 1 for i in range(len(seqs))]
    d = str(tmp_path)
    dazzdb.write_db(d, "R", seqs, block_size=60000, cutoff=3, all_reads=False, best=best)
    nb, _ = dazzdb.read_stub(os.path.join(d, "R.db"))
    assert nb >= 2
    seen = 0
    for part in range(0, nb + 1):
        host = dazzdb.load_db(os.path.join(d, "R.db"), part=part)
        pk = dazzdb.load_db(os.path.join(d, "R.db"), part=part, packed=True)
        assert pk.bases is None and not pk.db.bases and pk.packed is not None
        assert pk.nreads == host.nreads and (pk.reads == host.reads).all()
        assert pk.db.totlen == host.db.totlen and pk.db.maxlen == host.db.maxlen and pk.db.tfirst == host.db.tfirst
        assert (np.diff(pk.coff) > 0).all() and pk.coff[0] == 0
        for i in range(host.nreads):
            n = int(host.reads[i + 1]["rlen"])
            o = int(pk.coff[i])
            assert np.array_equal(dazzdb._unpack_2bit(pk.packed[o:o + (n + 3) // 4], n), host.seq(i))
        if part > 0:
            seen += host.nreads
    assert seen == sum(1 for s, b in zip(seqs, best) if len(s) >= 3 and b)
