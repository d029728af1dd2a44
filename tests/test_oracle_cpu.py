"""CPU suite: the oracle's restatements (oracle/oracle_py.py numpy stages, oracle_chain.c,
oracle_align.c) against the golden vectors generated from the compiled reference
(tests/golden/make_golden.py), and -- when oracle/_ref is present -- against the reference itself."""
import hashlib

import numpy as np
import pytest

from damapper_amd import dazzdb, las
import oracle_py
from util_golden import load, golden_cands
from util_pipeline import ref_blocks_of


@pytest.fixture(scope="module")
def G():
    return load()


def _kmerpos(c, r, p):
    arr = np.zeros(len(c), dtype=np.dtype([("code", "<u8"), ("rpos", "<i4"), ("read", "<i4")]))
    arr["code"], arr["rpos"], arr["read"] = c, p, r
    return arr


def test_kmer_index_matches_golden(G):
    z, genome, reads = G
    blk = dazzdb.make_block(reads)
    c, r, p = oracle_py.sort_kmers(blk, int(z["K"]))
    arr = _kmerpos(c, r, p)
    assert len(arr) == int(z["kmer_n"])
    assert hashlib.sha256(arr.tobytes()).digest() == z["kmer_sha"].tobytes()
    assert arr[:64].tobytes() == z["kmer_head"].tobytes()


def _passes(z, genome, reads):
    """Yield (pass index, comp, ref block) in damapper.c's order."""
    blocks = ref_blocks_of(genome, [int(x) for x in z["firsts"]])
    p = 0
    for blk in blocks:
        for comp in (0, 1):
            if comp:
                blk.complement_inplace()
            yield p, comp, blk
            p += 1
        blk.complement_inplace()


def test_seed_hits_and_candidates_match_golden(G):
    z, genome, reads = G
    K = int(z["K"])
    rblk = dazzdb.make_block(reads, path=b"./.READS")
    aidx = oracle_py.sort_kmers(rblk, K)
    lists = {}
    for p, comp, blk in _passes(z, genome, reads):
        bidx = oracle_py.sort_kmers(blk, K)
        limit, hits = oracle_py.seed_hits(aidx, bidx, rblk, blk, int(z["mem"]))
        want = z["seeds%d" % p]
        assert limit == 10000
        assert hits.shape == want.shape and np.array_equal(hits, want), "pass %d" % p
        oracle_py.filter_candidates(lists, hits, K, comp, blk.db.tfirst)
    want = golden_cands(z)
    got = {ar: [((c["score"], c["length"], c["bread"], c["comp"], c["afirst"], c["alast"], c["bfirst"], c["blast"]),
                 c["adisp"], c["bdisp"]) for c in L] for ar, L in lists.items() if L}
    assert got == want


def test_alignment_matches_golden(G):
    z, genome, reads = G
    from test_oracle_align import OracleAligner
    from util_align import comp_block, whole_ref_block
    rd = dazzdb.make_block(reads, path=b"./.READS")
    creads = comp_block(rd)
    ref = whole_ref_block(genome)
    oa = OracleAligner(0.85, 100, z["freq"], 1)
    ao, bo = z["align_aoff"], z["align_boff"]
    for i, (ar, br, cm, dg, ad) in enumerate(z["align_jobs"]):
        src = creads if cm else rd
        aoff = int(src.reads[ar + 1]["boff"]) + 1
        alen = int(src.reads[ar + 1]["rlen"])
        boff = int(ref.reads[br + 1]["boff"]) + 1
        blen = int(ref.reads[br + 1]["rlen"])
        a, at, b, bt = oa.align(src.bases, aoff, alen, ref.bases, boff, blen, int(cm), int(dg), int(ad))
        w = z["align_paths"][i]
        assert list(a) == list(w[:6]) and list(b) == list(w[6:]), "job %d" % i
        assert np.array_equal(at, z["align_atrace"][ao[i]:ao[i + 1]])
        assert np.array_equal(bt, z["align_btrace"][bo[i]:bo[i + 1]])


def test_golden_las_invariants(G):
    z, _, _ = G
    rows, to = z["las_rows"], z["las_toff"]
    assert len(rows) > 10
    for i, r in enumerate(rows):
        rec = las.LasRecord(*[int(x) for x in r], trace=bytes(z["las_trace"][to[i]:to[i + 1]]))
        assert las.check_trace_points(rec, 100)
        assert rec.flags & (las.START_FLAG | las.NEXT_FLAG)


def test_oracle_stages_match_reference_when_present(G):
    import ref_lib
    if not ref_lib.available():
        pytest.skip("oracle/_ref not built")
    z, genome, reads = G
    rng = np.random.default_rng(99)
    g2 = dazzdb.repeat_rich_genome(rng, [50000], family_len=500, tandem_unit=30, tandem_copies=10)
    blk = dazzdb.make_block(g2)
    for K, supp in ((20, 0), (14, 6)):
        ref_lib.lib().Set_Filter_Params(K, supp, 4)
        h, n, arr = ref_lib.sort_kmers(blk)
        c, r, p = oracle_py.sort_kmers(blk, K, supp)
        assert n == len(c)
        assert np.array_equal(arr["code"][:n], c) and np.array_equal(arr["read"][:n], r) and np.array_equal(arr["rpos"][:n], p)
        ref_lib.free(h)
