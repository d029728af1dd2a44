"""LAsort / LAcat / LAmerge equivalents (damapper_amd/tools/latools.cpp): the post-step the reference
driver runs after Reporter (damapper.c:893-910).  The DALIGNER originals are not in the reference tree
(parity unpinned); these tests pin what damapper's output contract fixes: chains stay intact, the
sorted stream is ordered by (A-read, A-position) of each chain's first record for -a, nothing is lost."""
import os
import random
import subprocess

import pytest

from damapper_amd import dazzdb, las
from util_pipeline import make_dataset


def _rand_chains(rng, nchains, areads):
    recs = []
    for _ in range(nchains):
        a = rng.randrange(areads)
        b = rng.randrange(5)
        comp = rng.randrange(2)
        n = rng.choice([1, 1, 1, 2, 3])
        pos = rng.randrange(0, 5000)
        for i in range(n):
            ln = rng.randrange(200, 900)
            tlen = 2 * ((pos + ln - 1) // 100 - pos // 100 + 1)
            tr = bytes(rng.randrange(0, 120) for _ in range(tlen))
            fl = comp | (las.NEXT_FLAG if i else las.START_FLAG | (las.BEST_FLAG if rng.random() < .7 else 0))
            recs.append(las.LasRecord(tlen, rng.randrange(50), pos, rng.randrange(10000), pos + ln,
                                      rng.randrange(10000), fl, a, b, tr))
            pos += ln + rng.randrange(0, 300)
    return recs


def _first_keys(recs, map_order=True):
    out = []
    for c in las.chains(recs):
        r = c[0]
        out.append((r.aread, r.abpos) if map_order else (r.aread, r.bread, r.flags & 1, r.abpos))
    return out


def _chain_multiset(recs):
    return sorted(tuple(r.key() for r in c) for c in las.chains(recs))


@pytest.mark.parametrize("map_order", [True, False])
def test_sort_cat_merge_keep_chains_and_order(tmp_path, map_order):
    rng = random.Random(7)
    d = str(tmp_path)
    parts = []
    for t in range(1, 5):
        recs = _rand_chains(rng, 60 if t != 3 else 0, 40)      # file 3 is empty
        las.write_las(os.path.join(d, "A.B.M%d.las" % t), 100, recs)
        parts.append(recs)
    flag = ["-a"] if map_order else []
    subprocess.check_call([las.tool_path("LAsort"), "-v"] + flag + [os.path.join(d, "A.B.M@.las")],
                          stdout=subprocess.DEVNULL)
    allrecs = []
    for t in range(1, 5):
        ts, s = las.read_las(os.path.join(d, "A.B.M%d.S.las" % t))
        assert ts == 100
        assert _chain_multiset(s) == _chain_multiset(parts[t - 1])
        ks = _first_keys(s, map_order)
        assert ks == sorted(ks)
        allrecs += s
    with open(os.path.join(d, "A.B.las"), "wb") as f:
        subprocess.check_call([las.tool_path("LAcat"), os.path.join(d, "A.B.M@.S")], stdout=f)
    ts, cat = las.read_las(os.path.join(d, "A.B.las"))
    assert ts == 100 and cat == allrecs
    subprocess.check_call([las.tool_path("LAmerge")] + flag + [os.path.join(d, "merged"), os.path.join(d, "A.B.M@.S.las")])
    ts, mg = las.read_las(os.path.join(d, "merged.las"))
    assert ts == 100 and len(mg) == len(allrecs)
    assert _chain_multiset(mg) == _chain_multiset(allrecs)
    ks = _first_keys(mg, map_order)
    assert ks == sorted(ks)


def test_sixteen_bit_traces_and_errors(tmp_path):
    d = str(tmp_path)
    r = las.LasRecord(4, 3, 10, 20, 300, 310, las.START_FLAG | las.BEST_FLAG, 0, 0, bytes(8))
    las.write_las(os.path.join(d, "x.las"), 126, [r, r])
    subprocess.check_call([las.tool_path("LAsort"), "-a", os.path.join(d, "x")])
    assert las.read_las(os.path.join(d, "x.S.las")) == (126, [r, r])
    open(os.path.join(d, "bad.las"), "wb").write(open(os.path.join(d, "x.las"), "rb").read()[:-3])
    p = subprocess.run([las.tool_path("LAsort"), os.path.join(d, "bad.las")], stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"truncated" in p.stderr
    p = subprocess.run([las.tool_path("LAsort"), os.path.join(d, "nothere@")], stderr=subprocess.PIPE)
    assert p.returncode == 1


def test_reference_driver_with_these_tools(tmp_path):
    """The reference driver end to end (CPU) with this package's LAsort/LAcat/LAmerge on its PATH: the
    final READS.REF.las / REF.READS.las hold exactly the chains of the per-thread files."""
    import run_ref
    if not run_ref.have_ref():
        pytest.skip("oracle/_ref not built")
    g, reads = make_dataset("repeat", 61, [120000, 100000], 30, dropouts=2)
    d = str(tmp_path)
    dazzdb.write_db(d, "REF", g, block_size=400000)
    dazzdb.write_db(d, "READS", reads)
    run_ref.run_reference(d, "REF", ["READS"], flags=["-C"], threads=4)                       # keeps M@/R@
    run_ref.run_reference(d, "REF", ["READS"], flags=["-C"], threads=4, keep="unused",
                          tools_dir=os.path.dirname(las.tool_path("LAsort")))
    for final, pat in (("READS.REF.las", "READS.REF.M@.las"), ("REF.READS.las", "REF.READS.R@.las")):
        ts, want = las.read_las_set(os.path.join(d, "las_ref"), pat)
        ts2, got = las.read_las(os.path.join(d, final))
        assert ts == ts2 and len(want) == len(got) and len(got) > 0
        assert _chain_multiset(want) == _chain_multiset(got)
        ks = _first_keys(got)
        assert ks == sorted(ks)
