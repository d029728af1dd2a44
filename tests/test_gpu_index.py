"""GPU parity: radix sort kernel and Sort_Kmers (k-mer emit + sort) against the oracle.
Bit-exact: the index must equal the reference's KmerPos list record for record."""
import numpy as np
import pytest

from damapper_amd import capi, dazzdb

pytestmark = pytest.mark.gpu


def _ref():
    import ref_lib
    if not ref_lib.available():
        pytest.skip("oracle/_ref not built")
    return ref_lib


@pytest.mark.parametrize("n,nbits", [(1, 8), (63, 8), (64, 16), (4097, 24), (100000, 40), (1 << 20, 40), (300007, 13)])
def test_sort_pairs_stable(n, nbits):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 1 << 40, size=n, dtype=np.uint64)
    if n > 1000:
        keys[: n // 3] &= np.uint64(0xFF)          # heavy duplicates: exercises stability
    vals = np.arange(n, dtype=np.uint32)
    mask = np.uint64((1 << nbits) - 1)
    order = np.argsort(keys & mask, kind="stable")
    k, v = keys.copy(), vals.copy()
    capi.sort_pairs(k, v, nbits)
    assert np.array_equal(k, keys[order])
    assert np.array_equal(v, vals[order])


def _kmer_index_numpy(block, K):
    """Oracle restatement in numpy of map.c:545-576 + the stable sort: (code, read, rpos) order."""
    codes, reads, rpos = [], [], []
    for i in range(block.nreads):
        s = block.seq(i).astype(np.uint64)
        n = len(s)
        if n < K:
            continue
        c = np.zeros(n - K + 1, dtype=np.uint64)
        for j in range(K):
            c = (c << np.uint64(2)) | s[j:n - K + 1 + j]
        codes.append(c)
        reads.append(np.full(n - K + 1, i, dtype=np.int32))
        rpos.append(np.arange(K - 1, n, dtype=np.int32))
    c = np.concatenate(codes)
    r = np.concatenate(reads)
    p = np.concatenate(rpos)
    o = np.argsort(c, kind="stable")
    return c[o], r[o], p[o]


@pytest.mark.parametrize("K,lens", [(20, [5000, 300, 20, 19, 7000]), (14, [1000] * 37), (32, [4000, 4100]), (5, [64, 5, 4, 100])])
def test_sort_kmers_matches_numpy(K, lens):
    rng = np.random.default_rng(K)
    seqs = dazzdb.random_genome(rng, lens)
    seqs[0][100:400] = 0          # poly-A: long runs of one code
    blk = dazzdb.make_block(seqs)
    capi.lib().Set_Filter_Params(K, 0, 4)
    h, n = capi.sort_kmers(blk)
    c, r, p = _kmer_index_numpy(blk, K)
    assert n == len(c)
    got = capi.index_download(h, n)
    assert np.array_equal(got["code"], c)
    assert np.array_equal(got["read"], r)
    assert np.array_equal(got["rpos"], p)
    capi.lib().dmh_free_index(h)


@pytest.mark.parametrize("K,supp", [(20, 0), (20, 5), (12, 3)])
def test_sort_kmers_matches_reference(K, supp):
    ref = _ref()
    rng = np.random.default_rng(7)
    g = dazzdb.repeat_rich_genome(rng, [120000, 80000])
    blk = dazzdb.make_block(g)
    ref.lib().Set_Filter_Params(K, supp, 4)
    rh, rn, rarr = ref.sort_kmers(blk)
    capi.lib().Set_Filter_Params(K, supp, 4)
    h, n = capi.sort_kmers(blk)
    assert n == rn
    got = capi.index_download(h, n)
    assert np.array_equal(got["code"], rarr["code"][:n])
    assert np.array_equal(got["read"], rarr["read"][:n])
    assert np.array_equal(got["rpos"], rarr["rpos"][:n])
    capi.lib().dmh_free_index(h)
    ref.free(rh)


def _random_mask(rng, lens, K, density=6):
    """Sorted disjoint [begin,end) intervals per read, incl. edge shapes: touching read ends,
    zero-length, shorter than k, gaps shorter than k."""
    out = []
    for n in lens:
        cuts = sorted(set(int(x) for x in rng.integers(0, n + 1, size=2 * int(rng.integers(0, density)))))
        if len(cuts) % 2:
            cuts = cuts[:-1]
        iv = [(cuts[j], cuts[j + 1]) for j in range(0, len(cuts), 2)]
        if n > 200 and rng.random() < .5:
            iv = sorted(set(iv + [(0, 7)])) if not iv or iv[0][0] > 7 else iv
        if n > 200 and rng.random() < .5 and (not iv or iv[-1][1] < n - K - 5):
            iv.append((n - 3, n))
        out.append(iv)
    return out


@pytest.mark.parametrize("K,supp,comp", [(20, 0, False), (20, 4, True), (12, 0, True)])
def test_masked_sort_kmers_matches_reference(K, supp, comp):
    """-m: k-mers touching a masked interval are left out (map.c:481-543), also after the in-place
    complement, which mirrors the intervals (damapper.c:505-518)."""
    ref = _ref()
    rng = np.random.default_rng(17 + K)
    g = dazzdb.repeat_rich_genome(rng, [60000, 30000, 500, 25, 40000])
    blk = dazzdb.make_block(g)
    blk.set_mask(_random_mask(rng, [len(s) for s in g], K))
    if comp:
        blk.complement_inplace()
    ref.lib().Set_Filter_Params(K, supp, 4)
    rh, rn, rarr = ref.sort_kmers(blk)
    capi.lib().Set_Filter_Params(K, supp, 4)
    h, n = capi.sort_kmers(blk)
    assert n == rn and 0 < n < sum(len(s) for s in g)
    got = capi.index_download(h, n)
    assert np.array_equal(got["code"], rarr["code"][:n])
    assert np.array_equal(got["read"], rarr["read"][:n])
    assert np.array_equal(got["rpos"], rarr["rpos"][:n])
    # every surviving k-mer lies inside one unmasked gap
    iv = blk.mask_intervals()
    for r, p in zip(got["read"][::97], got["rpos"][::97]):
        assert all(not (b <= p and e > p - (K - 1)) for b, e in iv[int(r)])
    capi.lib().dmh_free_index(h)
    ref.free(rh)


def test_fully_masked_block_gives_empty_index():
    rng = np.random.default_rng(3)
    g = dazzdb.random_genome(rng, [3000, 2000])
    blk = dazzdb.make_block(g)
    blk.set_mask([[(0, 3000)], [(0, 2000)]])
    capi.lib().Set_Filter_Params(20, 0, 4)
    h, n = capi.sort_kmers(blk)
    assert n == 0 and not h


@pytest.mark.parametrize("n", [1, 63, 4096, 4097, 25_001, 1_000_003, (1 << 20), (1 << 20) + 5, 9_000_001])
def test_exclusive_scan_matches_numpy(n):
    """The device-wide scan (single look-back kernel up to 1 M values, three kernels above) against numpy; run a few
    times over so that the look-back's never-cleared status words (epochs) are exercised."""
    import ctypes as C
    rng = np.random.default_rng(n)
    a = rng.integers(0, 200, n, dtype=np.uint32)
    want = np.concatenate(([0], np.cumsum(a[:-1], dtype=np.uint64))).astype(np.uint32)
    buf = a.copy()
    tot = C.c_uint32(0)
    assert capi.lib().dmh_exclusive_scan(buf.ctypes.data, n, 5, C.byref(tot), None) == 0
    assert tot.value == int(a.sum(dtype=np.uint64)) & 0xffffffff
    assert np.array_equal(buf, want)
