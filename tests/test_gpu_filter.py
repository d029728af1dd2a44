"""GPU parity for Match_Filter: seed hits (merge + seed sort) and the per-read candidate chains
(chain scoring + domination + Jump encoding) must equal the compiled reference's, bit for bit,
over every REF block and both orientations."""
import numpy as np
import pytest

from util_pipeline import Api, make_dataset, run_filter

pytestmark = pytest.mark.gpu

CASES = [
    # kind, seed, contigs, nreads, dropouts, K, supp, mem_limit, firsts
    ("random", 1, [300000, 200000, 250000], 60, 0, 20, 0, 8 << 30, [0, 1, 2, 3]),
    ("random", 2, [150000, 150000, 100000, 90000], 40, 0, 14, 0, 8 << 30, [0, 2, 4]),
    ("repeat", 3, [250000, 250000], 40, 2, 20, 0, 8 << 30, [0, 2]),
    ("repeat", 4, [250000, 200000], 30, 1, 20, 8, 8 << 30, [0, 1, 2]),
    ("repeat", 5, [300000], 30, 0, 16, 0, 24 << 20, [0, 1]),
]


def _ref_available():
    import ref_lib
    return ref_lib.available()


@pytest.mark.parametrize("cluster", [0, 1], ids=["pairs", "clusters"])
@pytest.mark.parametrize("case", CASES, ids=[str(i) for i in range(len(CASES))])
def test_seeds_and_candidates_match_reference(case, cluster):
    """`cluster` forces the chain groups: whole (aread, bread) pairs, or their B-clusters (dmh_merge.h)."""
    if not _ref_available():
        pytest.skip("oracle/_ref not built")
    from damapper_amd import capi
    kind, seed, contigs, nreads, drop, K, supp, mem, firsts = case
    g, reads = make_dataset(kind, seed, contigs, nreads, dropouts=drop)
    ref = run_filter(Api("ref"), reads, g, firsts, K=K, supp=supp, mem_limit=mem)
    capi.lib().dmh_test_option(b"cluster", cluster)
    try:
        hip = run_filter(Api("hip"), reads, g, firsts, K=K, supp=supp, mem_limit=mem)
    finally:
        capi.lib().dmh_test_option(b"cluster", -1)
    assert len(ref["seeds"]) == len(hip["seeds"])
    for p, (a, b) in enumerate(zip(ref["seeds"], hip["seeds"])):
        assert len(a) == len(b), "pass %d: %d vs %d seed hits" % (p, len(a), len(b))
        for f in ("aread", "bread", "apos", "diag"):
            assert np.array_equal(a[f], b[f]), "pass %d field %s" % (p, f)
    nc = 0
    for ar, ((ca, ja), (cb, jb)) in enumerate(zip(ref["cands"], hip["cands"])):
        assert ca.shape == cb.shape, "read %d: %d vs %d candidates\n%s\n%s" % (ar, len(ca), len(cb), ca, cb)
        assert np.array_equal(ca, cb), "read %d candidates differ\n%s\n%s" % (ar, ca, cb)
        for k, ((a1, b1), (a2, b2)) in enumerate(zip(ja, jb)):
            assert np.array_equal(a1, a2) and np.array_equal(b1, b2), "read %d cand %d jumps" % (ar, k)
        nc += len(ca)
    assert nc > 0


COVER_CASES = [
    # kind, seed, contigs, nreads, dropouts, spacing, firsts
    ("repeat", 6, [250000, 250000], 40, 1, 100, [0, 1, 2]),
    ("random", 7, [200000, 150000, 100000], 40, 2, 126, [0, 2, 3]),
    ("repeat", 8, [300000], 25, 0, 100, [0, 1]),
]


@pytest.mark.parametrize("case", COVER_CASES, ids=[str(i) for i in range(len(COVER_CASES))])
def test_profile_deltas_match_reference(case):
    """R11 (map.c:1654-1666): the -p coverage deltas of every read after all Match_Filter passes -- every EXAMINED
    chain counts, dominated or not -- equal the reference's stage vector (ref_harness.c ref_read_cover), before
    Reporter turns them into special_log bytes."""
    if not _ref_available():
        pytest.skip("oracle/_ref not built")
    kind, seed, contigs, nreads, drop, spacing, firsts = case
    g, reads = make_dataset(kind, seed, contigs, nreads, dropouts=drop)
    from damapper_amd import capi
    ref = run_filter(Api("ref"), reads, g, firsts, profile=1, spacing=spacing, want_seeds=False)
    capi.lib().dmh_test_option(b"cluster", seed & 1)        # (both kinds of chain groups over the cases)
    try:
        hip = run_filter(Api("hip"), reads, g, firsts, profile=1, spacing=spacing, want_seeds=False)
    finally:
        capi.lib().dmh_test_option(b"cluster", -1)
    nz = 0
    for ar, (a, b) in enumerate(zip(ref["cover"], hip["cover"])):
        assert len(a) == len(b) == (len(reads[ar]) - 1) // spacing + 2, "read %d: %d vs %d slots" % (ar, len(a), len(b))
        assert np.array_equal(a, b), "read %d cover deltas differ\n%s\n%s" % (ar, a, b)
        nz += int(np.count_nonzero(a))
    assert nz > 0
    for (ca, _), (cb, _) in zip(ref["cands"], hip["cands"]):
        assert np.array_equal(ca, cb)


@pytest.mark.parametrize("limit", [16000, 40000], ids=["ranges_of_a_few_reads", "two_ranges"])
def test_pass_beyond_the_hit_range_is_made_in_ranges_of_reads(limit):
    """Hits are numbered in 32 bits behind the merge, the reference counts them in 64 (map.c:876, 2899): a pass of 2^32
    hits and more is made in RANGES OF READS, each a batch of its own -- hits of different reads never meet before the
    domination rule, which is per read (dmh_merge.hip, merge_pass).  The "hit_limit" test option moves the range limit
    to a few thousand hits (the read opposite the tandem array of this data set meets 15,341 in one pass, and a read is never split): seeds per pass, candidate lists and Jump chains against the reference, as for a pass made
    in one go (round 3 refused such a pass)."""
    if not _ref_available():
        pytest.skip("oracle/_ref not built")
    from damapper_amd import capi
    g, reads = make_dataset("repeat", 3, [250000], 30)
    ref = run_filter(Api("ref"), reads, g, [0, 1], K=20, supp=0, mem_limit=8 << 30)
    L = capi.lib()
    L.dmh_test_option(b"cluster", 1)
    L.dmh_test_option(b"hit_limit", limit)
    try:
        hip = run_filter(Api("hip"), reads, g, [0, 1], K=20, supp=0, mem_limit=8 << 30)
    finally:
        L.dmh_test_option(b"cluster", -1)
        L.dmh_test_option(b"hit_limit", 0)
    assert len(ref["seeds"]) == len(hip["seeds"]) == 2
    assert max(len(s) for s in ref["seeds"]) > limit                # (made of more than one range)
    for p, (a, b) in enumerate(zip(ref["seeds"], hip["seeds"])):
        assert len(a) == len(b), "pass %d: %d vs %d seed hits" % (p, len(a), len(b))
        for f in ("aread", "bread", "apos", "diag"):
            assert np.array_equal(a[f], b[f]), "pass %d field %s" % (p, f)
    nc = 0
    for ar, ((ca, ja), (cb, jb)) in enumerate(zip(ref["cands"], hip["cands"])):
        assert np.array_equal(ca, cb), "read %d candidates differ\n%s\n%s" % (ar, ca, cb)
        for k, ((a1, b1), (a2, b2)) in enumerate(zip(ja, jb)):
            assert np.array_equal(a1, a2) and np.array_equal(b1, b2), "read %d cand %d jumps" % (ar, k)
        nc += len(ca)
    assert nc > 0


@pytest.mark.gpu
def test_library_says_so_when_the_runtime_has_too_few_hardware_queues():
    """The library keeps nine streams busy and asks the HIP runtime for twelve hardware queues (dmh_core.hip init_device);
    a process whose runtime was started with the default of four gets one pile stream instead of four and a line on stderr
    (probe_hw_queues: eight kernels of 1 ms on eight streams, timed)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "profiles", "scripts", "x_qprobe.py")
    for queues, want in (("4", 4), ("12", 8)):
        env = dict(os.environ, GPU_MAX_HW_QUEUES=queues, DMH_STATS="1")
        r = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert "sorted True" in r.stdout
        assert ("about %d run side by side" % want) in r.stderr, r.stderr
        assert ("only about" in r.stderr) == (want < 8), r.stderr
