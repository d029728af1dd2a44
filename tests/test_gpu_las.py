"""GPU end-to-end parity: the per-thread .las record streams (and the -p track) written by the HIP
pipeline must be identical to the compiled reference's on the same DBs (bytes 36..39 of every
record are uninitialised struct padding in the reference and are not compared)."""
import os
import shutil

import numpy as np
import pytest

from damapper_amd import dazzdb, driver, las
from util_pipeline import make_dataset

pytestmark = pytest.mark.gpu


def _run_ref():
    import run_ref
    if not run_ref.have_ref():
        pytest.skip("oracle/_ref not built")
    return run_ref


CASES = [
    # name, kind, seed, contigs, nreads, dropouts, block_size, flags(ref), kwargs(hip)
    ("plain", "random", 41, [300000, 200000, 250000], 120, 0, 400000, [], {}),
    ("dropout_n95_p", "repeat", 42, [300000, 250000], 80, 2, 1000000, ["-n.95", "-p"], dict(best_tie=0.95, profile=1)),
    ("k14_e75", "random", 43, [200000, 150000], 50, 1, 1000000, ["-k14", "-e.75"], dict(kmer=14, ave_corr=0.75)),
    ("s126_t20", "repeat", 44, [250000], 50, 1, 1000000, ["-s126", "-t20"], dict(spacing=126, suppress=20)),
    ("T1", "random", 45, [150000, 100000], 30, 3, 120000, ["-T1"], dict(threads=1)),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_las_identical_to_reference(case, tmp_path):
    run_ref = _run_ref()
    name, kind, seed, contigs, nreads, drop, bsize, flags, kw = case
    g, reads = make_dataset(kind, seed, contigs, nreads, dropouts=drop)
    d = str(tmp_path)
    dazzdb.write_db(d, "REF", g, block_size=bsize)
    dazzdb.write_db(d, "READS", reads)
    threads = kw.pop("threads", 4)
    mem = 8 << 30
    run_ref.run_reference(d, "REF", ["READS"], flags=list(flags) + ["-M8"], threads=threads)
    prof_ref = None
    if "-p" in flags:
        # The reference's special_log() fills a static table lazily without a lock (map.c:2270-2283):
        # with more than one report thread a second thread can read the half-filled table and emit
        # bogus 40s.  The track itself is -T invariant, so take it from a single-threaded run.
        os.remove(os.path.join(d, ".READS.prof.anno"))
        os.remove(os.path.join(d, ".READS.prof.data"))
        run_ref.run_reference(d, "REF", ["READS"], flags=list(flags) + ["-M8"], threads=1, keep="las_ref_T1")
        prof_ref = (open(os.path.join(d, ".READS.prof.anno"), "rb").read(),
                    open(os.path.join(d, ".READS.prof.data"), "rb").read())
        os.remove(os.path.join(d, ".READS.prof.anno"))
        os.remove(os.path.join(d, ".READS.prof.data"))
    driver.map_db(d, "REF", "READS", threads=threads, mem_limit=mem, **kw)
    ts_r, rr = las.read_las_set(os.path.join(d, "las_ref"), "READS.REF.M@.las")
    ts_h, hr = las.read_las_set(os.path.join(d, "las_hip"), "READS.REF.M@.las")
    assert ts_r == ts_h
    assert len(rr) == len(hr), "%d vs %d records" % (len(rr), len(hr))
    assert len(rr) > 0
    for i, (a, b) in enumerate(zip(rr, hr)):
        assert a == b, "record %d differs:\n%s\n%s" % (i, a, b)
    # per-file record counts must match too (same read partition per thread file)
    for t in range(1, threads + 1):
        _, fa = las.read_las(os.path.join(d, "las_ref", "READS.REF.M%d.las" % t))
        _, fb = las.read_las(os.path.join(d, "las_hip", "READS.REF.M%d.las" % t))
        assert len(fa) == len(fb)
    if prof_ref is not None:
        got = (open(os.path.join(d, ".READS.prof.anno"), "rb").read(),
               open(os.path.join(d, ".READS.prof.data"), "rb").read())
        assert got[0] == prof_ref[0]
        if got[1] != prof_ref[1]:      # keep the evidence (the run-time box merges gpurun_out/ back)
            out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
            os.makedirs(out, exist_ok=True)
            open(os.path.join(out, "prof_ref_%s.bin" % name), "wb").write(prof_ref[1])
            open(os.path.join(out, "prof_hip_%s.bin" % name), "wb").write(got[1])
            a = np.frombuffer(prof_ref[1], dtype=np.uint8)
            b = np.frombuffer(got[1], dtype=np.uint8)
            idx = np.nonzero(a != b)[0] if len(a) == len(b) else []
            raise AssertionError("-p track differs at %s: ref %s hip %s" % (list(idx[:20]), list(a[idx[:20]]), list(b[idx[:20]])))


COVER_CASES = [
    # name, kind, seed, contigs, nreads, dropouts, flags(ref), mflag, extra kwargs
    ("C", "repeat", 51, [300000, 250000], 100, 2, ["-C"], 3, {}),
    ("C_N", "random", 52, [250000, 200000], 80, 1, ["-C", "-N"], 2, {}),
    ("C_s126", "repeat", 53, [250000], 60, 1, ["-C", "-s126"], 3, dict(spacing=126)),
]


@pytest.mark.parametrize("case", COVER_CASES, ids=[c[0] for c in COVER_CASES])
def test_cover_files_identical_to_reference(case, tmp_path):
    """-C / -N (map.c FLAG_DOB): the REF-vs-READS files REF.READS.R@.las carry the B view of every
    reported alignment, complemented chains in reversed order."""
    run_ref = _run_ref()
    name, kind, seed, contigs, nreads, drop, flags, mflag, kw = case
    g, reads = make_dataset(kind, seed, contigs, nreads, dropouts=drop)
    d = str(tmp_path)
    dazzdb.write_db(d, "REF", g, block_size=400000)
    dazzdb.write_db(d, "READS", reads)
    run_ref.run_reference(d, "REF", ["READS"], flags=list(flags) + ["-M8"], threads=4)
    driver.map_db(d, "REF", "READS", threads=4, mem_limit=8 << 30, mflag=mflag, **kw)
    pats = ["REF.READS.R@.las"] + (["READS.REF.M@.las"] if (mflag & 1) else [])
    for pat in pats:
        ts_r, rr = las.read_las_set(os.path.join(d, "las_ref"), pat)
        ts_h, hr = las.read_las_set(os.path.join(d, "las_hip"), pat)
        assert ts_r == ts_h
        assert len(rr) == len(hr) and len(rr) > 0, "%s: %d vs %d records" % (pat, len(rr), len(hr))
        for i, (a, b) in enumerate(zip(rr, hr)):
            assert a == b, "%s record %d differs:\n%s\n%s" % (pat, i, a, b)
        for t in range(1, 5):
            fn = pat.replace("@", str(t))
            assert len(las.read_las(os.path.join(d, "las_ref", fn))[1]) == len(las.read_las(os.path.join(d, "las_hip", fn))[1])
    if not (mflag & 1):
        assert not os.path.exists(os.path.join(d, "las_hip", "READS.REF.M1.las"))


def test_ragged_and_degenerate_reads(tmp_path):
    """Reads the simulator never makes: unmappable noise, reads barely longer than k, error-free copies
    of the reference on both strands (bands of width 1, very long snakes), a 40 kb read next to 30 bp
    ones (ragged block), a low-complexity read, a read that is a whole contig, reads at contig ends."""
    run_ref = _run_ref()
    rng = np.random.default_rng(1091)        # (not 91: the noise must not repeat the genome's stream)
    g, reads = make_dataset("random", 91, [120000, 9000, 60000], 12)
    rc = lambda s: (3 - s[::-1]).astype(np.uint8)
    extra = [rng.integers(0, 4, 5000, dtype=np.uint8),                  # noise
             rng.integers(0, 4, 21, dtype=np.uint8),                    # k+1 bases of noise
             g[0][1000:1030].copy(), g[0][5000:5020].copy(),            # 30 / 20 bp exact
             g[0][20000:60000].copy(),                                  # 40 kb exact
             rc(g[2][10000:25000]),                                     # exact, other strand
             g[1].copy(),                                               # a whole contig
             g[0][:3000].copy(), rc(g[2][-2500:]),                      # contig ends
             np.zeros(4000, dtype=np.uint8),                            # poly-A
             np.tile(np.array([0, 1], dtype=np.uint8), 1500),           # dinucleotide repeat
             np.concatenate([g[0][70000:74000], g[2][3000:8000]])]      # chimera of two contigs
    reads = list(reads) + extra
    d = str(tmp_path)
    dazzdb.write_db(d, "REF", g, block_size=130000)
    dazzdb.write_db(d, "READS", reads)
    run_ref.run_reference(d, "REF", ["READS"], flags=["-M8"], threads=4)
    driver.map_db(d, "REF", "READS", threads=4, mem_limit=8 << 30)
    ts_r, rr = las.read_las_set(os.path.join(d, "las_ref"), "READS.REF.M@.las")
    ts_h, hr = las.read_las_set(os.path.join(d, "las_hip"), "READS.REF.M@.las")
    assert ts_r == ts_h and len(rr) == len(hr) and len(rr) >= 15
    for i, (a, b) in enumerate(zip(rr, hr)):
        assert a == b, "record %d differs:\n%s\n%s" % (i, a, b)
    mapped = {r.aread for r in hr}
    assert 12 + 4 in mapped and 12 + 5 in mapped and 12 + 6 in mapped          # the exact copies map
    assert 12 not in mapped and 13 not in mapped                               # the noise does not


@pytest.mark.parametrize("opts", [dict(align_g=16), dict(align_level=1), dict(align_level=2), dict(pool_div=400),
                                  dict(chain_g=64), dict(chain_g=32), dict(chain_g=16),
                                  dict(chain_pile=16), dict(chain_pile=16, pile_cap=128), dict(chain_pile=0),
                                  dict(merge_dense=1), dict(merge_dense=2),
                                  dict(cluster=1), dict(cluster=1, chain_pile=16), dict(cluster=1, chain_g=64),
                                  dict(chain_small=0), dict(cluster=1, chain_small=0), dict(cluster=1, chain_small=7),
                                  dict(chain_lane=0), dict(cluster=1, chain_lane=0), dict(batch=0),
                                  dict(cluster=1, cluster_sort=0), dict(expand=0), dict(cluster=1, in_flight=0),
                                  dict(cluster=1, seed_word=0), dict(cluster=1, seed_word=0, expand=0), dict(cluster=1, expand=0),
                                  dict(cluster=1, fin_stream=0), dict(fin_stream=0), dict(cluster=1, alloc_fail=30), dict(cluster=1, alloc_fail=90), dict(alloc_fail=150),
                                  dict(cluster=1, oom_finish=2), dict(oom_finish=1), dict(cluster=1, read_sort=0),
                                  dict(cluster=1, read_sort=2), dict(cluster=1, read_sort=3), dict(cluster=1, read_sort=5),
                                  dict(cluster=1, read_sort=7), dict(cluster=1, defer_close=1), dict(cluster=1, tails=0),
                                  dict(chain_simple=0), dict(cluster=1, chain_simple=0),
                                  dict(cluster=1, fuse=0), dict(cluster=1, fuse=0, read_sort=3), dict(cluster=1, fuse=0, read_sort=5),
                                  dict(cluster=1, fuse=0, chain_simple=0), dict(cluster=1, chain_classes=0), dict(chain_classes=0),
                                  dict(cluster=1, defer_close=0), dict(defer_close=0),
                                  dict(cluster=1, chain_pile=0), dict(match_compact=0), dict(cluster=1, match_compact=0),
                                  dict(sort_ballot=1)],
                         ids=["four_per_wavefront", "wave64", "global_band", "pool_grows", "chain64", "chain32", "chain16",
                              "chain_piles", "chain_piles_spill", "chain_no_piles", "merge_dense_table", "merge_low16_probes",
                              "clusters", "clusters_piles", "clusters_chain64",
                              "no_lds_groups", "clusters_no_lds_groups", "clusters_lds_groups_7",
                              "no_lane_groups", "clusters_no_lane_groups", "no_pass_batches", "clusters_global_sort", "expand_per_entry", "clusters_nothing_in_flight",
                              "clusters_key_value_sort", "clusters_key_value_sort_per_entry", "clusters_expand_per_entry",
                              "clusters_finish_on_main_stream", "finish_on_main_stream", "clusters_memory_runs_out_30", "clusters_memory_runs_out_90", "memory_runs_out_150",
                              "clusters_memory_runs_out_inside_finish", "memory_runs_out_inside_finish", "clusters_global_seed_sort",
                              "clusters_read_sort_large_workgroups", "clusters_read_sort_oversize_through_global_memory", "clusters_read_sort_ballot_ranks",
                              "clusters_read_sort_oversize_small_workgroups", "clusters_close_deferred_to_next_pass",
                              "clusters_nothing_on_the_tail_stream",
                              "no_simple_groups", "clusters_no_simple_groups",
                              "clusters_unfused_round3_route", "clusters_unfused_oversize_through_global_memory",
                              "clusters_unfused_ballot_ranks", "clusters_unfused_no_simple_groups",
                              "clusters_no_diagonal_classes", "no_diagonal_classes",
                              "clusters_every_pass_closed_before_it_returns", "every_pass_closed_before_it_returns",
                              "clusters_no_piles", "match_item_by_item", "clusters_match_item_by_item", "sort_ranks_from_ballots"])
def test_every_kernel_variant_gives_the_same_files(opts, tmp_path):
    """The alignment kernel ladder, the chain group sizes and the pool-overflow retry are alternative routes
    to the same bytes: force each one and compare with the default route on data that has repeats, dropouts and
    low-complexity reads; the default route on this data is compared with the reference by
    test_variant_dataset_matches_reference below."""
    from damapper_amd import capi
    d = str(tmp_path)
    base = _variant_base()
    _variant_dbs(d)
    L = capi.lib()
    try:
        for k, v in opts.items():
            assert L.dmh_test_option(k.encode(), v) == 0
        alt = _variant_run(d, "las_alt")
    finally:
        for k, v in VARIANT_DEFAULTS.items():
            L.dmh_test_option(k.encode(), v)
    assert len(base[0][1]) > 50
    assert base == alt


def test_long_reads_identical_to_reference(tmp_path):
    """Reads of 38 kb (a log-normal library's tail): a pass of the d-wave drops more pebbles than a lane group's LDS
    holds (288: reads beyond ~29 kb) -- they spill into the group's arena in global memory instead of sending the
    candidate one rung up.  Files against the reference's; the switch off gives the same files, and the ladder's
    counters say which route ran."""
    from damapper_amd import capi
    run_ref = _run_ref()
    g, reads = make_dataset("repeat", 61, [500000, 400000], 16, read_len=38000, dropouts=1)
    d = str(tmp_path)
    dazzdb.write_db(d, "REF", g, block_size=1000000)
    dazzdb.write_db(d, "READS", reads)
    run_ref.run_reference(d, "REF", ["READS"], flags=["-M8"], threads=4)
    _, want = las.read_las_set(os.path.join(d, "las_ref"), "READS.REF.M@.las")
    assert len(want) >= 16
    L = capi.lib()
    counts = {}
    try:
        for name, opts in (("default", {}), ("off", dict(pebble_spill=0))):
            for k, v in opts.items():
                assert L.dmh_test_option(k.encode(), v) == 0
            L.dmh_timing_reset()
            L.dmh_timing_enable(1)
            driver.map_db(d, "REF", "READS", threads=4, mem_limit=8 << 30, sort_path=os.path.join(d, "las_" + name))
            L.dmh_sync()
            counts[name] = {k: v["launches"] for k, v in capi.timings().items()}
            L.dmh_timing_enable(0)
            _, got = las.read_las_set(os.path.join(d, "las_" + name), "READS.REF.M@.las")
            assert got == want, name
    finally:
        L.dmh_timing_enable(0)
        L.dmh_test_option(b"pebble_spill", -1)
    assert counts["off"].get("count:align32_pebbles_over_288", 0) > 0, counts["off"]
    assert counts["default"].get("count:align32_pebbles_over_288", 0) == 0, counts["default"]


VARIANT_DEFAULTS = dict(align_level=-1, align_g=-1, chain_g=0, pool_div=1, chain_pile=-1, pile_cap=0, merge_dense=-1,
                        cluster=-1, chain_small=-1, chain_lane=-1, batch=-1, cluster_sort=-1, expand=-1, in_flight=-1,
                        seed_word=-1, alloc_fail=0, fin_stream=-1, read_sort=-1, oom_finish=0, chain_simple=-1,
                        defer_close=-1, tails=-1, fuse=-1, chain_classes=-1, pebble_spill=-1, front_ahead=-1, match_compact=-1, sort_ballot=-1)
_VARIANT = {}


def _variant_dbs(d):
    g, reads = make_dataset("repeat", 97, [150000, 120000], 70, dropouts=2)
    reads = list(reads) + [np.zeros(3000, dtype=np.uint8), np.tile(np.array([0, 1, 1], dtype=np.uint8), 1200)]
    dazzdb.write_db(d, "REF", g, block_size=400000)
    dazzdb.write_db(d, "READS", reads)


def _variant_run(d, out):
    driver.map_db(d, "REF", "READS", threads=4, mem_limit=8 << 30, sort_path=os.path.join(d, out), profile=1)
    return (las.read_las_set(os.path.join(d, out), "READS.REF.M@.las"),
            open(os.path.join(d, ".READS.prof.data"), "rb").read())


def _variant_base():
    """The default route's files on the variant data set (made once per session)."""
    if "base" not in _VARIANT:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            _variant_dbs(d)
            _VARIANT["base"] = _variant_run(d, "las_default")
    return _VARIANT["base"]


@pytest.mark.parametrize("cluster", [-1, 1], ids=["auto", "clusters"])
def test_memory_runs_out_at_every_allocation(cluster, tmp_path):
    """The n-th device allocation "runs out of memory" (the batches in flight are finished, idle buffers and the pool
    go back to the driver, the allocation is made again), for every n over the first passes of the variant data set:
    whichever buffer is being grown at that moment -- the reclaim may look at it: a stale pointer in DevBuf::ensure
    was freed twice there until round 3 -- the files stay the same."""
    from damapper_amd import capi
    d = str(tmp_path)
    base = _variant_base()
    _variant_dbs(d)
    L = capi.lib()
    try:
        assert L.dmh_test_option(b"cluster", cluster) == 0
        for n in range(5, 200, 4):
            assert L.dmh_test_option(b"alloc_fail", n) == 0
            alt = _variant_run(d, "las_alt")
            assert base == alt, "alloc_fail = %d" % n
    finally:
        for k, v in VARIANT_DEFAULTS.items():
            L.dmh_test_option(k.encode(), v)


def test_variant_dataset_matches_reference(tmp_path):
    """Pins the 30-odd forced routes above: the default route's record stream and -p track on THEIR data set
    (repeat-rich genome, dropout reads, a poly-A and a trinucleotide read) against the compiled reference."""
    run_ref = _run_ref()
    d = str(tmp_path)
    _variant_dbs(d)
    run_ref.run_reference(d, "REF", ["READS"], flags=["-p", "-M8"], threads=4)
    # (the track from a single-threaded run: special_log() races with more than one report thread)
    run_ref.run_reference(d, "REF", ["READS"], flags=["-p", "-M8"], threads=1, keep="las_ref_T1")
    want_prof = open(os.path.join(d, ".READS.prof.data"), "rb").read()
    want = las.read_las_set(os.path.join(d, "las_ref"), "READS.REF.M@.las")
    base = _variant_base()
    assert base[0] == want
    assert base[1] == want_prof


def test_trimmed_db_view_identical_to_reference(tmp_path):
    """A READS DB split with a cutoff and without -a (DBsplit -x1000): short reads and reads that are not
    the best of their well are left out, blocks are cut over the survivors and A-read numbers are the
    trimmed ones (Trim_DB, DB.c:908) -- also with a mask made over the untrimmed reads."""
    run_ref = _run_ref()
    rng = np.random.default_rng(5)
    g, reads = make_dataset("random", 151, [150000, 120000], 40)
    short = [g[0][1000:1400].copy(), g[1][500:1300].copy(), rng.integers(0, 4, 300, dtype=np.uint8)]
    mixed, best = [], []
    for i, r in enumerate(reads):
        mixed.append(r)
        best.append(i % 7 != 3)
        if i % 9 == 4:
            mixed.append(short[i % 3])
            best.append(True)
    d = str(tmp_path)
    dazzdb.write_db(d, "REF", g, block_size=400000)
    dazzdb.write_db(d, "READS", mixed, block_size=150000, cutoff=1000, all_reads=False, best=best)
    dazzdb.write_mask_track(d, "READS", "dust", [[(len(s) // 3, len(s) // 3 + 200)] if len(s) > 2000 else [] for s in mixed])
    nb, _ = dazzdb.read_stub(os.path.join(d, "READS.db"))
    assert nb == 3
    blocks = ["READS.%d" % (k + 1) for k in range(nb)]
    run_ref.run_reference(d, "REF", blocks, flags=["-M8", "-mdust"], threads=4)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import subprocess
    import sys
    cmd = [sys.executable, "-m", "damapper_amd.dispatch", os.path.join(d, "REF.db"), os.path.join(d, "READS.db"),
           "--gpus", "1", "--out", os.path.join(d, "las_disp"), "-m", "dust"]
    p = subprocess.run(cmd, env=dict(os.environ, PYTHONPATH=root), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    total = 0
    for b in blocks:
        _, rr = las.read_las_set(os.path.join(d, "las_ref"), b + ".REF.M@.las")
        _, hr = las.read_las_set(os.path.join(d, "las_disp"), b + ".REF.M@.las")
        assert rr == hr, b
        total += len(rr)
    assert total >= 30


def test_mask_tracks_identical_to_reference(tmp_path):
    """-m<track>: two masks on the REF (merged, damapper.c:232-343), one on the READS, one that no DB
    has; multi-block REF so the per-block slices of the track are exercised."""
    run_ref = _run_ref()
    rng = np.random.default_rng(71)
    g, reads = make_dataset("repeat", 71, [220000, 180000, 150000], 90, dropouts=1)
    d = str(tmp_path)
    dazzdb.write_db(d, "REF", g, block_size=300000)
    dazzdb.write_db(d, "READS", reads)

    def masks(seqs, k):
        out = []
        for s in seqs:
            n = len(s)
            cuts = sorted(set(int(x) for x in rng.integers(0, n, size=2 * k)))
            cuts = cuts[:len(cuts) // 2 * 2]
            out.append([(cuts[j], cuts[j + 1]) for j in range(0, len(cuts), 2) if cuts[j + 1] - cuts[j] < n // 6])
        return out

    dazzdb.write_mask_track(d, "REF", "dust", masks(g, 40))
    dazzdb.write_mask_track(d, "REF", "tan", masks(g, 12))
    dazzdb.write_mask_track(d, "READS", "dust", masks(reads, 3))
    flags = ["-mdust", "-mtan", "-mnone"]
    run_ref.run_reference(d, "REF", ["READS"], flags=flags + ["-M8"], threads=4)
    run_ref.run_reference(d, "REF", ["READS"], flags=["-M8"], threads=4, keep="las_nomask")
    driver.map_db(d, "REF", "READS", threads=4, mem_limit=8 << 30, masks=["dust", "tan", "none"])
    ts_r, rr = las.read_las_set(os.path.join(d, "las_ref"), "READS.REF.M@.las")
    ts_h, hr = las.read_las_set(os.path.join(d, "las_hip"), "READS.REF.M@.las")
    _, nomask = las.read_las_set(os.path.join(d, "las_nomask"), "READS.REF.M@.las")
    assert len(rr) > 0 and rr != nomask, "the masks must change the result for this test to mean anything"
    assert ts_r == ts_h and len(rr) == len(hr)
    for i, (a, b) in enumerate(zip(rr, hr)):
        assert a == b, "record %d differs:\n%s\n%s" % (i, a, b)


def test_reference_driver_linked_against_hip_library(tmp_path):
    """Drop-in proof: the reference's own damapper.c/DB.c/QV.c, linked against libdamapper_hip.so instead
    of map.c + align.c (oracle/_ref/damapper_hip), writes the same .las files as the all-CPU reference."""
    run_ref = _run_ref()
    if not os.path.exists(os.path.join(run_ref.REF_DIR, "damapper_hip")):
        pytest.skip("oracle/_ref/damapper_hip not built")
    g, reads = make_dataset("repeat", 46, [200000, 150000, 120000], 60, dropouts=1)
    d = str(tmp_path)
    dazzdb.write_db(d, "REF", g, block_size=260000)
    dazzdb.write_db(d, "READS", reads, block_size=350000)      # two read blocks
    nb, _ = dazzdb.read_stub(os.path.join(d, "READS.db"))
    assert nb >= 2
    blocks = ["READS.%d" % (k + 1) for k in range(nb)]
    run_ref.run_reference(d, "REF", blocks, flags=["-v", "-p"], threads=4, keep="las_ref")
    run_ref.run_reference(d, "REF", blocks, flags=["-v", "-p"], threads=4, keep="las_drop", binary="damapper_hip")
    for b in blocks:
        ts_r, rr = las.read_las_set(os.path.join(d, "las_ref"), b + ".REF.M@.las")
        ts_h, hr = las.read_las_set(os.path.join(d, "las_drop"), b + ".REF.M@.las")
        assert len(rr) > 0 and rr == hr, "block %s: %d vs %d records" % (b, len(rr), len(hr))


def test_reference_driver_with_ref_index_cache(tmp_path, monkeypatch):
    """DMH_REF_CACHE=1: the unmodified driver binary re-indexes every REF block for every READS block; with
    the cache the second and later READS blocks find the REF indices in HBM.  Same files, and the
    verbose log shows the hits."""
    run_ref = _run_ref()
    if not os.path.exists(os.path.join(run_ref.REF_DIR, "damapper_hip")):
        pytest.skip("oracle/_ref/damapper_hip not built")
    g, reads = make_dataset("repeat", 146, [200000, 150000, 120000], 75, dropouts=1)
    d = str(tmp_path)
    dazzdb.write_db(d, "REF", g, block_size=260000)
    dazzdb.write_db(d, "READS", reads, block_size=260000)      # three read blocks
    nb, _ = dazzdb.read_stub(os.path.join(d, "READS.db"))
    assert nb >= 3
    blocks = ["READS.%d" % (k + 1) for k in range(nb)]
    run_ref.run_reference(d, "REF", blocks, flags=["-v"], threads=4, keep="las_ref")
    monkeypatch.setenv("DMH_REF_CACHE", "1")
    _, log = run_ref.run_reference(d, "REF", blocks, flags=["-v"], threads=4, keep="las_drop", binary="damapper_hip")
    nrb, _ = dazzdb.read_stub(os.path.join(d, "REF.db"))
    assert log.count("Index found in HBM") == 2 * nrb * (nb - 1)
    for b in blocks:
        _, rr = las.read_las_set(os.path.join(d, "las_ref"), b + ".REF.M@.las")
        _, hr = las.read_las_set(os.path.join(d, "las_drop"), b + ".REF.M@.las")
        assert len(rr) > 0 and rr == hr, "block %s: %d vs %d records" % (b, len(rr), len(hr))


def test_resident_driver_recipe_matches_reference(tmp_path):
    """INTEGRATION.md's driver change (REF blocks read, complemented and indexed ONCE, dmh_retain_ref_index(1);
    oracle/resident_driver.py applies it to the reference's damapper.c, oracle/_ref/damapper_hip_resident): three
    READS blocks against a three-block REF, -p and a mask-free -v run -- the same files as the all-CPU reference."""
    run_ref = _run_ref()
    if not os.path.exists(os.path.join(run_ref.REF_DIR, "damapper_hip_resident")):
        pytest.skip("oracle/_ref/damapper_hip_resident not built")
    g, reads = make_dataset("repeat", 246, [200000, 150000, 120000], 75, dropouts=1)
    d = str(tmp_path)
    dazzdb.write_db(d, "REF", g, block_size=260000)
    dazzdb.write_db(d, "READS", reads, block_size=260000)      # three read blocks
    nb, _ = dazzdb.read_stub(os.path.join(d, "READS.db"))
    nrb, _ = dazzdb.read_stub(os.path.join(d, "REF.db"))
    assert nb >= 3 and nrb >= 2
    blocks = ["READS.%d" % (k + 1) for k in range(nb)]
    run_ref.run_reference(d, "REF", blocks, flags=["-v", "-p"], threads=4, keep="las_ref")
    _, log = run_ref.run_reference(d, "REF", blocks, flags=["-v", "-p"], threads=4, keep="las_res",
                                   binary="damapper_hip_resident")
    assert log.count("Building indices for REF.") == nrb          # once per REF block, not once per READS block
    for b in blocks:
        _, rr = las.read_las_set(os.path.join(d, "las_ref"), b + ".REF.M@.las")
        _, hr = las.read_las_set(os.path.join(d, "las_res"), b + ".REF.M@.las")
        assert len(rr) > 0 and rr == hr, "block %s: %d vs %d records" % (b, len(rr), len(hr))


@pytest.mark.parametrize("flags", [["-p"], ["-v", "-C"]], ids=["p", "v_C"])
def test_fast_driver_recipe_matches_reference(flags, tmp_path):
    """The C host side that "stays", driving the library the way driver.py does (oracle/fast_driver.py applied to the
    reference's damapper.c, oracle/_ref/damapper_hip_fast): REF resident, READS blocks opened by DB.c but decoded on the
    device from page-locked .bps bytes, dmh_async_report(2) with two READS blocks open, dmh_next_pass, the LAsort / LAcat
    commands on a thread once dmh_reports_done() allows -- three READS blocks (one shorter than ten passes' worth)
    against a three-block REF: the same per-thread files (and -p track) as the all-CPU reference."""
    run_ref = _run_ref()
    if not os.path.exists(os.path.join(run_ref.REF_DIR, "damapper_hip_fast")):
        pytest.skip("oracle/_ref/damapper_hip_fast not built")
    g, reads = make_dataset("repeat", 446, [200000, 150000, 120000, 90000, 80000, 70000], 75, dropouts=1)
    d = str(tmp_path)
    dazzdb.write_db(d, "REF", g, block_size=130000)             # six REF blocks: twelve passes, Reporter ten passes late
    dazzdb.write_db(d, "READS", reads, block_size=260000)
    nb, _ = dazzdb.read_stub(os.path.join(d, "READS.db"))
    nrb, _ = dazzdb.read_stub(os.path.join(d, "REF.db"))
    assert nb >= 3 and nrb >= 5
    blocks = ["READS.%d" % (k + 1) for k in range(nb)]
    run_ref.run_reference(d, "REF", blocks, flags=flags, threads=4, keep="las_ref")
    prof = {}
    if "-p" in flags:
        # (the track from a single-threaded run: special_log() races with more than one report thread)
        run_ref.run_reference(d, "REF", blocks, flags=flags, threads=1, keep="las_ref_T1")
        for b in blocks:
            for x in ("anno", "data"):
                f = os.path.join(d, ".%s.prof.%s" % (b, x))
                prof[f] = open(f, "rb").read()
                os.remove(f)
    run_ref.run_reference(d, "REF", blocks, flags=flags, threads=4, keep="las_fast", binary="damapper_hip_fast")
    for b in blocks:
        _, rr = las.read_las_set(os.path.join(d, "las_ref"), b + ".REF.M@.las")
        _, hr = las.read_las_set(os.path.join(d, "las_fast"), b + ".REF.M@.las")
        assert len(rr) > 0 and rr == hr, "block %s: %d vs %d records" % (b, len(rr), len(hr))
        if "-C" in flags:
            _, rr = las.read_las_set(os.path.join(d, "las_ref"), "REF." + b + ".R@.las")
            _, hr = las.read_las_set(os.path.join(d, "las_fast"), "REF." + b + ".R@.las")
            assert len(rr) > 0 and rr == hr, "block %s (R files): %d vs %d records" % (b, len(rr), len(hr))
    for f, want in prof.items():
        assert open(f, "rb").read() == want, f


def test_next_pass_hint_gives_identical_files(tmp_path, monkeypatch):
    """dmh_next_pass (the driver names the pass that follows; the library then probes the forward and the complement
    index of a REF block in ONE sweep of the READ keys -- the driver's default; DMH_HINT_PASSES=0: every pass for itself):
    the same files either way -- through the bench's call sequence on a three-block reference, two read blocks."""
    import test_gpu_scale as scale
    run_ref = _run_ref()
    g, reads = make_dataset("repeat", 346, [200000, 150000, 120000], 75, dropouts=1)
    d = str(tmp_path)
    dazzdb.write_db(d, "REF", g, block_size=260000)
    dazzdb.write_db(d, "READS", reads, block_size=400000)
    nb, _ = dazzdb.read_stub(os.path.join(d, "READS.db"))
    blocks = ["READS.%d" % (k + 1) for k in range(nb)]
    run_ref.run_reference(d, "REF", blocks, flags=[], threads=4, keep="las_ref")
    from damapper_amd import capi
    L = capi.lib()
    # (with the hints the front of the REF block that FOLLOWS is made on a stream of its own, next to the per-read kernels
    #  of the block at hand -- "front_ahead" test option = 0: every front on the main stream when its pass comes)
    try:
        for hint, ahead in (("1", -1), ("1", 0), ("0", -1)):
            monkeypatch.setenv("DMH_HINT_PASSES", hint)
            assert L.dmh_test_option(b"front_ahead", ahead) == 0
            out = os.path.join(d, "las_hint%s_%d" % (hint, ahead))
            L.dmh_timing_reset()
            L.dmh_timing_enable(1)
            scale._map_blocks_like_bench(d, "REF", "READS", out, report_mode=2, overlap=True)
            L.dmh_sync()
            waits = capi.timings().get("host:wait_front_ahead", {}).get("launches", 0)
            L.dmh_timing_enable(0)
            assert (waits > 0) == (hint == "1" and ahead != 0), (hint, ahead, waits)
            for b in blocks:
                _, rr = las.read_las_set(os.path.join(d, "las_ref"), b + ".REF.M@.las")
                _, hr = las.read_las_set(out, b + ".REF.M@.las")
                assert len(rr) > 0 and rr == hr, "hints %s, ahead %d, block %s: %d vs %d records" % (hint, ahead, b, len(rr), len(hr))
    finally:
        L.dmh_timing_enable(0)
        L.dmh_test_option(b"front_ahead", -1)


def test_memory_runs_out_while_a_front_is_made_ahead(tmp_path):
    """The n-th device allocation "runs out of memory" (alloc_fail: the batches in flight are finished, idle buffers go
    back, the allocation is made again) while the block loop runs with the front of the following REF block on its own
    stream -- three REF blocks, two read blocks, the bench's call sequence: the files stay the reference's for every n."""
    import test_gpu_scale as scale
    from damapper_amd import capi
    run_ref = _run_ref()
    g, reads = make_dataset("repeat", 347, [200000, 150000, 120000], 60, dropouts=1)
    d = str(tmp_path)
    dazzdb.write_db(d, "REF", g, block_size=260000)
    dazzdb.write_db(d, "READS", reads, block_size=320000)
    nb, _ = dazzdb.read_stub(os.path.join(d, "READS.db"))
    blocks = ["READS.%d" % (k + 1) for k in range(nb)]
    assert nb >= 2
    run_ref.run_reference(d, "REF", blocks, flags=[], threads=4, keep="las_ref")
    L = capi.lib()
    try:
        for n in range(4, 160, 6):
            assert L.dmh_test_option(b"alloc_fail", n) == 0
            out = os.path.join(d, "las_fail%d" % n)
            scale._map_blocks_like_bench(d, "REF", "READS", out, report_mode=2, overlap=True)
            for b in blocks:
                _, rr = las.read_las_set(os.path.join(d, "las_ref"), b + ".REF.M@.las")
                _, hr = las.read_las_set(out, b + ".REF.M@.las")
                assert len(rr) > 0 and rr == hr, "alloc_fail %d, block %s: %d vs %d records" % (n, b, len(rr), len(hr))
    finally:
        L.dmh_test_option(b"alloc_fail", 0)


def test_dispatcher_maps_every_read_block(tmp_path):
    """damapper_amd.dispatch (read blocks -> GPUs, resume by output file) against the reference run on
    the same blocks."""
    import subprocess
    import sys
    run_ref = _run_ref()
    g, reads = make_dataset("random", 47, [180000, 160000], 50, dropouts=0)
    d = str(tmp_path)
    dazzdb.write_db(d, "REF", g, block_size=200000)
    dazzdb.write_db(d, "READS", reads, block_size=270000)
    nb, _ = dazzdb.read_stub(os.path.join(d, "READS.db"))
    assert nb >= 2
    blocks = ["READS.%d" % (k + 1) for k in range(nb)]
    run_ref.run_reference(d, "REF", blocks, flags=["-M8"], threads=4)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    cmd = [sys.executable, "-m", "damapper_amd.dispatch", os.path.join(d, "REF.db"), os.path.join(d, "READS.db"),
           "--gpus", "1", "--out", os.path.join(d, "las_disp")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    for b in blocks:
        _, rr = las.read_las_set(os.path.join(d, "las_ref"), b + ".REF.M@.las")
        _, hr = las.read_las_set(os.path.join(d, "las_disp"), b + ".REF.M@.las")
        assert len(rr) > 0 and rr == hr, b
    p2 = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p2.returncode == 0 and p2.stdout.count("skipped") == nb
    # a worker that died inside a block leaves <name>.tmp files (parts are renamed only when complete): the block
    # counts as not done and is mapped again
    first = os.path.join(d, "las_disp", blocks[0] + ".REF.M2.las")
    whole = open(first, "rb").read()
    os.rename(first, first + ".tmp")
    with open(first + ".tmp", "r+b") as f:
        f.truncate(len(whole) // 2)
    p2b = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p2b.returncode == 0 and p2b.stdout.count("skipped") == nb - 1, p2b.stdout[-800:]
    assert open(first, "rb").read() == whole and not os.path.exists(first + ".tmp")
    # two workers on "two GPUs" (device numbers are taken modulo the device count, so this runs on a one-GPU box):
    # the N > 1 partition of the read blocks, b mod N, every worker with its own replica of the reference index
    cmd2g = cmd[:-3] + ["2", "--out", os.path.join(d, "las_disp_2gpu")]
    p2g = subprocess.run(cmd2g, env=env, capture_output=True, text=True, timeout=600)
    assert p2g.returncode == 0, p2g.stdout[-1500:] + p2g.stderr[-1500:]
    assert "[gpu 0]" in p2g.stdout and "[gpu 1]" in p2g.stdout
    for b in blocks:
        _, rr = las.read_las_set(os.path.join(d, "las_ref"), b + ".REF.M@.las")
        _, hr = las.read_las_set(os.path.join(d, "las_disp_2gpu"), b + ".REF.M@.las")
        assert rr == hr, b
    # two workers sharing the GPU: same files
    cmd3 = cmd[:-1] + [os.path.join(d, "las_disp2"), "--workers-per-gpu", "2"]
    p3 = subprocess.run(cmd3, env=env, capture_output=True, text=True, timeout=600)
    assert p3.returncode == 0, p3.stdout[-1500:] + p3.stderr[-1500:]
    for b in blocks:
        _, rr = las.read_las_set(os.path.join(d, "las_ref"), b + ".REF.M@.las")
        _, hr = las.read_las_set(os.path.join(d, "las_disp2"), b + ".REF.M@.las")
        assert rr == hr, b
    # the reference's own loop shape: reads decoded on the host, REF index rebuilt per block, synchronous Reporter
    cmd4 = cmd[:-1] + [os.path.join(d, "las_disp3"), "--host-decode", "--rebuild-ref", "--report-mode", "0"]
    p4 = subprocess.run(cmd4, env=env, capture_output=True, text=True, timeout=600)
    assert p4.returncode == 0, p4.stdout[-1500:] + p4.stderr[-1500:]
    for b in blocks:
        for t in range(1, 5):
            fn = "%s.REF.M%d.las" % (b, t)
            assert open(os.path.join(d, "las_disp", fn), "rb").read() == open(os.path.join(d, "las_disp3", fn), "rb").read(), fn


@pytest.mark.parametrize("envvar", ["DMH_NO_PACKED_POS", "DMH_NO_PACK32", "DMH_CHAIN_TAIL", "DMH_BACK_TAIL", "DMH_INDEX_ON_TAIL",
                                    "DMH_NO_LDS_ORDER"])
def test_layout_switches_read_from_the_environment(envvar, tmp_path):
    """Switches that are read once per process: the dispatcher in a process of its own with the switch set, its files
    against the reference's.  DMH_NO_PACKED_POS: k-mer positions as block offsets, turned into (read, rpos) by a table
    look-up; DMH_NO_PACK32: every candidate aligned one per wavefront; the three tail-stream switches = 0: everything on
    the main stream; DMH_NO_LDS_ORDER: ranks from ballots (as on a device whose LDS atomics are not served in lane order)."""
    import subprocess
    import sys
    run_ref = _run_ref()
    g, reads = make_dataset("repeat", 147, [180000, 160000], 60, dropouts=1)
    d = str(tmp_path)
    dazzdb.write_db(d, "REF", g, block_size=200000)
    dazzdb.write_db(d, "READS", reads, block_size=330000)
    nb, _ = dazzdb.read_stub(os.path.join(d, "READS.db"))
    blocks = ["READS.%d" % (k + 1) for k in range(nb)]
    run_ref.run_reference(d, "REF", blocks, flags=["-M8"], threads=4)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    env[envvar] = "0" if envvar.endswith("TAIL") else "1"
    cmd = [sys.executable, "-m", "damapper_amd.dispatch", os.path.join(d, "REF.db"), os.path.join(d, "READS.db"),
           "--gpus", "1", "--out", os.path.join(d, "las_env")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    for b in blocks:
        _, rr = las.read_las_set(os.path.join(d, "las_ref"), b + ".REF.M@.las")
        _, hr = las.read_las_set(os.path.join(d, "las_env"), b + ".REF.M@.las")
        assert len(rr) > 0 and rr == hr, b


def test_device_side_block_decode_equals_load_all_reads(tmp_path):
    """dmh_preload_block_2bit (the .bps slice uploaded as it is and unpacked by a kernel) leaves the same
    device image as uploading the host-decoded block: read lengths of every residue mod 4, one-base reads,
    a trimmed view whose dropped reads leave gaps in the slice, several blocks."""
    rng = np.random.default_rng(5)
    lens = [1, 2, 3, 4, 5, 6, 7, 8, 4095, 4096, 4097, 16383, 16385, 40001] + [int(x) for x in rng.integers(1, 9000, 300)]
    seqs = [rng.integers(0, 4, n).astype(np.uint8) for n in lens]
    best = [i % 5 != 2 for i in range(len(seqs))]
    d = str(tmp_path)
    dazzdb.write_db(d, "R", seqs, block_size=600000, cutoff=3, all_reads=False, best=best)
    nb, _ = dazzdb.read_stub(os.path.join(d, "R.db"))
    assert nb >= 2
    total = 0
    for part in list(range(1, nb + 1)) + [0]:
        host = dazzdb.load_db(os.path.join(d, "R.db"), part=part)
        dev = dazzdb.load_db(os.path.join(d, "R.db"), part=part, packed=True)
        assert dev.bases is None and dev.nreads == host.nreads
        assert (dev.reads == host.reads).all()
        dev.preload()
        img = dev.download()
        L = int(host.reads[host.nreads + 1]["boff"])
        assert (img == host.bases[:L + 1]).all(), part
        dev.release()
        total += host.nreads
    kept = sum(1 for s, b in zip(seqs, best) if len(s) >= 3 and b)
    assert total == 2 * kept


def test_dispatcher_finish_writes_final_las(tmp_path):
    """--finish -C: per block, `<reads>.<b>.REF.las` and `REF.<reads>.<b>.las` hold the chains of the
    reference's per-thread files in map order (LAsort -a / LAcat / LAmerge equivalents)."""
    import subprocess
    import sys
    run_ref = _run_ref()
    g, reads = make_dataset("repeat", 49, [180000, 160000], 50, dropouts=1)
    d = str(tmp_path)
    dazzdb.write_db(d, "REF", g, block_size=400000)
    dazzdb.write_db(d, "READS", reads, block_size=270000)
    nb, _ = dazzdb.read_stub(os.path.join(d, "READS.db"))
    blocks = ["READS.%d" % (k + 1) for k in range(nb)]
    run_ref.run_reference(d, "REF", blocks, flags=["-M8", "-C"], threads=4)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    cmd = [sys.executable, "-m", "damapper_amd.dispatch", os.path.join(d, "REF.db"), os.path.join(d, "READS.db"),
           "--gpus", "1", "--out", os.path.join(d, "las_disp"), "-C", "--finish"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]

    def chainset(recs):
        return sorted(tuple(r.key() for r in c) for c in las.chains(recs))

    for b in blocks:
        for final, pat in ((b + ".REF.las", b + ".REF.M@.las"), ("REF." + b + ".las", "REF." + b + ".R@.las")):
            _, want = las.read_las_set(os.path.join(d, "las_ref"), pat)
            _, got = las.read_las(os.path.join(d, final))
            assert len(want) == len(got) and len(got) > 0
            assert chainset(want) == chainset(got)
            ks = [(c[0].aread, c[0].abpos) for c in las.chains(got)]
            assert ks == sorted(ks)
    p2 = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p2.returncode == 0 and p2.stdout.count("skipped") == nb


@pytest.mark.parametrize("ref_block", [260000, 1 << 30], ids=["ref_blocks", "one_ref_block"])
def test_retained_ref_index_and_pipelining_give_identical_files(ref_block, tmp_path):
    """The opt-in throughput modes of the driver (REF indices built once and kept resident, next block's
    index queued before Reporter, .las written by the worker thread, alignment kernel left running behind
    Reporter -- the library does that only for a one-block reference) must not change a byte."""
    from damapper_amd import capi
    g, reads = make_dataset("repeat", 48, [200000, 150000, 100000], 90, dropouts=1)
    d = str(tmp_path)
    dazzdb.write_db(d, "REF", g, block_size=ref_block)
    dazzdb.write_db(d, "READS", reads, block_size=320000)
    nb, _ = dazzdb.read_stub(os.path.join(d, "READS.db"))
    nrb, _ = dazzdb.read_stub(os.path.join(d, "REF.db"))
    assert nb >= 3

    def run(out, fast):
        cwd = os.getcwd()
        os.chdir(d)
        try:
            m = driver.Mapper(threads=4, mem_limit=8 << 30, sort_path=os.path.join(d, out), retain_ref=bool(fast))
            ref_blocks = [dazzdb.load_db("./REF.db", part=k + 1) for k in range(nrb)]
            ref_whole = dazzdb.load_db("./REF.db")
            blocks = [dazzdb.load_db("./READS.db", part=b + 1) for b in range(nb)]
            capi.lib().dmh_async_report(fast)
            for b, blk in enumerate(blocks):
                nxt = blocks[b + 1] if (fast and b + 1 < nb) else None
                m.map_block(blk, "READS.%d" % (b + 1), ref_blocks, ref_whole, "REF", next_reads=nxt)
                if fast == 2:
                    capi.lib().dmh_release_block(blk.ptr())     # no-op: the deferred Reporter holds its own copy
                    blk.bases[:] = 0                             # ... and never reads the host block again
            capi.lib().dmh_report_wait()
            capi.lib().dmh_async_report(0)
            m.release_ref()
            capi.lib().dmh_retain_ref_index(0)
        finally:
            os.chdir(cwd)

    run("las_plain", 0)
    run("las_fast", 1)
    run("las_deferred", 2)       # + the alignment kernel of block i runs next to the filter of block i+1
    # ... and the deferred kernel's retry rounds (pool too small, then the wider kernels) on its own stream
    L = capi.lib()
    try:
        L.dmh_test_option(b"pool_div", 400)
        run("las_deferred_retry", 2)
        L.dmh_test_option(b"pool_div", 1)
        L.dmh_test_option(b"align_level", 2)
        run("las_deferred_global_band", 2)
    finally:
        L.dmh_test_option(b"pool_div", 1)
        L.dmh_test_option(b"align_level", -1)
        L.dmh_test_option(b"align_g", -1)
    for b in range(nb):
        for t in range(1, 5):
            fn = "READS.%d.REF.M%d.las" % (b + 1, t)
            a = open(os.path.join(d, "las_plain", fn), "rb").read()
            for other in ("las_fast", "las_deferred", "las_deferred_retry", "las_deferred_global_band"):
                c = open(os.path.join(d, other, fn), "rb").read()
                assert a == c and len(a) > 12, (other, fn)
