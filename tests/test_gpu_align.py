"""GPU parity for the d-wave kernel: every Local_Alignment (end points, diffs, A- and B-view
traces) must equal the compiled reference's, including complemented reads, off-alignment seeds,
sequence-boundary cases and low-complexity sequence that widens the band past one wavefront."""
import ctypes as C

import numpy as np
import pytest

from damapper_amd import capi, dazzdb
from util_pipeline import Api, make_dataset, run_filter
from util_align import jobs_from_candidates, comp_block, random_jobs

pytestmark = pytest.mark.gpu


def _ref():
    import ref_lib
    if not ref_lib.available():
        pytest.skip("oracle/_ref not built")
    return ref_lib


CASES = [("random", 31, [200000, 150000], 40, 0, 100, False), ("repeat", 32, [250000], 30, 2, 100, False),
         ("random", 33, [120000, 100000], 25, 1, 126, False), ("repeat", 34, [150000, 90000], 20, 3, 50, True)]


@pytest.mark.parametrize("case", CASES, ids=[str(i) for i in range(len(CASES))])
def test_local_alignment_matches_reference(case):
    ref_lib = _ref()
    kind, seed, contigs, nreads, drop, tspace, lowcomplex = case
    rng = np.random.default_rng(seed)
    g, rd = make_dataset(kind, seed, contigs, nreads, dropouts=drop)
    if lowcomplex:
        for p in range(5000, len(g[0]) - 5000, 20000):
            g[0][p:p + 300] = 0
            g[0][p + 1000:p + 1400] = np.tile(np.array([0, 1], dtype=np.uint8), 200)
        rd, _ = dazzdb.simulate_reads(rng, g, nreads, dropouts=drop)
    res = run_filter(Api("ref"), rd, g, [0, len(contigs)], K=20, want_seeds=False)
    reads, ref, jobs = jobs_from_candidates(res, g, max_per_cand=8)
    jobs += random_jobs(rng, reads, ref, 200)
    creads = comp_block(reads)
    freq = [ref.db.freq[j] for j in range(4)]
    ra = ref_lib.Aligner(0.85, tspace, freq, 1)
    want = []
    gj = []
    for (ar, br, cm, dg, ad) in jobs:
        src = creads if cm else reads
        aoff = int(src.reads[ar + 1]["boff"]) + 1
        alen = int(src.reads[ar + 1]["rlen"])
        boff = int(ref.reads[br + 1]["boff"]) + 1
        blen = int(ref.reads[br + 1]["rlen"])
        want.append(ra.align(src.bases, aoff, alen, ref.bases, boff, blen, cm, dg, ad))
        # the kernel complements on the fly: it always gets the forward read
        gj.append((int(reads.reads[ar + 1]["boff"]) + 1, alen, boff, blen, cm, dg, ad))
    f = (C.c_float * 4)(*freq)
    spec = capi.lib().New_Align_Spec(0.85, tspace, f, 1)
    out, at, bt = capi.local_alignment(spec, reads.bases, ref.bases, gj, stride=2048)
    bad = 0
    for i, w in enumerate(want):
        o = out[i]
        assert o[7] == 0, "job %d status %d" % (i, o[7])
        ok = (tuple(int(x) for x in o[:6]) == w[0] and np.array_equal(at[i][:o[5]], w[1])
              and (int(o[8]), int(o[9]), int(o[10]), int(o[11])) == w[2][:4] and int(o[6]) == w[2][5]
              and np.array_equal(bt[i][:o[6]], w[3]))
        if not ok:
            bad += 1
            if bad < 5:
                print("MISMATCH", jobs[i], "\n ref", w[0], w[2], "\n got", o)
    assert bad == 0, "%d of %d alignments differ" % (bad, len(jobs))
    capi.lib().Free_Align_Spec(spec)


@pytest.mark.parametrize("lanes", [16, 32])
@pytest.mark.parametrize("case", CASES, ids=[str(i) for i in range(len(CASES))])
def test_lane_group_passes_match_reference(case, lanes):
    """The passes the candidate kernels run first -- four Local_Alignments per wavefront on 16-lane groups with two
    slots per lane (bands up to 14 diagonals on a 16-position ring, up to 30 on a 32-position one), or two on 32-lane
    groups -- through the stage-level API ("align_g" test option): A view (end points, diffs, trace) and the B end
    points against the reference; a band or pebble overflow falls through to the wavefront-per-job passes."""
    ref_lib = _ref()
    kind, seed, contigs, nreads, drop, tspace, lowcomplex = case
    rng = np.random.default_rng(seed)
    g, rd = make_dataset(kind, seed, contigs, nreads, dropouts=drop)
    if lowcomplex:
        for p in range(5000, len(g[0]) - 5000, 20000):
            g[0][p:p + 300] = 0
            g[0][p + 1000:p + 1400] = np.tile(np.array([0, 1], dtype=np.uint8), 200)
        rd, _ = dazzdb.simulate_reads(rng, g, nreads, dropouts=drop)
    res = run_filter(Api("ref"), rd, g, [0, len(contigs)], K=20, want_seeds=False)
    reads, ref, jobs = jobs_from_candidates(res, g, max_per_cand=8)
    jobs += random_jobs(rng, reads, ref, 200)
    creads = comp_block(reads)
    freq = [ref.db.freq[j] for j in range(4)]
    ra = ref_lib.Aligner(0.85, tspace, freq, 1)
    want = []
    gj = []
    for (ar, br, cm, dg, ad) in jobs:
        src = creads if cm else reads
        aoff = int(src.reads[ar + 1]["boff"]) + 1
        alen = int(src.reads[ar + 1]["rlen"])
        boff = int(ref.reads[br + 1]["boff"]) + 1
        blen = int(ref.reads[br + 1]["rlen"])
        want.append(ra.align(src.bases, aoff, alen, ref.bases, boff, blen, cm, dg, ad))
        gj.append((int(reads.reads[ar + 1]["boff"]) + 1, alen, boff, blen, cm, dg, ad))
    f = (C.c_float * 4)(*freq)
    L = capi.lib()
    spec = L.New_Align_Spec(0.85, tspace, f, 1)
    assert L.dmh_test_option(b"align_g", lanes) == 0
    try:
        out, at, bt = capi.local_alignment(spec, reads.bases, ref.bases, gj, stride=2048)
    finally:
        L.dmh_test_option(b"align_g", -1)
    bad = 0
    on_groups = 0
    for i, w in enumerate(want):
        o = out[i]
        assert o[7] == 0, "job %d status %d" % (i, o[7])
        ok = (tuple(int(x) for x in o[:6]) == w[0] and np.array_equal(at[i][:o[5]], w[1])
              and (int(o[8]), int(o[9]), int(o[10]), int(o[11])) == w[2][:4])
        on_groups += int(o[6] == 0)            # (the lane-group passes report no B trace)
        if not ok:
            bad += 1
            if bad < 5:
                print("MISMATCH", jobs[i], "\n ref", w[0], w[2], "\n got", o)
    assert bad == 0, "%d of %d alignments differ" % (bad, len(jobs))
    print("lanes %d: %d of %d jobs stayed on the lane groups" % (lanes, on_groups, len(jobs)))
    assert on_groups > len(jobs) // 4
    L.Free_Align_Spec(spec)
