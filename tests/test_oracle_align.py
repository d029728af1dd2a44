"""CPU: pins the oracle's wave-parallel restatement of Local_Alignment (oracle/oracle_align.c)
against the compiled reference (oracle/_ref) call by call: end points, diffs, and every trace
value of both the A view and the B view."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from util_pipeline import Api, make_dataset, run_filter
from util_align import jobs_from_candidates, comp_block, random_jobs

ORACLE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


def _ref():
    import ref_lib
    if not ref_lib.available():
        pytest.skip("oracle/_ref not built")
    return ref_lib


def oracle_lib():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"])
    L = C.CDLL(so)
    L.oracle_local_alignment.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
    L.oracle_align_spec.argtypes = [C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


class OracleAligner:
    def __init__(self, ave_corr, tspace, freq, reach=1):
        self.L = oracle_lib()
        self.tspace = tspace
        self.reach = reach
        self.score = np.zeros(32768, dtype=np.int16)
        self.table = np.zeros(32768, dtype=np.int16)
        ap = C.c_int(0)
        f = np.asarray(freq, dtype=np.float32)
        self.L.oracle_align_spec(ave_corr, f.ctypes.data, C.byref(ap), self.score.ctypes.data, self.table.ctypes.data)
        self.ave_path = ap.value

    def align(self, abuf, aoff, alen, bbuf, boff, blen, comp, dg, ad):
        ap = np.zeros(6, dtype=np.int32)
        bp = np.zeros(6, dtype=np.int32)
        n = 2 * (max(alen, blen) // self.tspace + 2) * 2 + 16
        at = np.zeros(n, dtype=np.uint16)
        bt = np.zeros(n, dtype=np.uint16)
        self.L.oracle_local_alignment(abuf.ctypes.data + aoff, alen, bbuf.ctypes.data + boff, blen, comp,
                                      self.tspace, self.ave_path, self.reach, self.score.ctypes.data,
                                      self.table.ctypes.data, dg, ad, ap.ctypes.data, at.ctypes.data,
                                      bp.ctypes.data, bt.ctypes.data)
        return tuple(int(x) for x in ap), at[:ap[5]].copy(), tuple(int(x) for x in bp), bt[:bp[5]].copy()


def run_jobs(aligner, reads, creads, ref, jobs):
    out = []
    for (ar, br, cm, dg, ad) in jobs:
        src = creads if cm else reads
        aoff = int(src.reads[ar + 1]["boff"]) + 1
        alen = int(src.reads[ar + 1]["rlen"])
        boff = int(ref.reads[br + 1]["boff"]) + 1
        blen = int(ref.reads[br + 1]["rlen"])
        out.append(aligner.align(src.bases, aoff, alen, ref.bases, boff, blen, cm, dg, ad))
    return out


CASES = [("random", 11, [200000, 150000], 40, 0, 100), ("repeat", 12, [250000], 30, 2, 100),
         ("random", 13, [120000, 100000], 25, 1, 126), ("repeat", 14, [150000, 90000], 20, 3, 50)]


@pytest.mark.parametrize("case", CASES, ids=[str(i) for i in range(len(CASES))])
def test_oracle_alignment_matches_reference(case):
    ref_lib = _ref()
    kind, seed, contigs, nreads, drop, tspace = case
    g, rd = make_dataset(kind, seed, contigs, nreads, dropouts=drop)
    res = run_filter(Api("ref"), rd, g, [0, len(contigs)], K=20, want_seeds=False)
    reads, ref, jobs = jobs_from_candidates(res, g)
    rng = np.random.default_rng(seed)
    jobs += random_jobs(rng, reads, ref, 150)
    creads = comp_block(reads)
    freq = [ref.db.freq[j] for j in range(4)]
    ra = ref_lib.Aligner(0.85, tspace, freq, 1)
    oa = OracleAligner(0.85, tspace, freq, 1)
    want = run_jobs(ra, reads, creads, ref, jobs)
    got = run_jobs(oa, reads, creads, ref, jobs)
    assert len(jobs) > 100
    bad = 0
    for j, (w, g_) in zip(jobs, zip(want, got)):
        ok = (w[0] == g_[0] and np.array_equal(w[1], g_[1]) and w[2] == g_[2] and np.array_equal(w[3], g_[3]))
        if not ok:
            bad += 1
            if bad < 4:
                print("MISMATCH job", j, "\n ref", w[0], w[2], "\n got", g_[0], g_[2])
    assert bad == 0, "%d of %d alignments differ" % (bad, len(jobs))
