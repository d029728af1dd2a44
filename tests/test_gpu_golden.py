"""GPU parity against the committed golden vectors (generated from the compiled reference by
tests/golden/make_golden.py): works on a box that has neither /root/reference nor oracle/_ref."""
import ctypes as C
import os

import numpy as np
import pytest

from damapper_amd import capi, dazzdb, driver, las
from util_golden import load, golden_cands
from util_pipeline import Api, run_filter

pytestmark = pytest.mark.gpu


def test_filter_stages_match_golden():
    z, genome, reads = load()
    res = run_filter(Api("hip"), reads, genome, [int(x) for x in z["firsts"]], K=int(z["K"]), mem_limit=int(z["mem"]))
    assert len(res["seeds"]) == int(z["npass"])
    for p, s in enumerate(res["seeds"]):
        got = np.stack([s["aread"], s["bread"], s["apos"], s["apos"] - s["diag"]], axis=1)
        assert np.array_equal(got, z["seeds%d" % p]), "pass %d" % p
    want = golden_cands(z)
    got = {}
    for ar, (c, js) in enumerate(res["cands"]):
        if len(c):
            got[ar] = [(tuple(int(x) for x in row), ja.tolist(), jb.tolist()) for row, (ja, jb) in zip(c, js)]
    assert got == want


def test_alignments_match_golden():
    z, genome, reads = load()
    rd = dazzdb.make_block(reads, path=b"./.READS")
    ref = dazzdb.make_block(genome, path=b"./.REF")
    jobs = []
    for (ar, br, cm, dg, ad) in z["align_jobs"]:
        jobs.append((int(rd.reads[ar + 1]["boff"]) + 1, int(rd.reads[ar + 1]["rlen"]),
                     int(ref.reads[br + 1]["boff"]) + 1, int(ref.reads[br + 1]["rlen"]), int(cm), int(dg), int(ad)))
    f = (C.c_float * 4)(*[float(x) for x in z["freq"]])
    spec = capi.lib().New_Align_Spec(0.85, 100, f, 1)
    out, at, bt = capi.local_alignment(spec, rd.bases, ref.bases, jobs, stride=1024)
    ao, bo = z["align_aoff"], z["align_boff"]
    for i, w in enumerate(z["align_paths"]):
        o = out[i]
        assert o[7] == 0
        assert [int(x) for x in o[:6]] == [int(x) for x in w[:6]], "job %d" % i
        assert [int(o[8]), int(o[9]), int(o[10]), int(o[11]), int(o[6])] == [int(w[6]), int(w[7]), int(w[8]), int(w[9]), int(w[11])]
        assert np.array_equal(at[i][:o[5]], z["align_atrace"][ao[i]:ao[i + 1]])
        assert np.array_equal(bt[i][:o[6]], z["align_btrace"][bo[i]:bo[i + 1]])
    capi.lib().Free_Align_Spec(spec)


def test_las_and_profile_match_golden(tmp_path):
    z, genome, reads = load()
    d = str(tmp_path)
    dazzdb.write_db(d, "REF", genome, block_firsts=[int(x) for x in z["firsts"]])
    dazzdb.write_db(d, "READS", reads)
    driver.map_db(d, "REF", "READS", threads=4, mem_limit=int(z["mem"]), profile=1, best_tie=0.95)
    ts, recs = las.read_las_set(os.path.join(d, "las_hip"), "READS.REF.M@.las")
    rows, to = z["las_rows"], z["las_toff"]
    assert len(recs) == len(rows)
    for i, r in enumerate(recs):
        assert [r.tlen, r.diffs, r.abpos, r.bbpos, r.aepos, r.bepos, r.flags, r.aread, r.bread] == [int(x) for x in rows[i]]
        assert r.trace == bytes(z["las_trace"][to[i]:to[i + 1]])
    assert open(os.path.join(d, ".READS.prof.anno"), "rb").read() == z["prof_anno"].tobytes()
    assert open(os.path.join(d, ".READS.prof.data"), "rb").read() == z["prof_data"].tobytes()
