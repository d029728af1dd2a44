"""Generates tests/golden/tiny.npz from the COMPILED REFERENCE (oracle/_ref, built by oracle/Makefile
from /root/reference).  Run in the build container:  python tests/golden/make_golden.py
The fixture holds inputs and the reference's outputs only (no reference code)."""
import hashlib
import io
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from damapper_amd import dazzdb, las  # noqa: E402
import ref_lib  # noqa: E402
import run_ref  # noqa: E402
from util_pipeline import Api, run_filter  # noqa: E402
from util_align import jobs_from_candidates, comp_block, random_jobs  # noqa: E402

K, MEM = 20, 8 << 30


def main():
    rng = np.random.default_rng(2024)
    genome = dazzdb.repeat_rich_genome(rng, [60000, 45000], family_len=600, tandem_unit=40, tandem_copies=12)
    reads, _ = dazzdb.simulate_reads(rng, genome, 14, read_len=4000)
    r2, _ = dazzdb.simulate_reads(rng, genome, 4, read_len=4000, dropouts=2)
    reads += r2
    firsts = [0, 1, 2]
    out = {"K": K, "mem": MEM, "firsts": np.array(firsts)}
    for i, c in enumerate(genome):
        out["contig%d" % i] = c
    out["ncontigs"] = len(genome)
    out["nreads"] = len(reads)
    for i, r in enumerate(reads):
        out["read%d" % i] = r

    res = run_filter(Api("ref"), reads, genome, firsts, K=K, mem_limit=MEM)
    # k-mer index of the reads block
    ref_lib.lib().Set_Filter_Params(K, 0, 4)
    h, n, arr = ref_lib.sort_kmers(res["reads"])
    out["kmer_n"] = n
    out["kmer_sha"] = np.frombuffer(hashlib.sha256(arr.tobytes()).digest(), dtype=np.uint8)
    out["kmer_head"] = arr[:64].copy().view(np.uint8)
    out["kmer_tail"] = arr[-64:].copy().view(np.uint8)
    for p, s in enumerate(res["seeds"]):
        out["seeds%d" % p] = np.stack([s["aread"], s["bread"], s["apos"], s["apos"] - s["diag"]], axis=1).astype(np.int32)
    out["npass"] = len(res["seeds"])
    cl, jl, jo = [], [], [0]
    for ar, (c, js) in enumerate(res["cands"]):
        for row, (ja, jb) in zip(c, js):
            cl.append([ar] + [int(x) for x in row])
            jl.extend([(int(a), int(b)) for a, b in zip(ja, jb)])
            jo.append(len(jl))
    out["cands"] = np.array(cl, dtype=np.int32)
    out["jumps"] = np.array(jl, dtype=np.int32).reshape(-1, 2)
    out["jump_off"] = np.array(jo, dtype=np.int64)

    # alignments
    rd, ref, jobs = jobs_from_candidates(res, genome, max_per_cand=5)
    jobs += random_jobs(np.random.default_rng(7), rd, ref, 40)
    creads = comp_block(rd)
    freq = [ref.db.freq[j] for j in range(4)]
    ra = ref_lib.Aligner(0.85, 100, freq, 1)
    paths, atr, ato, btr, bto = [], [], [0], [], [0]
    for (ar, br, cm, dg, ad) in jobs:
        src = creads if cm else rd
        aoff = int(src.reads[ar + 1]["boff"]) + 1
        alen = int(src.reads[ar + 1]["rlen"])
        boff = int(ref.reads[br + 1]["boff"]) + 1
        blen = int(ref.reads[br + 1]["rlen"])
        a, at, b, bt = ra.align(src.bases, aoff, alen, ref.bases, boff, blen, cm, dg, ad)
        paths.append(list(a) + list(b))
        atr.extend(at.tolist()); ato.append(len(atr))
        btr.extend(bt.tolist()); bto.append(len(btr))
    out["align_jobs"] = np.array(jobs, dtype=np.int32)
    out["align_paths"] = np.array(paths, dtype=np.int32)
    out["align_atrace"] = np.array(atr, dtype=np.uint16)
    out["align_aoff"] = np.array(ato, dtype=np.int64)
    out["align_btrace"] = np.array(btr, dtype=np.uint16)
    out["align_boff"] = np.array(bto, dtype=np.int64)
    out["freq"] = np.array(freq, dtype=np.float32)

    # end-to-end .las (+ -p track) from the reference binary
    with tempfile.TemporaryDirectory() as d:
        dazzdb.write_db(d, "REF", genome, block_firsts=firsts)
        dazzdb.write_db(d, "READS", reads)
        run_ref.run_reference(d, "REF", ["READS"], flags=["-M8", "-p", "-n.95"], threads=4)
        ts, recs = las.read_las_set(os.path.join(d, "las_ref"), "READS.REF.M@.las")
        rows, tr, to = [], [], [0]
        for r in recs:
            rows.append([r.tlen, r.diffs, r.abpos, r.bbpos, r.aepos, r.bepos, r.flags, r.aread, r.bread])
            tr.extend(list(r.trace)); to.append(len(tr))
        out["las_rows"] = np.array(rows, dtype=np.int32)
        out["las_trace"] = np.array(tr, dtype=np.uint8)
        out["las_toff"] = np.array(to, dtype=np.int64)
        out["prof_anno"] = np.frombuffer(open(os.path.join(d, ".READS.prof.anno"), "rb").read(), dtype=np.uint8)
        out["prof_data"] = np.frombuffer(open(os.path.join(d, ".READS.prof.data"), "rb").read(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "tiny.npz"), **out)
    print("wrote tiny.npz:", os.path.getsize(os.path.join(HERE, "tiny.npz")), "bytes;", len(cl), "candidates,",
          len(jobs), "alignments,", len(rows), "las records")


if __name__ == "__main__":
    main()
