"""Alignment job generation shared by the oracle (CPU) and GPU alignment tests: every retained
chain point of every candidate becomes a Local_Alignment seed exactly as report_thread forms it
(map.c:2487-2513), without the alast skipping, so that short/odd seeds are covered too."""
import numpy as np

from damapper_amd import dazzdb


def whole_ref_block(genome):
    return dazzdb.make_block(genome, path=b"./.REF")


def jobs_from_candidates(res, genome, max_per_cand=6):
    """res: output of util_pipeline.run_filter.  Returns (reads_block, ref_block, jobs) with jobs =
    list of (aread, bread, comp, dg, ad)."""
    reads = res["reads"]
    ref = whole_ref_block(genome)
    jobs = []
    for ar, (cands, jumps) in enumerate(res["cands"]):
        alen = int(reads.reads[ar + 1]["rlen"])
        for c, (ja, jb) in zip(cands, jumps):
            score, length, br, cm, af, al, bf, bl = [int(x) for x in c]
            blen = int(ref.reads[br + 1]["rlen"])
            apos, bpos = al, bl
            n = len(ja)
            pick = sorted(set(np.linspace(0, n - 1, min(n, max_per_cand)).astype(int).tolist())) if n else []
            for j in range(n):
                apos -= int(ja[j])
                bpos -= int(jb[j])
                if j not in pick:
                    continue
                if cm:
                    ac, bc = alen - apos, blen - bpos
                    dg, ad = ac - bc, ac + bc
                else:
                    dg, ad = apos - bpos, apos + bpos
                jobs.append((ar, br, cm, dg, ad))
    return reads, ref, jobs


def comp_block(reads):
    """Reverse-complemented copy of a reads block (what report_thread builds per read, map.c:1940-1948)."""
    seqs = [(3 - reads.seq(i)[::-1]).astype(np.uint8) for i in range(reads.nreads)]
    return dazzdb.make_block(seqs, path=b"./.CREADS")


def random_jobs(rng, reads, ref, n):
    """Seeds that mostly do NOT lie on a real alignment: exercises the short/both-short paths and
    boundary handling."""
    jobs = []
    for _ in range(n):
        ar = int(rng.integers(0, reads.nreads))
        br = int(rng.integers(0, ref.nreads))
        alen = int(reads.reads[ar + 1]["rlen"])
        blen = int(ref.reads[br + 1]["rlen"])
        cm = int(rng.integers(0, 2))
        m = rng.integers(0, 4)
        if m == 0:      # near the start corner
            a, b = int(rng.integers(0, 40)), int(rng.integers(0, 40))
        elif m == 1:    # near the end corner
            a, b = alen - int(rng.integers(0, 40)), blen - int(rng.integers(0, 40))
        else:
            a, b = int(rng.integers(0, alen + 1)), int(rng.integers(0, blen + 1))
        jobs.append((ar, br, cm, a - b, a + b))
    return jobs
