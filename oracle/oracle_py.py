"""TEST INFRASTRUCTURE ONLY: numpy restatements of the integer/byte stages of the hot path
(each function cites the reference lines it follows) + ctypes access to oracle/liboracle.so.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAXGRAM = 10000          # map.c:32
MIN_PIECE = 300          # map.c:37


def lib():
    so = os.path.join(HERE, "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", HERE, "oracle"])
    L = C.CDLL(so)
    L.oracle_local_alignment.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
    L.oracle_align_spec.argtypes = [C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_chain_group.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.oracle_chain_group.restype = C.c_int
    return L


def sort_kmers(block, K, suppress=0):
    """map.c:545-576 (rolling 2-bit code, rpos = last base) + stable sort on the code
    (map.c:181-444) + -t compaction (map.c:590-636): order (code, read, rpos)."""
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
    if not codes:
        z = np.zeros(0, dtype=np.uint64)
        return z, np.zeros(0, np.int32), np.zeros(0, np.int32)
    c = np.concatenate(codes)
    r = np.concatenate(reads)
    p = np.concatenate(rpos)
    o = np.argsort(c, kind="stable")
    c, r, p = c[o], r[o], p[o]
    if suppress > 0:
        _, inv, cnt = np.unique(c, return_inverse=True, return_counts=True)
        keep = cnt[inv] < suppress
        c, r, p = c[keep], r[keep], p[keep]
    return c, r, p


def sizeof_db(block):
    """DB.c:1044-1076 for a block without tracks."""
    db = block.db
    return 112 + 40 * (db.nreads + 2) + len(db.path) + 1 + (db.totlen + db.nreads + 4)


def seed_hits(a_idx, b_idx, ablock, bblock, mem_limit):
    """count_thread/merge_thread (map.c:881-1002) with the limit rule (map.c:2992-3055) and the
    seed sort (map.c:3110-3126).  Returns (limit, hits[n,4] = aread,bread,apos,bpos sorted)."""
    ac, ar, ap = a_idx
    bc, br, bp = b_idx
    ua, ia, ca = np.unique(ac, return_index=True, return_counts=True)
    ub, ib, cb = np.unique(bc, return_index=True, return_counts=True)
    common, xa, xb = np.intersect1d(ua, ub, assume_unique=True, return_indices=True)
    prod = ca[xa].astype(np.int64) * cb[xb].astype(np.int64)
    histo = np.bincount(prod[prod < MAXGRAM], minlength=MAXGRAM).astype(np.int64)
    alen, blen = len(ac), len(bc)
    if mem_limit > 0:
        avail = int((mem_limit - (sizeof_db(ablock) + sizeof_db(bblock))) // 16)
        if avail > alen + 2 * blen:
            avail = (avail - alen) // 2
        else:
            avail = avail - (alen + blen)
        avail = int(avail * .98)
        tom = 0
        limit = MAXGRAM
        for j in range(MAXGRAM):
            tom += j * int(histo[j])
            if tom > avail:
                limit = j
                break
    else:
        limit = 0x7fffffff
    out = []
    for g in np.nonzero(prod < limit)[0]:
        a0, na = ia[xa[g]], ca[xa[g]]
        b0, nb = ib[xb[g]], cb[xb[g]]
        for a in range(a0, a0 + na):
            for b in range(b0, b0 + nb):
                out.append((ar[a], br[b], ap[a], bp[b]))
    h = np.array(out, dtype=np.int64).reshape(-1, 4)
    if len(h):
        o = np.lexsort((h[:, 3], h[:, 2], h[:, 1], h[:, 0]))
        h = h[o]
    return limit, h


def chain_group(apos, bpos, K):
    """One (read, contig) group through oracle_chain.c.  Returns (chains[m,6], from[n])."""
    L = lib()
    a = np.ascontiguousarray(apos, dtype=np.int32)
    b = np.ascontiguousarray(bpos, dtype=np.int32)
    n = len(a)
    out = np.zeros((max(1, n), 6), dtype=np.int32)
    frm = np.zeros(max(1, n), dtype=np.int32)
    m = L.oracle_chain_group(a.ctypes.data, b.ctypes.data, n, K, out.ctypes.data, max(1, n), frm.ctypes.data)
    return out[:m], frm[:n]


def jumps_of(apos, bpos, frm, end):
    """chain_length + Jump encoding (map.c:1243-1260, 1741-1759)."""
    ja, jb = [], []
    x = end
    y = frm[x]
    while y >= 0:
        da = int(apos[x] - apos[y])
        db = int(bpos[x] - bpos[y])
        if da == db and da < 100:
            y = frm[y]
        else:
            ja.append(da & 0xffff)
            jb.append(db & 0xffff)
            x = y
            y = frm[x]
    return ja, jb


def filter_candidates(lists, hits, K, comp, bstart):
    """chain_thread's per-read bookkeeping (map.c:1606-1777): for every (aread,bread) group of the
    sorted hits score the chains, then apply the domination rule to the read's list (newest first).
    `lists` maps aread -> list of dicts and is updated in place."""
    if len(hits) == 0:
        return
    key = hits[:, 0] * (1 << 32) + hits[:, 1]
    bounds = np.nonzero(np.diff(key))[0] + 1
    starts = np.concatenate([[0], bounds])
    ends = np.concatenate([bounds, [len(hits)]])
    for s, e in zip(starts, ends):
        ar, br = int(hits[s, 0]), int(hits[s, 1])
        ap, bp = hits[s:e, 2], hits[s:e, 3]
        chains, frm = chain_group(ap, bp, K)
        L = lists.setdefault(ar, [])
        for (cost, ab, ae, bb, be, end) in chains:
            dropped = False
            d = 0
            while d < len(L):
                cd = L[d]
                d_has_c = cd["afirst"] < ab + MIN_PIECE and cd["alast"] > ae - MIN_PIECE
                c_has_d = ab < cd["afirst"] + MIN_PIECE and ae > cd["alast"] - MIN_PIECE
                if d_has_c and .9 * cd["score"] >= cost:
                    dropped = True
                    break
                if c_has_d and cd["score"] <= .9 * cost:
                    del L[d]
                    continue
                d += 1
            if dropped:
                continue
            ja, jb = jumps_of(ap + 1, bp + 1, frm, int(end))
            L.insert(0, dict(score=int(cost), length=len(ja), bread=br + bstart, comp=comp, afirst=int(ab),
                             alast=int(ae), bfirst=int(bb), blast=int(be), adisp=ja, bdisp=jb))
