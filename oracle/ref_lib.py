"""TEST INFRASTRUCTURE: ctypes access to the compiled reference (oracle/_ref/libdamapper_ref.so,
built by oracle/Makefile from the sources under /root/reference).  Only tests, smoke() and
bench.py's cpu_baseline leg may import this."""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from damapper_amd.dazzdb import DAZZ_DB, Block  # noqa: E402

LIB_PATH = os.path.join(HERE, "_ref", "libdamapper_ref.so")

KMERPOS = np.dtype([("code", "<u8"), ("rpos", "<i4"), ("read", "<i4")])
SEEDPAIR = np.dtype([("diag", "<i4"), ("apos", "<i4"), ("bread", "<i4"), ("aread", "<i4")])


class Path(C.Structure):          # align.h:89-95
    _fields_ = [("trace", C.c_void_p), ("tlen", C.c_int), ("diffs", C.c_int),
                ("abpos", C.c_int), ("bbpos", C.c_int), ("aepos", C.c_int), ("bepos", C.c_int)]


class Alignment(C.Structure):     # align.h:145-152
    _fields_ = [("path", C.POINTER(Path)), ("flags", C.c_uint32), ("aseq", C.c_void_p),
                ("bseq", C.c_void_p), ("alen", C.c_int), ("blen", C.c_int)]


_lib: Optional[C.CDLL] = None


def available() -> bool:
    return os.path.exists(LIB_PATH)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)
    P = C.POINTER
    L.Set_Filter_Params.argtypes = [C.c_int, C.c_int, C.c_int]
    L.Sort_Kmers.argtypes = [P(DAZZ_DB), P(C.c_int)]
    L.Sort_Kmers.restype = C.c_void_p
    L.Match_Filter.argtypes = [P(DAZZ_DB), P(DAZZ_DB), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.Match_Filter.restype = None
    L.Reporter.argtypes = [C.c_char_p, P(DAZZ_DB), C.c_char_p, P(DAZZ_DB), C.c_void_p, C.c_int]
    L.Reporter.restype = None
    L.New_Align_Spec.argtypes = [C.c_double, C.c_int, P(C.c_float), C.c_int]
    L.New_Align_Spec.restype = C.c_void_p
    L.New_Work_Data.restype = C.c_void_p
    L.Free_Work_Data.argtypes = [C.c_void_p]
    L.Local_Alignment.argtypes = [P(Alignment), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.Local_Alignment.restype = P(Path)
    L.ref_set_globals.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_char_p, C.c_uint64, C.c_uint64]
    L.ref_last_seeds.argtypes = [P(C.c_int64)]
    L.ref_last_seeds.restype = C.c_void_p
    L.ref_free.argtypes = [C.c_void_p]
    L.ref_read_candidates.argtypes = [P(DAZZ_DB), C.c_int, C.c_void_p, C.c_int]
    L.ref_read_candidates.restype = C.c_int
    L.ref_candidate_jumps.argtypes = [P(DAZZ_DB), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.ref_candidate_jumps.restype = C.c_int
    L.ref_read_cover.argtypes = [P(DAZZ_DB), C.c_int, C.c_void_p, C.c_int]
    L.ref_read_cover.restype = C.c_int
    L.ref_cover.argtypes = [C.c_int, P(C.c_int)]
    L.ref_cover.restype = C.c_void_p
    L.ref_nthreads.restype = C.c_int
    _lib = L
    return L


def set_globals(verbose=0, profile=0, spacing=100, best_tie=1.0, sort_path="/tmp", mem_limit=0, mem_physical=0):
    lib().ref_set_globals(verbose, profile, spacing, best_tie, sort_path.encode(), mem_limit, mem_physical)


def sort_kmers(block: Block, copy: bool = True) -> Tuple[Optional[int], int, Optional[np.ndarray]]:
    """Returns (handle, len, numpy copy of the KmerPos records)."""
    n = C.c_int(0)
    h = lib().Sort_Kmers(block.ptr(), C.byref(n))
    arr = None
    if h and copy:
        arr = np.ctypeslib.as_array(C.cast(h, C.POINTER(C.c_uint8)), shape=(n.value * 16,)).view(KMERPOS).copy()
    return h, n.value, arr


def match_filter(reads: Block, ref: Block, rtab, rlen, ftab, flen, comp, start):
    lib().Match_Filter(reads.ptr(), ref.ptr(), rtab, rlen, ftab, flen, comp, start)


def last_seeds() -> np.ndarray:
    n = C.c_int64(0)
    p = lib().ref_last_seeds(C.byref(n))
    if n.value == 0:
        return np.zeros(0, dtype=SEEDPAIR)
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value * 16,)).view(SEEDPAIR).copy()


def read_candidates(reads: Block, ar: int, maxc: int = 4096) -> np.ndarray:
    out = np.zeros((maxc, 9), dtype=np.int32)
    n = lib().ref_read_candidates(reads.ptr(), ar, out.ctypes.data, maxc)
    return out[:max(n, 0)]


def candidate_jumps(reads: Block, ar: int, k: int, maxj: int = 65536):
    a = np.zeros(maxj, dtype=np.uint16)
    b = np.zeros(maxj, dtype=np.uint16)
    n = lib().ref_candidate_jumps(reads.ptr(), ar, k, a.ctypes.data, b.ctypes.data, maxj)
    return a[:max(n, 0)], b[:max(n, 0)]


def read_cover(reads: Block, ar: int, maxn: int = 4096) -> np.ndarray:
    """-p coverage deltas of one read as chain_thread leaves them (before Reporter's special_log pass)."""
    out = np.zeros(maxn, dtype=np.int16)
    n = lib().ref_read_cover(reads.ptr(), ar, out.ctypes.data, maxn)
    return out[:max(n, 0)]


def free(h):
    if h:
        lib().ref_free(h)


class Aligner:
    """Direct Local_Alignment calls (align.c:1727-1946) as report_thread issues them
    (map.c:2474-2513)."""

    def __init__(self, ave_corr: float, tspace: int, freq, reach: int = 1):
        f = (C.c_float * 4)(*[float(x) for x in freq])
        self.spec = lib().New_Align_Spec(ave_corr, tspace, f, reach)
        self.work = lib().New_Work_Data()

    def align(self, abuf: np.ndarray, aoff: int, alen: int, bbuf: np.ndarray, boff: int, blen: int,
              comp: int, dg: int, ad: int):
        """abuf/bbuf: uint8 arrays with a 4 before and after each sequence; aoff/boff index the
        first base.  For comp != 0 the caller passes the already reverse-complemented A."""
        path = Path()
        al = Alignment()
        al.path = C.pointer(path)
        al.flags = 0x2 if comp else 0          # ACOMP_FLAG, align.h:128
        al.aseq = abuf.ctypes.data + aoff
        al.bseq = bbuf.ctypes.data + boff
        al.alen = alen
        al.blen = blen
        bp = lib().Local_Alignment(C.byref(al), self.work, self.spec, dg, dg, ad, -1, -1)
        at = np.ctypeslib.as_array(C.cast(path.trace, C.POINTER(C.c_uint16)), shape=(max(path.tlen, 1),))[:path.tlen].copy()
        b = bp.contents
        bt = np.ctypeslib.as_array(C.cast(b.trace, C.POINTER(C.c_uint16)), shape=(max(b.tlen, 1),))[:b.tlen].copy()
        return ((path.abpos, path.bbpos, path.aepos, path.bepos, path.diffs, path.tlen), at,
                (b.abpos, b.bbpos, b.aepos, b.bepos, b.diffs, b.tlen), bt)
