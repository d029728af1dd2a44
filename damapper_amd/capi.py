"""ctypes binding of libdamapper_hip.so (include/damapper_hip.h).  This is plumbing: it mirrors
the reference's filter interface (map.h:26-37) one to one and adds the stage-level `dmh_*`
entry points.  There is no CPU fallback: loading fails loudly if the HIP library is missing."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

import numpy as np

from .dazzdb import DAZZ_DB, Block

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DMH_LIB") or os.path.join(_HERE, "libdamapper_hip.so")      # (DMH_LIB: an instrumented build)

KMERPOS = np.dtype([("code", "<u8"), ("rpos", "<i4"), ("read", "<i4")])               # map.c:78-82
SEEDPAIR = np.dtype([("diag", "<i4"), ("apos", "<i4"), ("bread", "<i4"), ("aread", "<i4")])  # map.c:84-89

FLAG_DOA = 0x1
FLAG_DOB = 0x2

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libdamapper_hip.so is not built (run __graft_entry__.build()); "
                           "damapper_amd has no CPU fallback")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    P = C.POINTER
    L.Set_Filter_Params.argtypes = [C.c_int, C.c_int, C.c_int]
    L.Set_Filter_Params.restype = C.c_int
    L.Sort_Kmers.argtypes = [P(DAZZ_DB), P(C.c_int)]
    L.Sort_Kmers.restype = C.c_void_p
    L.dmh_ref_cache.argtypes = [C.c_int, C.c_double]
    L.dmh_ref_cache.restype = None
    L.dmh_test_option.argtypes = [C.c_char_p, C.c_int]
    L.dmh_test_option.restype = C.c_int
    L.dmh_next_pass.argtypes = [C.c_void_p, C.c_void_p]
    L.dmh_next_pass.restype = None
    L.dmh_sort_kmers_async.argtypes = [P(DAZZ_DB), P(C.c_int)]
    L.dmh_sort_kmers_async.restype = C.c_void_p
    L.Match_Filter.argtypes = [P(DAZZ_DB), P(DAZZ_DB), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.Match_Filter.restype = None
    L.Reporter.argtypes = [C.c_char_p, P(DAZZ_DB), C.c_char_p, P(DAZZ_DB), C.c_void_p, C.c_int]
    L.Reporter.restype = None
    L.New_Align_Spec.argtypes = [C.c_double, C.c_int, P(C.c_float), C.c_int]
    L.New_Align_Spec.restype = C.c_void_p
    L.Free_Align_Spec.argtypes = [C.c_void_p]
    L.Free_Align_Spec.restype = None
    L.Trace_Spacing.argtypes = [C.c_void_p]
    L.Trace_Spacing.restype = C.c_int
    L.dmh_device_count.restype = C.c_int
    L.dmh_last_error.restype = C.c_char_p
    L.dmh_preload_block.argtypes = [P(DAZZ_DB)]
    L.dmh_device_memory.argtypes = [P(C.c_ulonglong), P(C.c_ulonglong)]
    L.dmh_device_memory.restype = C.c_int
    L.dmh_prepare_block.argtypes = [P(DAZZ_DB), C.c_int]
    L.dmh_prepare_block.restype = C.c_int
    L.dmh_preload_block_2bit.argtypes = [P(DAZZ_DB), C.c_void_p, C.c_int64, C.c_void_p]
    L.dmh_preload_block_2bit.restype = C.c_int
    L.dmh_download_block.argtypes = [P(DAZZ_DB), C.c_void_p]
    L.dmh_download_block.restype = C.c_int
    L.dmh_release_block.argtypes = [P(DAZZ_DB)]
    L.dmh_retain_ref_index.argtypes = [C.c_int]
    L.dmh_free_index.argtypes = [C.c_void_p]
    L.dmh_index_download.argtypes = [C.c_void_p, C.c_void_p]
    L.dmh_index_download.restype = C.c_int
    L.dmh_capture_seeds.argtypes = [C.c_int]
    L.dmh_last_seed_count.restype = C.c_int64
    L.dmh_last_seeds.argtypes = [C.c_void_p]
    L.dmh_last_seeds.restype = C.c_int
    L.dmh_last_limit.restype = C.c_int
    L.dmh_read_candidates.argtypes = [P(DAZZ_DB), C.c_int, C.c_void_p, C.c_int]
    L.dmh_read_candidates.restype = C.c_int
    L.dmh_candidate_jumps.argtypes = [P(DAZZ_DB), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.dmh_candidate_jumps.restype = C.c_int
    L.dmh_read_cover.argtypes = [P(DAZZ_DB), C.c_int, C.c_void_p, C.c_int]
    L.dmh_read_cover.restype = C.c_int
    L.dmh_sort_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
    L.dmh_sort_pairs.restype = C.c_int
    L.dmh_exclusive_scan.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.dmh_exclusive_scan.restype = C.c_int
    L.dmh_local_alignment.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.dmh_local_alignment.restype = C.c_int
    L.dmh_timing_enable.argtypes = [C.c_int]
    L.dmh_async_report.argtypes = [C.c_int]
    L.dmh_timing_count.restype = C.c_int
    L.dmh_timing_get.argtypes = [C.c_int, C.c_char_p, C.c_int, P(C.c_double), P(C.c_int64), P(C.c_double)]
    L.dmh_timing_get.restype = C.c_int
    _lib = L
    return L


def exported_symbols() -> List[str]:
    """Every symbol include/damapper_hip.h declares (checked by the CPU test-suite)."""
    return ["Set_Filter_Params", "Sort_Kmers", "Match_Filter", "Reporter", "Clean_Exit",
            "New_Align_Spec", "Free_Align_Spec", "Trace_Spacing", "Average_Correlation",
            "Base_Frequencies", "Overlap_If_Possible",
            "VERBOSE", "PROFILE", "SPACING", "BEST_TIE", "SORT_PATH", "MEM_LIMIT", "MEM_PHYSICAL",
            "dmh_device_count", "dmh_last_error", "dmh_reset", "dmh_preload_block", "dmh_release_block",
            "dmh_retain_ref_index", "dmh_free_index", "dmh_index_download", "dmh_capture_seeds",
            "dmh_last_seed_count", "dmh_last_seeds", "dmh_last_limit", "dmh_read_candidates",
            "dmh_candidate_jumps", "dmh_read_cover", "dmh_sort_pairs", "dmh_exclusive_scan", "dmh_local_alignment",
            "dmh_timing_enable", "dmh_timing_reset", "dmh_timing_count", "dmh_timing_get", "dmh_sync",
            "dmh_async_report", "dmh_report_wait", "dmh_reports_done", "dmh_sort_kmers_async", "dmh_host_alloc", "dmh_host_free", "dmh_test_option", "dmh_next_pass", "dmh_ref_cache",
            "dmh_preload_block_2bit", "dmh_download_block", "dmh_prepare_block", "dmh_device_memory"]


def set_globals(verbose: int = 0, profile: int = 0, spacing: int = 100, best_tie: float = 1.0,
                sort_path: str = "/tmp", mem_limit: int = 0, mem_physical: int = 0) -> None:
    """The globals damapper.c owns (damapper.c:58-65)."""
    L = lib()
    C.c_int.in_dll(L, "VERBOSE").value = verbose
    C.c_int.in_dll(L, "PROFILE").value = profile
    C.c_int.in_dll(L, "SPACING").value = spacing
    C.c_double.in_dll(L, "BEST_TIE").value = best_tie
    buf = C.create_string_buffer(sort_path.encode())
    _keep.append(buf)
    C.c_void_p.in_dll(L, "SORT_PATH").value = C.addressof(buf)
    C.c_uint64.in_dll(L, "MEM_LIMIT").value = mem_limit
    C.c_uint64.in_dll(L, "MEM_PHYSICAL").value = mem_physical


_keep: list = []


def sort_kmers(block: Block, deferred: bool = False) -> Tuple[Optional[int], int]:
    """Sort_Kmers; deferred=True only queues the k-mer emit, the radix passes follow in the gaps of the
    work on the current READS block (dmh_sort_kmers_async)."""
    n = C.c_int(0)
    f = lib().dmh_sort_kmers_async if deferred else lib().Sort_Kmers
    h = f(block.ptr(), C.byref(n))
    return h, n.value


def index_download(handle, n: int) -> np.ndarray:
    out = np.zeros(n, dtype=KMERPOS)
    if n:
        got = lib().dmh_index_download(handle, out.ctypes.data)
        assert got == n
    return out


def match_filter(reads: Block, ref: Block, rtab, rlen: int, ftab, flen: int, comp: int, start: int) -> None:
    lib().Match_Filter(reads.ptr(), ref.ptr(), rtab, rlen, ftab, flen, comp, start)


def last_seeds() -> np.ndarray:
    n = lib().dmh_last_seed_count()
    out = np.zeros(n, dtype=SEEDPAIR)
    if n:
        lib().dmh_last_seeds(out.ctypes.data)
    return out


def read_candidates(reads: Block, ar: int, maxc: int = 4096) -> np.ndarray:
    out = np.zeros((maxc, 9), dtype=np.int32)
    n = lib().dmh_read_candidates(reads.ptr(), ar, out.ctypes.data, maxc)
    return out[:max(n, 0)]


def candidate_jumps(reads: Block, ar: int, k: int, maxj: int = 65536) -> Tuple[np.ndarray, np.ndarray]:
    a = np.zeros(maxj, dtype=np.uint16)
    b = np.zeros(maxj, dtype=np.uint16)
    n = lib().dmh_candidate_jumps(reads.ptr(), ar, k, a.ctypes.data, b.ctypes.data, maxj)
    return a[:max(n, 0)], b[:max(n, 0)]


def read_cover(reads: Block, ar: int, maxn: int = 4096) -> np.ndarray:
    """-p coverage deltas of one read (map.c:1654-1666), (rlen-1)/SPACING+2 int16 values."""
    out = np.zeros(maxn, dtype=np.int16)
    n = lib().dmh_read_cover(reads.ptr(), ar, out.ctypes.data, maxn)
    return out[:max(n, 0)]


def local_alignment(spec, abuf: np.ndarray, bbuf: np.ndarray, jobs, stride: int = 1024, want_b: bool = True):
    """jobs: list of (aoff, alen, boff, blen, comp, dg, ad) with offsets into abuf/bbuf (uint8, 4-delimited).
    Returns (out[n,12], atrace[n,stride], btrace[n,stride])."""
    n = len(jobs)
    J = np.asarray(jobs, dtype=np.int64).reshape(n, 7)
    aoff = np.ascontiguousarray(J[:, 0]); alen = np.ascontiguousarray(J[:, 1].astype(np.int32))
    boff = np.ascontiguousarray(J[:, 2]); blen = np.ascontiguousarray(J[:, 3].astype(np.int32))
    comp = np.ascontiguousarray(J[:, 4].astype(np.int32))
    dg = np.ascontiguousarray(J[:, 5].astype(np.int32)); ad = np.ascontiguousarray(J[:, 6].astype(np.int32))
    out = np.zeros((n, 16), dtype=np.int32)
    at = np.zeros((n, stride), dtype=np.uint16)
    bt = np.zeros((n, stride), dtype=np.uint16)
    lib().dmh_local_alignment(spec, abuf.ctypes.data, abuf.nbytes, bbuf.ctypes.data, bbuf.nbytes, n,
                              aoff.ctypes.data, alen.ctypes.data, boff.ctypes.data, blen.ctypes.data,
                              comp.ctypes.data, dg.ctypes.data, ad.ctypes.data, out.ctypes.data,
                              at.ctypes.data, bt.ctypes.data if want_b else None, stride)
    return out, at, bt


def sort_pairs(keys: np.ndarray, vals: np.ndarray, nbits: int) -> None:
    assert keys.dtype == np.uint64 and vals.dtype == np.uint32 and len(keys) == len(vals)
    lib().dmh_sort_pairs(keys.ctypes.data, vals.ctypes.data, len(keys), nbits)


def timings() -> dict:
    L = lib()
    out = {}
    name = C.create_string_buffer(128)
    for i in range(L.dmh_timing_count()):
        ms, ln, by = C.c_double(), C.c_int64(), C.c_double()
        L.dmh_timing_get(i, name, 128, C.byref(ms), C.byref(ln), C.byref(by))
        out[name.value.decode()] = {"ms": ms.value, "launches": ln.value, "bytes": by.value}
    return out
