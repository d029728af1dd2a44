"""Host-side mirror of damapper.c's main() block loops (damapper.c:825-880) over the C-ABI:
for every READS block: index it, run Match_Filter against every REF block in both orientations,
then Reporter against the whole REF.  Pure plumbing -- all compute is in libdamapper_hip.so."""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import capi
from .dazzdb import Block, load_db, read_stub


class Mapper:
    def __init__(self, kmer: int = 20, suppress: int = 0, threads: int = 4, ave_corr: float = 0.85,
                 spacing: int = 100, best_tie: float = 1.0, profile: int = 0, verbose: int = 0,
                 mem_limit: int = 0, sort_path: str = "/tmp", retain_ref: bool = False,
                 prefetch_deferred: bool = True):
        self.L = capi.lib()
        self.kmer, self.suppress, self.threads = kmer, suppress, threads
        self.ave_corr, self.spacing = ave_corr, spacing
        self.sort_path = sort_path
        os.makedirs(sort_path, exist_ok=True)
        if mem_limit == 0:
            try:
                mem_limit = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")   # getMemorySize()
            except (ValueError, OSError):
                mem_limit = 64 << 30
        capi.set_globals(verbose=verbose, profile=profile, spacing=spacing, best_tie=best_tie,
                         sort_path=sort_path, mem_limit=mem_limit, mem_physical=mem_limit)
        if self.L.Set_Filter_Params(kmer, suppress, threads):
            raise ValueError("Illegal combination of filter parameters")
        self.spec = None
        # north_star: "reference k-mer index replicated per GPU": build both orientations of every REF
        # block once and keep them in HBM for all read blocks (Match_Filter then does not consume them)
        self.retain_ref = retain_ref
        self.prefetch_deferred = prefetch_deferred
        self._ref_cache = None
        self.L.dmh_retain_ref_index(1 if retain_ref else 0)
        # overlap mode of map_block: the block whose Reporter call is still to come, and the number of passes of the
        # following block that are queued ahead of it
        self._pending = None
        self.overlap_passes = int(os.environ.get("DMH_OVERLAP_PASSES", "10"))
        self.prefetch_overlap = os.environ.get("DMH_PREFETCH_OVERLAP", "1") != "0"
        # dmh_next_pass: the library is told which (READS index, REF index) pair the NEXT Match_Filter call will bring, and
        # -- when that is the complement of the REF block at hand, as in damapper.c:851,861 -- probes both orientations'
        # indices in ONE sweep of the READ keys (results unchanged; DMH_HINT_PASSES=0: every pass sweeps for itself)
        self.hint_passes = os.environ.get("DMH_HINT_PASSES", "1") != "0"

    def release_ref(self) -> None:
        self.finish()
        self._drop_prefetched()
        if self._ref_cache:
            for (h, n, comp, k) in self._ref_cache:
                if h:
                    self.L.dmh_free_index(h)
        self._ref_cache = None

    def add_ref_block(self, blk: Block, k: int, drop_bases: bool = False) -> None:
        """Both orientations of REF block k, built now and kept in HBM.  The device copy of the block's bases is
        released as soon as its two indices exist (Reporter reads the whole-reference copy); with `drop_bases`
        the host copy goes too -- Match_Filter only reads the block's read records."""
        if self._ref_cache is None:
            self._ref_cache = []
        h0, n0 = capi.sort_kmers(blk)
        blk.complement_inplace()
        h1, n1 = capi.sort_kmers(blk)
        blk.complement_inplace()
        self.L.dmh_sync()
        self.L.dmh_release_block(blk.ptr())
        if drop_bases:
            blk.bases = None
            blk.db.bases = None
        self._ref_cache.append((h0, n0, 0, k))
        self._ref_cache.append((h1, n1, 1, k))

    def _ref_indices(self, ref_blocks, drop_bases: bool = False):
        if self._ref_cache is None:
            self._ref_cache = []
            for k, blk in enumerate(ref_blocks):
                self.add_ref_block(blk, k, drop_bases)
        return self._ref_cache

    def build_ref(self, ref_blocks, drop_bases: bool = False) -> None:
        """Build the resident REF indices now (retain_ref) instead of at the first map_block."""
        if self.retain_ref:
            self._ref_indices(ref_blocks, drop_bases)

    def _drop_prefetched(self) -> None:
        pre = getattr(self, "_prefetched", None)
        if pre is not None and pre[1]:
            self.L.dmh_sync()              # its deferred stages may still be queued
            self.L.dmh_free_index(pre[1])
        self._prefetched = None

    def threads_pow2(self) -> int:
        """NTHREADS of map.c:142-147: the number of per-"thread" output files."""
        n = 1
        while 2 * n <= self.threads:
            n *= 2
        return n

    def make_spec(self, freq: Sequence[float]):
        f = (C.c_float * 4)(*[float(x) for x in freq])
        if self.spec:
            self.L.Free_Align_Spec(self.spec)
        self.spec = self.L.New_Align_Spec(self.ave_corr, self.spacing, f, 1)    # damapper.c:796
        return self.spec

    def _report(self, pend) -> None:
        """Reporter for a block whose filter passes are done (map.c:3227), then what main() does with the block."""
        reads, reads_name, ref_whole, ref_name, mflag, rh, release = pend
        maxlen = reads.db.maxlen
        self.L.Reporter(reads_name.encode(), reads.ptr(), ref_name.encode(), ref_whole.ptr(), self.spec, mflag)
        reads.db.maxlen = maxlen           # Reporter adds SPACING (map.c:3246); a block mapped again must not grow
        if rh:
            self.L.dmh_free_index(rh)      # Reporter released the device side; this frees the stub
        if release:
            reads.release()

    def finish(self) -> None:
        """Overlap mode: Reporter for the last block mapped (nothing follows it)."""
        if self._pending is not None:
            self._report(self._pending)
            self._pending = None

    def map_block(self, reads: Block, reads_name: str, ref_blocks: Sequence[Block], ref_whole: Block,
                  ref_name: str, mflag: int = capi.FLAG_DOA, restore: bool = True,
                  next_reads: Optional[Block] = None, overlap: bool = False, release: bool = False) -> Dict[str, float]:
        """One iteration of main()'s outer loop (damapper.c:825-880).  With `next_reads` the index of
        the following read block is queued on the GPU before this block's Reporter, so the device
        never idles while the host selects and writes alignments.
        overlap: the block's Reporter call is left to the NEXT map_block (or finish()): at a block boundary the last
        batches of the filter are still being finished on their own stream (repeat piles: tens of milliseconds), and
        Reporter has to wait for them; with the first `overlap_passes` Match_Filter passes of the following block
        queued ahead of it the GPU has work meanwhile (the library keeps the candidate lists of two READS blocks).
        release: reads.release() once its Reporter call has been made.  Results are unchanged either way.
        (software pipelining of the block
        loop; results are unchanged)."""
        t = {}
        t0 = time.time()
        t["between_blocks"] = (t0 - self._last_end) if getattr(self, "_last_end", None) else 0.0
        if self.spec is None:
            self.make_spec([ref_whole.db.freq[j] for j in range(4)])
        for blk in (reads, next_reads):
            if blk is not None and blk.packed is not None:
                blk.preload()              # 2-bit upload + device-side decode (dmh_preload_block_2bit)
        pre = getattr(self, "_prefetched", None)
        if pre is not None and pre[0] is reads:
            rh, rn = pre[1], pre[2]
            self._prefetched = None
        else:
            self._drop_prefetched()        # a prefetched index for a block that did not come next
            rh, rn = capi.sort_kmers(reads)
        # (overlap mode: the library runs the stages of that build on its tail stream, one behind each pass of this
        #  block -- DMH_PREFETCH_OVERLAP=0: the index is built when the block's own turn comes, as before.
        #  With few passes per block most stages would be left for the block boundary anyway, behind the tail stream's
        #  other work: ecoli (2 passes) 8.3 against 9.2 Gbp/s -- so only from a dozen passes on.)
        if next_reads is not None and self.prefetch_deferred and \
           (not overlap or (self.prefetch_overlap and 2 * len(ref_blocks) >= 12)):
            # the next block's index build is deferred: its stages are queued in the gaps of this
            # block's filter and report phases (dmh_sort_kmers_async)
            h2, n2 = capi.sort_kmers(next_reads, deferred=True)
            self._prefetched = (next_reads, h2, n2)
        t["index_reads"] = time.time() - t0
        t1 = time.time()
        npass = 0
        t["report"] = 0.0

        def passed():
            # overlap mode: the previous block's Reporter call, a few passes into this block
            nonlocal npass
            npass += 1
            if self._pending is not None and npass == self.overlap_passes:
                tr = time.time()
                self._report(self._pending)
                self._pending = None
                t["report"] += time.time() - tr

        if self.retain_ref:
            refs = self._ref_indices(ref_blocks)
            for j, (h, n, comp, k) in enumerate(refs):
                # (the pass after this one: when it is this REF block's complement, one sweep of the READ keys serves both)
                if self.hint_passes:
                    self.L.dmh_next_pass(rh, refs[j + 1][0]) if j + 1 < len(refs) else self.L.dmh_next_pass(None, None)
                capi.match_filter(reads, ref_blocks[k], rh, rn, h, n, comp, 1 if (k == 0 and comp == 0) else 0)
                passed()
        else:
            for k, blk in enumerate(ref_blocks):
                for comp in (0, 1):
                    if comp:
                        blk.complement_inplace()                                      # damapper.c:853
                    h, n = capi.sort_kmers(blk)
                    capi.match_filter(reads, blk, rh, rn, h, n, comp, 1 if (k == 0 and comp == 0) else 0)
                    passed()
                if restore:
                    blk.complement_inplace()
        if self._pending is not None:        # (fewer passes than overlap_passes)
            tr = time.time()
            self._report(self._pending)
            self._pending = None
            t["report"] += time.time() - tr
        t["filter"] = time.time() - t1 - t["report"]
        if overlap:
            self._pending = (reads, reads_name, ref_whole, ref_name, mflag, rh, release)
            self._last_end = time.time()
            t["total"] = self._last_end - t0
            return t
        t2 = time.time()
        if next_reads is not None and not self.prefetch_deferred:
            h2, n2 = capi.sort_kmers(next_reads)
            self._prefetched = (next_reads, h2, n2)
        self._report((reads, reads_name, ref_whole, ref_name, mflag, rh, release))
        t["report"] += time.time() - t2
        self._last_end = time.time()
        t["total"] = self._last_end - t0
        return t


def finish_block(sort_path: str, reads_name: str, ref_name: str, mflag: int = capi.FLAG_DOA,
                 map_order: bool = True, outdir: str = ".", keep_parts: bool = False,
                 nparts: Optional[int] = None) -> Dict[str, str]:
    """The post-step of main() (damapper.c:887-910) with this package's LAsort/LAcat/LAmerge
    (tools/latools.cpp): per-thread files -> `<reads>.<ref>.las` (and `<ref>.<reads>.las` for -C).
    All queued Reporter output must be on disk: call after dmh_report_wait() when reports are async."""
    import glob
    import subprocess
    from . import las
    out = {}
    a = ["-a"] if map_order else []

    def cleanup(pattern):
        if not keep_parts:
            for f in glob.glob(pattern):
                os.remove(f)

    # the parts M1..M<nparts> / R1..R<nparts> by name (not every numbered file that happens to lie in sort_path: an
    # earlier run with a larger -T may have left higher-numbered ones), the finals written aside and renamed when whole
    def parts(stem, suffix):
        if nparts is None:
            return [stem + "@" + suffix]
        return [stem + "%d%s" % (i + 1, suffix) for i in range(nparts)]

    if mflag & capi.FLAG_DOA:
        stem = os.path.join(sort_path, "%s.%s.M" % (reads_name, ref_name))
        subprocess.check_call([las.tool_path("LAsort")] + a + parts(stem, ".las"))
        final = os.path.join(outdir, "%s.%s.las" % (reads_name, ref_name))
        with open(final + ".tmp", "wb") as f:
            subprocess.check_call([las.tool_path("LAcat")] + parts(stem, ".S"), stdout=f)
        os.replace(final + ".tmp", final)
        cleanup(stem + "*.las")
        out["map"] = final
    if mflag & capi.FLAG_DOB:
        stem = os.path.join(sort_path, "%s.%s.R" % (ref_name, reads_name))
        subprocess.check_call([las.tool_path("LAsort")] + a + parts(stem, ".las"))
        final = os.path.join(outdir, "%s.%s" % (ref_name, reads_name))
        subprocess.check_call([las.tool_path("LAmerge")] + a + [final + ".tmp"] + parts(stem, ".S.las"))
        os.replace(final + ".tmp.las", final + ".las")
        cleanup(stem + "*.las")
        out["cover"] = final + ".las"
    return out


def map_db(workdir: str, ref_root: str, read_root: str, sort_path: Optional[str] = None,
           mflag: int = capi.FLAG_DOA, masks: Sequence[str] = (), **kw) -> Dict[str, float]:
    """damapper <ref> <reads> on DBs written by dazzdb.write_db (all blocks of the reads DB).
    mflag: FLAG_DOA (default), FLAG_DOA|FLAG_DOB (-C) or FLAG_DOB (-C -N), damapper.c:709-720."""
    sort_path = sort_path or os.path.join(workdir, "las_hip")
    m = Mapper(sort_path=sort_path, **kw)
    cwd = os.getcwd()
    os.chdir(workdir)        # the -p track files go to the current directory (map.c:3301-3302)
    try:
        nb, _ = read_stub(ref_root + ".db")
        used: set = set()
        ref_blocks = [load_db("./" + ref_root + ".db", part=k + 1, masks=masks, used=used) for k in range(nb)]
        ref_whole = load_db("./" + ref_root + ".db")           # Reporter's copy carries no mask (damapper.c:790)
        reads = load_db("./" + read_root + ".db", masks=masks, used=used)
        out = m.map_block(reads, read_root, ref_blocks, ref_whole, ref_root, mflag=mflag)
        for name in masks:
            if name not in used:
                print("damapper: Warning: Track %s given but never used." % name)      # damapper.c:916-918
        return out
    finally:
        os.chdir(cwd)
