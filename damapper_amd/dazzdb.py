"""DAZZ_DB containers for the damapper hot path: on-disk writer/reader, in-memory block
builder (the struct layout `Sort_Kmers`/`Match_Filter`/`Reporter` receive) and the
synthetic genome / PacBio-like read simulator used by tests and bench.

Formats follow the reference's DB.h (cited by line, nothing is copied):
  * stub text file  <root>.db            DB.h:431-435, parsed DB.c:766-806
  * index           .<root>.idx          raw DAZZ_DB struct (DB.h:390-420) + DAZZ_READ[] (DB.h:285-295)
  * bases           .<root>.bps          2 bit/base, first base in the top bits (DB.c:319-363)
  * in-memory block                      DB.c:1402-1437 (Load_All_Reads): [4] r0 [4] r1 ... [4]
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

DB_BEST = 0x800  # DB.h:279 read flag: survives trimming when all=0
DB_ALL = 0x1


class DAZZ_READ(C.Structure):
    """DB.h:285-295 (40 bytes on x86-64)."""
    _fields_ = [
        ("origin", C.c_int), ("rlen", C.c_int), ("fpulse", C.c_int), ("_pad0", C.c_int),
        ("boff", C.c_int64), ("coff", C.c_int64), ("flags", C.c_int), ("_pad1", C.c_int),
    ]


class DAZZ_DB(C.Structure):
    """DB.h:390-420 (112 bytes on x86-64)."""
    _fields_ = [
        ("ureads", C.c_int), ("treads", C.c_int), ("cutoff", C.c_int), ("allarr", C.c_int),
        ("freq", C.c_float * 4),
        ("maxlen", C.c_int), ("_pad0", C.c_int), ("totlen", C.c_int64),
        ("nreads", C.c_int), ("trimmed", C.c_int), ("part", C.c_int), ("ufirst", C.c_int),
        ("tfirst", C.c_int), ("_pad1", C.c_int),
        ("path", C.c_char_p), ("loaded", C.c_int), ("_pad2", C.c_int),
        ("bases", C.c_void_p), ("reads", C.c_void_p), ("tracks", C.c_void_p),
    ]


class DAZZ_TRACK(C.Structure):
    """DB.h:306-316 (64 bytes on x86-64)."""


DAZZ_TRACK._fields_ = [
    ("next", C.POINTER(DAZZ_TRACK)), ("name", C.c_char_p), ("size", C.c_int), ("nreads", C.c_int),
    ("anno", C.c_void_p), ("alen", C.c_void_p), ("data", C.c_void_p), ("loaded", C.c_int), ("_pad0", C.c_int),
    ("dmax", C.c_int64),
]

assert C.sizeof(DAZZ_READ) == 40 and C.sizeof(DAZZ_DB) == 112 and C.sizeof(DAZZ_TRACK) == 64

READ_DTYPE = np.dtype({
    "names": ["origin", "rlen", "fpulse", "boff", "coff", "flags"],
    "formats": ["<i4", "<i4", "<i4", "<i8", "<i8", "<i4"],
    "offsets": [0, 4, 8, 16, 24, 32],
    "itemsize": 40,
})


# ------------------------------------------------------------------------------------------
#  Simulator
# ------------------------------------------------------------------------------------------

def random_genome(rng: np.random.Generator, contig_lens: Sequence[int]) -> List[np.ndarray]:
    """Uniform random ACGT contigs (values 0..3)."""
    return [rng.integers(0, 4, size=int(n), dtype=np.uint8) for n in contig_lens]


def repeat_rich_genome(rng: np.random.Generator, contig_lens: Sequence[int], family_len: int = 2000,
                       divergence: float = 0.03, tandem_unit: int = 150, tandem_copies: int = 60
                       ) -> List[np.ndarray]:
    """Random contigs seeded with a diverged repeat family on both strands every 1-6 kb and
    one tandem array per contig (SURVEY.md section 8d generator)."""
    fam = rng.integers(0, 4, size=family_len, dtype=np.uint8)
    out = []
    for n in contig_lens:
        s = rng.integers(0, 4, size=int(n), dtype=np.uint8)
        p = int(rng.integers(500, 3000))
        while p + family_len < n:
            cp = fam.copy()
            mut = rng.random(family_len) < divergence
            cp[mut] = (cp[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
            if rng.random() < 0.5:
                cp = (3 - cp[::-1]).astype(np.uint8)
            s[p:p + family_len] = cp
            p += family_len + int(rng.integers(1000, 6000))
        tl = tandem_unit * tandem_copies
        if n > 4 * tl:
            unit = rng.integers(0, 4, size=tandem_unit, dtype=np.uint8)
            q = int(rng.integers(0, n - tl))
            s[q:q + tl] = np.tile(unit, tandem_copies)
        out.append(s)
    return out


def _mutate(rng: np.random.Generator, seq: np.ndarray, err: float) -> np.ndarray:
    """i.i.d. per-base errors, ins:del:sub = 60:25:15 (vectorised)."""
    n = len(seq)
    r = rng.random(n)
    is_ins = r < err * 0.60
    is_del = (r >= err * 0.60) & (r < err * 0.85)
    is_sub = (r >= err * 0.85) & (r < err)
    base = seq.copy()
    nsub = int(is_sub.sum())
    if nsub:
        base[is_sub] = (base[is_sub] + rng.integers(1, 4, size=nsub, dtype=np.uint8)) & 3
    keep = ~is_del
    counts = keep.astype(np.int64) + is_ins.astype(np.int64)
    total = int(counts.sum())
    out = np.empty(total, dtype=np.uint8)
    ends = np.cumsum(counts)
    starts = ends - counts
    # inserted base first, then the (possibly substituted) template base
    ins_pos = starts[is_ins]
    out[ins_pos] = rng.integers(0, 4, size=len(ins_pos), dtype=np.uint8)
    kp = (starts + is_ins.astype(np.int64))[keep]
    out[kp] = base[keep]
    return out


def simulate_reads(rng: np.random.Generator, contigs: Sequence[np.ndarray], nreads: int,
                   read_len: int = 10000, err: float = 0.15, dropouts: int = 0
                   ) -> Tuple[List[np.ndarray], List[Tuple[int, int, int]]]:
    """Sample (contig, position, strand) uniformly; fixed template length `read_len`.
    `dropouts` > 0 inserts that many random 200-900 bp junk segments per read (exercises
    the chain linker). Returns reads and their (contig, pos, strand)."""
    lens = np.array([len(c) for c in contigs], dtype=np.int64)
    ok = lens >= read_len
    w = np.where(ok, lens - read_len + 1, 0).astype(np.float64)
    w /= w.sum()
    reads, truth = [], []
    for _ in range(nreads):
        ci = int(rng.choice(len(contigs), p=w))
        pos = int(rng.integers(0, lens[ci] - read_len + 1))
        strand = int(rng.integers(0, 2))
        t = contigs[ci][pos:pos + read_len]
        if strand:
            t = (3 - t[::-1]).astype(np.uint8)
        r = _mutate(rng, t, err)
        for _d in range(dropouts):
            q = int(rng.integers(500, max(501, len(r) - 500)))
            junk = rng.integers(0, 4, size=int(rng.integers(200, 900)), dtype=np.uint8)
            r = np.concatenate([r[:q], junk, r[q:]])
        reads.append(r)
        truth.append((ci, pos, strand))
    return reads, truth


# ------------------------------------------------------------------------------------------
#  On-disk DB
# ------------------------------------------------------------------------------------------

def _pack_2bit(seq: np.ndarray) -> bytes:
    n = len(seq)
    pad = (-n) % 4
    s = np.concatenate([seq, np.zeros(pad, dtype=np.uint8)]) if pad else seq
    s = s.reshape(-1, 4).astype(np.uint8)
    return ((s[:, 0] << 6) | (s[:, 1] << 4) | (s[:, 2] << 2) | s[:, 3]).astype(np.uint8).tobytes()


def _unpack_2bit(buf: np.ndarray, n: int) -> np.ndarray:
    out = np.empty(len(buf) * 4, dtype=np.uint8)
    out[0::4] = buf >> 6
    out[1::4] = (buf >> 4) & 3
    out[2::4] = (buf >> 2) & 3
    out[3::4] = buf & 3
    return out[:n]


def split_blocks(lens: Sequence[int], block_size: int) -> List[int]:
    """First-read index of each block plus the end (DBsplit-like: reads are never split;
    a block closes once its bases reach `block_size`)."""
    firsts = [0]
    tot = 0
    for i, n in enumerate(lens):
        if tot > 0 and tot + n > block_size:
            firsts.append(i)
            tot = 0
        tot += n
    firsts.append(len(lens))
    return firsts


def write_db(directory: str, root: str, seqs: Sequence[np.ndarray], block_size: int = 200_000_000,
             block_firsts: Optional[Sequence[int]] = None, cutoff: int = 0, all_reads: bool = True,
             best: Optional[Sequence[bool]] = None) -> List[int]:
    """Write <root>.db, .<root>.idx, .<root>.bps as fasta2DB + DBsplit -x<cutoff> [-a] leave them.
    Returns the block boundaries (untrimmed read indices).  With a cutoff (or all_reads=False and
    some reads not flagged `best`) the DB is a TRIMMED view: blocks are cut over the surviving reads
    and the stub lists (untrimmed first, trimmed first) per block (DB.c:766-806, Trim_DB DB.c:908)."""
    os.makedirs(directory, exist_ok=True)
    n = len(seqs)
    lens = [len(s) for s in seqs]
    best = [True] * n if best is None else [bool(x) for x in best]
    keep = [lens[i] >= cutoff and (all_reads or best[i]) for i in range(n)]
    if block_firsts is not None:
        firsts = list(block_firsts)
    else:                       # DBsplit: a block closes once the bases of its SURVIVING reads reach the size
        firsts, tot = [0], 0
        for i in range(n):
            if not keep[i]:
                continue
            if tot > 0 and tot + lens[i] > block_size:
                firsts.append(i)
                tot = 0
            tot += lens[i]
        firsts.append(n)
    nblocks = len(firsts) - 1
    tfirsts = [sum(1 for i in range(b) if keep[i]) for b in firsts]

    recs = np.zeros(n, dtype=READ_DTYPE)
    off = 0
    with open(os.path.join(directory, "." + root + ".bps"), "wb") as f:
        for i, s in enumerate(seqs):
            b = _pack_2bit(np.asarray(s, dtype=np.uint8))
            f.write(b)
            recs[i] = (i, lens[i], 0, off, 0, DB_BEST if best[i] else 0)
            off += len(b)

    allb = np.concatenate([np.asarray(s, dtype=np.uint8) for s in seqs]) if n else np.zeros(0, np.uint8)
    cnt = np.bincount(allb, minlength=4).astype(np.float64)
    freq = cnt / max(1.0, cnt.sum())

    hdr = DAZZ_DB()
    hdr.ureads = n
    hdr.treads = sum(keep)
    hdr.cutoff = cutoff
    hdr.allarr = DB_ALL if all_reads else 0
    for j in range(4):
        hdr.freq[j] = float(freq[j])
    hdr.maxlen = max(lens) if lens else 0
    hdr.totlen = int(sum(lens))
    hdr.nreads = n
    hdr.trimmed = 0
    with open(os.path.join(directory, "." + root + ".idx"), "wb") as f:
        f.write(bytes(hdr))
        f.write(recs.tobytes())

    with open(os.path.join(directory, root + ".db"), "w") as f:
        f.write("files = %9d\n" % 1)
        f.write("  %9d %s %s\n" % (n, root, root))
        f.write("blocks = %9d\n" % nblocks)
        f.write("size = %11d cutoff = %9d all = %1d\n" % (block_size, cutoff, 1 if all_reads else 0))
        for b, t in zip(firsts, tfirsts):
            f.write(" %9d %9d\n" % (b, t))
    return firsts


Intervals = Sequence[Sequence[Tuple[int, int]]]      # per read: sorted, disjoint [begin, end) pairs


def write_mask_track(directory: str, root: str, name: str, intervals: Intervals) -> None:
    """Interval ("mask") track files .<root>.<name>.anno / .data as DBdust, TANmask or REPmask leave
    them: anno = int32 nreads, int32 0 (size 0 marks a mask track, DB.c:1677), int64 byte offsets
    [nreads+1]; data = int32 begin/end pairs (DB.c:1853-1957)."""
    n = len(intervals)
    off = np.zeros(n + 1, dtype=np.int64)
    flat: List[int] = []
    for i, iv in enumerate(intervals):
        for b, e in iv:
            flat += [int(b), int(e)]
        off[i + 1] = 4 * len(flat)
    with open(os.path.join(directory, ".%s.%s.anno" % (root, name)), "wb") as f:
        f.write(np.array([n, 0], dtype=np.int32).tobytes())
        f.write(off.tobytes())
    with open(os.path.join(directory, ".%s.%s.data" % (root, name)), "wb") as f:
        f.write(np.array(flat, dtype=np.int32).tobytes())


def read_mask_track(directory: str, root: str, name: str, lo: int, hi: int, total: int) -> Optional[List[List[Tuple[int, int]]]]:
    """Intervals of reads [lo, hi) of a whole-DB mask track, or None when the DB has no such track
    (Check_Track returning -2, damapper.c:352-366)."""
    an = os.path.join(directory, ".%s.%s.anno" % (root, name))
    if not os.path.exists(an):
        return None
    with open(an, "rb") as f:
        tracklen, size = np.frombuffer(f.read(8), dtype=np.int32)
        if size != 0:
            raise ValueError("%s track is not a mask track." % name)            # damapper.c:355-358
        if tracklen != total:
            print("Warning: %s track not sync'd with db %s, ignored." % (name, root))
            return None
        off = np.frombuffer(f.read(8 * (total + 1)), dtype=np.int64)
    data = np.fromfile(os.path.join(directory, ".%s.%s.data" % (root, name)), dtype=np.int32)
    out = []
    for r in range(lo, hi):
        d = data[off[r] // 4:off[r + 1] // 4]
        out.append([(int(d[j]), int(d[j + 1])) for j in range(0, len(d), 2)])
    return out


def union_intervals(tracks: Sequence[Intervals]) -> List[List[Tuple[int, int]]]:
    """Union of several masks per read (merge_tracks, damapper.c:232-343: an event sweep that keeps
    the outermost begin/end of every covered stretch)."""
    n = len(tracks[0])
    out: List[List[Tuple[int, int]]] = []
    for r in range(n):
        ev = sorted(iv for t in tracks for iv in t[r])
        cur: List[Tuple[int, int]] = []
        for b, e in ev:
            if cur and b <= cur[-1][1]:
                if e > cur[-1][1]:
                    cur[-1] = (cur[-1][0], e)
            else:
                cur.append((b, e))
        out.append(cur)
    return out


@dataclass
class Block:
    """An in-memory DAZZ_DB block laid out exactly as Load_All_Reads leaves it
    (DB.c:1402-1437); keeps the numpy buffers alive for the ctypes struct."""
    db: DAZZ_DB
    bases: np.ndarray        # uint8, [4] r0 [4] r1 ... [4]; db.bases points at element 1
    reads: np.ndarray        # READ_DTYPE, nreads+2 records; db.reads points at record 1
    path: bytes
    track: Optional[DAZZ_TRACK] = None       # the k-mer mask (-m), if any
    track_anno: Optional[np.ndarray] = None  # int64[nreads+1], in ints (damapper.c:386-388)
    track_data: Optional[np.ndarray] = None  # int32 begin/end pairs
    track_alen: Optional[np.ndarray] = None
    packed: Optional[np.ndarray] = None      # load_db(packed=True): the block's slice of the .bps file; bases is None
    coff: Optional[np.ndarray] = None        # int64[nreads]: byte offset of each read's record inside `packed`
    resident: bool = False                   # the device holds the block (preload() done)
    keep: Optional[np.ndarray] = None        # the uploaded slice: a page-locked one goes up asynchronously and has
                                             # to stay as it is until a call that uses the block has returned

    def preload(self) -> None:
        """Make the block resident in HBM before it is used: a packed block is uploaded as the 2-bit bytes
        of the .bps file and decoded by a kernel (dmh_preload_block_2bit, the device-side Load_All_Reads);
        an ordinary block is uploaded as it is.  Released by release()."""
        from . import capi
        if self.resident:
            return
        if self.packed is not None:
            capi.lib().dmh_preload_block_2bit(self.ptr(), self.packed.ctypes.data, len(self.packed), self.coff.ctypes.data)
            self.keep, self.packed = self.packed, None      # the device copy is the block now
        else:
            capi.lib().dmh_preload_block(self.ptr())
        self.resident = True

    def release(self) -> None:
        from . import capi
        if self.resident:
            capi.lib().dmh_release_block(self.ptr())
            self.resident = False
            self.keep = None

    def download(self) -> np.ndarray:
        """The device image of a resident block (separator + bases), for tests."""
        from . import capi
        out = np.empty(int(self.reads[self.nreads + 1]["boff"]) + 1, dtype=np.uint8)
        if capi.lib().dmh_download_block(self.ptr(), out.ctypes.data) != 0:
            raise RuntimeError("block is not resident")
        return out

    def set_mask(self, intervals: Optional[Intervals], merged: bool = False) -> None:
        """Attach the (already merged) mask as block->tracks, the way read_DB leaves it: a single
        track keeps Open_Track's size 8, a merge of several is named "merge" with size 4
        (damapper.c:243-262)."""
        if intervals is None:
            self.track = None
            self.db.tracks = None
            return
        n = self.nreads
        assert len(intervals) == n
        anno = np.zeros(n + 1, dtype=np.int64)
        flat: List[int] = []
        for i, iv in enumerate(intervals):
            for b, e in iv:
                flat += [int(b), int(e)]
            anno[i + 1] = len(flat)
        self.track_anno = anno
        self.track_data = np.array(flat + [0, 0], dtype=np.int32)
        self.track_alen = np.diff(anno).astype(np.int32)
        t = DAZZ_TRACK()
        t.name = b"merge" if merged else b"mask"
        t.size = 4 if merged else 8
        t.nreads = n
        t.anno = self.track_anno.ctypes.data
        t.alen = self.track_alen.ctypes.data
        t.data = self.track_data.ctypes.data
        t.loaded = 1
        t.dmax = int(self.track_alen.max()) * 4 if n else 0
        self.track = t
        self.db.tracks = C.cast(C.pointer(t), C.c_void_p)

    def mask_intervals(self) -> Optional[List[List[Tuple[int, int]]]]:
        if self.track is None:
            return None
        a, d = self.track_anno, self.track_data
        return [[(int(d[j]), int(d[j + 1])) for j in range(int(a[i]), int(a[i + 1]), 2)] for i in range(self.nreads)]

    @property
    def nreads(self) -> int:
        return int(self.db.nreads)

    def ptr(self):
        return C.byref(self.db)

    def seq(self, i: int) -> np.ndarray:
        r = self.reads[i + 1]
        o = int(r["boff"]) + 1
        return self.bases[o:o + int(r["rlen"])]

    def complement_inplace(self) -> None:
        """damapper.c:433-469 (complement_DB, inplace): reverse-complement every read and
        swap the base frequencies."""
        try:
            L = _sim_lib()
            boff = np.ascontiguousarray(self.reads["boff"][1:self.nreads + 2])
            L.dmh_host_complement(self.bases.ctypes.data + 1, boff.ctypes.data, self.nreads)
        except (RuntimeError, AttributeError, OSError):
            for i in range(self.nreads):
                r = self.reads[i + 1]
                o = int(r["boff"]) + 1
                n = int(r["rlen"])
                self.bases[o:o + n] = 3 - self.bases[o:o + n][::-1]
        f = self.db.freq
        f[0], f[3] = f[3], f[0]
        f[1], f[2] = f[2], f[1]
        if self.track is not None:              # damapper.c:505-518: intervals mirror with the read
            a, d = self.track_anno, self.track_data
            for i in range(self.nreads):
                lo, hi = int(a[i]), int(a[i + 1])
                if hi > lo:
                    d[lo:hi] = int(self.reads[i + 1]["rlen"]) - d[lo:hi][::-1]


_pin_pool: List[Tuple[int, int]] = []       # (capacity, address) of page-locked buffers waiting for re-use
_PIN_GRAIN = 16 << 20
_PIN_KEEP = 6


def _pin_return(L, cap: int, p: int) -> None:
    if len(_pin_pool) < _PIN_KEEP:
        _pin_pool.append((cap, p))
    else:
        L.dmh_host_free(p)


def host_bytes(n: int, pinned: bool) -> np.ndarray:
    """n uint8s; pinned=True asks libdamapper_hip for page-locked memory (dmh_host_alloc) so that the
    block is uploaded by one DMA without staging.  Page-locking is slow (and stalls other HIP calls), so
    buffers are handed back to a small pool when their array dies and re-used for the next block.
    Falls back to ordinary memory without a device."""
    if pinned:
        try:
            import weakref
            so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdamapper_hip.so")
            L = C.CDLL(so, mode=C.RTLD_GLOBAL)
            L.dmh_host_alloc.argtypes = [C.c_size_t]
            L.dmh_host_alloc.restype = C.c_void_p
            L.dmh_host_free.argtypes = [C.c_void_p]
            L.dmh_device_count.restype = C.c_int
            p, cap = None, 0
            fit = [(c, a) for (c, a) in _pin_pool if c >= n]
            if fit:
                cap, p = min(fit)
                _pin_pool.remove((cap, p))
            elif L.dmh_device_count() > 0:
                cap = (max(n, 1) + _PIN_GRAIN - 1) // _PIN_GRAIN * _PIN_GRAIN
                p = L.dmh_host_alloc(cap)
            if p:
                arr = np.ctypeslib.as_array((C.c_uint8 * n).from_address(p))
                weakref.finalize(arr, _pin_return, L, cap, p)
                return arr
        except (OSError, AttributeError):
            pass
    return np.empty(n, dtype=np.uint8)


def make_block(seqs: Sequence[np.ndarray], tfirst: int = 0, part: int = 0,
               path: bytes = b"./.mem", freq: Optional[Sequence[float]] = None, pinned: bool = False,
               lens: Optional[np.ndarray] = None) -> Block:
    """The in-memory block of `seqs`.  With `lens` given instead (seqs empty) only the read records are
    laid out and `bases` stays None: the shape of a block whose bases are decoded on the device."""
    headless = lens is not None
    n = len(lens) if headless else len(seqs)
    if not headless:
        lens = np.array([len(s) for s in seqs], dtype=np.int64)
    lens = np.asarray(lens, dtype=np.int64)
    tot = int(lens.sum())
    reads = np.zeros(n + 2, dtype=READ_DTYPE)
    offs = np.concatenate(([0], np.cumsum(lens + 1)))
    reads["origin"][1:n + 1] = tfirst + np.arange(n)
    reads["rlen"][1:n + 1] = lens
    reads["boff"][1:n + 2] = offs
    reads["flags"][1:n + 1] = DB_BEST
    if headless:
        bases = None
        if freq is None:
            freq = [0.25] * 4
    else:
        bases = host_bytes(tot + n + 4, pinned)
        bases[:] = 4
        for i, s in enumerate(seqs):
            o = int(offs[i])
            bases[1 + o:1 + o + len(s)] = s
    db = DAZZ_DB()
    db.ureads = db.treads = n
    db.cutoff = 0
    db.allarr = DB_ALL
    if freq is None:
        cnt = np.bincount(bases[bases < 4], minlength=4).astype(np.float64)
        freq = cnt / max(1.0, cnt.sum())
    for j in range(4):
        db.freq[j] = float(freq[j])
    db.maxlen = int(lens.max()) if n else 0
    db.totlen = tot
    db.nreads = n
    db.trimmed = 1
    db.part = part
    db.ufirst = db.tfirst = tfirst
    db.path = path
    db.loaded = 1
    db.bases = (bases.ctypes.data + 1) if bases is not None else None
    db.reads = reads.ctypes.data + 40
    db.tracks = None
    # Open_DB's kludge (DB.c:853-854): block read counts in the two ints before reads[0]
    iv = reads.view(np.int32)
    iv[9] = n    # ((int *) reads)[-1]
    iv[8] = n    # ((int *) reads)[-2]
    return Block(db=db, bases=bases, reads=reads, path=path)


_sim = None


def _sim_lib():
    global _sim
    if _sim is None:
        so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdmh_sim.so")
        if not os.path.exists(so):
            raise RuntimeError("libdmh_sim.so is not built (run __graft_entry__.build())")
        L = C.CDLL(so)
        L.dmh_sim_lengths.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_void_p]
        L.dmh_sim_fill.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_void_p, C.c_void_p]
        L.dmh_host_complement.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.dmh_sim_genome.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_double,
                                     C.c_int, C.c_int, C.c_int, C.c_int]
        L.dmh_sim_lengths_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double,
                                            C.c_uint64, C.c_void_p, C.c_double]
        L.dmh_sim_fill_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double,
                                         C.c_uint64, C.c_void_p, C.c_void_p, C.c_double]
        L.dmh_pack_bps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _sim = L
    return _sim


def simulate_block(seed: int, contig: np.ndarray, nreads: int, read_len: int = 10000, err: float = 0.15,
                   tfirst: int = 0, part: int = 0, path: bytes = b"./.mem", pinned: bool = False) -> Block:
    """Large synthetic read block straight into the in-memory DAZZ_DB layout (native, threaded
    generator in tools/simulate.c: uniform position and strand on one contig, i.i.d. errors
    ins:del:sub = 60:25:15; per-read RNG streams, so the block depends on `seed` only)."""
    L = _sim_lib()
    g = np.ascontiguousarray(contig, dtype=np.uint8)
    rlen = np.zeros(nreads, dtype=np.int32)
    L.dmh_sim_lengths(g.ctypes.data, len(g), nreads, read_len, err, seed, rlen.ctypes.data)
    boff = np.zeros(nreads + 1, dtype=np.int64)
    np.cumsum(rlen.astype(np.int64) + 1, out=boff[1:])
    tot = int(rlen.sum())
    bases = host_bytes(tot + nreads + 4, pinned)
    bases[-4:] = 4
    L.dmh_sim_fill(g.ctypes.data, len(g), nreads, read_len, err, seed, boff.ctypes.data, bases.ctypes.data)
    reads = np.zeros(nreads + 2, dtype=READ_DTYPE)
    reads["origin"][1:nreads + 1] = tfirst + np.arange(nreads)
    reads["rlen"][1:nreads + 1] = rlen
    reads["boff"][1:nreads + 2] = boff
    reads["flags"][1:nreads + 1] = DB_BEST
    db = DAZZ_DB()
    db.ureads = db.treads = nreads
    db.cutoff = 0
    db.allarr = DB_ALL
    cnt = np.bincount(bases[:1 << 24], minlength=5)[:4].astype(np.float64)
    for j in range(4):
        db.freq[j] = float(cnt[j] / max(1.0, cnt.sum()))
    db.maxlen = int(rlen.max())
    db.totlen = tot
    db.nreads = nreads
    db.trimmed = 1
    db.part = part
    db.ufirst = db.tfirst = tfirst
    db.path = path
    db.loaded = 1
    db.bases = (bases.ctypes.data + 1) if bases is not None else None
    db.reads = reads.ctypes.data + 40
    db.tracks = None
    iv = reads.view(np.int32)
    iv[9] = nreads
    iv[8] = nreads
    return Block(db=db, bases=bases, reads=reads, path=path)


@dataclass
class Genome:
    """A synthetic reference held in the in-memory layout of the WHOLE reference DB: [4] c0 [4] c1 ... [4]
    (DB.c:1402-1437), so that the whole-DB block Reporter needs is a view and not a copy."""
    layout: np.ndarray           # uint8
    lens: np.ndarray             # int64 contig lengths
    offs: np.ndarray             # int64 index of each contig's first base inside `layout`
    kind: str = "random"

    @property
    def contigs(self) -> List[np.ndarray]:
        return [self.layout[int(o):int(o) + int(n)] for o, n in zip(self.offs, self.lens)]

    def __len__(self) -> int:
        return len(self.lens)

    @property
    def total(self) -> int:
        return int(self.lens.sum())


def native_genome(seed: int, contig_lens: Sequence[int], kind: str = "repeat", family_len: int = 2000,
                  divergence: float = 0.03, gap: Tuple[int, int] = (1000, 6000), tandem_unit: int = 150,
                  tandem_copies: int = 60, families: int = 1) -> Genome:
    """SURVEY 8d's reference generators at genome scale (tools/simulate.c, threaded; the result depends on
    `seed` only): kind "random" = uniform ACGT; "repeat" = the repeat-rich model of repeat_rich_genome --
    `families` consensus sequences of `family_len` bases shared by all contigs, a copy of a randomly drawn
    family mutated at `divergence` on a random strand every gap[0]..gap[1] bases, and one tandem array per
    contig.  families = 1 is repeat_rich_genome's model as it is."""
    L = _sim_lib()
    lens = np.asarray(contig_lens, dtype=np.int64)
    n = len(lens)
    offs = 1 + np.concatenate(([0], np.cumsum(lens + 1)))[:n]
    layout = np.empty(int(lens.sum()) + n + 4, dtype=np.uint8)
    families = max(1, int(families))
    fam = np.random.default_rng(seed).integers(0, 4, size=max(1, family_len) * families, dtype=np.uint8)
    rep = kind == "repeat"
    for c in range(n):
        o = int(offs[c])
        layout[o - 1] = 4
        L.dmh_sim_genome(layout.ctypes.data + o, int(lens[c]), (seed * 1000003 + c) & 0xffffffffffffffff, fam.ctypes.data, families,
                         family_len if rep else 0, divergence, gap[0], gap[1],
                         tandem_unit if rep else 0, tandem_copies if rep else 0)
    layout[int(offs[-1]) + int(lens[-1]):] = 4
    return Genome(layout=layout, lens=lens, offs=offs, kind=kind)


def genome_from_contigs(seqs: Sequence[np.ndarray], kind: str = "fasta") -> Genome:
    """Contigs (arrays of 0..3) in the whole-DB layout a Genome keeps."""
    lens = np.asarray([len(x) for x in seqs], dtype=np.int64)
    n = len(lens)
    offs = 1 + np.concatenate(([0], np.cumsum(lens + 1)))[:n]
    layout = np.full(int(lens.sum()) + n + 4, 4, dtype=np.uint8)
    for o, x in zip(offs, seqs):
        layout[int(o):int(o) + len(x)] = x
    return Genome(layout=layout, lens=lens, offs=offs, kind=kind)


def genome_from_fasta(path: str, min_len: int = 1) -> Genome:
    """A FASTA on disk as a Genome: entries cut at runs of N into contigs as fasta2DAM does, contigs shorter than
    `min_len` left out (a simulated read must fit its contig)."""
    seqs, _ = read_fasta(path, split_n=True, min_len=min_len)
    if not seqs:
        raise ValueError("%s holds no contig of %d bases or more" % (path, min_len))
    return genome_from_contigs(seqs)


def genome_block(G: Genome, first: int = 0, last: Optional[int] = None, part: int = 0, path: bytes = b"./.REF",
                 copy: bool = False, freq: Optional[Sequence[float]] = None) -> Block:
    """Contigs [first, last) of a Genome as a DAZZ_DB block (tfirst = first).  Without `copy` the block's bases are
    a view into the genome (a block that is complemented in place needs a copy)."""
    last = len(G) if last is None else last
    n = last - first
    lens = G.lens[first:last]
    lo = int(G.offs[first]) - 1
    hi = int(G.offs[last - 1]) + int(lens[-1]) + 1          # one past the final separator
    bases = G.layout[lo:hi]
    if copy:
        b2 = np.empty(hi - lo + 3, dtype=np.uint8)
        b2[:hi - lo] = bases
        b2[hi - lo:] = 4
        bases = b2
    if freq is None:
        cnt = np.bincount(bases[1:1 + (1 << 24)], minlength=5)[:4].astype(np.float64)
        freq = cnt / max(1.0, cnt.sum())
    blk = make_block([], tfirst=first, part=part, path=path, freq=freq, lens=lens)
    blk.bases = bases
    blk.db.bases = bases.ctypes.data + 1
    return blk


def simulate_block_genome(seed: int, G: Genome, nreads: int, read_len: int = 10000, err: float = 0.15,
                          tfirst: int = 0, part: int = 0, path: bytes = b"./.mem", pinned: bool = False,
                          lengths: str = "fixed") -> Block:
    """simulate_block with the template start uniform over ALL contigs of a Genome (a read never spans two).
    lengths = "lognormal": CLR-like template lengths, log-normal with mean `read_len` (sigma 0.4, clipped to
    read_len / 10 .. 4 read_len) instead of the fixed `read_len`."""
    L = _sim_lib()
    nc = len(G)
    coff = np.ascontiguousarray(np.concatenate((G.offs, [int(G.offs[-1]) + int(G.lens[-1])])), dtype=np.int64)
    sigma = 0.4 if lengths == "lognormal" else 0.0
    valid = np.maximum(G.lens - (4 * read_len if sigma > 0 else read_len) + 1, 0)
    cum = np.ascontiguousarray(np.concatenate(([0], np.cumsum(valid))), dtype=np.int64)
    if cum[-1] <= 0:
        raise ValueError("no contig is as long as a read")
    rlen = np.zeros(nreads, dtype=np.int32)
    g = G.layout
    L.dmh_sim_lengths_multi(g.ctypes.data, coff.ctypes.data, cum.ctypes.data, nc, nreads, read_len, err, seed, rlen.ctypes.data, sigma)
    boff = np.zeros(nreads + 1, dtype=np.int64)
    np.cumsum(rlen.astype(np.int64) + 1, out=boff[1:])
    tot = int(rlen.sum())
    bases = host_bytes(tot + nreads + 4, pinned)
    bases[-4:] = 4
    L.dmh_sim_fill_multi(g.ctypes.data, coff.ctypes.data, cum.ctypes.data, nc, nreads, read_len, err, seed,
                         boff.ctypes.data, bases.ctypes.data, sigma)
    cnt = np.bincount(bases[:1 << 24], minlength=5)[:4].astype(np.float64)
    blk = make_block([], tfirst=tfirst, part=part, path=path, freq=cnt / max(1.0, cnt.sum()), lens=rlen)
    blk.bases = bases
    blk.db.bases = bases.ctypes.data + 1
    return blk


def pack_block(blk: Block, pinned: bool = True) -> Block:
    """The block as it lies in a .bps file (4 bases per byte, every read starting on a byte) with Load_All_Reads'
    read records and no bases: what the block loop uploads (dmh_preload_block_2bit decodes it on the device)."""
    n = blk.nreads
    lens = blk.reads["rlen"][1:n + 1].astype(np.int64)
    boff = np.ascontiguousarray(blk.reads["boff"][1:n + 2], dtype=np.int64)
    clen = (lens + 3) // 4
    coff = np.ascontiguousarray(np.concatenate(([0], np.cumsum(clen))), dtype=np.int64)
    out = host_bytes(int(coff[-1]), pinned)
    _sim_lib().dmh_pack_bps(blk.bases.ctypes.data, boff.ctypes.data, coff.ctypes.data, n, out.ctypes.data)
    hb = make_block([], tfirst=int(blk.db.tfirst), part=int(blk.db.part), path=blk.path,
                    freq=[blk.db.freq[j] for j in range(4)], lens=lens)
    hb.packed = out
    hb.coff = np.ascontiguousarray(coff[:-1])
    return hb


def write_db_blocks(directory: str, root: str, blocks: Sequence[Block], block_size: int = 400_000_000) -> None:
    """write_db for the reads of several in-memory blocks, one DB block each (native 2-bit packing): the DB that
    DBsplit would leave if it cut exactly there."""
    os.makedirs(directory, exist_ok=True)
    n = sum(b.nreads for b in blocks)
    recs = np.zeros(n, dtype=READ_DTYPE)
    firsts, at, off, maxlen, totlen = [0], 0, 0, 0, 0
    with open(os.path.join(directory, "." + root + ".bps"), "wb") as f:
        for blk in blocks:
            hb = pack_block(blk, pinned=False)
            m = blk.nreads
            lens = blk.reads["rlen"][1:m + 1].astype(np.int64)
            recs["origin"][at:at + m] = np.arange(at, at + m)
            recs["rlen"][at:at + m] = lens
            recs["boff"][at:at + m] = hb.coff + off
            recs["flags"][at:at + m] = DB_BEST
            f.write(hb.packed.tobytes())
            off += len(hb.packed)
            at += m
            firsts.append(at)
            maxlen = max(maxlen, int(lens.max()) if m else 0)
            totlen += int(lens.sum())
    hdr = DAZZ_DB()
    hdr.ureads = hdr.treads = n
    hdr.cutoff = 0
    hdr.allarr = DB_ALL
    for j in range(4):
        hdr.freq[j] = blocks[0].db.freq[j]
    hdr.maxlen = maxlen
    hdr.totlen = totlen
    hdr.nreads = n
    hdr.trimmed = 0
    with open(os.path.join(directory, "." + root + ".idx"), "wb") as f:
        f.write(bytes(hdr))
        f.write(recs.tobytes())
    with open(os.path.join(directory, root + ".db"), "w") as f:
        f.write("files = %9d\n" % 1)
        f.write("  %9d %s %s\n" % (n, root, root))
        f.write("blocks = %9d\n" % len(blocks))
        f.write("size = %11d cutoff = %9d all = %1d\n" % (block_size, 0, 1))
        for b in firsts:
            f.write(" %9d %9d\n" % (b, b))


def write_db_block(directory: str, root: str, blk: Block, block_size: int = 400_000_000) -> None:
    """write_db for the reads of ONE in-memory block."""
    write_db_blocks(directory, root, [blk], block_size)


def block_reads(blk: Block, lo: int = 0, hi: Optional[int] = None) -> List[np.ndarray]:
    """Copies of reads [lo, hi) of a block as separate arrays."""
    hi = blk.nreads if hi is None else hi
    return [blk.seq(i).copy() for i in range(lo, hi)]


def read_stub_params(path: str) -> Tuple[int, int, bool]:
    """(block size, cutoff, all) of a stub file (the DBsplit parameters, DB.h:431-435)."""
    with open(path) as f:
        lines = f.read().split("\n")
    k = 1 + int(lines[0].split("=")[1])
    w = lines[k + 1].replace("=", " ").split()        # size N cutoff C all A
    return int(w[1]), int(w[3]), bool(int(w[5]))


def read_stub(path: str) -> Tuple[int, List[Tuple[int, int]]]:
    """Parse a stub file: returns (nblocks, [(ufirst, tfirst)] * (nblocks+1))."""
    with open(path) as f:
        lines = f.read().split("\n")
    nfiles = int(lines[0].split("=")[1])
    k = 1 + nfiles
    nblocks = int(lines[k].split("=")[1])
    firsts = []
    for ln in lines[k + 2:k + 2 + nblocks + 1]:
        a, b = ln.split()
        firsts.append((int(a), int(b)))
    return nblocks, firsts


def load_db(path: str, part: int = 0, masks: Sequence[str] = (), used: Optional[set] = None,
            pinned: bool = False, packed: bool = False) -> Block:
    """Open_DB + Trim_DB + Load_All_Reads equivalent (DB.c:600-1007,1402-1437).  `path` is
    'dir/root.db'; part k > 0 loads block k.  The DBsplit cutoff / all flags of the stub trim the view:
    reads shorter than the cutoff (and, without -a, reads that are not the best of their well) are left
    out and the block's reads are numbered from the stub's trimmed first.  `masks` are -m track names:
    those the DB has are loaded (over the untrimmed or the trimmed reads, whichever the track was made
    for), merged into one mask and attached (read_DB, damapper.c:345-414); their names are added to
    `used` (the MSTAT bookkeeping).  packed=True skips Load_All_Reads' decode: the block carries its slice
    of the .bps file (`Block.packed`, `Block.coff`) and no `bases`; `Block.preload()` has the device decode it."""
    d, base = os.path.split(path)
    d = d or "."
    root = base[:-3] if base.endswith(".db") else base
    nblocks, firsts = read_stub(os.path.join(d, root + ".db"))
    with open(os.path.join(d, "." + root + ".idx"), "rb") as f:
        hdr = DAZZ_DB.from_buffer_copy(f.read(112))
        recs = np.frombuffer(f.read(), dtype=READ_DTYPE)
    _, cutoff, all_reads = read_stub_params(os.path.join(d, root + ".db"))
    if part > 0:
        lo, hi = firsts[part - 1][0], firsts[part][0]
        tfirst = firsts[part - 1][1]
    else:
        lo, hi = 0, int(hdr.ureads)
        tfirst = 0
    keep_all = (recs["rlen"] >= cutoff) & (all_reads | ((recs["flags"] & DB_BEST) != 0))
    keep = keep_all[lo:hi]
    kept = recs[lo:hi][keep]
    bps_path = os.path.join(d, "." + root + ".bps")
    if len(kept) == 0:
        first = last = 0
    else:
        first = int(kept["boff"][0])
        last = int(kept["boff"][-1]) + (int(kept["rlen"][-1]) + 3) // 4
    if packed and len(kept) > 0:
        buf = host_bytes(last - first, pinned)
        with open(bps_path, "rb") as f:
            f.seek(first)
            got = f.readinto(memoryview(buf))
        if got != last - first:
            raise IOError("%s: short read" % bps_path)
        blk = make_block([], tfirst=tfirst, part=part, path=os.path.join(d, "." + root).encode(),
                         freq=[hdr.freq[j] for j in range(4)], lens=kept["rlen"].astype(np.int64))
        blk.packed = buf
        blk.coff = np.ascontiguousarray(kept["boff"].astype(np.int64) - first)
    else:
        bps = np.fromfile(bps_path, dtype=np.uint8, count=last - first, offset=first)
        seqs = []
        for r in kept:
            n = int(r["rlen"])
            o = int(r["boff"]) - first
            seqs.append(_unpack_2bit(bps[o:o + (n + 3) // 4], n))
        blk = make_block(seqs, tfirst=tfirst, part=part,
                         path=os.path.join(d, "." + root).encode(),
                         freq=[hdr.freq[j] for j in range(4)], pinned=pinned)
    blk.db.cutoff = cutoff
    blk.db.allarr = DB_ALL if all_reads else 0
    found = []
    treads = int(keep_all.sum())
    for name in masks:
        iv = read_mask_track(d, root, name, lo, hi, int(hdr.ureads))
        if iv is not None:                                   # a track over the untrimmed reads
            iv = [x for x, k in zip(iv, keep) if k]
        elif treads != int(hdr.ureads):                      # ... or over the trimmed ones
            iv = read_mask_track(d, root, name, tfirst, tfirst + int(keep.sum()), treads)
        if iv is not None:
            found.append(iv)
            if used is not None:
                used.add(name)
    if len(found) == 1:
        blk.set_mask(found[0])
    elif len(found) > 1:
        blk.set_mask(union_intervals(found), merged=True)
    return blk


# ------------------------------------------------------------------------------------------
#  FASTA import (the fasta2DB / fasta2DAM + DBsplit subset needed to feed the mapper)
# ------------------------------------------------------------------------------------------

_CODE = np.full(256, 4, dtype=np.uint8)
for _i, _ch in enumerate("acgt"):
    _CODE[ord(_ch)] = _i
    _CODE[ord(_ch.upper())] = _i


def read_fasta(path: str, split_n: bool = True, min_len: int = 1) -> Tuple[List[np.ndarray], List[str]]:
    """Sequences of a FASTA file as 0..3 arrays.  Runs of non-ACGT letters split an entry into
    contigs (what fasta2DAM does with scaffolds); with split_n=False such an entry is an error
    (fasta2DB's behaviour).  Returns (sequences, names) -- a split entry yields name/1, name/2 ..."""
    seqs: List[np.ndarray] = []
    names: List[str] = []

    def flush(name, chunks):
        if name is None:
            return
        raw = np.frombuffer("".join(chunks).encode("ascii", "replace"), dtype=np.uint8)
        code = _CODE[raw]
        bad = code > 3
        if not bad.any():
            parts = [code]
        elif not split_n:
            raise ValueError("%s: entry %s contains letters other than ACGT" % (path, name))
        else:
            edges = np.flatnonzero(np.diff(np.concatenate(([1], bad.astype(np.int8), [1]))))
            parts = [code[edges[j]:edges[j + 1]] for j in range(0, len(edges) - 1, 2)]
        keep = [p for p in parts if len(p) >= min_len]
        for j, p in enumerate(keep):
            seqs.append(np.ascontiguousarray(p))
            names.append(name if len(keep) == 1 else "%s/%d" % (name, j + 1))

    name, chunks = None, []
    with open(path) as f:
        for line in f:
            if line.startswith(">"):
                flush(name, chunks)
                name, chunks = line[1:].split()[0] if len(line) > 1 and line[1:].split() else "seq%d" % len(names), []
            else:
                chunks.append(line.strip())
    flush(name, chunks)
    return seqs, names


def fasta_to_db(fasta: str, directory: str, root: str, block_size: int = 200_000_000, split_n: bool = True,
                min_len: int = 1) -> List[int]:
    """fasta2DB/fasta2DAM followed by DBsplit -s<block_size>: writes <root>.db and its hidden files."""
    seqs, _ = read_fasta(fasta, split_n=split_n, min_len=min_len)
    if not seqs:
        raise ValueError("%s holds no sequence" % fasta)
    return write_db(directory, root, seqs, block_size=block_size)


def _main(argv=None) -> int:
    import argparse
    ap = argparse.ArgumentParser(prog="damapper_amd.dazzdb", description="build a DAZZ_DB from FASTA")
    ap.add_argument("db", help="path of the DB to create, e.g. dir/REF or dir/REF.db")
    ap.add_argument("fasta")
    ap.add_argument("-s", type=float, default=200.0, help="block size in Mbp (DBsplit -s)")
    ap.add_argument("-x", type=int, default=1, help="drop sequences shorter than this (DBsplit -x)")
    ap.add_argument("--strict", action="store_true", help="reject non-ACGT letters instead of splitting contigs there")
    a = ap.parse_args(argv)
    d, base = os.path.split(a.db)
    root = base[:-3] if base.endswith(".db") else base
    firsts = fasta_to_db(a.fasta, d or ".", root, block_size=int(a.s * 1e6), split_n=not a.strict, min_len=a.x)
    print("%s: %d sequences in %d block(s)" % (root, firsts[-1], len(firsts) - 1))
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(_main())
