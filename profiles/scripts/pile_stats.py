import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from damapper_amd import capi
if os.environ.get('PROF'):
    capi.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'build', 'libdamapper_hip_prof.so')
from damapper_amd import dazzdb, driver
rng = np.random.default_rng(77)
g = dazzdb.repeat_rich_genome(rng, [7_500_000] * 8)
seqs = []
for ci, c in enumerate(g):
    seqs += dazzdb.block_reads(dazzdb.simulate_block(100 + ci, c, 1000))
reads = dazzdb.make_block(seqs)
ref = dazzdb.make_block(g, path=b"./.REF")
L = capi.lib()
m = driver.Mapper(threads=4, mem_limit=64 << 30)
L.dmh_capture_seeds(1)
L.dmh_timing_enable(1)
rh, rn = capi.sort_kmers(reads)
h, n = capi.sort_kmers(ref)
capi.match_filter(reads, ref, rh, rn, h, n, 0, 1)
s = capi.last_seeds()
print("seeds", len(s), s.dtype)
key = s["aread"].astype(np.int64) << 32 | s["bread"].astype(np.int64)
edges = np.flatnonzero(np.diff(key)) + 1
starts = np.concatenate(([0], edges)); ends = np.concatenate((edges, [len(s)]))
sz = ends - starts
print("groups", len(sz), "max", sz.max(), "top10", np.sort(sz)[-10:])
for t in (64, 256, 1024, 4096, 16384, 65536):
    print(">= %d: %d groups, %d hits (%.1f%%)" % (t, (sz >= t).sum(), sz[sz >= t].sum(), 100.0 * sz[sz >= t].sum() / len(s)))
# window occupancy of the largest group: hits within MAX_GAP=1000 in apos
gi = int(np.argmax(sz)); a = s["apos"][starts[gi]:ends[gi]].astype(np.int64)
lo = np.searchsorted(a, a - 1000, side="left")
occ = np.arange(len(a)) - lo
print("largest group: window occupancy (all hits within 1000 bp in A) mean %.0f max %d" % (occ.mean(), occ.max()))
print({k: round(v["ms"], 1) for k, v in sorted(capi.timings().items(), key=lambda kv: -kv[1]["ms"])[:8]})

if os.environ.get('PROF'):
    import ctypes as C
    out = (C.c_ulonglong * 16)()
    L.dmh_chain_profile(out, 1)
    v = list(out)
    h = max(1, v[6])
    print("pile hits %d, mean list %d, compactions %d, spills %d; cycles per hit: scan %.0f reduce %.0f barrier1 %.0f serial %.0f barrier2 %.0f compaction %.0f" %
          (v[6], v[7] / h, v[8], v[9], v[0] / h, v[1] / h, v[2] / h, v[3] / h, v[4] / h, v[5] / h))
