"""Phase clock of k_read_sort (make -C damapper_amd/csrc profile; DMH_RS_PROFILE): runs a few read blocks of the bench
workload against ONE REF block (both orientations) with the instrumented library and prints cycles per phase."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from damapper_amd import capi
capi.LIB_PATH = os.path.join(ROOT, "build", "libdamapper_hip_prof.so")
from damapper_amd import dazzdb, driver
import bench

clen, ncontig, kind, _ = bench.WORKLOADS["grch38"]
ncontig = 3
G = dazzdb.native_genome(20, [clen] * ncontig, kind, families=496)
L = capi.lib()
m = driver.Mapper(threads=4, mem_limit=64 << 30, sort_path="/tmp/rsprof", retain_ref=True)
ref_whole = dazzdb.genome_block(G, path=b"./.REF")
m.make_spec([ref_whole.db.freq[j] for j in range(4)])
blk = dazzdb.genome_block(G, 0, ncontig, part=1, path=b"./.REF", copy=True)
m.add_ref_block(blk, 0, drop_bases=True)
L.dmh_timing_enable(1)
out = (C.c_ulonglong * 16)()
L.dmh_rs_profile.argtypes = [C.c_void_p, C.c_int]
for b in range(3):
    reads = dazzdb.simulate_block_genome(1000 + b, G, 25000, tfirst=0, part=1, path=b"./.READS")
    if b == 1:
        L.dmh_rs_profile(out, 1)
        L.dmh_timing_reset()
    m.map_block(reads, "READS.1", [blk], ref_whole, "REF")
L.dmh_rs_profile(out, 0)
v = list(out)
names = ["load / rows -> words", "rank (ballots)", "digit scan", "scatter + reload", "flags + scan", "convert + records", "cluster ranks + store",
         "plain chains: the sweep"]
tot = sum(v[:8]) + v[15]
print("workgroups %d, hits %d (%.0f per workgroup)" % (v[8], v[9], v[9] / max(1, v[8])))
for i, nm in enumerate(names):
    print("%-24s %6.1f %%  %8.0f cycles per workgroup" % (nm, 100.0 * v[i] / max(1, tot), v[i] / max(1, v[8])))
print("%-24s %6.1f %%  %8.0f cycles per workgroup" % ("plain chains: per cluster", 100.0 * v[15] / max(1, tot), v[15] / max(1, v[8])))
print("hits in clusters left to the chain kernels, by cluster size <=16 <=32 <=64 more:", v[10:14], "of", v[9])
print("... of clusters up to 1023 hits: in clusters of two and more diagonal classes (1024-wide bins, a free bin between):", v[14])
print({k: round(x["ms"], 1) for k, x in sorted(capi.timings().items(), key=lambda kv: -kv[1]["ms"])[:10]})

lp = (C.c_ulonglong * 24)()
L.dmh_lane_profile.argtypes = [C.c_void_p, C.c_int]
L.dmh_lane_profile(lp, 0)
v = list(lp)
tot = max(1, sum(v[:9]))
print("k_chain_lane: hits by live window entries before the hit:", " ".join("%d:%.2f%%" % (i, 100.0 * v[i] / tot) for i in range(9)))
print("k_chain_lane: hits by live trees:", " ".join("%d:%.2f%%" % (i, 100.0 * v[12 + i] / tot) for i in range(6)))

cp = (C.c_ulonglong * 16)()
L.dmh_chain_profile.argtypes = [C.c_void_p, C.c_int]
L.dmh_chain_profile(cp, 0)
v = list(cp)
h = max(1, v[6])
print("k_chain_pile: hits %d, mean list %d, compactions %d, spills %d; cycles per hit: scan %.0f reduce %.0f barrier1 %.0f serial %.0f barrier2 %.0f compaction %.0f" %
      (v[6], v[7] / h, v[8], v[9], v[0] / h, v[1] / h, v[2] / h, v[3] / h, v[4] / h, v[5] / h))
