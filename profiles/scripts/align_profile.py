import sys, os, ctypes as C, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from damapper_amd import capi
capi.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "build", "libdamapper_hip_prof.so")
from damapper_amd import dazzdb, driver
L = capi.lib()
g = dazzdb.random_genome(np.random.default_rng(20), [4_600_000])
blocks = [dazzdb.simulate_block(1000 + b, g[0], 25000, read_len=10000, err=0.15, tfirst=b * 25000, part=b + 1, path=b"./.READS") for b in range(2)]
ref_blocks = [dazzdb.make_block(g, tfirst=0, part=1, path=b"./.REF")]
ref_whole = dazzdb.make_block(g, path=b"./.REF")
m = driver.Mapper(kmer=20, ave_corr=0.85, spacing=100, threads=4, sort_path=tempfile.mkdtemp(), mem_limit=64 << 30, retain_ref=True)
m.make_spec([ref_whole.db.freq[j] for j in range(4)])
out = (C.c_ulonglong * 16)()
for i in range(3):
    t0 = time.time()
    m.map_block(blocks[i % 2], "READS.%d" % (i % 2 + 1), ref_blocks, ref_whole, "REF")
    dt = time.time() - t0
    L.dmh_align_profile(out, 1)
    v = list(out)
    print("step %d %.1f ms: reg_pass %.3g cyc (%d passes), fallback %.3g cyc (%d passes), trace %.3g cyc, pebbles %d, job cyc %.3g, jobs %d" %
          (i, dt * 1e3, v[0], v[3], v[1], v[4], v[2], v[5], v[6], v[7]))
    print("   max band per pass: <=14: %d  <=30: %d  <=46: %d  <=62: %d ; mean band %.2f ; wave steps %d (%.0f cyc/step)" % (v[8], v[9], v[10], v[11], v[12] / max(1, v[13]), v[13], v[0] / max(1, v[13])))
    print("   wave steps with a band of more than 14 diagonals: %.2f%% ; in wide mode with hysteresis (in > 14, out <= 10): %.2f%%" % (100.0 * v[14] / max(1, v[13]), 100.0 * v[15] / max(1, v[13])))
    print("   (16-lane kernel: [12] = groups active per wave step: %.2f of 4; [14] = wave steps with a wide group: %.1f%%)" % (v[12] / max(1, v[13]), 100.0 * v[14] / max(1, v[13])))
    print("   (32-lane kernel: groups active per wave step: %.2f of 2 over %d wave steps)" % (v[11] / max(1, v[10]), v[10]))
    print("   cycles between consecutive wave steps of a wavefront: %.0f (16-lane: over %d steps; 32-lane: over %d steps)" % (v[15] / max(1, v[13] if v[10] < 100000 else v[10]), v[13], v[10]))
    tot = max(1, v[6])
    print("   share of job time: reg %.1f%%  fallback %.1f%%  trace %.1f%%  other %.1f%%" %
          (100 * v[0] / tot, 100 * v[1] / tot, 100 * v[2] / tot, 100 * (tot - v[0] - v[1] - v[2]) / tot))
