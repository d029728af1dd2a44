"""Larger-than-unit-test parity run (not collected by pytest): a multi-block reference of `--ref-mbp`
Mbp (random or repeat-rich), `--reads` simulated reads, HIP pipeline vs the compiled reference.
Usage: python tests/scale_check.py --ref-mbp 100 --reads 20000 [--repeat] [--blocks 2]"""
import argparse, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle'))
import numpy as np
from damapper_amd import dazzdb, driver, las, capi
import run_ref

ap = argparse.ArgumentParser()
ap.add_argument("--ref-mbp", type=float, default=100)
ap.add_argument("--contigs", type=int, default=8)
ap.add_argument("--blocks", type=int, default=2)
ap.add_argument("--reads", type=int, default=20000)
ap.add_argument("--repeat", action="store_true")
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--seed", type=int, default=77)
ap.add_argument("--report-mode", type=int, default=0, help="dmh_async_report mode while mapping (0/1/2)")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
clen = int(a.ref_mbp * 1e6 / a.contigs)
g = dazzdb.repeat_rich_genome(rng, [clen] * a.contigs) if a.repeat else dazzdb.random_genome(rng, [clen] * a.contigs)
big = np.concatenate(g)
# reads are sampled from the concatenation restricted to one contig each: use contig 0..n via per-contig blocks
blocks = []
per = a.reads // a.contigs
seqs = []
for ci, c in enumerate(g):
    b = dazzdb.simulate_block(100 + ci + 1000 * (a.seed - 77), c, per)
    seqs += dazzdb.block_reads(b)
d = tempfile.mkdtemp()
t = time.time()
dazzdb.write_db(d, "REF", g, block_size=int(a.ref_mbp * 1e6 / a.blocks) + 1)
dazzdb.write_db(d, "READS", seqs)
print("db written %.1fs, %d reads, %.0f Mbp ref" % (time.time() - t, len(seqs), sum(map(len, g)) / 1e6), flush=True)
capi.lib().dmh_timing_enable(1)
capi.lib().dmh_timing_reset()
t = time.time()
capi.lib().dmh_async_report(a.report_mode)
tm = driver.map_db(d, "REF", "READS", threads=4, mem_limit=64 << 30)
capi.lib().dmh_report_wait()
capi.lib().dmh_async_report(0)
t_hip = time.time() - t
print({k: round(v["ms"], 1) for k, v in sorted(capi.timings().items(), key=lambda kv: -kv[1]["ms"])[:14]})
print("hip %.2fs (incl. DB load)" % t_hip, tm, flush=True)
t = time.time()
dt, out = run_ref.run_reference(d, "REF", ["READS"], flags=["-M64"], threads=a.threads)
print("ref -T%d %.2fs" % (a.threads, dt), flush=True)
_, rr = las.read_las_set(os.path.join(d, "las_ref"), "READS.REF.M@.las")
_, hr = las.read_las_set(os.path.join(d, "las_hip"), "READS.REF.M@.las")
# the reference ran with -T16: per-thread files differ in count but the concatenated stream does not
bad = sum(1 for x, y in zip(rr, hr) if x != y)
print("records ref %d hip %d differing %d" % (len(rr), len(hr), bad))
print("PARITY", "OK" if (len(rr) == len(hr) and bad == 0) else "FAIL")
