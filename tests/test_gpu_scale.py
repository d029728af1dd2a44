"""GPU parity at BASELINE.json's scale: the per-thread .las record streams of the HIP path -- in the configuration
bench.py times (REF indices resident, read blocks uploaded as 2-bit .bps bytes and decoded on the device, the next
block's index built in the gaps, Reporter's last stage on a worker thread / deferred kernel) -- against the compiled
reference `damapper -T16` on the same DAZZ_DBs on disk.

  configs[1]  4.6 Mbp reference + 100 k reads in four DBsplit-sized blocks of 25 k reads
  configs[2]  a 250 Mbp repeat-rich reference (one REF block: the genome-block merge path -- 2^29-bucket table,
              four-key probes on the 2-byte low-bits array -- un-forced) + one block of 25 k reads
  configs[4]  a three-block repeat-rich reference kept resident, two read blocks, -n.95 -p (near-optimal chains
              and the repeat profile track)
  one-family  the repeat-rich generator as it is (one 2 kb family): thousands of hits per read and pass, B-clusters,
              repeat piles and k-mers beyond MAXGRAM, in cluster mode and without
"""
import os

import numpy as np
import pytest

from damapper_amd import capi, dazzdb, driver, las

pytestmark = pytest.mark.gpu


def _run_ref():
    import run_ref
    if not run_ref.have_ref():
        pytest.skip("oracle/_ref not built")
    return run_ref


def _threads():
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        n = os.cpu_count() if q == "max" else max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        n = os.cpu_count() or 4
    p = 1
    while 2 * p <= min(n, 16):
        p *= 2
    return p


def _map_blocks_like_bench(d, ref_root, reads_root, out, report_mode=2, **kw):
    """The block loop of bench.py / damapper_amd.dispatch, in process: resident REF indices, packed read blocks from
    the DB on disk, deferred index build of the next block, asynchronous Reporter."""
    L = capi.lib()
    cwd = os.getcwd()
    os.chdir(d)
    try:
        nrb, _ = dazzdb.read_stub(ref_root + ".db")
        nb, _ = dazzdb.read_stub(reads_root + ".db")
        m = driver.Mapper(threads=4, sort_path=os.path.join(d, out), retain_ref=True, **kw)
        ref_blocks = [dazzdb.load_db("./" + ref_root + ".db", part=k + 1) for k in range(nrb)]
        ref_whole = dazzdb.load_db("./" + ref_root + ".db")
        m.build_ref(ref_blocks, drop_bases=True)
        blocks = [dazzdb.load_db("./" + reads_root + ".db", part=b + 1, pinned=True, packed=True) for b in range(nb)]
        L.dmh_async_report(report_mode)
        for b, blk in enumerate(blocks):
            nxt = blocks[b + 1] if b + 1 < nb else None
            m.map_block(blk, "%s.%d" % (reads_root, b + 1), ref_blocks, ref_whole, ref_root, next_reads=nxt)
            blk.release()
        L.dmh_report_wait()
        L.dmh_async_report(0)
        m.release_ref()
        L.dmh_retain_ref_index(0)
        L.dmh_release_block(ref_whole.ptr())
        return nb
    finally:
        os.chdir(cwd)


def _compare(d, ref_dir, hip_dir, names, min_records):
    total = 0
    for name in names:
        ts_r, rr = las.read_las_set(os.path.join(d, ref_dir), name + ".REF.M@.las")
        ts_h, hr = las.read_las_set(os.path.join(d, hip_dir), name + ".REF.M@.las")
        assert ts_r == ts_h
        assert len(rr) == len(hr), "%s: %d vs %d records" % (name, len(rr), len(hr))
        bad = [i for i, (a, b) in enumerate(zip(rr, hr)) if a != b]
        assert not bad, "%s: %d records differ, first %d:\n%s\n%s" % (name, len(bad), bad[0], rr[bad[0]], hr[bad[0]])
        total += len(rr)
    assert total >= min_records, "only %d records" % total
    return total


def test_configs1_ecoli_100k_reads(tmp_path):
    run_ref = _run_ref()
    d = str(tmp_path)
    G = dazzdb.native_genome(20, [4_600_000], "random")
    dazzdb.write_db_block(d, "REF", dazzdb.genome_block(G))
    blocks = [dazzdb.simulate_block_genome(1000 + b, G, 25000, tfirst=b * 25000, part=b + 1) for b in range(4)]
    dazzdb.write_db_blocks(d, "READS", blocks)
    del blocks
    names = ["READS.%d" % (b + 1) for b in range(4)]
    run_ref.run_reference(d, "REF", names, flags=["-k20", "-e.85", "-M64"], threads=_threads())
    assert _map_blocks_like_bench(d, "REF", "READS", "las_hip", mem_limit=64 << 30) == 4
    _compare(d, "las_ref", "las_hip", names, 95000)


def test_configs2_chr1_sized_repeat_rich_reference(tmp_path):
    run_ref = _run_ref()
    d = str(tmp_path)
    G = dazzdb.native_genome(21, [125_000_000, 125_000_000], "repeat", families=333)
    dazzdb.write_db_block(d, "REF", dazzdb.genome_block(G))
    dazzdb.write_db_block(d, "READS", dazzdb.simulate_block_genome(2000, G, 25000))
    run_ref.run_reference(d, "REF", ["READS.1"], flags=["-k20", "-e.85", "-M64"], threads=_threads())
    _map_blocks_like_bench(d, "REF", "READS", "las_hip", mem_limit=64 << 30)
    _compare(d, "las_ref", "las_hip", ["READS.1"], 25000)


def test_configs4_resident_multi_block_reference_n95_p(tmp_path):
    run_ref = _run_ref()
    d = str(tmp_path)
    G = dazzdb.native_genome(22, [25_000_000] * 3, "repeat", families=33)
    dazzdb.write_db_blocks(d, "REF", [dazzdb.genome_block(G, k, k + 1) for k in range(3)])
    # the second block holds 1.5 kb reads: one that lies inside a copy of a repeat family maps to the family's other
    # copies nearly as well -- the near-optimal chains -n.95 keeps (START without BEST) and -p counts
    blocks = [dazzdb.simulate_block_genome(3000, G, 4000, tfirst=0, part=1),
              dazzdb.simulate_block_genome(3001, G, 12000, read_len=1500, tfirst=4000, part=2)]
    dazzdb.write_db_blocks(d, "READS", blocks)
    names = ["READS.1", "READS.2"]
    flags = ["-k20", "-e.85", "-n.95", "-p", "-M64"]
    run_ref.run_reference(d, "REF", names, flags=flags, threads=_threads())
    # the -p track from a single-threaded run: the reference's special_log() races with more than one report thread
    # (DESIGN.md 4); the track is -T invariant
    prof_ref = {}
    for name in names[:1]:
        for f in (".%s.prof.anno" % name, ".%s.prof.data" % name):
            os.remove(os.path.join(d, f))
        run_ref.run_reference(d, "REF", [name], flags=flags, threads=1, keep="las_ref_T1")
        prof_ref[name] = tuple(open(os.path.join(d, f), "rb").read() for f in (".%s.prof.anno" % name, ".%s.prof.data" % name))
    for name in names:
        for f in (".%s.prof.anno" % name, ".%s.prof.data" % name):
            if os.path.exists(os.path.join(d, f)):
                os.remove(os.path.join(d, f))
    _map_blocks_like_bench(d, "REF", "READS", "las_hip", mem_limit=64 << 30, best_tie=0.95, profile=1)
    n = _compare(d, "las_ref", "las_hip", names, 12000)
    flagset = set()
    for name in names:
        flagset |= {r.flags for r in las.read_las_set(os.path.join(d, "las_hip"), name + ".REF.M@.las")[1]}
    assert any(f & las.START_FLAG and not f & las.BEST_FLAG for f in flagset), "no near-optimal chain in %d records" % n
    for name, (anno, data) in prof_ref.items():
        assert open(os.path.join(d, ".%s.prof.anno" % name), "rb").read() == anno
        got = np.frombuffer(open(os.path.join(d, ".%s.prof.data" % name), "rb").read(), dtype=np.uint8)
        want = np.frombuffer(data, dtype=np.uint8)
        assert len(got) == len(want) and np.array_equal(got, want), "-p track of %s differs at %s" % (name, np.nonzero(got != want)[0][:10])
        assert want.max() >= 10                      # repeat coverage shows


@pytest.mark.parametrize("cluster", [-1, 0], ids=["auto", "pairs"])
def test_one_family_repeat_stress(cluster, tmp_path):
    run_ref = _run_ref()
    d = str(tmp_path)
    G = dazzdb.native_genome(23, [4_000_000, 4_000_000], "repeat", families=1)
    dazzdb.write_db_block(d, "REF", dazzdb.genome_block(G))
    dazzdb.write_db_block(d, "READS", dazzdb.simulate_block_genome(4000, G, 1500))
    run_ref.run_reference(d, "REF", ["READS.1"], flags=["-k20", "-e.85", "-M64"], threads=_threads())
    capi.lib().dmh_test_option(b"cluster", cluster)
    try:
        _map_blocks_like_bench(d, "REF", "READS", "las_hip", mem_limit=64 << 30)
    finally:
        capi.lib().dmh_test_option(b"cluster", -1)
    _compare(d, "las_ref", "las_hip", ["READS.1"], 1500)
