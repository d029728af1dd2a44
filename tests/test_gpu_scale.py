"""GPU parity at BASELINE.json's scale: the per-thread .las record streams of the HIP path -- in the configuration
bench.py times (REF indices resident, read blocks uploaded as 2-bit .bps bytes and decoded on the device, the next
block's index built in the gaps, Reporter's last stage on a worker thread / deferred kernel) -- against the compiled
reference `damapper -T16` on the same DAZZ_DBs on disk.

  configs[1]  4.6 Mbp reference + 100 k reads in four DBsplit-sized blocks of 25 k reads
  configs[2]  a 250 Mbp repeat-rich reference (one REF block: the genome-block merge path -- 2^29-bucket table,
              four-key probes on the 2-byte low-bits array -- un-forced) + one block of 25 k reads
  configs[3]  the GRCh38-SIZED reference of bench.py (3.1 Gbp, 9 REF blocks, 18 resident indices), two read blocks
  configs[4]  the same reference with -n.95 -p (near-optimal chains and the repeat profile track); also a
              three-block repeat-rich reference kept resident, two read blocks, with the track from a -T1 run
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


def _map_blocks_like_bench(d, ref_root, reads_root, out, report_mode=2, parts=None, overlap=True, **kw):
    """The block loop of bench.py / damapper_amd.dispatch, in process: resident REF indices, packed read blocks from
    the DB on disk, asynchronous Reporter, and -- overlap -- the Reporter call of a block made a few passes into the
    next one (Mapper.map_block(overlap=True)).  `parts`: the read blocks to map (1-based; default all)."""
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
        parts = list(parts) if parts is not None else list(range(1, nb + 1))
        blocks = [dazzdb.load_db("./" + reads_root + ".db", part=b, pinned=True, packed=True) for b in parts]
        L.dmh_async_report(report_mode)
        for x, blk in enumerate(blocks):
            nxt = blocks[x + 1] if x + 1 < len(blocks) else None
            m.map_block(blk, "%s.%d" % (reads_root, parts[x]), ref_blocks, ref_whole, ref_root, next_reads=nxt,
                        overlap=overlap, release=True)
        m.finish()
        L.dmh_report_wait()
        L.dmh_async_report(0)
        m.release_ref()
        L.dmh_retain_ref_index(0)
        L.dmh_release_block(ref_whole.ptr())
        return len(blocks)
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
    # two READS blocks that differ in the number of reads AND in the longest read (log-normal lengths in the second): the
    # byte counts of the seed key (map.c:2920-2936) and every per-block size change at the block boundary, at this size
    blocks = [dazzdb.simulate_block_genome(2000, G, 19000, tfirst=0, part=1),
              dazzdb.simulate_block_genome(2001, G, 6000, tfirst=19000, part=2, lengths="lognormal")]
    dazzdb.write_db_blocks(d, "READS", blocks)
    del blocks
    names = ["READS.1", "READS.2"]
    run_ref.run_reference(d, "REF", names, flags=["-k20", "-e.85", "-M64"], threads=_threads())
    assert _map_blocks_like_bench(d, "REF", "READS", "las_hip", mem_limit=64 << 30) == 2
    _compare(d, "las_ref", "las_hip", names, 25000)


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


@pytest.mark.parametrize("cluster,read_sort", [(-1, -1), (0, -1), (-1, 0), (-1, 2), (-1, 5), (-1, 6)],
                         ids=["auto", "pairs", "auto_global_seed_sort", "auto_large_workgroups", "auto_ballot_ranks",
                              "auto_no_1024_thread_workgroups"])
def test_one_family_repeat_stress(cluster, read_sort, tmp_path):
    run_ref = _run_ref()
    d = str(tmp_path)
    G = dazzdb.native_genome(23, [4_000_000, 4_000_000], "repeat", families=1)
    dazzdb.write_db_block(d, "REF", dazzdb.genome_block(G))
    dazzdb.write_db_block(d, "READS", dazzdb.simulate_block_genome(4000, G, 1500))
    run_ref.run_reference(d, "REF", ["READS.1"], flags=["-k20", "-e.85", "-M64"], threads=_threads())
    capi.lib().dmh_test_option(b"cluster", cluster)
    capi.lib().dmh_test_option(b"read_sort", read_sort)
    try:
        _map_blocks_like_bench(d, "REF", "READS", "las_hip", mem_limit=64 << 30)
    finally:
        capi.lib().dmh_test_option(b"cluster", -1)
        capi.lib().dmh_test_option(b"read_sort", -1)
    _compare(d, "las_ref", "las_hip", ["READS.1"], 1500)


# ---- BASELINE configs[3] / configs[4] at their size: the reference of bench.py's default workload -------------------

@pytest.fixture(scope="module")
def grch38_dbs(tmp_path_factory):
    """bench.py's GRCh38-sized reference (3.1 Gbp, 25 contigs, 9 REF blocks <= 375 Mbp, 496 repeat families) as a
    DAZZ_DB on disk, and a READS DB of three blocks: two of 10 kb reads (configs[3]) and one of 1.5 kb reads -- a read
    inside one copy of a repeat family maps to the family's other copies nearly as well: the near-optimal chains that
    -n.95 keeps and -p counts (configs[4]).  The reference's time is REF indexing (about 90 s per read block on 16
    cores, whatever the block holds), so the blocks are small."""
    import bench
    d = str(tmp_path_factory.mktemp("grch38"))
    clen, ncontig, kind, _ = bench.WORKLOADS["grch38"]
    lens = [clen] * ncontig
    ranges = bench.ref_block_ranges(lens)
    fam = max(1, int(round(max(sum(lens[a:b]) for a, b in ranges) / 750_000)))
    G = dazzdb.native_genome(20, lens, kind, families=fam)
    dazzdb.write_db_blocks(d, "REF", [dazzdb.genome_block(G, a, b) for a, b in ranges])
    blocks = [dazzdb.simulate_block_genome(5000, G, 4000, tfirst=0, part=1),
              dazzdb.simulate_block_genome(5001, G, 4000, tfirst=4000, part=2, lengths="lognormal"),
              dazzdb.simulate_block_genome(5002, G, 10000, read_len=1500, tfirst=8000, part=3)]
    dazzdb.write_db_blocks(d, "READS", blocks)
    del blocks, G
    return d, len(ranges)


def test_configs3_grch38_resident(grch38_dbs):
    """The pipeline bench.py times, at the metric's size: 18 resident REF indices, the passes of a read block batched
    and in flight, finish stream, deferred alignment kernel, prefetch of the next block -- against the reference's whole
    loop (damapper.c:825-880) over all 9 REF blocks.  The second block has CLR-like (log-normal) read lengths."""
    run_ref = _run_ref()
    d, nrb = grch38_dbs
    assert nrb == 9
    names = ["READS.1", "READS.2"]
    run_ref.run_reference(d, "REF", names, flags=["-k20", "-e.85", "-M64"], threads=_threads(), keep="las_ref3")
    _map_blocks_like_bench(d, "REF", "READS", "las_hip3", mem_limit=64 << 30, parts=[1, 2])
    _compare(d, "las_ref3", "las_hip3", names, 8000)


def test_configs4_grch38_resident_n95_p(grch38_dbs):
    run_ref = _run_ref()
    d, nrb = grch38_dbs
    names = ["READS.2", "READS.3"]
    flags = ["-k20", "-e.85", "-n.95", "-p", "-M64"]
    run_ref.run_reference(d, "REF", names, flags=flags, threads=_threads(), keep="las_ref4")
    prof_ref = {}
    for name in names:
        fs = [os.path.join(d, ".%s.prof.%s" % (name, x)) for x in ("anno", "data")]
        prof_ref[name] = tuple(open(f, "rb").read() for f in fs)
        for f in fs:
            os.remove(f)
    _map_blocks_like_bench(d, "REF", "READS", "las_hip4", mem_limit=64 << 30, best_tie=0.95, profile=1, parts=[2, 3])
    n = _compare(d, "las_ref4", "las_hip4", names, 12000)
    flagset = set()
    for name in names:
        flagset |= {r.flags for r in las.read_las_set(os.path.join(d, "las_hip4"), name + ".REF.M@.las")[1]}
    assert any(f & las.START_FLAG and not f & las.BEST_FLAG for f in flagset), "no near-optimal chain in %d records" % n
    # The -p track.  A -T1 reference run at this size would take half an hour, and with more than one report thread
    # the reference's special_log() (map.c:2270-2302) may search its table while another thread is still filling it:
    # entries not yet written read as 0, so such a call returns a value that is too LARGE -- only during the first calls
    # of a run.  Hence: equal, except for at most 128 of the track's 2.7 M bytes in which the reference's value is the
    # larger one (how many calls fall into that window differs from run to run: 0-32 in most, 40 seen once in round 4).
    for name, (anno, data) in prof_ref.items():
        assert open(os.path.join(d, ".%s.prof.anno" % name), "rb").read() == anno
        got = np.frombuffer(open(os.path.join(d, ".%s.prof.data" % name), "rb").read(), dtype=np.uint8)
        want = np.frombuffer(data, dtype=np.uint8)
        assert len(got) == len(want)
        bad = np.nonzero(got != want)[0]
        assert len(bad) <= 128 and np.all(want[bad] > got[bad]), "-p track of %s differs at %s" % (name, bad[:10])
        assert want.max() >= 10                      # repeat coverage shows


def test_bench_two_ranks_gloo_strong_scaling(tmp_path):
    """bench.py under torch.distributed.run with two ranks (gloo: the ranks share the one device of a 1-GPU box; on a
    node RCCL is only used for the barrier and the two reductions of the contract, the data path has no collective):
    the N > 1 control flow -- b mod N sharding, barrier, max-time / sum-bases reductions, one JSON line from rank 0 with
    the parity of the timed output."""
    import json
    import subprocess
    import sys
    _run_ref()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DMH_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", TMPDIR=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "ecoli", "--steps", "2",
           "--warmup", "1"]
    p = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 2
    assert out["config"]["read_blocks_timed"] == 2 * 8
    assert out["parity"] is not None and out["parity"]["differing"] == 0 and out["parity"]["records"] > 20000
    assert out["cpu_baseline"] is None              # (N = 1 only, by the contract)
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "ecoli", "--steps", "2", "--warmup", "1"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
    o1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert o1["n_gpus"] == 1 and o1["parity"]["differing"] == 0 and o1["parity"]["records"] == out["parity"]["records"]
    assert o1["config"]["read_bases_per_step"] == out["config"]["read_bases_per_step"]
    assert o1["cpu_baseline"] is not None and "cached" in o1["cpu_baseline"]["sample"]


def test_soak_deferred_pipeline_with_poisoned_allocations(tmp_path):
    """Twenty read blocks through the deferred pipeline (mode 2: batches in flight, finish stream, alignment kernel
    next to the following block's filter, pool blocks changing streams) in a process of its own with every device
    allocation pre-filled with a poison pattern (DMH_POISON): a kernel that reads memory it has not written, or a
    buffer handed to another stream too early, shows as a difference from the reference or as a fault."""
    import subprocess
    import sys
    _run_ref()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'oracle')); sys.path.insert(0, os.path.join(%r, 'tests'))
import run_ref
from damapper_amd import dazzdb, las
import test_gpu_scale as T
d = %r
G = dazzdb.native_genome(31, [6_000_000, 5_000_000, 4_000_000], 'repeat', families=20)
dazzdb.write_db_blocks(d, 'REF', [dazzdb.genome_block(G, k, k + 1) for k in range(3)])
blocks = [dazzdb.simulate_block_genome(6000 + b, G, 600 + 40 * (b %% 5), tfirst=sum(600 + 40 * (x %% 5) for x in range(b)), part=b + 1,
                                       lengths='lognormal' if b %% 2 else 'fixed') for b in range(20)]
dazzdb.write_db_blocks(d, 'READS', blocks)
names = ['READS.%%d' %% (b + 1) for b in range(20)]
run_ref.run_reference(d, 'REF', names, flags=['-k20', '-e.85', '-M64'], threads=T._threads())
T._map_blocks_like_bench(d, 'REF', 'READS', 'las_hip', mem_limit=64 << 30)
print('records', T._compare(d, 'las_ref', 'las_hip', names, 10000))
""" % (root, root, root, str(tmp_path))
    env = dict(os.environ, DMH_POISON="0xa5")
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "records" in p.stdout


def test_bench_four_ranks_gloo_rehearsal(tmp_path):
    """The N = 8 run of the driver's scaling bench must not be the first many-rank run: bench.py under torchrun with FOUR
    ranks on this box's one device (gloo).  A GPU box of this pool kills a job with more than six processes on its card --
    the test runner is one, the launcher's children and a finished run's stragglers count too: six ranks were killed by
    that guard, so four is what can be rehearsed here; the control flow has no place where 8 differs from 4: b mod N
    sharding of the fixed batch of 8 blocks, two blocks per rank (one per rank at N = 8, which is the per_step == 1 path the
    two-handle trick of bench.py exists for and test_block_sharding_two_ranks_gloo covers on the CPU).  Checks: one JSON
    line, n_gpus, the batch, every rank's kept files equal to the N = 1 run's (ports, temp dirs, pinned memory and the core
    quota are shared by the processes), parity of rank 0's block against the reference, and the CPU seconds per rank."""
    import json
    import subprocess
    import sys
    _run_ref()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    keep1, keepn = str(tmp_path / "keep1"), str(tmp_path / "keepn")
    base = [os.path.join(root, "bench.py"), "--workload", "ecoli", "--steps", "2", "--warmup", "1", "--block-reads", "4000"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TMPDIR=str(tmp_path), DMH_BENCH_KEEP=keep1)
    p = subprocess.run([sys.executable] + base + ["--gpus", "1"], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    env = dict(os.environ, DMH_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", TMPDIR=str(tmp_path), DMH_BENCH_KEEP=keepn, DMH_YIELD="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", "29531"] + base + ["--gpus", "4"]
    p = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 4 and out["scaling"] == "strong"
    assert out["config"]["read_blocks_timed"] == 2 * 8
    assert out["parity"]["differing"] == 0 and out["parity"]["records"] > 1000
    assert len(out["host_cpu_seconds_per_rank"]) == 4
    busy = out["host_cores_busy_per_rank"]
    assert sum(busy) <= 16.5, "four ranks keep %s cores busy" % busy      # (DMH_YIELD=1: the waits an 8-rank node would use)
    # every block's files: the four-rank run's equal the one-rank run's
    names = sorted(os.listdir(keep1))
    assert names == sorted(os.listdir(keepn)) and len(names) >= 8 * 4
    for f in names:
        assert open(os.path.join(keep1, f), "rb").read() == open(os.path.join(keepn, f), "rb").read(), f
