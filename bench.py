#!/usr/bin/env python3
"""bench.py -- Gbp of reads mapped per second through the HIP hot path, on the metric's configuration.

Workload (default `--workload grch38`, BASELINE.json configs[3]): a GRCh38-SIZED reference, 3.1 Gbp in 25 contigs
of 124 Mbp, cut into 9 REF blocks of <= 375 Mbp as DBsplit would (whole contigs, < 2^31 k-mers), built by SURVEY
8d's repeat-rich generator scaled up -- 2 kb repeat families at 3 % divergence on both strands every 1-6 kb plus a
60 x 150 bp tandem array per contig, ONE FAMILY PER 750 kb OF A REF BLOCK (496 families), which keeps the per-family
copy number each Match_Filter pass sees (about 136) and the seed-hit density (0.3-0.4 hits per read base and
orientation, 12-16 x a random genome) of the survey's one-family 750 kb probe genome; there is no network for the
real assembly.  Reads: simulated PacBio CLR, 10 kb, 15 % error (ins:del:sub 60:25:15), uniform over the genome and
both strands, in DBsplit-sized blocks of 25,000 reads (0.26 Gbp).  `--workload grch38-stress` is the same model
with ONE family genome-wide (560 k copies: the generator scaled up literally; ~3.5 hits per read base and
orientation), `chr1` a 250 Mbp one-block reference, `ecoli` the 4.6 Mbp random reference of configs[1].

A STEP maps a fixed batch of `--blocks-per-step` (8) read blocks: for every block Sort_Kmers(reads), Match_Filter
against both orientations of every REF block (18 passes; the REF indices are built once per GPU and stay in HBM,
BASELINE north_star) and Reporter (.las written), i.e. damapper.c:825-880 per block.  The batch is FIXED and its
blocks are dealt to the ranks b mod N ("strong" scaling; no collective on the data path, every rank holds the whole
reference index).  Read blocks are NOT resident: every visit uploads the block's 2-bit .bps bytes from page-locked
host memory and decodes them on the device inside the timed region (the steady state of the block loop).

After the timed region (rank 0) the compiled reference (oracle/_ref/damapper) maps read block 1 of the batch
through its own whole loop -- all REF blocks, both orientations, Reporter -- on the host cores: its wall time is
`cpu_baseline` (measured, nothing extrapolated), and its per-thread .las files are compared record for record with
the files the LAST TIMED STEP wrote for that block -> `parity` (the timed pipeline's own output is what is checked).

`roofline` is reported for the kernel with the largest share of the timed GPU time (HIP events on the stream each
kernel runs on), against its algorithmic bytes (the reference's 16-byte records); `hbm_stage` repeats
it for the HBM-bound radix scatter kernel when that is not the dominant one.
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")      # (before anything touches the GPU: damapper_amd/__init__.py)


ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from damapper_amd import capi, dazzdb, driver, las  # noqa: E402

READ_LEN = 10_000
HBM_PEAK_GBS = 8000.0
REF_BLOCK_MAX = 375_000_000

WORKLOADS = {
    # name: (contig length, contigs, kind, families per genome (0 = one per 750 kb of a REF block))
    "grch38": (124_000_000, 25, "repeat", 0),
    "grch38-stress": (124_000_000, 25, "repeat", 1),
    "grch38-random": (124_000_000, 25, "random", 1),
    "chr1": (125_000_000, 2, "repeat", 0),
    "ecoli": (4_600_000, 1, "random", 1),
}

# --flags: the mapper's options on both sides (BASELINE configs[3] = default, configs[4] = n95p: near-optimal chains and
# the -p repeat profile)
FLAG_SETS = {
    "default": {"ref": [], "mapper": {}, "profile": False, "text": ""},
    "n95p": {"ref": ["-n.95", "-p"], "mapper": {"best_tie": 0.95, "profile": 1}, "profile": True, "text": " -n.95 -p"},
}

# kernels whose time is HBM streaming: algorithmic bytes are meaningful as a bandwidth
STAGE_KERNEL = {
    "k_radix_scatter": "k_radix_scatter", "k_radix_hist": "k_radix_hist", "radix_scan": "k_scan_*",
    "kmer_emit": "k_kmer_emit", "merge_match": "k_merge_match", "merge_match2": "k_merge_match (both orientations of a REF block in one sweep)", "merge_fixup": "k_merge_fixup (+reduce, scan)",
    "merge_emit": "k_merge_expand (+sizes, scan)", "merge_rows": "k_row_list (+scan, classes)",
    "seed_groups": "k_group_flag/write (+scan)", "chain_small_groups": "k_chain<16>", "chain_listed_groups": "k_chain<64>",
 "chain_lane_groups": "k_chain_lane", "chain_simple_groups": "k_chain_simple", "chain_lds_groups": "k_chain_small",
    "chain_class_groups": "k_chain_classes", "chain_piles": "k_chain_pile", "chain_dominate": "k_order_summaries + k_dominate (+ scans)", "seed_clusters": "k_cluster_flag/rekey (+scan)",
    "seed_pairs": "k_pair_flag/write (+scan)", "chain_spilled_piles": "k_chain<64> (spilled piles)", "chain_jumps": "k_chain_jumps",
    "align": "k_align_cands32", "align_wave64": "k_align_cands<128>", "align_global_band": "k_align_cands<8192>",
    "read_sort": "k_read_sort (fused: rows -> ordered hits -> plain chains)", "unpack_2bit": "k_unpack_2bit", "pack_bases": "k_pack_fwd", "pack_revcomp": "k_pack_rc", "merge_buckets": "k_bucket_bounds",
}
# enclosing stages (their kernels are timed one by one as well): not candidates for "dominant kernel"
ENCLOSING = {"kmer_sort", "seed_sort", "cluster_sort", "chain", "sort_pairs"}
SIDE_STREAM = {"chain_piles"}
# stages that are ONE kernel on the main stream: the candidates for "the dominant kernel" (the finish stream's stages --
# chain_jumps, chain_dominate -- and the alignment stream's run next to these, a dozen small kernels each)
SINGLE_MAIN = {"merge_match", "merge_match2", "chain_simple_groups", "chain_lane_groups", "chain_small_groups", "chain_listed_groups",
               "chain_lds_groups", "k_radix_scatter", "k_radix_hist", "kmer_emit", "merge_buckets"}
# kernel symbols as rocprofv3 names them, for the PMC traffic summary under profiles/
PMC_SYMBOL = {"k_radix_scatter": "dmh::k_radix_scatter", "k_radix_hist": "dmh::k_radix_hist", "chain_lane_groups": "dmh::k_chain_lane",
              "kmer_emit": "dmh::k_kmer_emit", "chain_piles": "dmh::k_chain_pile", "read_sort": "dmh::k_read_sort",
              "merge_match": "dmh::k_merge_match", "merge_match2": "dmh::k_merge_match", "align": "dmh::k_align_cands32", "chain_simple_groups": "dmh::k_chain_simple"}


def ref_block_ranges(lens):
    """REF blocks as DBsplit cuts them: whole contigs, at most REF_BLOCK_MAX bases per block."""
    out, first, acc = [], 0, 0
    for ci in range(len(lens) + 1):
        if ci == len(lens) or (acc > 0 and acc + int(lens[ci]) > REF_BLOCK_MAX):
            out.append((first, ci))
            first, acc = ci, 0
        if ci < len(lens):
            acc += int(lens[ci])
    return out


def shard_blocks(nblocks, rank, world):
    """Read blocks are independent units: block b goes to rank b mod world."""
    return [b for b in range(nblocks) if b % world == rank]


def host_threads():
    """(cores this process may use, largest power of two <= that): the affinity mask capped by the cgroup's CPU
    quota (a 1-GPU box of this pool shows 256 CPUs in the mask and a cpu.max of 16 cores: beyond the quota more
    threads only time-slice).  The reference rounds -T down to a power of two (map.c:142-147) and its own radix
    sort crashes for -T >= 64, so the power of two is capped at 32."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    p = 1
    while 2 * p <= min(n, 32):
        p *= 2
    return n, p


def parity_and_cpu_baseline(G, ranges, args, sort_path, threads, cores, world):
    """The reference's whole block loop (damapper.c:825-880) on read block 1 of the timed batch: `damapper -k20
    -e.85 -T<threads> -M<mem_gb> REF READS.1` against ALL REF blocks in both orientations, wall time of the process
    (LAsort/LAcat are shims) -> cpu_baseline, measured, nothing extrapolated.  Its per-thread .las files
    are then compared record for record with the files the LAST TIMED STEP of this run left for the same block
    (resident REF indices, batched passes, finish stream, deferred alignment kernel: the pipeline that was timed)
    -> parity.  The reference's files are cached under the temp directory (keyed by the workload), so that the N > 1
    runs on the same box do not spend the CPU time again.  Returns (cpu_baseline, parity)."""
    try:
        import run_ref
    except Exception:
        return None, None
    if not run_ref.have_ref():
        return None, None
    import hashlib

    def file_sha(path):
        try:
            return hashlib.sha1(open(path, "rb").read()).hexdigest()
        except OSError:
            return "-"
    # everything the reference's files and its wall time depend on: the workload, the flags, -T (the per-thread file
    # split and the time), the reference binary and the simulator that makes the inputs
    ref_flags = ["-k20", "-e.85", "-M%d" % args.mem_gb] + FLAG_SETS[args.flags]["ref"]
    key = hashlib.sha1(repr((args.workload, args.genome_mbp, args.families, args.block_reads, args.mem_gb, args.read_lengths,
                             [int(x) for x in G.lens], ref_flags, threads, args.ref_fasta,
                             file_sha(os.path.join(run_ref.REF_DIR, "damapper")),
                             file_sha(os.path.join(ROOT, "damapper_amd", "libdmh_sim.so")))).encode()).hexdigest()[:16]
    cache = os.path.join(tempfile.gettempdir(), "dmh_bench_refcache_%s" % key)
    cpu = None
    if args.refresh_ref_cache:
        shutil.rmtree(cache, ignore_errors=True)
    if not os.path.exists(os.path.join(cache, "cpu.json")):
        d = tempfile.mkdtemp(prefix="dmh_bench_cpu_")
        try:
            dazzdb.write_db_blocks(d, "REF", [dazzdb.genome_block(G, a, b) for a, b in ranges])
            sample = simulate_read_block(G, 0, args)
            dazzdb.write_db_block(d, "READS", sample)
            bases = float(sample.db.totlen)
            del sample
            dt, _ = run_ref.run_reference(d, "REF", ["READS.1"], flags=ref_flags, threads=threads)
            cpu = {"value": bases / dt / 1e9, "unit": "Gbp/s", "cores": threads, "kind": "reference",
                   "host_cores_available": cores, "cached": False,
                   "sample": "oracle/_ref/damapper " + " ".join(ref_flags[3:]) + " -k20 -e.85 -T%d -M%d REF READS.1: read block 1 of the timed batch (%d reads, "
                             "%.0f Mbp) through the reference's whole loop damapper.c:825-880 -- Sort_Kmers + Match_Filter "
                             "against all %d REF blocks (%.0f Mbp) in both orientations, Reporter -- in %.1f s wall, DB "
                             "loading included, measured (no extrapolation); -T is the largest power of two <= the host "
                             "cores, capped at 32 (the reference's radix sort crashes for -T >= 64)"
                             % (threads, args.mem_gb, args.block_reads, bases / 1e6, len(ranges), G.total / 1e6, dt)}
            os.makedirs(cache, exist_ok=True)
            for f in os.listdir(os.path.join(d, "las_ref")):
                shutil.copy(os.path.join(d, "las_ref", f), os.path.join(cache, f))
            for x in ("anno", "data"):                      # (-p: the reference leaves the track next to the DBs)
                f = os.path.join(d, ".READS.1.prof.%s" % x)
                if os.path.exists(f):
                    shutil.copy(f, os.path.join(cache, "ref.prof.%s" % x))
            json.dump(cpu, open(os.path.join(cache, "cpu.json"), "w"))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    else:
        cpu = json.load(open(os.path.join(cache, "cpu.json")))
        cpu["cached"] = True                                # (not measured in THIS run: --refresh-ref-cache measures again)
        cpu["sample"] += " [cached from an earlier run on this box]"
    ts_r, want = las.read_las_set(cache, "READS.1.REF.M@.las")
    ts_h, got = las.read_las_set(sort_path, "READS.1.REF.M@.las")
    bad = sum(1 for x, y in zip(want, got) if x != y) + abs(len(want) - len(got)) + (0 if ts_r == ts_h else 1)
    parity = {"records": len(want), "records_hip": len(got), "differing": bad,
              "what": "timed output, %d REF blocks: the per-thread .las files the last timed step wrote for read block 1 "
                      "(report mode %d, REF indices %s, %d ranks) against the reference's files for the same block; record "
                      "streams compared record for record, bytes 36-39 (struct padding) masked"
                      % (len(ranges), args.report_mode, "rebuilt" if args.rebuild_ref else "resident", world)}
    if FLAG_SETS[args.flags]["profile"]:
        # the -p track of the same block.  With more than one report thread the reference's special_log() (map.c:2270-2302)
        # may read its lazily filled table while another thread fills it (entries not yet written read as 0: a value that
        # is too LARGE, only during the first calls of a run), and a -T1 run of this size takes half an hour: equal, except
        # for at most 128 bytes in which the reference's value is the larger one (tests/test_gpu_scale.py pins the track
        # exactly against -T1 at test size)
        import numpy as np
        try:
            anno_r = open(os.path.join(cache, "ref.prof.anno"), "rb").read()
            want = np.frombuffer(open(os.path.join(cache, "ref.prof.data"), "rb").read(), dtype=np.uint8)
            anno_h = open(".READS.1.prof.anno", "rb").read()
            got = np.frombuffer(open(".READS.1.prof.data", "rb").read(), dtype=np.uint8)
            diff = np.nonzero(got != want)[0] if len(got) == len(want) else None
            ok = anno_r == anno_h and diff is not None and len(diff) <= 128 and bool(np.all(want[diff] > got[diff]))
            parity["prof_track"] = {"bytes": int(len(want)), "differing": int(len(diff)) if diff is not None else -1, "ok": ok,
                                    "rule": "identical but for <= 128 bytes in which the -T%d reference's racy special_log() value is the larger" % threads}
            if not ok:
                parity["differing"] += 1
        except OSError as e:
            parity["prof_track"] = {"ok": False, "error": str(e)}
            parity["differing"] += 1
    return (cpu if world == 1 else None), parity


def simulate_read_block(G, b, args, pinned=False):
    """Read block b of the fixed batch (seeded: the same block whenever it is asked for)."""
    return dazzdb.simulate_block_genome(1000 + b, G, args.block_reads, read_len=READ_LEN, tfirst=b * args.block_reads,
                                        part=b + 1, path=b"./.READS", pinned=pinned, lengths=args.read_lengths)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="grch38")
    ap.add_argument("--genome-mbp", type=float, default=None, help="override the reference size (contig count scales)")
    ap.add_argument("--families", type=int, default=None, help="override the number of repeat families")
    ap.add_argument("--block-reads", type=int, default=25000)
    ap.add_argument("--blocks-per-step", type=int, default=8,
                    help="read blocks in the fixed batch one step maps (dealt to the ranks b mod N)")
    ap.add_argument("--read-lengths", choices=["fixed", "lognormal"], default="fixed",
                    help="template lengths: fixed 10 kb (BASELINE north_star) or CLR-like log-normal with mean 10 kb "
                         "(sigma 0.4, clipped to 1-40 kb; SURVEY 8d for configs[3]/[4])")
    ap.add_argument("--flags", choices=sorted(FLAG_SETS), default="default",
                    help="n95p = BASELINE configs[4]: -n.95 (near-optimal chains) and -p (repeat profile track, compared too)")
    ap.add_argument("--ref-fasta", default=None,
                    help="map against this FASTA (a real genome on the box's disk: contigs as fasta2DAM cuts them, REF blocks of "
                         "whole contigs <= 375 Mbp) instead of the synthetic reference; reads are still simulated from it")
    ap.add_argument("--refresh-ref-cache", action="store_true",
                    help="run the CPU reference again even if this box has its files for the same inputs cached")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the reference run (then there is no parity check either)")
    ap.add_argument("--mem-gb", type=int, default=64, help="-M of both sides (the limit rule of map.c:2992-3055 depends on it)")
    ap.add_argument("--roofline-only", action="store_true",
                    help="HIP events around k_radix_scatter only (default: around every kernel, about 1 ms per read block)")
    ap.add_argument("--no-stage-timing", action="store_true", help="no HIP events at all (no roofline)")
    ap.add_argument("--no-alone", action="store_true", help="skip the extra untimed visit that measures the kernels with the GPU to themselves")
    ap.add_argument("--event-blocks", type=int, default=8,
                    help="one read block in this many of a rank's blocks carries the HIP events of the stage timers (0 or 1: every block)")
    ap.add_argument("--no-prefetch", action="store_true", help="do not queue the next block's index build early")
    ap.add_argument("--no-overlap", action="store_true",
                    help="Reporter of a block right behind its last pass (default: a few passes into the next block, "
                         "Mapper.map_block(overlap=True): the GPU does not idle while the block's last batches finish)")
    ap.add_argument("--resident-reads", action="store_true",
                    help="read blocks pre-uploaded and pre-packed (inputs resident; the round-1 bench configuration)")
    ap.add_argument("--report-mode", type=int, default=2, choices=[0, 1, 2],
                    help="0 = Reporter synchronous, 1 = .las selection/writing on a worker thread, 2 (default) = also the "
                         "alignment kernel of block i deferred next to the filter of block i+1 (one-block references only)")
    ap.add_argument("--rebuild-ref", action="store_true",
                    help="rebuild the REF indices for every read block as damapper.c does")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_mod
        torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        # RCCL ("nccl") on a node with one GPU per rank; DMH_BENCH_BACKEND=gloo only serves to rehearse the
        # N > 1 control flow on a box with fewer GPUs than ranks (the ranks then share a device)
        dist_mod.init_process_group(backend=os.environ.get("DMH_BENCH_BACKEND", "nccl"))
        dist = dist_mod
    cores, tpow = host_threads()
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, cores // max(1, world))))

    L = capi.lib()
    if L.dmh_device_count() < 1:
        raise SystemExit("bench.py needs a GPU: libdamapper_hip has no CPU fallback")

    # ---- synthetic workload (seeded): the reference is the same on all ranks, the batch of read blocks is fixed
    t_setup = time.time()
    clen, ncontig, kind, fam = WORKLOADS[args.workload]
    if args.genome_mbp is not None:
        ncontig = max(1, int(round(args.genome_mbp * 1e6 / clen)))
        if args.genome_mbp * 1e6 < clen:
            clen = int(args.genome_mbp * 1e6)
    lens = [clen] * ncontig
    ranges = ref_block_ranges(lens)
    if fam == 0:
        fam = max(1, int(round(max(sum(lens[a:b]) for a, b in ranges) / 750_000)))
    if args.families is not None:
        fam = args.families
    if args.ref_fasta:
        G = dazzdb.genome_from_fasta(args.ref_fasta, min_len=4 * READ_LEN if args.read_lengths == "lognormal" else READ_LEN)
        lens = [int(x) for x in G.lens]
        ncontig, kind, fam = len(lens), "fasta", 0
        ranges = ref_block_ranges(lens)
    else:
        G = dazzdb.native_genome(20, lens, kind, families=fam)
    ref_whole = dazzdb.genome_block(G, path=b"./.REF")
    freq = [ref_whole.db.freq[j] for j in range(4)]

    sort_path = tempfile.mkdtemp(prefix="dmh_bench_%d_" % rank)
    mapper_kw = dict(kmer=20, ave_corr=0.85, spacing=100, threads=4, **FLAG_SETS[args.flags]["mapper"])
    if FLAG_SETS[args.flags]["profile"]:
        os.chdir(sort_path)               # (the -p track goes to the working directory, map.c:3301-3302)
    m = driver.Mapper(sort_path=sort_path, mem_limit=args.mem_gb << 30, retain_ref=not args.rebuild_ref, **mapper_kw)
    m.make_spec(freq)
    L.dmh_preload_block(ref_whole.ptr())          # Reporter's copy of the whole reference: resident
    L.dmh_prepare_block(ref_whole.ptr(), 0)
    ref_blocks = []
    for k, (a, b) in enumerate(ranges):
        blk = dazzdb.genome_block(G, a, b, part=k + 1, path=b"./.REF", copy=True, freq=freq)
        ref_blocks.append(blk)
        if not args.rebuild_ref:
            m.add_ref_block(blk, k, drop_bases=True)      # (one block's host copy at a time)
    t_ref = time.time() - t_setup

    nbatch = args.blocks_per_step
    mine = shard_blocks(nbatch, rank, world)
    blocks = {}
    for b in mine:
        raw = simulate_read_block(G, b, args, pinned=args.resident_reads)
        if args.resident_reads:
            blocks[b] = (raw, raw)
            L.dmh_preload_block(raw.ptr())
            L.dmh_prepare_block(raw.ptr(), 1)
        else:
            # two handles on one page-locked slice: a rank that owns a single block of the batch can then
            # prefetch its next visit like any other (the handle is the library's identity of a block)
            h0 = dazzdb.pack_block(raw, pinned=True)
            h0.slice = h0.packed
            h1 = dazzdb.make_block([], tfirst=int(h0.db.tfirst), part=int(h0.db.part), path=h0.path,
                                   freq=[h0.db.freq[j] for j in range(4)], lens=h0.reads["rlen"][1:h0.nreads + 1])
            h1.slice, h1.coff = h0.slice, h0.coff
            blocks[b] = (h0, h1)
            del raw
    L.dmh_sync()
    t_setup = time.time() - t_setup

    seq = [b for _ in range(args.warmup + args.steps) for b in mine]       # this rank's visits, in order
    per_step = len(mine)

    def visit(i):
        blk = blocks[seq[i]][(i // per_step) & 1]
        nxt = blocks[seq[i + 1]][((i + 1) // per_step) & 1] if (i + 1 < len(seq) and not args.no_prefetch) else None
        if nxt is blk:
            nxt = None
        if not args.resident_reads:
            for bb in (blk, nxt):                # map_block uploads (2-bit) and decodes what carries a slice
                if bb is not None and not bb.resident:
                    bb.packed = bb.slice
        t = m.map_block(blk, "READS.%d" % (seq[i] + 1), ref_blocks, ref_whole, "REF", next_reads=nxt,
                        overlap=not args.no_overlap, release=not args.resident_reads)
        return t

    # the warm-up runs the configuration that is timed (worker thread, streams, pinned pools and the allocator's
    # blocks are created by it, not inside the timed region)
    L.dmh_async_report(args.report_mode)
    for i in range(args.warmup * per_step):
        visit(i)
    m.finish()
    L.dmh_report_wait()
    tlevel = 0 if args.no_stage_timing else (2 if args.roofline_only else 1)
    t_from = int(os.environ.get("DMH_BENCH_TIMING_FROM", "0"))       # (experiment: events only from this timed step on)
    L.dmh_timing_enable(tlevel if t_from == 0 else 0)
    L.dmh_timing_reset()
    if dist is not None:
        import torch
        dist.barrier()
        torch.cuda.synchronize()
    L.dmh_sync()
    t0 = time.time()
    cpu0 = time.process_time()            # (CPU time of this rank's process, all its threads: the node's core budget at N ranks)
    bases = 0
    host = {"index_reads": 0.0, "filter": 0.0, "report": 0.0, "between_blocks": 0.0}
    step_t = []
    # HIP events around every stage of every launch cost 3 % of a step (11 us per pair of records, 3,600 pairs per
    # step): by default ONE read block in eight of a rank's timed visits carries them -- every launch of that block, on
    # the streams the kernels run on -- and the others none (--event-blocks 0: all of them)
    ev_every = args.event_blocks if args.event_blocks > 0 else 1
    sampled_blocks = 0
    for i in range(args.warmup * per_step, len(seq)):
        k = i - args.warmup * per_step
        if k % per_step == 0:
            step_t.append(time.time())
        if t_from == 0 or len(step_t) - 1 >= t_from:
            nvis = len(seq) - args.warmup * per_step
            on = k % ev_every == min(ev_every // 2, nvis - 1)      # (of the rank's visits, whatever the number of ranks)
            sampled_blocks += 1 if (on and tlevel) else 0
            L.dmh_timing_enable(tlevel if on else 0)
        t = visit(i)
        for k in host:
            host[k] += t[k]
        bases += int(blocks[seq[i]][0].db.totlen)
    m.finish()
    L.dmh_report_wait()
    L.dmh_sync()
    if dist is not None:
        import torch
        torch.cuda.synchronize()
        dist.barrier()
    dt = time.time() - t0
    cpu_s = time.process_time() - cpu0
    if os.environ.get("DMH_BENCH_STEP_TIMES") and rank == 0:
        step_t.append(time.time())
        sys.stderr.write("step times (ms): %s\n" % " ".join("%.1f" % ((b - a) * 1e3) for a, b in zip(step_t, step_t[1:])))

    tot_bases = float(bases)
    if dist is not None:
        import torch
        dev = "cpu" if dist.get_backend() == "gloo" else "cuda"
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tb = torch.tensor([tot_bases], dtype=torch.float64, device=dev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        tot_bases = float(tb.item())
        tc = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(tc, torch.tensor([cpu_s], dtype=torch.float64, device=dev))
        cpu_per_rank = [round(float(x.item()), 3) for x in tc]
    else:
        cpu_per_rank = [round(cpu_s, 3)]

    timings = capi.timings()
    L.dmh_timing_enable(0)
    # The timed pipeline runs the kernels of a pass's second half next to the following pass's merge kernels (tail
    # stream): a kernel's duration there is not what it takes with the GPU to itself.  One more visit of the rank's
    # LAST block, outside the clock, with nothing moved to the tail stream and events on every launch: the "alone"
    # figures next to the live ones in `roofline`.
    alone = {}
    if tlevel == 1 and per_step > 1 and not args.no_alone:
        L.dmh_test_option(b"tails", 0)
        L.dmh_timing_reset()
        L.dmh_timing_enable(1)
        blk = blocks[mine[-1]][0]
        if not args.resident_reads and not blk.resident:
            blk.packed = blk.slice
        m.map_block(blk, "READS.%d" % (mine[-1] + 1), ref_blocks, ref_whole, "REF", overlap=False, release=not args.resident_reads)
        L.dmh_report_wait()
        L.dmh_sync()
        alone = capi.timings()
        L.dmh_timing_enable(0)
        L.dmh_test_option(b"tails", -1)
    if os.environ.get("DMH_BENCH_DUMP_TIMERS") and rank == 0:
        for k, v in sorted(timings.items(), key=lambda kv: -kv[1]["ms"]):
            sys.stderr.write("timer %-28s %10.2f ms %8d calls\n" % (k, v["ms"], v["launches"]))

    if rank == 0:
        value = tot_bases / dt / 1e9
        kern = {k: v for k, v in timings.items() if not k.startswith("host:") and not k.startswith("count:")
                and k not in ENCLOSING and v["ms"] > 0}
        counters = {k[6:]: int(v["launches"]) for k, v in timings.items() if k.startswith("count:")}
        gpu_ms = sum(v["ms"] for v in kern.values())

        def roof_of(name):
            v = kern.get(name)
            if not v or v["launches"] <= 0 or v["ms"] <= 0 or v["bytes"] <= 0:
                return None
            achieved = v["bytes"] / (v["ms"] * 1e-3) / 1e9          # GB/s, algorithmic bytes
            r = {"bound": "hbm", "kernel": STAGE_KERNEL.get(name, name), "achieved": round(achieved, 4 if achieved < 10 else 1),
                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": float("%.3g" % (achieved / HBM_PEAK_GBS)), "traffic": None,
                 "launches": int(v["launches"]), "avg_ms": round(v["ms"] / v["launches"], 4),
                 "alg_bytes_per_launch": v["bytes"] / v["launches"],
                 "share_of_gpu_time": round(v["ms"] / gpu_ms, 3) if gpu_ms > 0 else None}
            # HBM bytes of this kernel from the PMC passes committed under profiles/ (FETCH_SIZE doubled as
            # MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as is): per record for the scatter kernel (whose
            # launches differ in size), per launch -- of the same workload -- for the others
            try:
                import glob
                latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fetch_write_summary.json")))[-1]
                pm = json.load(open(latest))
                if name == "k_radix_scatter":
                    r["traffic"] = round(pm["derived"]["scatter_traffic_bytes_per_record"] * (v["bytes"] / v["launches"] / 32.0))
                elif pm.get("workload") == args.workload and PMC_SYMBOL.get(name) in pm.get("kernels", {}):
                    r["traffic"] = round(pm["kernels"][PMC_SYMBOL[name]]["traffic_bytes_per_launch"])
            except Exception:
                pass
            return r

        # k_chain_pile walks one repeat pile per WORKGROUP (a handful of workgroups per launch, on a stream of their
        # own, next to the following passes' kernels): its duration is latency, not machine time, so it is reported
        # separately and the dominant kernel is picked among the ones that have the GPU to themselves
        side = {k: kern.pop(k) for k in list(kern) if k in SIDE_STREAM}
        single = {k: v for k, v in kern.items() if k in SINGLE_MAIN}
        dominant = max(single, key=lambda k: single[k]["ms"]) if single else None
        roof = roof_of(dominant) if dominant else None
        if roof is None and dominant:
            v = kern[dominant]
            roof = {"bound": "hbm", "kernel": STAGE_KERNEL.get(dominant, dominant), "achieved": None, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": None, "traffic": None, "launches": int(v["launches"]),
                    "avg_ms": round(v["ms"] / max(1, v["launches"]), 4),
                    "share_of_gpu_time": round(v["ms"] / gpu_ms, 3) if gpu_ms > 0 else None}
        if roof is not None and dominant in alone and alone[dominant]["launches"] > 0 and alone[dominant]["ms"] > 0 and alone[dominant]["bytes"] > 0:
            v = alone[dominant]
            ach = v["bytes"] / (v["ms"] * 1e-3) / 1e9
            roof["alone"] = {"achieved": round(ach, 1), "frac": float("%.3g" % (ach / HBM_PEAK_GBS)), "launches": int(v["launches"]),
                             "avg_ms": round(v["ms"] / v["launches"], 4),
                             "note": "the same kernel on the same inputs with the GPU to itself: one more read block mapped outside "
                                     "the clock with nothing on the side streams (in the timed pipeline it runs on the front stream "
                                     "next to the per-read kernels of the REF block before -- which fill the vector ALUs -- and the "
                                     "previous passes' ordering and chain kernels, which is what `achieved` measures; "
                                     "DMH_FRONT_AHEAD=0 puts it back on the main stream: shorter launches, longer read blocks)"}
        hbm_stage = roof_of("k_radix_scatter") if dominant != "k_radix_scatter" else None
        stages = {k: round(v["ms"], 2) for k, v in sorted(timings.items(), key=lambda kv: -kv[1]["ms"]) if not k.startswith("count:")}

        # the whole path against SURVEY 8(d)'s algorithmic bytes per read block (the reference's 16-byte records):
        # N_r + 16 K_r (1 + 2 P_k) + 16 K_r G + 16 H (1 + 2 P_s + 1), P_k = 5 byte passes of the k-mer sort, P_s = 5 of
        # the seed sort (2 bytes of maxlen, 1 of the REF block's contigs, 2 of the READS block's reads), G = passes,
        # H = hits of the block (from the expansion stage of the sampled blocks)
        path_roof = None
        nblk_total = args.steps * nbatch
        hit_stage = "merge_rows" if "merge_rows" in timings else "merge_emit"     # (fused route / round 3's: 16 B per hit either way)
        if sampled_blocks > 0 and hit_stage in timings and nblk_total > 0:
            n_r = tot_bases / (args.steps * nbatch)
            k_r = n_r - args.block_reads * 19.0
            hits = (timings[hit_stage]["bytes"] + (timings["merge_emit"]["bytes"] if hit_stage == "merge_rows" and "merge_emit" in timings else 0.0)) / 16.0 / sampled_blocks
            g = 2 * len(ranges)
            b_alg = n_r + 16.0 * k_r * 11 + 16.0 * k_r * g + 16.0 * hits * 12
            sec = dt / nblk_total * world
            path_roof = {"bound": "hbm", "alg_bytes_per_read_block": round(b_alg), "hits_per_read_block": round(hits),
                         "achieved": round(b_alg / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": float("%.3g" % (b_alg / sec / 1e9 / HBM_PEAK_GBS)),
                         "formula": "SURVEY 8(d): N_r + 16 K_r (1 + 2*5) + 16 K_r G + 16 H (1 + 2*5 + 1) per read block / wall time per read block"}

        cpu, parity = None, None
        if not args.no_cpu_baseline:
            cpu, parity = parity_and_cpu_baseline(G, ranges, args, sort_path, tpow, cores, world)
        nblk_total = args.steps * nbatch
        out = {
            "metric": "Gbp reads mapped/sec (whole node) on a GRCh38-sized reference, -k20 -e.85; .las parity-checked vs CPU reference"
                      if args.workload.startswith("grch38") else
                      "Gbp reads mapped/sec (whole node), -k20 -e.85; .las parity-checked vs CPU reference",
            "value": round(value, 4), "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / max(1, args.steps) * 1e3, 2), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u64/u8 integer", "data": "synthetic" if not args.ref_fasta else "real reference (FASTA), simulated reads",
            "config": {"workload": "%s: %.0f Mbp %s reference (%d contigs, %d REF blocks <= 375 Mbp, %s) + simulated PacBio "
                                   "reads (10 kb, 15%% err) in blocks of %d reads; a step = a fixed batch of %d read blocks "
                                   "dealt to the ranks b mod N, each through %d Match_Filter passes + Reporter; -k20 -e.85 "
                                   "-s100 -M%d%s; REF index %s; read blocks %s" %
                                   (args.workload, G.total / 1e6, kind, ncontig, len(ranges),
                                    ("%d repeat families" % fam) if kind == "repeat" else ("FASTA " + os.path.basename(args.ref_fasta) if kind == "fasta" else "uniform ACGT"),
                                    args.block_reads, nbatch, 2 * len(ranges), args.mem_gb, FLAG_SETS[args.flags]["text"],
                                    "rebuilt per read block" if args.rebuild_ref else "built once per GPU, resident",
                                    "resident in HBM" if args.resident_reads else
                                    "uploaded as 2-bit .bps bytes from page-locked memory and decoded on the device inside the timed region"),
                       "read_bases_per_step": int(tot_bases / max(1, args.steps)), "read_blocks_timed": nblk_total,
                       "ms_per_read_block": round(dt / max(1, nblk_total) * world * 1e3, 2),
                       "parallelism": "read blocks sharded x%d" % world,
                       "setup_seconds": {"reference_and_indices": round(t_ref, 1), "total": round(t_setup, 1)}},
            "roofline": roof, "hbm_stage": hbm_stage, "path_roofline": path_roof,
            "stage_events": {"read_blocks_with_events": sampled_blocks, "read_blocks_timed": nblk_total // max(1, world),
                             "note": "HIP events (hipEventRecord on the kernels' own streams) around every stage of every "
                                     "launch of the sampled read blocks, inside the timed region; launches / ms in roofline, "
                                     "hbm_stage and device_stage_ms are those blocks' (events on every block: --event-blocks 0, "
                                     "3 % slower)"},
            "side_stream_kernels": {STAGE_KERNEL.get(k, k): {"ms": round(v["ms"], 2), "launches": int(v["launches"]),
                                                            "avg_ms": round(v["ms"] / max(1, v["launches"]), 3),
                                                            "alg_bytes_per_launch": v["bytes"] / max(1, v["launches"]),
                                                            "note": "one workgroup per repeat pile, sequential walk (latency-bound), "
                                                                    "runs next to the following passes' kernels"}
                                    for k, v in side.items()},
            "cpu_baseline": cpu, "parity": parity,
            "host_seconds": {k: round(v, 3) for k, v in host.items()},
            # CPU seconds of every rank's process over the timed region (all threads): cores busy per rank = this / wall
            "host_cpu_seconds_per_rank": cpu_per_rank,
            "host_cores_busy_per_rank": [round(c / dt, 2) for c in cpu_per_rank],
            "device_stage_ms": stages,
            # the same stages with the GPU to themselves (the extra untimed visit of ONE read block with nothing on the tail
            # stream): what a read block costs in machine time when nothing overlaps
            "device_stage_ms_alone_one_block": {k: round(v["ms"], 2) for k, v in sorted(alone.items(), key=lambda kv: -kv[1]["ms"])
                                                if not k.startswith("count:") and not k.startswith("host:") and k not in ENCLOSING},
            # the d-wave ladder on the sampled blocks: candidates each rung ran, and why some had to run again one rung up
            "alignment_ladder": counters,
        }
        print(json.dumps(out))
        sys.stdout.flush()
        if parity is not None and parity["differing"] != 0:
            raise SystemExit("bench.py: HIP .las differs from the reference's on the parity sample")
    keep = os.environ.get("DMH_BENCH_KEEP")       # (tests: every rank's per-thread files of its blocks' last visit)
    if keep:
        os.makedirs(keep, exist_ok=True)
        for f in os.listdir(sort_path):
            if f.endswith(".las"):
                shutil.copy(os.path.join(sort_path, f), os.path.join(keep, f))
    shutil.rmtree(sort_path, ignore_errors=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
