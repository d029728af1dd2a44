#!/usr/bin/env python3
"""bench.py -- Gbp of reads mapped per second through the HIP hot path.

A "step" maps ONE read block (default 25,000 simulated PacBio reads x 10 kb, 15 % error = 0.25 Gbp,
a DBsplit-sized block) against the reference in both orientations, exactly as one iteration of
damapper.c's main loop does: Sort_Kmers(reads) -> for every REF block {Sort_Kmers, Match_Filter,
complement, Sort_Kmers, Match_Filter} -> Reporter (.las written).  The workload is BASELINE.json's
config[1]: an E. coli-sized reference (4.6 Mbp; uniform random ACGT with a fixed seed, there is no
network for the real K-12 FASTA) and 100 k reads = 4 blocks; K steps cycle over the blocks.

Read blocks shard across GPUs (one process per GPU, no collective on the data path; the reference
index is rebuilt per GPU): rank r maps its own blocks, so per-GPU work is fixed ("weak").
Inputs (block bases) are resident in HBM before the timed region starts.

The JSON line also carries
  roofline     -- the dominant HBM-bound kernel (one LSD radix-sort pass, k_radix_scatter): the
                  reference's algorithmic bytes per pass (32 B per record) over the kernel's mean
                  duration measured with HIP events on the library's stream;
  cpu_baseline -- the compiled reference (oracle/_ref/damapper -T<P>) timed on this box's host
                  cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from damapper_amd import capi, dazzdb, driver  # noqa: E402

GENOME_LEN = 4_600_000
READ_LEN = 10_000
HBM_PEAK_GBS = 8000.0


def make_block(genome, nreads, seed, tfirst=0, part=0, pinned=False):
    return dazzdb.simulate_block(seed, genome[part % len(genome)], nreads, read_len=READ_LEN, err=0.15, tfirst=tfirst, part=part,
                                 path=b"./.READS", pinned=pinned)


def pack_block(blk):
    """The block as it lies in a .bps file (4 bases per byte) with Load_All_Reads' read records and no bases:
    what `--pcie packed` uploads (dmh_preload_block_2bit decodes it on the device)."""
    n = blk.nreads
    lens = blk.reads["rlen"][1:n + 1].astype(np.int64)
    offs = blk.reads["boff"][1:n + 1].astype(np.int64)
    clen = (lens + 3) // 4
    coff = np.concatenate(([0], np.cumsum(clen)))
    out = dazzdb.host_bytes(int(coff[-1]), True)
    for i in range(n):
        s = np.zeros(int(clen[i]) * 4, dtype=np.uint8)
        s[:lens[i]] = blk.bases[1 + offs[i]:1 + offs[i] + lens[i]]
        q = s.reshape(-1, 4)
        out[coff[i]:coff[i + 1]] = (q[:, 0] << 6) | (q[:, 1] << 4) | (q[:, 2] << 2) | q[:, 3]
    hb = dazzdb.make_block([], tfirst=int(blk.db.tfirst), part=int(blk.db.part), path=blk.path,
                           freq=[blk.db.freq[j] for j in range(4)], lens=lens)
    hb.slice = out
    hb.coff = np.ascontiguousarray(coff[:-1])
    return hb


def shard_blocks(nblocks, rank, world):
    """Read blocks are independent units: block b goes to rank b mod world."""
    return [b for b in range(nblocks) if b % world == rank]


def cpu_baseline(genome, sample_reads, threads):
    """Reference `damapper -k20 -e.85 -T<threads>` on a bounded sample; wall time of the process
    (LAsort/LAcat are shims)."""
    try:
        import run_ref
    except Exception:
        return None
    if not run_ref.have_ref():
        return None
    with tempfile.TemporaryDirectory() as d:
        dazzdb.write_db(d, "REF", genome, block_size=400_000_000)
        dazzdb.write_db(d, "READS", sample_reads)
        dt, _ = run_ref.run_reference(d, "REF", ["READS"], flags=["-k20", "-e.85"], threads=threads)
    bases = float(sum(len(r) for r in sample_reads))
    return {"value": bases / dt / 1e9, "unit": "Gbp/s", "cores": threads, "kind": "reference",
            "sample": "%d reads x 10 kb (%.0f Mbp) vs the same 4.6 Mbp reference, damapper -k20 -e.85 -T%d, %.1f s wall"
                      % (len(sample_reads), bases / 1e6, threads, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--block-reads", type=int, default=25000)
    ap.add_argument("--nblocks", type=int, default=4)
    ap.add_argument("--cpu-sample-reads", type=int, default=100000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-timing", action="store_true", help="do not record HIP events around any kernel (no roofline)")
    ap.add_argument("--stage-timing", action="store_true",
                    help="HIP events around EVERY kernel and host timers (device_stage_ms; costs about 1 ms per step); "
                         "by default only the roofline kernel is timed")
    ap.add_argument("--no-prefetch", action="store_true", help="do not queue the next block's index before Reporter")
    ap.add_argument("--serial-prefetch", action="store_true",
                    help="queue the next block's whole index build before Reporter instead of spreading its "
                         "stages over the gaps of the current block")
    ap.add_argument("--genome-mbp", type=float, default=GENOME_LEN / 1e6,
                    help="reference size in Mbp (default 4.6 = configs[1]; 250 approximates configs[2], one contig per "
                         "~125 Mbp, one REF block)")
    ap.add_argument("--pcie", choices=["pinned", "pageable", "packed"], default=None,
                    help="PCIe-inclusive rate: read blocks are NOT resident, every step uploads its block from "
                         "page-locked (dmh_host_alloc) or ordinary host memory inside the timed region; packed = "
                         "as the 2-bit bytes of the .bps file (a quarter of the volume), decoded by a kernel")
    ap.add_argument("--sync-report", action="store_true", help="write the .las files inside Reporter (drop-in behaviour)")
    ap.add_argument("--report-mode", type=int, default=2, choices=[0, 1, 2],
                    help="0 = Reporter synchronous, 1 = .las selection/writing on a worker thread, 2 (default) = also the "
                         "alignment kernel of block i deferred, at half grid, next to the filter of block i+1")
    ap.add_argument("--rebuild-ref", action="store_true",
                    help="rebuild the REF indices for every read block as damapper.c does (default: build both "
                         "orientations once per GPU and keep them resident, BASELINE.json north_star)")
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

    L = capi.lib()
    if L.dmh_device_count() < 1:
        raise SystemExit("bench.py needs a GPU: libdamapper_hip has no CPU fallback")

    # ---- synthetic workload (seeded): reference shared by all ranks, reads per rank
    g_rng = np.random.default_rng(20)
    glen = int(args.genome_mbp * 1e6)
    ncontig = max(1, int(round(glen / 125e6)))
    genome = dazzdb.random_genome(g_rng, [glen // ncontig] * ncontig)
    nblk = min(args.nblocks, max(1, args.steps + args.warmup))
    blocks = []
    for b in range(nblk):
        blocks.append(make_block(genome, args.block_reads, 1000 + 97 * rank + b, tfirst=b * args.block_reads, part=b + 1,
                                 pinned=(args.pcie == "pinned")))
        if args.pcie == "packed":
            blocks[-1] = pack_block(blocks[-1])
    # REF blocks as DBsplit would cut them: whole contigs, at most ~375 Mbp (< 2^31 k-mers) per block
    ref_blocks, first, acc = [], 0, 0
    for ci in range(len(genome) + 1):
        if ci == len(genome) or (acc > 0 and acc + len(genome[ci]) > 375_000_000):
            ref_blocks.append(dazzdb.make_block(genome[first:ci], tfirst=first, part=len(ref_blocks) + 1, path=b"./.REF"))
            first, acc = ci, 0
        if ci < len(genome):
            acc += len(genome[ci])
    ref_whole = dazzdb.make_block(genome, path=b"./.REF")

    sort_path = tempfile.mkdtemp(prefix="dmh_bench_%d_" % rank)
    m = driver.Mapper(kmer=20, ave_corr=0.85, spacing=100, threads=4, sort_path=sort_path, mem_limit=64 << 30,
                      retain_ref=not args.rebuild_ref, prefetch_deferred=not args.serial_prefetch)
    m.make_spec([ref_whole.db.freq[j] for j in range(4)])

    # inputs resident in HBM before the clock starts
    if args.pcie is None:
        for blk in blocks:
            L.dmh_preload_block(blk.ptr())
            L.dmh_prepare_block(blk.ptr(), 1)
    L.dmh_preload_block(ref_whole.ptr())
    L.dmh_prepare_block(ref_whole.ptr(), 0)
    pcie_t0 = time.time()
    L.dmh_sync()

    last = args.warmup + args.steps - 1

    def step(i):
        blk = blocks[i % nblk]
        nxt = blocks[(i + 1) % nblk] if (i < last and not args.no_prefetch and nblk > 1) else None
        if args.pcie == "packed":
            for b in (blk, nxt):                 # map_block uploads (2-bit) and decodes what carries a slice
                if b is not None and not b.resident:
                    b.packed = b.slice
        t = m.map_block(blk, "READS.%d" % (i % nblk + 1), ref_blocks, ref_whole, "REF", next_reads=nxt)
        if args.pcie == "packed":
            blk.release()
        elif args.pcie is not None:
            L.dmh_release_block(blk.ptr())       # the next visit uploads it again
        return t

    # the warm-up steps run the configuration that is timed (worker thread, second stream, page-locked pool
    # buffers and the allocator's blocks are created by them, not inside the timed region)
    L.dmh_async_report(0 if args.sync_report else args.report_mode)   # block i's alignments and .las writing overlap block i+1's filter
    for i in range(args.warmup):
        step(i)
    L.dmh_report_wait()
    L.dmh_timing_enable(0 if args.no_stage_timing else (1 if args.stage_timing else 2))
    L.dmh_timing_reset()
    if dist is not None:
        import torch
        dist.barrier()
        torch.cuda.synchronize()
    L.dmh_sync()
    t0 = time.time()
    bases = 0
    host = {"index_reads": 0.0, "filter": 0.0, "report": 0.0, "between_blocks": 0.0}
    for i in range(args.steps):
        t = step(args.warmup + i)
        for k in host:
            host[k] += t[k]
        bases += int(blocks[(args.warmup + i) % nblk].db.totlen)
    L.dmh_report_wait()
    L.dmh_sync()
    if dist is not None:
        import torch
        torch.cuda.synchronize()
        dist.barrier()
    dt = time.time() - t0

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

    timings = capi.timings()
    L.dmh_timing_enable(0)

    if rank == 0:
        value = tot_bases / dt / 1e9
        sc = timings.get("k_radix_scatter", None)
        roof = None
        if sc and sc["launches"] > 0 and sc["ms"] > 0:
            achieved = sc["bytes"] / (sc["ms"] * 1e-3) / 1e9          # GB/s, algorithmic bytes
            # HBM bytes per record of this kernel from the PMC passes committed under profiles/
            # (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as is)
            traffic = None
            try:
                import glob
                latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fetch_write_summary.json")))[-1]
                pm = json.load(open(latest))
                traffic = pm["derived"]["scatter_traffic_bytes_per_record"] * (sc["bytes"] / sc["launches"] / 32.0)
            except Exception:
                pass
            roof = {"bound": "hbm", "kernel": "k_radix_scatter", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": None if traffic is None else round(traffic),
                    "launches": int(sc["launches"]), "avg_ms": round(sc["ms"] / sc["launches"], 4),
                    "alg_bytes_per_launch": sc["bytes"] / sc["launches"]}
        stages = {k: round(v["ms"], 2) for k, v in sorted(timings.items(), key=lambda kv: -kv[1]["ms"])}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                ncores = len(os.sched_getaffinity(0))
            except AttributeError:
                ncores = os.cpu_count() or 1
            ncores = min(ncores, 16)        # a 1-GPU box shares its host: 16 cores are ours
            p = 1
            while 2 * p <= ncores:
                p *= 2
            sample = dazzdb.block_reads(make_block(genome, args.cpu_sample_reads, 4242))
            cpu = cpu_baseline(genome, sample, p)
        wl_name = ("configs[1]: E.coli-sized random reference (4.6 Mbp)" if abs(args.genome_mbp - 4.6) < 1e-6 else
                   "random reference of %.0f Mbp in %d contig(s)" % (args.genome_mbp, ncontig))
        out = {
            "metric": "Gbp reads mapped/sec (whole node), -k20 -e.85; .las parity-checked vs CPU reference",
            "value": round(value, 4), "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / max(1, args.steps) * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64/u8 integer", "data": "synthetic",
            "config": {"workload": wl_name + " + simulated PacBio reads "
                                   "(10 kb, 15%% err), %d reads per block/step, %d distinct blocks per GPU, "
                                   "-k20 -e.85 -s100, REF index %s" %
                                   (args.block_reads, nblk, ("rebuilt per read block" if args.rebuild_ref else "built once per GPU, resident") +
                                    ("" if args.pcie is None else ("; read blocks uploaded inside the timed region " +
                                     ("as 2-bit .bps bytes from page-locked memory and decoded on the device" if args.pcie == "packed"
                                      else "from %s host memory" % args.pcie)))),
                       "read_bases_per_step": int(blocks[0].db.totlen), "parallelism": "read blocks sharded x%d" % world},
            "roofline": roof, "cpu_baseline": cpu,
            "host_seconds": {k: round(v, 3) for k, v in host.items()},
            "device_stage_ms": stages,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
