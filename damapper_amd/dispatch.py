"""Read-block dispatcher for one node (the MI355X counterpart of HPC.damapper's job lists,
HPC.damapper.c:369-442): every read block of a DAZZ_DB is an independent unit, block b goes to GPU
b mod N, the reference index is rebuilt on each GPU, and a block whose output files already exist
is skipped (resume granularity = read block, as HPC.damapper.c:329-354).

    python -m damapper_amd.dispatch REF.db READS.db --gpus 8 [-k20 -e.85 -s100 -T4 -p -C -N -z -m<track>...] [--finish]

--finish runs the LAsort/LAcat/LAmerge post-step (damapper.c:887-910) per block, leaving
`<reads>.<b>.<ref>.las` (and `<ref>.<reads>.<b>.las` for -C) next to the reference DB.

One worker process per GPU (DMH_DEVICE selects it; --workers-per-gpu 2 runs two), no communication between workers."""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
import time

from .dazzdb import load_db, read_stub


def _worker(args) -> None:
    from . import capi, driver
    d, ref_root = os.path.split(args.ref)
    d = d or "."
    ref_root = ref_root[:-3] if ref_root.endswith(".db") else ref_root
    rd, read_root = os.path.split(args.reads)
    read_root = read_root[:-3] if read_root.endswith(".db") else read_root
    nrb, _ = read_stub(os.path.join(rd or ".", read_root + ".db"))
    nworkers = args.gpus * args.workers_per_gpu
    mine = [b for b in range(1, nrb + 1) if (b - 1) % nworkers == args.rank]
    out = args.out or os.path.join(d, "las_hip")
    nb, _ = read_stub(os.path.join(d, ref_root + ".db"))
    used: set = set()
    ref_blocks = [load_db(os.path.join(d, ref_root + ".db"), part=k + 1, masks=args.m, used=used) for k in range(nb)]
    ref_whole = load_db(os.path.join(d, ref_root + ".db"))
    # resident REF indices: both orientations of every block, 12 B per k-mer + the bucket table and its low-bits array
    # (up to 6.5 B per k-mer for a genome-sized block), next to ~6 GB of working set per worker
    retain = not args.rebuild_ref
    if retain:
        import ctypes as C
        free_b, total_b = C.c_ulonglong(0), C.c_ulonglong(0)
        need = 2 * 18.5 * float(ref_whole.db.totlen) + 6e9
        if capi.lib().dmh_device_memory(C.byref(free_b), C.byref(total_b)) == 0 and \
                need * args.workers_per_gpu > 0.85 * float(total_b.value):
            print("[gpu %d] resident reference indices would need %.0f GB x %d worker(s) of %.0f GB: rebuilding them "
                  "per read block instead (--rebuild-ref)" % (args.rank, need / 1e9, args.workers_per_gpu,
                                                               total_b.value / 1e9), flush=True)
            retain = False
    m = driver.Mapper(kmer=args.k, suppress=args.t, threads=args.T, ave_corr=args.e, spacing=args.s,
                      best_tie=args.n, profile=int(args.p), verbose=int(args.v), sort_path=out,
                      retain_ref=retain)
    mflag = (capi.FLAG_DOB if args.N else capi.FLAG_DOA | capi.FLAG_DOB) if args.C else capi.FLAG_DOA
    todo = []
    for b in mine:
        name = "%s.%d" % (read_root, b)
        if args.finish:
            finals = ([os.path.join(d, "%s.%s.las" % (name, ref_root))] if mflag & capi.FLAG_DOA else []) + \
                     ([os.path.join(d, "%s.%s.las" % (ref_root, name))] if mflag & capi.FLAG_DOB else [])
            done = all(os.path.exists(f) for f in finals)
        else:
            parts = (["%s.%s.M%d.las" % (name, ref_root, i + 1) for i in range(m.threads_pow2())] if mflag & capi.FLAG_DOA else []) + \
                    (["%s.%s.R%d.las" % (ref_root, name, i + 1) for i in range(m.threads_pow2())] if mflag & capi.FLAG_DOB else [])
            done = all(os.path.exists(os.path.join(out, f)) for f in parts)
        if done and not args.force:
            print("[gpu %d] %s: output present, skipped" % (args.rank, name), flush=True)
            continue
        todo.append(b)

    # The block loop, software-pipelined three deep: while block i is mapped, a loader thread reads block
    # i+2's slice of the .bps file into page-locked memory (no decode on the host: the 2-bit bytes are
    # uploaded and unpacked by a kernel), block i+1's index is built in the gaps of block i's filter, and
    # block i-1's alignment kernel / .las writing are still finishing (dmh_async_report).
    from concurrent.futures import ThreadPoolExecutor
    loader = ThreadPoolExecutor(1)

    def load(b):
        return load_db(os.path.join(rd or ".", read_root + ".db"), part=b, masks=args.m, used=used, pinned=True,
                       packed=not args.host_decode)

    pending = [loader.submit(load, b) for b in todo[:2]]
    capi.lib().dmh_async_report(args.report_mode)
    mapped = []
    cur = pending.pop(0).result() if pending else None
    t_all = t_loop = time.time()
    for i, b in enumerate(todo):
        name = "%s.%d" % (read_root, b)
        nxt = pending.pop(0).result() if pending else None
        if i + 2 < len(todo):
            pending.append(loader.submit(load, todo[i + 2]))
        t0 = time.time()
        tm = m.map_block(cur, name, ref_blocks, ref_whole, ref_root, mflag=mflag, next_reads=nxt)
        cur.release()
        mapped.append(name)
        dt = time.time() - t0
        print("[gpu %d] %s: %d reads, %.1f Mbp in %.2f s (%.3f Gbp/s)%s" %
              (args.rank, name, cur.nreads, cur.db.totlen / 1e6, dt, cur.db.totlen / dt / 1e9,
               "  [wait for loader %.3f, index %.3f, filter %.3f, report %.3f]" %
               (t0 - t_loop, tm["index_reads"], tm["filter"], tm["report"]) if args.v else ""), flush=True)
        t_loop = time.time()
        cur = nxt
    capi.lib().dmh_report_wait()
    capi.lib().dmh_async_report(0)
    loader.shutdown()
    if todo and args.v:
        print("[gpu %d] %d blocks in %.2f s" % (args.rank, len(todo), time.time() - t_all), flush=True)
    if args.finish:
        for name in mapped:
            driver.finish_block(out, name, ref_root, mflag=mflag, map_order=not args.z, outdir=d, nparts=m.threads_pow2())


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="damapper_amd.dispatch")
    ap.add_argument("ref")
    ap.add_argument("reads")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workers-per-gpu", type=int, default=1,
                    help="worker processes per GPU (each with its own replica of the reference index): two fill each "
                         "other's host gaps, +25%% aggregate throughput measured on one MI355X")
    ap.add_argument("--rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--out", default=None)
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-k", type=int, default=20)
    ap.add_argument("-t", type=int, default=0)
    ap.add_argument("-e", type=float, default=0.85)
    ap.add_argument("-s", type=int, default=100)
    ap.add_argument("-n", type=float, default=1.0)
    ap.add_argument("-T", type=int, default=4)
    ap.add_argument("-p", action="store_true")
    ap.add_argument("-v", action="store_true")
    ap.add_argument("-m", action="append", default=[], metavar="TRACK", help="k-mer mask track (repeatable)")
    ap.add_argument("-C", action="store_true", help="also write the reference-vs-reads files")
    ap.add_argument("-N", action="store_true", help="with -C: no reads-vs-reference files")
    ap.add_argument("-z", action="store_true", help="--finish sorts by A,B-read piles instead of map order")
    ap.add_argument("--finish", action="store_true", help="run the LAsort/LAcat/LAmerge post-step")
    ap.add_argument("--report-mode", type=int, default=2, choices=[0, 1, 2],
                    help="dmh_async_report: 0 synchronous Reporter, 1 .las writing on a worker thread, 2 (default) also the "
                         "alignment kernel left running next to the following block's filter")
    ap.add_argument("--host-decode", action="store_true",
                    help="decode the 2-bit reads on the host (Load_All_Reads' way) instead of on the GPU")
    ap.add_argument("--rebuild-ref", action="store_true",
                    help="rebuild the reference index for every read block (the reference's loop) instead of keeping "
                         "both orientations of every reference block resident in HBM")
    args = ap.parse_args(argv)
    if args.N and not args.C:
        raise SystemExit("damapper_amd.dispatch: Cannot specify N flag without C also")     # damapper.c:716
    if args.N and args.p:
        raise SystemExit("damapper_amd.dispatch: Cannot specify both N and p flags together")
    if args.rank >= 0:
        _worker(args)
        return 0
    procs = []
    for r in range(args.gpus * args.workers_per_gpu):
        env = dict(os.environ)
        env["DMH_DEVICE"] = str(r // args.workers_per_gpu)
        cmd = [sys.executable, "-m", "damapper_amd.dispatch"] + (argv if argv is not None else sys.argv[1:]) + ["--rank", str(r)]
        procs.append(subprocess.Popen(cmd, env=env))
    rc = 0
    for p in procs:
        rc |= p.wait()
    return rc


if __name__ == "__main__":
    sys.exit(main())
