#!/usr/bin/env python3
"""The drop-in binaries' own rate on the bench workload: the reference's damapper.c driver linked
against libdamapper_hip.so, on DBs on disk, as a user of the reference would run it.

  damapper_hip            the UNMODIFIED driver (oracle/_ref/damapper_hip), with DMH_REF_CACHE=1: the REF indices are
                          found in HBM from the second READS block on, but the driver still reads and complements every
                          REF block on the host for every READS block (damapper.c:842-857)
  damapper_hip_resident   the driver with INTEGRATION.md's change (oracle/resident_driver.py): REF blocks read,
                          complemented and indexed once
  damapper_hip_fast       the driver with ALL of INTEGRATION.md section 3 (oracle/fast_driver.py): also READS blocks
                          decoded on the device, Reporter deferred with two blocks open, pass hints, LAsort / LAcat on
                          a thread

Each is run on k (--base-blocks) and on N READS blocks; the marginal wall time per READS block is (T_N - T_k) / (N - k).  The .las files
of READS.1 are compared with the reference's when bench.py has left them in its cache on this box.

usage: python3 profiles/scripts/dropin_rate.py [--blocks 3] [--workload grch38] [--out FILE.json]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench            # noqa: E402  (the workload generator only)
from damapper_amd import dazzdb, las   # noqa: E402
import run_ref          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=3)
    ap.add_argument("--workload", default="grch38")
    ap.add_argument("--block-reads", type=int, default=25000)
    ap.add_argument("--mem-gb", type=int, default=64)
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None, help="comma-separated binaries to run (default: all three)")
    ap.add_argument("--base-blocks", type=int, default=1,
                    help="the shorter run: the marginal time per block is (T_blocks - T_base) / (blocks - base).  The first few "
                         "READS blocks of a process also pay for the library's buffers (11 ring slots growing to the largest "
                         "pass they meet: 2.7 s of hipMalloc over the first four blocks), so the steady state needs base >= 4")
    a = ap.parse_args()
    a.read_lengths = "fixed"
    a.genome_mbp = None
    a.families = None
    clen, ncontig, kind, fam = bench.WORKLOADS[a.workload]
    lens = [clen] * ncontig
    ranges = bench.ref_block_ranges(lens)
    if fam == 0:
        fam = max(1, int(round(max(sum(lens[x:y]) for x, y in ranges) / 750_000)))
    G = dazzdb.native_genome(20, lens, kind, families=fam)
    cores, tpow = bench.host_threads()
    d = tempfile.mkdtemp(prefix="dmh_dropin_")
    res = {"workload": a.workload, "read_blocks": a.blocks, "threads": tpow, "runs": {}}
    try:
        t0 = time.time()
        dazzdb.write_db_blocks(d, "REF", [dazzdb.genome_block(G, x, y) for x, y in ranges])
        blocks = [bench.simulate_read_block(G, b, a) for b in range(a.blocks)]
        bases = [float(b.db.totlen) for b in blocks]
        dazzdb.write_db_blocks(d, "READS", blocks)
        del blocks
        res["setup_s"] = round(time.time() - t0, 1)
        flags = ["-k20", "-e.85", "-M%d" % a.mem_gb]
        names = ["READS.%d" % (k + 1) for k in range(a.blocks)]
        runs = (("damapper_hip_fast", {}), ("damapper_hip_resident", {}), ("damapper_hip", {"DMH_REF_CACHE": "1"}))
        if a.only:
            runs = tuple(r for r in runs if r[0] in a.only.split(","))
        os.environ.setdefault("DMH_FAST_TIMES", "1")
        for binary, env in runs:
            if not os.path.exists(os.path.join(run_ref.REF_DIR, binary)):
                continue
            for k, v in env.items():
                os.environ[k] = v
            t1, _ = run_ref.run_reference(d, "REF", names[:a.base_blocks], flags=flags, threads=tpow, keep="las_%s_1" % binary, binary=binary)
            tn, log = run_ref.run_reference(d, "REF", names, flags=flags, threads=tpow, keep="las_%s_n" % binary, binary=binary,
                                            capture=True)
            if os.environ.get("DMH_TIMING_AT_EXIT"):
                sys.stderr.write("\n".join(x for x in log.splitlines() if x.startswith("dmh:")) + "\n")
            for k in env:
                os.environ.pop(k, None)
            t_end = time.time()
            per = (tn - t1) / max(1, a.blocks - a.base_blocks)
            res["runs"][binary] = {"env": env, "wall_%d_block_s" % a.base_blocks: round(t1, 2), "wall_%d_blocks_s" % a.blocks: round(tn, 2),
                                   "marginal_s_per_read_block": round(per, 3),
                                   "marginal_gbp_per_s": round(bases[1] / per / 1e9, 4) if per > 0 else None}
            # the fast driver's own clock (DMH_FAST_TIMES): the block loop in the steady state, and what the process does
            # before and after it (start-up: HIP context, 18 REF indices, the whole reference on the device; exit)
            import re
            after = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r"after \S+\.(\d+): .* wall ([0-9.]+)", log)}
            mend = re.search(r"wall at end ([0-9.]+)", log)
            if a.base_blocks in after and a.blocks in after:
                loop = (after[a.blocks] - after[a.base_blocks]) / (a.blocks - a.base_blocks)
                res["runs"][binary]["loop_s_per_read_block"] = round(loop, 4)
                res["runs"][binary]["loop_gbp_per_s"] = round(bases[1] / loop / 1e9, 4)
                res["runs"][binary]["loop_note"] = ("wall clock of the driver's own block loop, READS blocks %d..%d of one run: load "
                                                    "(Open_DB, Trim_DB, .bps slice), Sort_Kmers, %d Match_Filter passes, Reporter of the "
                                                    "block before, Close_DB; LAsort/LAcat on the driver's thread"
                                                    % (a.base_blocks + 1, a.blocks, 2 * len(ranges)))
                res["runs"][binary]["startup_s"] = round(after[1] - (t_end - tn), 2)
                if mend:
                    res["runs"][binary]["exit_s"] = round(t_end - float(mend.group(1)), 2)
            sys.stderr.write("%s: %s\n" % (binary, json.dumps(res["runs"][binary])))
            # parity of READS.1 with the reference's files, when bench.py has cached them on this box
            import glob
            for cache in glob.glob(os.path.join(tempfile.gettempdir(), "dmh_bench_refcache_*")):
                try:
                    if json.load(open(os.path.join(cache, "cpu.json")))["sample"].find(" -n.95") >= 0:
                        continue                       # (another flag set's files)
                    ts_r, want = las.read_las_set(cache, "READS.1.REF.M@.las")
                    ts_h, got = las.read_las_set(os.path.join(d, "las_%s_n" % binary), "READS.1.REF.M@.las")
                except Exception:
                    continue
                if len(want) == len(got):
                    bad = sum(1 for x, y in zip(want, got) if x != y)
                    res["runs"][binary]["parity_vs_reference"] = {"records": len(want), "differing": bad}
        print(json.dumps(res))
        if a.out:
            json.dump(res, open(a.out, "w"), indent=1)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
