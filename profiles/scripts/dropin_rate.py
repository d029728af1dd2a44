#!/usr/bin/env python3
"""The drop-in binaries' own rate on the bench workload: the reference's damapper.c driver linked
against libdamapper_hip.so, on DBs on disk, as a user of the reference would run it.

  damapper_hip            the UNMODIFIED driver (oracle/_ref/damapper_hip), with DMH_REF_CACHE=1: the REF indices are
                          found in HBM from the second READS block on, but the driver still reads and complements every
                          REF block on the host for every READS block (damapper.c:842-857)
  damapper_hip_resident   the driver with INTEGRATION.md's change (oracle/resident_driver.py): REF blocks read,
                          complemented and indexed once

Each is run on 1 and on N READS blocks; the marginal wall time per READS block is (T_N - T_1) / (N - 1).  The .las files
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
        for binary, env in (("damapper_hip_resident", {}), ("damapper_hip", {"DMH_REF_CACHE": "1"})):
            if not os.path.exists(os.path.join(run_ref.REF_DIR, binary)):
                continue
            for k, v in env.items():
                os.environ[k] = v
            t1, _ = run_ref.run_reference(d, "REF", names[:1], flags=flags, threads=tpow, keep="las_%s_1" % binary, binary=binary)
            tn, log = run_ref.run_reference(d, "REF", names, flags=flags, threads=tpow, keep="las_%s_n" % binary, binary=binary,
                                            capture=True)
            if os.environ.get("DMH_TIMING_AT_EXIT"):
                sys.stderr.write("\n".join(x for x in log.splitlines() if x.startswith("dmh:")) + "\n")
            for k in env:
                os.environ.pop(k, None)
            per = (tn - t1) / max(1, a.blocks - 1)
            res["runs"][binary] = {"env": env, "wall_1_block_s": round(t1, 2), "wall_%d_blocks_s" % a.blocks: round(tn, 2),
                                   "marginal_s_per_read_block": round(per, 3),
                                   "marginal_gbp_per_s": round(bases[1] / per / 1e9, 4) if per > 0 else None}
            sys.stderr.write("%s: %s\n" % (binary, json.dumps(res["runs"][binary])))
            # parity of READS.1 with the reference's files, when bench.py has cached them on this box
            import glob
            for cache in glob.glob(os.path.join(tempfile.gettempdir(), "dmh_bench_refcache_*")):
                try:
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
