"""Repeats the variant data set under one option set and counts the runs whose files differ from the first default run.
usage: x_flake.py <n> key=value ...   (environment switches are read by the library once per process)"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import test_gpu_las as T
from damapper_amd import capi
n = int(sys.argv[1])
opts = dict(kv.split("=") for kv in sys.argv[2:])
L = capi.lib()
with tempfile.TemporaryDirectory() as d:
    T._variant_dbs(d)
    base = T._variant_run(d, "las_default")
    for k, v in opts.items():
        assert L.dmh_test_option(k.encode(), int(v)) == 0
    bad = 0
    for i in range(n):
        sys.stderr.write("=== run %d\n" % i); sys.stderr.flush()
        alt = T._variant_run(d, "las_alt")
        sys.stderr.write("=== run %d %s\n" % (i, "ok" if alt == base else "BAD")); sys.stderr.flush()
        if alt != base:
            bad += 1
            a = np.frombuffer(base[1], dtype=np.uint8); b = np.frombuffer(alt[1], dtype=np.uint8)
            idx = np.nonzero(a != b)[0] if len(a) == len(b) else []
            print("run", i, "las equal:", alt[0] == base[0], "prof bytes differing:", len(idx), list(idx[:8]), list(a[idx[:8]]), list(b[idx[:8]]))
    print("MODE", os.environ.get("DMH_WIDE_MODE"), opts, "failures", bad, "of", n)
