"""CPU suite: the C-ABI library loads without a GPU and exports every symbol the header declares;
host-only entry points behave like the reference's (map.c:124-150, align.c:222-287); the read-block
sharding used by bench.py is covered with a 2-process gloo run."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np

from damapper_amd import capi, dazzdb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_header_symbols():
    L = capi.lib()
    hdr = open(os.path.join(ROOT, "include", "damapper_hip.h")).read()
    declared = set(re.findall(r"\b((?:dmh_[a-z_]+)|Set_Filter_Params|Sort_Kmers|Match_Filter|Reporter|New_Align_Spec|"
                              r"Free_Align_Spec|Trace_Spacing|Average_Correlation|Base_Frequencies|"
                              r"Overlap_If_Possible|Clean_Exit)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for s in sorted(declared | set(capi.exported_symbols())):
        assert hasattr(L, s), "missing symbol " + s


def test_set_filter_params_contract():
    L = capi.lib()
    assert L.Set_Filter_Params(1, 0, 4) == 1        # map.c:125-126
    assert L.Set_Filter_Params(20, 0, 4) == 0


def test_align_spec_accessors():
    L = capi.lib()
    L.Average_Correlation.restype = C.c_double
    L.Average_Correlation.argtypes = [C.c_void_p]
    f = (C.c_float * 4)(.25, .25, .25, .25)
    sp = L.New_Align_Spec(0.85, 100, f, 1)
    assert L.Trace_Spacing(sp) == 100
    assert abs(L.Average_Correlation(sp) - 0.85) < 1e-12
    L.Free_Align_Spec(sp)


def test_db_roundtrip(tmp_path):
    rng = np.random.default_rng(3)
    seqs = dazzdb.random_genome(rng, [1000, 37, 4001, 20])
    firsts = dazzdb.write_db(str(tmp_path), "X", seqs, block_size=1500)
    whole = dazzdb.load_db(os.path.join(str(tmp_path), "X.db"))
    assert whole.nreads == 4
    for i, s in enumerate(seqs):
        assert np.array_equal(whole.seq(i), s)
    nb, fs = dazzdb.read_stub(os.path.join(str(tmp_path), "X.db"))
    assert nb == len(firsts) - 1
    b2 = dazzdb.load_db(os.path.join(str(tmp_path), "X.db"), part=2)
    assert b2.db.tfirst == firsts[1]
    assert np.array_equal(b2.seq(0), seqs[firsts[1]])
    blk = dazzdb.make_block(seqs)
    before = blk.bases.copy()
    blk.complement_inplace()
    blk.complement_inplace()
    assert np.array_equal(before, blk.bases)


def test_simulator_is_seed_deterministic():
    rng = np.random.default_rng(4)
    g = dazzdb.random_genome(rng, [50000])
    a = dazzdb.simulate_block(11, g[0], 50, read_len=2000)
    b = dazzdb.simulate_block(11, g[0], 50, read_len=2000)
    assert np.array_equal(a.bases, b.bases)
    assert a.bases[0] == 4 and a.bases[-1] == 4
    err_len = abs(a.db.totlen / 50 / 2000 - 1.0525)     # E[len] = 1 + .15*(.6 - .25)
    assert err_len < 0.02


GLOO_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from bench import shard_blocks
dist.init_process_group(backend="gloo")
r, w = dist.get_rank(), dist.get_world_size()
mine = shard_blocks(11, r, w)
t = torch.zeros(11, dtype=torch.int64)
for b in mine: t[b] += 1
dist.all_reduce(t)
assert t.tolist() == [1] * 11, t.tolist()
tm = torch.tensor([float(r + 1)], dtype=torch.float64)
dist.all_reduce(tm, op=dist.ReduceOp.MAX)
assert tm.item() == float(w)
dist.destroy_process_group()
print("ok", r)
"""


def test_block_sharding_two_ranks_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(GLOO_WORKER % ROOT)
    env = dict(os.environ)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert p.stdout.count("ok") == 2


def test_header_compiles_as_c(tmp_path):
    """include/damapper_hip.h is a C header (the boundary is a C ABI): gcc -fsyntax-only on a C file that includes it."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "h.c"
    src.write_text('#include "damapper_hip.h"\nint main(void) { dmh_next_pass(0, 0); return 0; }\n')
    p = subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-I", os.path.join(root, "include"), str(src)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr
