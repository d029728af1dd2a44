"""TEST INFRASTRUCTURE: run the compiled reference (oracle/_ref/damapper) end to end on DBs
written by damapper_amd.dazzdb and keep its per-thread .las files."""
from __future__ import annotations

import os
import shutil
import subprocess
import time
from typing import List, Optional, Sequence

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")


def have_ref() -> bool:
    return os.path.exists(os.path.join(REF_DIR, "damapper"))


def run_reference(workdir: str, ref_root: str, read_roots: Sequence[str], flags: Sequence[str] = (),
                  threads: int = 4, keep: str = "las_ref", binary: str = "damapper",
                  capture: bool = False, tools_dir: Optional[str] = None):
    """cd workdir; damapper <flags> -T<threads> <ref_root> <read_roots...>.  Returns
    (elapsed seconds, stdout text).  Per-thread files end up in workdir/<keep>/."""
    env = dict(os.environ)
    # LAsort/LAcat/LAmerge: by default the keep-and-exit shims; tools_dir selects real ones
    env["PATH"] = (tools_dir or os.path.join(REF_DIR, "bin")) + ":" + env.get("PATH", "")
    keepdir = os.path.join(workdir, keep)
    if os.path.isdir(keepdir):
        shutil.rmtree(keepdir)
    env["LAS_KEEP"] = keepdir
    tmp = os.path.join(workdir, "tmp_sort")
    os.makedirs(tmp, exist_ok=True)
    cmd = [os.path.join(REF_DIR, binary), "-T%d" % threads, "-P" + tmp] + list(flags) + [ref_root] + list(read_roots)
    t0 = time.time()
    p = subprocess.run(cmd, cwd=workdir, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    dt = time.time() - t0
    if p.returncode != 0:
        raise RuntimeError("reference failed (%d): %s\n%s" % (p.returncode, " ".join(cmd), p.stderr[-2000:]))
    return dt, (p.stdout + "\n" + p.stderr if capture else p.stdout)
