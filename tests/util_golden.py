import os

import numpy as np

from damapper_amd import dazzdb

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny.npz")


def load():
    z = np.load(GOLDEN)
    genome = [z["contig%d" % i] for i in range(int(z["ncontigs"]))]
    reads = [z["read%d" % i] for i in range(int(z["nreads"]))]
    return z, genome, reads


def golden_cands(z):
    """{aread: [(8 ints, adisp, bdisp), ...]} in list order."""
    out = {}
    jo = z["jump_off"]
    for i, row in enumerate(z["cands"]):
        j = z["jumps"][jo[i]:jo[i + 1]]
        out.setdefault(int(row[0]), []).append((tuple(int(x) for x in row[1:9]), j[:, 0].tolist(), j[:, 1].tolist()))
    return out
