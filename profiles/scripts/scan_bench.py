"""Device time of the exclusive scan by size (idle GPU): DMH_SCAN3=1 for the three-kernel version."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from damapper_amd import capi
L = capi.lib()
for n in (25_000, 131_072, 1_000_000, 2_500_000, 3_600_000, 4_194_304, 5_800_000, 16_400_000):
    a = np.ones(n, dtype=np.uint32)
    ms = C.c_double(0)
    L.dmh_exclusive_scan(a.ctypes.data, n, 50, None, C.byref(ms))
    print("%10d values: %7.1f us per scan" % (n, ms.value * 1e3))
