"""What the library's probe of the runtime's hardware queues says (dmh_core.hip probe_hw_queues): run under different
GPU_MAX_HW_QUEUES with DMH_STATS=1."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from damapper_amd import capi
k = np.arange(1000, 0, -1, dtype=np.uint64)
v = np.arange(1000, dtype=np.uint32)
capi.lib().dmh_sort_pairs(k.ctypes.data, v.ctypes.data, len(k), 12)
print("sorted", bool((np.diff(k.astype(np.int64)) >= 0).all()))
