"""damapper_amd: MI355X-native hot path of damapper (k-mer index, seed merge, chain scoring, d-wave
local alignment) behind the reference's map.h C API.  All compute lives in libdamapper_hip.so
(HIP, gfx950); the Python modules are the ctypes binding, the DAZZ_DB containers and the block-loop
driver used by tests and bench."""
import os as _os

# the library's streams want a hardware queue each (dmh_core.hip init_device); the HIP runtime reads this once, when the
# process first uses the GPU -- torch may do that before the library is loaded, so it is set when the package is imported
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

__all__ = ["capi", "dazzdb", "driver", "las"]
