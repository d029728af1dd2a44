"""damapper_amd: MI355X-native hot path of damapper (k-mer index, seed merge, chain scoring, d-wave
local alignment) behind the reference's map.h C API.  All compute lives in libdamapper_hip.so
(HIP, gfx950); the Python modules are the ctypes binding, the DAZZ_DB containers and the block-loop
driver used by tests and bench."""
__all__ = ["capi", "dazzdb", "driver", "las"]
