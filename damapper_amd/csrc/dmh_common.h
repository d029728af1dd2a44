// dmh_common.h -- internal helpers shared by the HIP translation units of libdamapper_hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/damapper_hip.h"

// -DDMH_DEBUG_LAUNCH (make debug): every kernel launch prints its name and waits for the device, so that a device
// fault is reported right behind the kernel that caused it (never part of the library build).
#ifdef DMH_DEBUG_LAUNCH
#include <stdio.h>
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, ...)                                                          \
  do { fprintf(stderr, "dmh: launch %s\n", #kernelName); fflush(stderr);                            \
       hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__);                                        \
       (void) hipDeviceSynchronize();                                                                \
  } while (0)
#endif

namespace dmh {

void fatal(const char *fmt, ...) __attribute__((format(printf, 1, 2), noreturn));
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define HIP_CHECK(expr)                                                                  \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess)                                                                \
      dmh::fatal("HIP error %s at %s:%d: %s", hipGetErrorName(_e), __FILE__, __LINE__,   \
                 hipGetErrorString(_e));                                                 \
  } while (0)

bool        host_crowded();        // more ranks on this node than half its cores: host waits yield instead of spinning (dmh_core.hip)
hipStream_t stream();              // the stream this thread's launches go to: the main stream unless a StreamScope is open
hipStream_t main_stream();
void        join_side_streams();   // the main stream waits for everything queued on the finish stream so far
void        bg_pump_tail(int nstages);   // stages of a deferred index build on the tail stream (dmh_core.hip)
int         hw_queues_seen();      // streams the runtime was found to run side by side (1..8; probed once, dmh_core.hip)
hipStream_t tail_stream();         // where the chain stage's last, latency-bound launches of a batch run (dmh_core.hip)
hipStream_t front_stream();        // where the match kernels of the REF block AFTER the one at hand run (dmh_merge.hip); the main stream when off
hipStream_t finish_stream();       // where the filter finishes its batches (chain_finish); the main stream when switched off
struct StreamScope                 // stream() = s for the calling thread while the scope lives
{ explicit StreamScope(hipStream_t s);
  ~StreamScope();
  hipStream_t prev;
};

// Deferred index builds (dmh_sort_kmers_async): the sort of the NEXT READS block is cut into stages
// (k-mer emit, then per radix pass a histogram stage and a scatter stage) that are queued on the
// stream at the points where the host is about to do host-only work between two GPU round trips of
// the CURRENT block, so the GPU has something to run while the host computes.  bg_pump(n) queues up
// to n stages; consumers of an index flush whatever is left.
void bg_pump(int nstages = 1);
void bg_flush();
bool bg_on_tail();                 // the deferred index build that is waiting runs on the tail stream (bg_pump_tail)
// true while a deferred Reporter's alignment kernel is running on the alignment stream: background
// stages are then held back (the HBM-bound sort kernels keep the GPU to themselves, before the alignment)
bool report_in_flight();
bool report_deferred_mode();          // dmh_async_report(2): the alignment kernel of a block runs next to the following block's filter
// Called by the filter at a moment when the GPU has work queued: if the previous READS block's deferred alignment
// kernel has finished, its records are fetched now and the candidates that need a wider kernel are re-launched --
// instead of at the next Reporter call, where the GPU would wait for both.
void report_poll();
void report_finish_pending();
bool report_adopt_block(const DAZZ_DB *block);   // a deferred Reporter takes over the resident copy it reads

// Device allocation that aborts through Clean_Exit on failure (the reference's behaviour
// for a failed Malloc, e.g. map.c:689-690).
void *dev_alloc(size_t bytes, const char *what);
void  dev_free(void *p);
// Called when device memory runs out, before giving up: returns true when it has released something (the filter
// registers one that finishes the batches in flight and drops the buffers of the idle ring slots).
void  set_reclaim_hook(bool (*fn)());
void  simulate_out_of_memory();       // runs the hook as a failed allocation would (test options)

// Host <-> device transfers go through one persistent pinned staging buffer: handing pageable host
// memory to hipMemcpyAsync makes the runtime pin and unpin those pages around the copy, and the
// unpin was observed to delay the next kernel dispatch by ~10 ms.  Both calls are synchronous.
void copy_d2h(void *host, const void *dev, size_t bytes, int pump_stages = 0);   // pump: bg_pump() behind the copy
void copy_h2d(void *dev, const void *host, size_t bytes);

template <typename T> struct DevBuf
{ T     *p   = nullptr;
  size_t cap = 0;      // elements
  unsigned gen = 0;    // allocations so far (whoever keeps state INSIDE the buffer between calls checks this)
  void ensure(size_t n, const char *what)
  { if (n > cap)
      { if (p) dev_free(p);
        p = nullptr; cap = 0;            // (dev_alloc may run the memory reclaim, which may look at this buffer)
        const size_t want = n + n / 8 + 256;
        p   = (T *) dev_alloc(want * sizeof(T), what);
        cap = want;
        gen += 1;
      }
  }
  void release()
  { if (p) dev_free(p);
    p = nullptr; cap = 0;
  }
};

// ---- stage timing (HIP events on our stream) -----------------------------------------
struct StageTimer
{ StageTimer(const char *name, double bytes = 0.0, hipStream_t on = nullptr);    // on = nullptr: the main stream
  ~StageTimer();
  int    idx;
  bool   active;
  const char *tname;
  hipStream_t s;
  hipEvent_t e0, e1;
};

bool   sort_scratch_release();                      // memory ran out: the k-mer sorts' partner buffers (dmh_core.hip)
double dev_live_gb();                               // device memory this library holds (buffers handed out, not the pool)
void timing_count(const char *name, int64_t n);     // a counter in the same table (reported as `launches`, 0 ms)

// wall-clock time of a host section, reported next to the device stages as "host:<name>"
struct HostTimer
{ HostTimer(const char *name);
  ~HostTimer();
  int    idx;
  bool   active;
  double t0;
};

static inline int ceil_div(int64_t a, int64_t b) { return (int) ((a + b - 1) / b); }

// ---- sort / scan primitives (dmh_sort.hip) ---------------------------------------------
struct SortScratch
{ DevBuf<uint32_t> hist;     // 256 x ntiles digit histogram / offsets
  DevBuf<uint32_t> bsum;     // scan block sums
  DevBuf<uint64_t> keys_alt;
  DevBuf<uint32_t> vals_alt;
};

// Stable LSD radix sort of (key,val) pairs on key bits [lo_bit, lo_bit+nbits).  Uses
// keys_alt/vals_alt as the ping-pong partner; on return *keys/*vals point at the buffers
// holding the result (either the originals or the alternates) and *okeys/*ovals at the others.
// vals = ovals = nullptr: the keys alone (8 instead of 12 bytes per record and pass).
void radix_sort_pairs(uint64_t **keys, uint32_t **vals, uint64_t **okeys, uint32_t **ovals,
                      int64_t n, int lo_bit, int nbits, SortScratch &scr, const char *stage);

void radix_pass_hist(const uint64_t *keys, int64_t n, int shift, int w, SortScratch &scr, int part, int nparts);
void radix_pass_scatter(const uint64_t *keys, const uint32_t *vals, uint64_t *okeys, uint32_t *ovals, int64_t n,
                        int shift, int w, SortScratch &scr, int part, int nparts);

// Exclusive scan of n uint32 values in place; returns nothing, total is written to *d_total
// (device pointer to one uint32) if not null.  n < 2^32, sum < 2^32.
void exclusive_scan_u32(uint32_t *d_data, int64_t n, DevBuf<uint32_t> &bsum, uint32_t *d_total);
bool lds_atomics_ordered();       // LDS atomic adds of a wavefront are served in lane order (checked on the device once, dmh_merge.hip)
void scan_state_reset();           // call when a scan scratch buffer (SortScratch::bsum) has been released
// ... of d_src into d_data (the two may be the same array)
void exclusive_scan_u32(const uint32_t *d_src, uint32_t *d_data, int64_t n, DevBuf<uint32_t> &bsum, uint32_t *d_total);

}  // namespace dmh
