// dmh_core.hip -- library state, device memory, stage timing, Set_Filter_Params, Sort_Kmers
// and the small stage-level helpers of the C-ABI (see include/damapper_hip.h).

#include "dmh_state.h"
#include <sched.h>
#include <unistd.h>
#include "dmh_merge.h"
#include <deque>
#include "dmh_device.h"
#include <stdarg.h>
#include <time.h>
#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <thread>

// ---- globals shared with the driver (damapper.c:58-65; map.h:17-24) ----------------------
extern "C" {
int      VERBOSE      = 0;
int      PROFILE      = 0;
int      SPACING      = 100;
double   BEST_TIE     = 1.0;
char    *SORT_PATH    = (char *) "/tmp";
uint64_t MEM_LIMIT    = 0;
uint64_t MEM_PHYSICAL = 0;

__attribute__((weak)) void Clean_Exit(int val) { exit(val); }
}

namespace dmh {

static char       g_err[1024] = "";
static hipStream_t g_stream    = nullptr;
static hipStream_t g_upstream  = nullptr;     // block uploads from page-locked host memory (DMA only, no kernels)
static bool        g_init      = false;

State &S()
{ static State *s = new State();
  return *s;
}

void set_error(const char *fmt, ...)
{ va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

void fatal(const char *fmt, ...)
{ va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  fprintf(stderr, "damapper_hip: %s\n", g_err);
  fflush(stderr);
  Clean_Exit(1);
  exit(1);
}

extern "C" void dmh_timing_enable(int on);
static void timing_dump_at_exit();

// Cores this process may use: the affinity mask, capped by the cgroup's CPU quota (a GPU box of this pool shows 256 CPUs
// in the mask and a quota of 16).
static int host_cores()
{ int n = 0;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
  if (n <= 0) n = (int) sysconf(_SC_NPROCESSORS_ONLN);
  FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
  if (f != nullptr)
    { char q[64];
      long long per = 0;
      if (fscanf(f, "%63s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0)
        { const long long c = atoll(q) / per;
          if (c >= 1 && c < n) n = (int) c;
        }
      fclose(f);
    }
  return n > 0 ? n : 1;
}

// One process per GPU keeps two threads busy at the peak (the block loop, which spins on the device's answers, and the
// report worker).  When the ranks of a node outnumber half its cores -- eight ranks on a 16-core quota -- the waits give
// the core up instead (blocking synchronisation, a yield in the finish's tag wait): a spinning rank would then only take
// the time of another rank's worker.  DMH_YIELD=0 / 1 forces either.
bool host_crowded()
{ static int v = -1;
  if (v < 0)
    { const char *e = getenv("DMH_YIELD"), *ws = getenv("WORLD_SIZE");
      if (e != nullptr) v = atoi(e) != 0;
      else v = (ws != nullptr && 2 * atoi(ws) > host_cores()) ? 1 : 0;
    }
  return v == 1;
}

// How many of the library's streams run side by side?  (dmh_core.hip init_device asks the runtime for twelve hardware
// queues, which only works when the process has not used HIP before.)  Eight streams get a kernel each that spins for
// 1 ms of the constant-rate clock: side by side they are done in 1.2 ms, on the runtime's default of four queues in 2.2,
// on one in 8.  Once per process; the answer sizes the pile streams (dmh_chain.hip) and a
// host that gets the default is told so.
__global__ void k_queue_probe(unsigned long long ticks)
{ const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

static int g_hw_queues_seen = 0;       // 0: not probed; else the streams found to run side by side (1..8)
int hw_queues_seen() { return g_hw_queues_seen; }

static void probe_hw_queues()
{ static const bool off = getenv("DMH_QUEUE_PROBE") != nullptr && atoi(getenv("DMH_QUEUE_PROBE")) == 0;
  if (off) { g_hw_queues_seen = 8; return; }
  constexpr int NP = 8;
  hipStream_t ps[NP];
  for (int i = 0; i < NP; i++) HIP_CHECK(hipStreamCreateWithFlags(&ps[i], hipStreamNonBlocking));
  // (the first launch of a kernel loads its code object, the first on a stream makes its queue: once untimed)
  for (int i = 0; i < NP; i++) hipLaunchKernelGGL(k_queue_probe, dim3(1), dim3(64), 0, ps[i], 100ull);
  HIP_CHECK(hipDeviceSynchronize());
  const unsigned long long ticks = 100000ull;           // 1 ms at 100 MHz
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < NP; i++) hipLaunchKernelGGL(k_queue_probe, dim3(1), dim3(64), 0, ps[i], ticks);
  HIP_CHECK(hipDeviceSynchronize());
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  for (int i = 0; i < NP; i++) HIP_CHECK(hipStreamDestroy(ps[i]));
  const int rounds = std::max(1, (int) ((us + 400.0) / 1000.0));       // kernels that ran one behind the other (+ 0.2 ms of launches)
  g_hw_queues_seen = std::max(1, std::min(NP, (NP + rounds - 1) / rounds));
  if (getenv("DMH_STATS") != nullptr)
    fprintf(stderr, "dmh: eight probe kernels on eight streams took %.0f us: about %d run side by side\n", us, g_hw_queues_seen);
  if (g_hw_queues_seen < NP)
    fprintf(stderr, "libdamapper_hip: the HIP runtime of this process runs only about %d streams side by side (its default is four "
            "hardware queues); the library keeps nine busy.  Export GPU_MAX_HW_QUEUES=12 before the process starts (INTEGRATION.md 3): "
            "a read block takes a quarter longer without.\n", g_hw_queues_seen);
}

static void init_device()
{ if (g_init) return;
  int n = 0;
  // The filter keeps six streams busy next to each other (main, tail, finish, piles, alignment, upload) and the HIP runtime
  // maps streams onto FOUR hardware queues by default, in creation order: two streams on one queue run one behind the
  // other -- with the tail stream behind the pile stream's 30 ms kernels a read block took 141 ms instead of 124, and which
  // streams meet depends on the order the code happens to create them in.  Twelve queues: every stream its own (main, upload, tail, finish, alignment, four for the piles, the caller's) (read when the
  // runtime starts: no effect when the process has used HIP before this library, see damapper_amd/__init__.py)
  setenv("GPU_MAX_HW_QUEUES", "12", 0);
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    fatal("no HIP device available (%s): libdamapper_hip has no CPU fallback",
          e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  int dev = 0;
  const char *lr = getenv("LOCAL_RANK");
  if (lr != nullptr && getenv("DMH_IGNORE_LOCAL_RANK") == nullptr)
    dev = atoi(lr) % n;
  const char *dv = getenv("DMH_DEVICE");
  if (dv != nullptr) dev = atoi(dv) % n;
  HIP_CHECK(hipSetDevice(dev));
  if (getenv("DMH_SPIN") != nullptr) (void) hipSetDeviceFlags(hipDeviceScheduleSpin);
  else if (host_crowded())
    { if (hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) (void) hipGetLastError(); }
  probe_hw_queues();
  HIP_CHECK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
  HIP_CHECK(hipStreamCreateWithFlags(&g_upstream, hipStreamNonBlocking));
  g_init = true;
  // DMH_TIMING_AT_EXIT=1: stage and host timers on from the start, printed when the process ends -- for drivers that
  // do not know the dmh_timing_* calls (the reference's damapper.c linked against this library)
  if (getenv("DMH_TIMING_AT_EXIT") != nullptr)
    { dmh_timing_enable(1);
      atexit(timing_dump_at_exit);
    }
}

static thread_local hipStream_t g_stream_scope = nullptr;
static hipStream_t g_fstream = nullptr;

hipStream_t stream()
{ init_device();
  return g_stream_scope != nullptr ? g_stream_scope : g_stream;
}

hipStream_t main_stream()
{ init_device();
  return g_stream;
}

hipStream_t finish_stream()
{ init_device();
  static const int env_fs = getenv("DMH_FIN_STREAM") ? atoi(getenv("DMH_FIN_STREAM")) : 1;
  const int on = S().opt_fin_stream >= 0 ? S().opt_fin_stream : env_fs;
  if (on == 0) return g_stream;
  if (g_fstream == nullptr)
    { // High priority: its kernels are small and the host waits for them (three round trips per batch); behind the
      // workgroups of a sort or a chain kernel they took a third longer (DMH_FIN_PRIO=0: default priority, -1: lowest)
      static const int prio = getenv("DMH_FIN_PRIO") ? atoi(getenv("DMH_FIN_PRIO")) : 1;
      if (prio != 0)
        { int least = 0, greatest = 0;
          HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
          HIP_CHECK(hipStreamCreateWithPriority(&g_fstream, hipStreamNonBlocking, prio > 0 ? greatest : least));
        }
      else
        HIP_CHECK(hipStreamCreateWithFlags(&g_fstream, hipStreamNonBlocking));
    }
  return g_fstream;
}

// The chain stage's tail -- the lane-group and wavefront-per-group launches that walk what k_chain_lane has handed
// on: a few hundred groups, their time the LENGTH of the longest walk, the GPU nearly empty -- runs here, next to the
// following pass's match kernels (DMH_CHAIN_TAIL=0 or the "fin_stream" test option = 0: on the main stream).
static hipStream_t g_tstream = nullptr;
hipStream_t tail_stream()
{ init_device();
  static const int env_ts = getenv("DMH_CHAIN_TAIL") ? atoi(getenv("DMH_CHAIN_TAIL")) : 1;
  if (env_ts == 0 || S().opt_fin_stream == 0 || S().opt_tails == 0) return g_stream;
  // (default priority: at the highest 149, at the lowest 150.5 against 146 ms per read block)
  if (g_tstream == nullptr) HIP_CHECK(hipStreamCreateWithFlags(&g_tstream, hipStreamNonBlocking));
  return g_tstream;
}

// The front of the NEXT REF block (its match kernels, fix-ups, hit sums) next to the per-read kernels of the block at hand:
// "front_ahead" test option / DMH_FRONT_AHEAD = 0: everything on the main stream, one REF block after the other.
static hipStream_t g_frstream = nullptr;
hipStream_t front_stream()
{ init_device();
  static const int env_fa = getenv("DMH_FRONT_AHEAD") ? atoi(getenv("DMH_FRONT_AHEAD")) : 1;
  const int on = S().opt_front_ahead >= 0 ? S().opt_front_ahead : env_fa;
  if (on == 0 || S().opt_tails == 0 || hw_queues_seen() < 8) return g_stream;
  if (g_frstream == nullptr)
    { static const int prio = getenv("DMH_FRONT_PRIO") ? atoi(getenv("DMH_FRONT_PRIO")) : 0;
      if (prio != 0)
        { int least = 0, greatest = 0;
          HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
          HIP_CHECK(hipStreamCreateWithPriority(&g_frstream, hipStreamNonBlocking, prio > 0 ? greatest : least));
        }
      else
        HIP_CHECK(hipStreamCreateWithFlags(&g_frstream, hipStreamNonBlocking));
    }
  return g_frstream;
}

void join_side_streams()
{ static hipEvent_t ev = nullptr;
  if (ev == nullptr) HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  if (g_frstream != nullptr)
    { HIP_CHECK(hipEventRecord(ev, g_frstream));
      HIP_CHECK(hipStreamWaitEvent(g_stream, ev, 0));
    }
  if (g_tstream != nullptr)
    { HIP_CHECK(hipEventRecord(ev, g_tstream));
      HIP_CHECK(hipStreamWaitEvent(g_stream, ev, 0));
    }
  if (g_fstream != nullptr)
    { HIP_CHECK(hipEventRecord(ev, g_fstream));
      HIP_CHECK(hipStreamWaitEvent(g_stream, ev, 0));
    }
}

StreamScope::StreamScope(hipStream_t s) : prev(g_stream_scope) { g_stream_scope = (s == g_stream) ? nullptr : s; }
StreamScope::~StreamScope() { g_stream_scope = prev; }

static void sync_all_streams()
{ init_device();
  HIP_CHECK(hipStreamSynchronize(g_upstream));
  if (g_tstream != nullptr) HIP_CHECK(hipStreamSynchronize(g_tstream));
  if (g_frstream != nullptr) HIP_CHECK(hipStreamSynchronize(g_frstream));
  if (g_fstream != nullptr) HIP_CHECK(hipStreamSynchronize(g_fstream));
  HIP_CHECK(hipStreamSynchronize(g_stream));
}

// Device memory comes from a small caching pool: hipMalloc/hipFree of multi-GB buffers cost
// milliseconds each and the block loops allocate the same sizes over and over.
// A block goes back to the pool while kernels queued earlier on the stream of the code that frees it may still use it:
// harmless when the next user queues its work on the same stream, not when it is the filter's finish stream (or the
// main stream, the other way round).  Every free block therefore remembers the stream it was freed under, blocks of the
// asking stream are preferred, and a block that changes streams is only handed out once its old stream has drained.
// (ev: recorded on the owner stream when the block was freed -- once it has completed nobody can be using the block any
//  more and it changes streams without anybody waiting.  Without it a finish-stream allocation that picked a block the
//  main stream had freed made the host wait until the main stream ran dry: 3 ms of idle GPU, three times per read block.)
struct PoolBlock { void *p; size_t cap; hipStream_t owner; hipEvent_t ev; };
static std::vector<PoolBlock>    g_free;
static std::vector<hipEvent_t>   g_pool_events;      // spare events

static hipEvent_t pool_event()
{ if (!g_pool_events.empty())
    { hipEvent_t e = g_pool_events.back();
      g_pool_events.pop_back();
      return e;
    }
  hipEvent_t e;
  HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return e;
}

static bool pool_block_idle(const PoolBlock &b)
{ if (b.ev == nullptr) return false;
  const hipError_t e = hipEventQuery(b.ev);
  if (e == hipSuccess) return true;
  (void) hipGetLastError();
  return false;
}

static void pool_drop(PoolBlock &b)          // the block goes back to the driver
{ (void) hipFree(b.p);
  if (b.ev != nullptr) g_pool_events.push_back(b.ev);
  b.ev = nullptr;
}
static std::map<void *, size_t>  g_live;
static std::map<void *, const char *> g_live_what;     // (labels, for the out-of-memory message)
static size_t                    g_pooled = 0;
static const size_t              POOL_MAX = (size_t) 96 << 30;

static bool (*g_reclaim)() = nullptr;
static bool g_reclaiming = false;

// DMH_ZERO_FILL=1 (debugging): every block handed out starts as zeros -- device memory comes back from the driver, and
// from this pool, with whatever its last user left in it.  (Was the default for a few hours while a device fault that
// looked like a read of uninitialised memory was hunted; the cause was an over-read in k_kmer_emit)
static const bool g_zero_fill = getenv("DMH_ZERO_FILL") != nullptr && getenv("DMH_NO_ZERO_FILL") == nullptr;
static int g_alloc_fail = 0;       // test option "alloc_fail" = n: the n-th allocation from now behaves as if memory had run out

// what an allocation that fails does before it retries, without the failure (tests: "alloc_fail", "oom_finish")
void simulate_out_of_memory()
{ if (g_reclaim == nullptr || g_reclaiming) return;
  g_reclaiming = true;
  S().simulated_oom = true;
  (void) g_reclaim();
  S().simulated_oom = false;
  g_reclaiming = false;
  for (auto &b : g_free) pool_drop(b);
  g_free.clear();
  g_pooled = 0;
}

double dev_live_gb();

void *dev_alloc(size_t bytes, const char *what)
{ init_device();
  if (bytes == 0) bytes = 16;
  if (g_alloc_fail > 0 && --g_alloc_fail == 0) simulate_out_of_memory();
  // quantise sizes (8 steps per octave above 1 MB) so that near-equal requests -- read blocks
  // differ by a fraction of a percent -- land in the same pool bucket instead of in hipMalloc
  { size_t g = 256;
    if (bytes > ((size_t) 1 << 20))
      { size_t p2 = (size_t) 1 << (63 - __builtin_clzll((unsigned long long) bytes));
        g = p2 >> 3;
      }
    bytes = (bytes + g - 1) / g * g;
  }
  int best = -1;
  const hipStream_t me = stream();
  bool idle = true;
  for (int pass = 0; pass < 3 && best < 0; pass++)          // first the blocks this stream freed itself, then the ones
    for (size_t i = 0; i < g_free.size(); i++)              // whose last user has finished, then the others (and wait)
      if (g_free[i].cap >= bytes && g_free[i].cap <= 2 * bytes + (1 << 20) &&
          (pass == 2 || g_free[i].owner == me || (pass == 1 && pool_block_idle(g_free[i]))))
        if (best < 0 || g_free[i].cap < g_free[best].cap) { best = (int) i; idle = (pass < 2); }
  // DMH_POISON=<byte> (debugging): every allocation starts out filled with that byte, so that a kernel that reads what nobody wrote
  // misbehaves every time and not only when the memory happens to hold another process's leftovers
  // DMH_POISON_WHAT=a|b|c: only the allocations whose label contains one of these; DMH_POISON_LOG: print every label
  static const char *pwhat = getenv("DMH_POISON_WHAT");
  static const bool  plog  = getenv("DMH_POISON_LOG") != nullptr;
  bool poison = getenv("DMH_POISON") != nullptr;
  if (poison && pwhat != nullptr)
    { poison = false;
      std::string list(pwhat);
      size_t a = 0;
      while (a <= list.size())
        { size_t b = list.find('|', a);
          if (b == std::string::npos) b = list.size();
          if (b > a && strstr(what, list.substr(a, b - a).c_str()) != nullptr) poison = true;
          a = b + 1;
        }
    }
  if (plog) { fprintf(stderr, "dmh: alloc [%s]\n", what); fflush(stderr); }
  // DMH_ZERO_WHAT=a|b|c: these allocations are zero-filled whatever else is set (who READS the leftovers?)
  static const char *zwhat = getenv("DMH_ZERO_WHAT");
  bool zero_this = false;
  if (zwhat != nullptr)
    { std::string list(zwhat);
      size_t a = 0;
      while (a <= list.size())
        { size_t b = list.find('|', a);
          if (b == std::string::npos) b = list.size();
          if (b > a && strstr(what, list.substr(a, b - a).c_str()) != nullptr) zero_this = true;
          a = b + 1;
        }
      if (zero_this) poison = false;
    }
  static const int  pbyte = getenv("DMH_POISON") ? (int) strtol(getenv("DMH_POISON"), nullptr, 0) & 0xff : 0;      // (the byte to fill with)
  if (best >= 0)
    { PoolBlock b = g_free[best];
      g_free.erase(g_free.begin() + best);
      g_pooled -= b.cap;
      g_live[b.p] = b.cap;
      g_live_what[b.p] = what;
      if (!idle)
        { HostTimer ht("pool_owner_wait");
          HIP_CHECK(hipStreamSynchronize(b.owner));
        }
      if (b.ev != nullptr) g_pool_events.push_back(b.ev);
      if (poison)
        { HIP_CHECK(hipMemsetAsync(b.p, pbyte, b.cap, me));
          HIP_CHECK(hipStreamSynchronize(me));
        }
      else if (g_zero_fill || zero_this)
        { HIP_CHECK(hipMemsetAsync(b.p, 0, b.cap, me));          // (finished before the block is handed out: its first
          HIP_CHECK(hipStreamSynchronize(me));                   //  writer may well be another stream, e.g. an upload)
        }
      return b.p;
    }
  void *p = nullptr;
  hipError_t e;
  { HostTimer ht("pool_hipMalloc");
    e = hipMalloc(&p, bytes);
  }
  { static const bool log_miss = getenv("DMH_POOL_LOG") != nullptr;        // (which buffers the pool does not hold)
    if (log_miss) fprintf(stderr, "dmh: pool miss %10.1f MB  [%s]  pooled %.1f GB in %zu blocks\n", bytes / 1048576.0, what, g_pooled / 1e9, g_free.size());
  }
  if (e != hipSuccess)
    { // give the pool back and retry once
      (void) hipGetLastError();            // (the failure must not surface at some later launch check)
      for (auto &b : g_free) pool_drop(b);
      g_free.clear();
      g_pooled = 0;
      e = hipMalloc(&p, bytes);
      if (e != hipSuccess) (void) hipGetLastError();
    }
  if (e != hipSuccess && getenv("DMH_STATS") != nullptr)
    { size_t fr = 0, tot = 0;
      (void) hipMemGetInfo(&fr, &tot);
      fprintf(stderr, "dmh: allocation of %zu bytes (%s) failed; reclaim hook %s, %s; device %.1f of %.1f GB free, library holds %.1f GB\n",
              bytes, what, g_reclaim != nullptr ? "set" : "not set", g_reclaiming ? "already reclaiming" : "free", fr / 1e9, tot / 1e9,
              dev_live_gb());
    }
  if (e != hipSuccess && g_reclaim != nullptr && !g_reclaiming)
    { // last resort: what other stages can do without (batches in flight are finished, their buffers dropped)
      (void) hipGetLastError();
      g_reclaiming = true;
      const bool got = g_reclaim();
      g_reclaiming = false;
      if (got)
        { for (auto &b : g_free) pool_drop(b);
          g_free.clear();
          g_pooled = 0;
          e = hipMalloc(&p, bytes);
          if (e != hipSuccess) (void) hipGetLastError();
        }
    }
  if (e != hipSuccess)
    { size_t fr = 0, tot = 0, live = 0;
      (void) hipMemGetInfo(&fr, &tot);
      for (auto &kv : g_live) live += kv.second;
      { // the largest holders by label
        std::map<std::string, double> by;
        for (auto &kv : g_live) by[g_live_what.count(kv.first) ? g_live_what[kv.first] : "?"] += kv.second / 1e9;
        std::vector<std::pair<double, std::string>> top;
        for (auto &kv : by) top.push_back({-kv.second, kv.first});
        std::sort(top.begin(), top.end());
        for (size_t i = 0; i < top.size() && i < 16; i++)
          fprintf(stderr, "damapper_hip:   %8.2f GB  %s\n", -top[i].first, top[i].second.c_str());
      }
      fatal("out of device memory allocating %zu bytes (%s): %s [device: %.1f of %.1f GB free; this library holds %.1f GB in %zu "
            "buffers, %.1f GB pooled]", bytes, what, hipGetErrorString(e), fr / 1e9, tot / 1e9, live / 1e9, g_live.size(),
            g_pooled / 1e9);
    }
  g_live[p] = bytes;
  g_live_what[p] = what;
  if (poison)
    { HIP_CHECK(hipMemsetAsync(p, pbyte, bytes, me));
      HIP_CHECK(hipStreamSynchronize(me));
    }
  else if (g_zero_fill || zero_this)
    { HIP_CHECK(hipMemsetAsync(p, 0, bytes, me));
      HIP_CHECK(hipStreamSynchronize(me));
    }
  return p;
}

void set_reclaim_hook(bool (*fn)()) { g_reclaim = fn; }

double dev_live_gb()
{ double b = 0;
  for (auto &kv : g_live) b += (double) kv.second;
  return b / 1e9;
}

void dev_free(void *p)
{ if (p == nullptr) return;
  auto it = g_live.find(p);
  if (it == g_live.end())
    { HIP_CHECK(hipFree(p));
      return;
    }
  size_t cap = it->second;
  g_live.erase(it);
  if (g_pooled + cap > POOL_MAX)
    { HIP_CHECK(hipStreamSynchronize(stream()));
      HIP_CHECK(hipFree(p));
      return;
    }
  hipEvent_t ev = pool_event();
  HIP_CHECK(hipEventRecord(ev, stream()));
  g_free.push_back({p, cap, stream(), ev});
  g_pooled += cap;
}

static void pool_release()
{ for (auto &b : g_free) pool_drop(b);
  g_free.clear();
  g_pooled = 0;
}

static uint8_t *g_pin = nullptr;
static const size_t PIN_CHUNK = (size_t) 32 << 20;

static void pin_init()
{ if (g_pin == nullptr)
    HIP_CHECK(hipHostMalloc((void **) &g_pin, PIN_CHUNK, hipHostMallocDefault));
}

void copy_d2h(void *host, const void *dev, size_t bytes, int pump_stages)
{ pin_init();
  static hipEvent_t ev = nullptr;
  size_t off = 0;
  do
    { size_t c = std::min(PIN_CHUNK, bytes - off);
      if (c > 0)
        HIP_CHECK(hipMemcpyAsync(g_pin, (const uint8_t *) dev + off, c, hipMemcpyDeviceToHost, stream()));
      if (pump_stages > 0 && off + c >= bytes)
        { // wait for the copy only; the background stages queued behind it keep the GPU busy while
          // the caller digests the data on the host
          if (ev == nullptr) HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
          HIP_CHECK(hipEventRecord(ev, stream()));
          bg_pump(pump_stages);
          HIP_CHECK(hipEventSynchronize(ev));
        }
      else
        HIP_CHECK(hipStreamSynchronize(stream()));
      if (c > 0) memcpy((uint8_t *) host + off, g_pin, c);
      off += c;
    }
  while (off < bytes);
}

void copy_h2d(void *dev, const void *host, size_t bytes)
{ pin_init();
  if (bytes <= PIN_CHUNK)
    { memcpy(g_pin, host, bytes);
      HIP_CHECK(hipMemcpyAsync(dev, g_pin, bytes, hipMemcpyHostToDevice, stream()));
      HIP_CHECK(hipStreamSynchronize(stream()));
      return;
    }
  // large transfers (whole blocks from pageable memory): two staging buffers, the host fills one
  // while the DMA engine drains the other
  static uint8_t   *pin2 = nullptr;
  static hipEvent_t ev[2] = {nullptr, nullptr};
  if (pin2 == nullptr)
    { HIP_CHECK(hipHostMalloc((void **) &pin2, PIN_CHUNK, hipHostMallocDefault));
      HIP_CHECK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
      HIP_CHECK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    }
  uint8_t *buf[2] = {g_pin, pin2};
  bool     used[2] = {false, false};
  size_t   off = 0;
  int      k = 0;
  while (off < bytes)
    { size_t c = std::min(PIN_CHUNK, bytes - off);
      if (used[k]) HIP_CHECK(hipEventSynchronize(ev[k]));
      memcpy(buf[k], (const uint8_t *) host + off, c);
      HIP_CHECK(hipMemcpyAsync((uint8_t *) dev + off, buf[k], c, hipMemcpyHostToDevice, stream()));
      HIP_CHECK(hipEventRecord(ev[k], stream()));
      used[k] = true;
      off += c;
      k ^= 1;
    }
  HIP_CHECK(hipStreamSynchronize(stream()));
}

// Is p page-locked host memory known to the runtime (dmh_host_alloc, hipHostMalloc, hipHostRegister)?
static bool is_pinned_host(const void *p)
{ hipPointerAttribute_t at;
  hipError_t e = hipPointerGetAttributes(&at, p);
  if (e != hipSuccess)
    { (void) hipGetLastError();          // pageable memory: not an error for us
      return false;
    }
  return at.type == hipMemoryTypeHost;
}

// ---- stage timing -----------------------------------------------------------------------

struct TimerSlot { std::string name; double ms = 0; int64_t launches = 0; double bytes = 0; };
struct PendingEv { int idx; hipEvent_t e0, e1; };
static std::mutex g_tmu;                 // timers are also used from the report worker thread
static std::vector<TimerSlot> g_slots;
static std::vector<PendingEv> g_pending;
static bool g_timing = false;
static bool g_timing_roofline_only = false;     // dmh_timing_enable(2): events around the roofline kernel only

static std::vector<hipEvent_t> g_timer_events;       // spare timing events
static size_t g_harvest_at = 256;

static hipEvent_t timer_event()          // (g_tmu held)
{ if (!g_timer_events.empty())
    { hipEvent_t e = g_timer_events.back();
      g_timer_events.pop_back();
      return e;
    }
  hipEvent_t e;
  HIP_CHECK(hipEventCreate(&e));
  return e;
}

// The event pairs that have completed are read and go back to the spare list.  (Left to pile up until the timings were
// asked for -- 40,000 live events after six bench steps -- every step ran slower than the one before: 1341, 1378,
// 1399 ms against 1300 without events.)
static void harvest_timers()             // (g_tmu held)
{ size_t keep = 0;
  for (size_t i = 0; i < g_pending.size(); i++)
    { PendingEv &p = g_pending[i];
      const hipError_t e = hipEventQuery(p.e1);
      if (e == hipSuccess)
        { float ms = 0;
          HIP_CHECK(hipEventElapsedTime(&ms, p.e0, p.e1));
          g_slots[p.idx].ms += ms;
          g_timer_events.push_back(p.e0);
          g_timer_events.push_back(p.e1);
        }
      else
        { (void) hipGetLastError();
          g_pending[keep++] = p;
        }
    }
  g_pending.resize(keep);
  g_harvest_at = keep + 256;
}

static int slot_of(const char *name)
{ for (size_t i = 0; i < g_slots.size(); i++)
    if (g_slots[i].name == name) return (int) i;
  g_slots.push_back(TimerSlot());
  g_slots.back().name = name;
  return (int) g_slots.size() - 1;
}

// DMH_TRACE_STAGES (debugging a device fault): every stage prints its name and waits for its stream at its end
static const bool g_trace_stages = getenv("DMH_TRACE_STAGES") != nullptr;
static const char *g_trace_prefix = getenv("DMH_TRACE_PREFIX");        // only the stages whose names start like this wait

StageTimer::StageTimer(const char *name, double bytes, hipStream_t on)
{ active = g_timing && (!g_timing_roofline_only || strcmp(name, "k_radix_scatter") == 0);
  idx = -1;
  s = on ? on : stream();
  tname = name;
  if (g_trace_stages && g_trace_prefix == nullptr) { fprintf(stderr, "dmh: stage %s ...\n", name); fflush(stderr); }
  if (!active) return;
  { std::lock_guard<std::mutex> lk(g_tmu);
    idx = slot_of(name);
    g_slots[idx].bytes += bytes;
    g_slots[idx].launches += 1;
  }
  { std::lock_guard<std::mutex> lk(g_tmu);
    if (g_pending.size() >= g_harvest_at) harvest_timers();
    e0 = timer_event();
    e1 = timer_event();
  }
  HIP_CHECK(hipEventRecord(e0, s));
}

StageTimer::~StageTimer()
{ if (g_trace_stages && (g_trace_prefix == nullptr || strncmp(tname, g_trace_prefix, strlen(g_trace_prefix)) == 0))
    { HIP_CHECK(hipStreamSynchronize(s));
      if (g_trace_prefix == nullptr) { fprintf(stderr, "dmh: stage %s done\n", tname); fflush(stderr); }
    }
  if (!active) return;
  HIP_CHECK(hipEventRecord(e1, s));
  std::lock_guard<std::mutex> lk(g_tmu);
  g_pending.push_back({idx, e0, e1});
}

static double now_ms()
{ struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

HostTimer::HostTimer(const char *name)
{ active = g_timing && !g_timing_roofline_only;
  idx = -1;
  t0 = 0;
  if (!active) return;
  { std::lock_guard<std::mutex> lk(g_tmu);        // host timers also run on the report threads
    idx = slot_of((std::string("host:") + name).c_str());
    g_slots[idx].launches += 1;
  }
  t0 = now_ms();
}

HostTimer::~HostTimer()
{ if (!active) return;
  double dt = now_ms() - t0;
  std::lock_guard<std::mutex> lk(g_tmu);
  if (idx < (int) g_slots.size()) g_slots[idx].ms += dt;     // (the table may have been reset meanwhile)
}

// a counter in the timing table: name, 0 ms, n "launches" (e.g. the candidates the d-wave kernel has to run again, by cause)
void timing_count(const char *name, int64_t n)
{ if (!g_timing || g_timing_roofline_only) return;
  std::lock_guard<std::mutex> lk(g_tmu);
  g_slots[slot_of(name)].launches += n;
}

static void drain_timers()
{ if (g_pending.empty()) return;
  report_finish_pending();          // (its kernel's events live on the alignment stream)
  sync_all_streams();
  std::lock_guard<std::mutex> lk(g_tmu);
  for (auto &p : g_pending)
    { float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, p.e0, p.e1));
      g_slots[p.idx].ms += ms;
      g_timer_events.push_back(p.e0);
      g_timer_events.push_back(p.e1);
    }
  g_pending.clear();
  g_harvest_at = 256;
}

// ---- block upload -----------------------------------------------------------------------

static uint64_t block_stamp(const DAZZ_DB *b)
{ const uint8_t *p = (const uint8_t *) b->bases;
  int64_t L = b->reads[b->nreads].boff;
  uint64_t h = 1469598103934665603ull ^ (uint64_t) L ^ ((uint64_t) b->nreads << 40);
  int64_t step = L / 257 + 1;
  for (int64_t i = 0; i < L; i += step)
    { int64_t e = std::min<int64_t>(L, i + 16);
      for (int64_t j = i; j < e; j++) h = (h ^ p[j]) * 1099511628211ull;
    }
  return h;
}

void block_ready(BlockDev &bd)
{ if (bd.uploaded == nullptr) return;
  HIP_CHECK(hipStreamWaitEvent(stream(), bd.uploaded, 0));
  HIP_CHECK(hipEventDestroy(bd.uploaded));
  bd.uploaded = nullptr;
  for (void *&t : bd.d_tmp)          // (the main stream is behind the decode now: whoever gets this memory next, too)
    { dev_free(t); t = nullptr; }
}

void free_blockdev(BlockDev &bd)
{ if (bd.uploaded != nullptr)
    { HIP_CHECK(hipEventSynchronize(bd.uploaded));
      HIP_CHECK(hipEventDestroy(bd.uploaded));
      bd.uploaded = nullptr;
    }
  for (void *&t : bd.d_tmp)
    { dev_free(t); t = nullptr; }
  dev_free(bd.d_raw);
  dev_free(bd.d_boff);
  dev_free(bd.d_pkraw);
  dev_free(bd.d_rcraw);
  bd.d_raw = nullptr;
  bd.d_boff = nullptr;
  bd.d_pkraw = nullptr;
  bd.d_rcraw = nullptr;
}

// ---- 2-bit packing for the alignment kernel ------------------------------------------------

__global__ void k_pack_fwd(const uint8_t *__restrict__ bases, int64_t L, uint64_t *__restrict__ pk, int64_t nwords)
{ int64_t w = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  uint64_t v = 0;
  int64_t  q = w << 5;
#pragma unroll 8
  for (int t = 0; t < 32; t++)
    { uint64_t b = (q + t < L) ? bases[q + t] : 0;
      v |= (b & 3) << (2 * t);
    }
  pk[w] = v;
}

// Packed reverse complement of every read, stored at the read's own offsets:
// rc[boff[r]+i] = 3 - a[boff[r] + rlen_r - 1 - i]   (map.c:1940-1948 builds the same per read)
__global__ void k_pack_rc(const uint8_t *__restrict__ bases, int64_t L, const int64_t *__restrict__ boff,
                          int nreads, uint64_t *__restrict__ pk, int64_t nwords)
{ int64_t w = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  int64_t q0 = w << 5;
  int lo = 0, hi = nreads;                 // largest r with boff[r] <= q0
  while (lo < hi)
    { int m = (lo + hi + 1) >> 1;
      if (boff[m] > q0) hi = m - 1; else lo = m;
    }
  int r = lo;
  uint64_t v = 0;
  for (int t = 0; t < 32; t++)
    { int64_t q = q0 + t;
      if (q >= L) break;
      while (r < nreads && q >= boff[r + 1]) r += 1;
      if (r >= nreads) break;
      int64_t s = boff[r], rl = boff[r + 1] - s - 1;
      int64_t i = q - s;
      if (i >= 0 && i < rl)
        { uint64_t b = 3 - (uint64_t) (bases[s + rl - 1 - i] & 3);
          v |= b << (2 * t);
        }
    }
  pk[w] = v;
}

void pack_bases(const uint8_t *d_bases, int64_t L, uint64_t *d_pkraw, hipStream_t on)
{ if (on == nullptr) on = stream(); int64_t nwords = (L >> 5) + 1;
  HIP_CHECK(hipMemsetAsync(d_pkraw, 0, sizeof(uint64_t) * (nwords + 4), on));
  StageTimer tm("pack_bases", (double) L, on);
  hipLaunchKernelGGL(k_pack_fwd, dim3(ceil_div(nwords, 256)), dim3(256), 0, on, d_bases, L, d_pkraw + 2, nwords);
  HIP_CHECK(hipGetLastError());
}

void pack_revcomp(const uint8_t *d_bases, int64_t L, const int64_t *d_boff, int nreads, uint64_t *d_rcraw, hipStream_t on)
{ if (on == nullptr) on = stream(); int64_t nwords = (L >> 5) + 1;
  HIP_CHECK(hipMemsetAsync(d_rcraw, 0, sizeof(uint64_t) * (nwords + 4), on));
  StageTimer tm("pack_revcomp", (double) L, on);
  hipLaunchKernelGGL(k_pack_rc, dim3(ceil_div(nwords, 256)), dim3(256), 0, on, d_bases, L, d_boff, nreads,
                     d_rcraw + 2, nwords);
  HIP_CHECK(hipGetLastError());
}

void ensure_packed(BlockDev &bd, bool want_rc)
{ block_ready(bd); int64_t nwords = (bd.L >> 5) + 1;
  if (bd.d_pkraw == nullptr)
    { bd.d_pkraw = (uint64_t *) dev_alloc(sizeof(uint64_t) * (nwords + 4), "packed bases");
      pack_bases(bd.d_bases(), bd.L, bd.d_pkraw, stream());
    }
  if (want_rc && bd.d_rcraw == nullptr)
    { bd.d_rcraw = (uint64_t *) dev_alloc(sizeof(uint64_t) * (nwords + 4), "packed reverse complements");
      pack_revcomp(bd.d_bases(), bd.L, bd.d_boff, bd.nreads, bd.d_rcraw, stream());
    }
}

BlockDev &upload_block(const DAZZ_DB *block, bool pin)
{ State &st = S();
  auto it = st.blocks.find(block);
  if (it != st.blocks.end() && it->second.decoded)       // dmh_preload_block_2bit: there are no host bases
    { if (it->second.nreads != block->nreads) fatal("block changed after dmh_preload_block_2bit");
      return it->second;
    }
  if (block->bases == nullptr) fatal("block has no bases (not loaded, or released after dmh_preload_block_2bit)");
  uint64_t stamp = block_stamp(block);
  if (it != st.blocks.end())
    { if (it->second.stamp == stamp && it->second.nreads == block->nreads)
        { if (pin) it->second.pinned = true;
          return it->second;
        }
      free_blockdev(it->second);
      st.blocks.erase(it);
    }
  BlockDev bd;
  int n = block->nreads;
  bd.nreads = n;
  bd.L = block->reads[n].boff;
  bd.stamp = stamp;
  bd.pinned = pin;
  bd.d_raw = (uint8_t *) dev_alloc((size_t) bd.L + 1 + 64, "block bases");     // (+64: k_kmer_emit reads whole 16-byte lines)
  // the byte before bases[0] is the leading separator (DB.c:1411)
  const uint8_t *src = ((const uint8_t *) block->bases) - 1;
  bd.d_boff = (int64_t *) dev_alloc(sizeof(int64_t) * (n + 1), "block read offsets");
  if (is_pinned_host(src))
    { // the driver loaded the block into page-locked memory (dmh_host_alloc): one DMA straight from it
      // on the upload stream, next to whatever kernels the main stream is running; the first kernel
      // that reads the copy waits for it (block_ready).  The caller keeps the block unchanged until
      // a library call that uses it has returned (Match_Filter and Reporter synchronise).  Nothing
      // here waits for the main stream.
      static int64_t   *h_off = nullptr;       // page-locked staging for the offsets, reused per upload
      static size_t     h_cap = 0;
      static hipEvent_t h_free = nullptr;
      if (h_free == nullptr) HIP_CHECK(hipEventCreateWithFlags(&h_free, hipEventDisableTiming));
      else HIP_CHECK(hipEventSynchronize(h_free));
      if ((size_t) n + 1 > h_cap)
        { if (h_off) (void) hipHostFree(h_off);
          h_cap = (size_t) n + 1 + 1024;
          HIP_CHECK(hipHostMalloc((void **) &h_off, sizeof(int64_t) * h_cap, hipHostMallocDefault));
        }
      for (int i = 0; i <= n; i++) h_off[i] = block->reads[i].boff;
      HIP_CHECK(hipMemcpyAsync(bd.d_boff, h_off, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, g_upstream));
      HIP_CHECK(hipEventRecord(h_free, g_upstream));
      HIP_CHECK(hipMemcpyAsync(bd.d_raw, src, (size_t) bd.L + 1, hipMemcpyHostToDevice, g_upstream));
      HIP_CHECK(hipEventCreateWithFlags(&bd.uploaded, hipEventDisableTiming));
      HIP_CHECK(hipEventRecord(bd.uploaded, g_upstream));
    }
  else
    { copy_h2d(bd.d_raw, src, (size_t) bd.L + 1);
      std::vector<int64_t> off(n + 1);
      for (int i = 0; i <= n; i++) off[i] = block->reads[i].boff;
      copy_h2d(bd.d_boff, off.data(), sizeof(int64_t) * (n + 1));
      HIP_CHECK(hipStreamSynchronize(stream()));
    }
  st.blocks[block] = bd;
  return st.blocks[block];
}

void drop_block(const DAZZ_DB *block)
{ State &st = S();
  auto it = st.blocks.find(block);
  if (it == st.blocks.end()) return;
  free_blockdev(it->second);
  st.blocks.erase(it);
}

// ---- Load_All_Reads on the device (DB.c:1389-1437 with Uncompress_Read, DB.c:342-361) -----------------
// The block's slice of the .bps file: 4 bases per byte, the first in the two top bits, every read starting on
// a byte; `coff[r]` = first byte of read r inside the slice (ascending; bytes of reads the trimmed view leaves
// out lie between).  One workgroup decodes UNPACK_SPAN bytes of the slice: it looks up the read its first
// byte belongs to once, each thread walks on from there.
constexpr int UNPACK_THREADS = 256;
constexpr int UNPACK_SPAN    = 4096;

__global__ __launch_bounds__(UNPACK_THREADS) void k_unpack_2bit(const uint8_t *__restrict__ bps, int64_t nbytes,
                                                                const int64_t *__restrict__ coff,
                                                                const int64_t *__restrict__ boff, int nreads,
                                                                uint8_t *__restrict__ raw)
{ __shared__ int s_first;
  const int64_t span0 = (int64_t) blockIdx.x * UNPACK_SPAN;
  if (threadIdx.x == 0)
    { int lo = 0, hi = nreads;              // last read with coff <= span0 (0 if none)
      while (hi - lo > 1)
        { int mid = (lo + hi) >> 1;
          if (coff[mid] <= span0) lo = mid; else hi = mid;
        }
      s_first = lo;
    }
  __syncthreads();
  int r = s_first;
  for (int64_t p = span0 + threadIdx.x; p < span0 + UNPACK_SPAN && p < nbytes; p += UNPACK_THREADS)
    { while (r + 1 < nreads && coff[r + 1] <= p) r++;
      const int64_t c = p - coff[r];
      const int64_t o = boff[r];
      const int64_t rlen = boff[r + 1] - o - 1;
      if (c < 0 || 4 * c >= rlen) continue;                    // a byte between two reads of the view
      const uint32_t b = bps[p];
      uint8_t *out = raw + 1 + o + 4 * c;
      const int  m = (int) ((rlen - 4 * c < 4) ? (rlen - 4 * c) : 4);
      out[0] = (uint8_t) (b >> 6);
      if (m > 1) out[1] = (uint8_t) ((b >> 4) & 3);
      if (m > 2) out[2] = (uint8_t) ((b >> 2) & 3);
      if (m > 3) out[3] = (uint8_t) (b & 3);
    }
}

// the separators: raw[0] and the byte behind every read (DB.c:1411,1428)
__global__ void k_unpack_seps(const int64_t *__restrict__ boff, int nreads, uint8_t *__restrict__ raw)
{ int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) raw[0] = 4;
  if (i < nreads) raw[boff[i + 1]] = 4;          // raw + 1 + boff[i] + rlen
}

static int preload_block_2bit(DAZZ_DB *block, const uint8_t *bps, int64_t nbytes, const int64_t *coff)
{ State &st = S();
  const int n = block->nreads;
  if (n <= 0 || nbytes <= 0) fatal("dmh_preload_block_2bit: empty block");
  for (int i = 0; i < n; i++)
    { if (block->reads[i + 1].boff - block->reads[i].boff != (int64_t) block->reads[i].rlen + 1)
        fatal("dmh_preload_block_2bit: reads[%d].boff does not follow Load_All_Reads' layout", i);
      if (coff[i] < 0 || coff[i] + (block->reads[i].rlen + 3) / 4 > nbytes || (i > 0 && coff[i] < coff[i - 1]))
        fatal("dmh_preload_block_2bit: compressed offset of read %d out of range or out of order", i);
    }
  if (st.blocks.count(block))
    { report_finish_pending();         // (a block handle re-used while a deferred kernel may read the old copy)
      drop_block(block);
    }
  BlockDev bd;
  bd.nreads  = n;
  bd.L       = block->reads[n].boff;
  bd.pinned  = true;
  bd.decoded = true;
  bd.d_raw   = (uint8_t *) dev_alloc((size_t) bd.L + 1 + 64, "block bases");
  bd.d_boff  = (int64_t *) dev_alloc(sizeof(int64_t) * (n + 1), "block read offsets");
  uint8_t *d_bps  = (uint8_t *) dev_alloc((size_t) nbytes, "compressed block");
  int64_t *d_coff = (int64_t *) dev_alloc(sizeof(int64_t) * n, "compressed read offsets");
  // a slice in page-locked memory (dmh_host_alloc) goes up on the upload stream, decode included, without
  // the host or the main stream waiting: the first kernel that reads the block waits for it (block_ready)
  const bool async = is_pinned_host(bps);
  hipStream_t us = async ? g_upstream : stream();
  if (async)
    { static int64_t   *h_off = nullptr;       // page-locked staging for both offset arrays, reused per call
      static size_t     h_cap = 0;
      static hipEvent_t h_free = nullptr;
      if (h_free == nullptr) HIP_CHECK(hipEventCreateWithFlags(&h_free, hipEventDisableTiming));
      else HIP_CHECK(hipEventSynchronize(h_free));
      if ((size_t) 2 * n + 1 > h_cap)
        { if (h_off) (void) hipHostFree(h_off);
          h_cap = (size_t) 2 * n + 1 + 2048;
          HIP_CHECK(hipHostMalloc((void **) &h_off, sizeof(int64_t) * h_cap, hipHostMallocDefault));
        }
      for (int i = 0; i <= n; i++) h_off[i] = block->reads[i].boff;
      memcpy(h_off + n + 1, coff, sizeof(int64_t) * n);
      HIP_CHECK(hipMemcpyAsync(bd.d_boff, h_off, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, us));
      HIP_CHECK(hipMemcpyAsync(d_coff, h_off + n + 1, sizeof(int64_t) * n, hipMemcpyHostToDevice, us));
      HIP_CHECK(hipEventRecord(h_free, us));
      HIP_CHECK(hipMemcpyAsync(d_bps, bps, (size_t) nbytes, hipMemcpyHostToDevice, us));
    }
  else
    { std::vector<int64_t> off(n + 1);
      for (int i = 0; i <= n; i++) off[i] = block->reads[i].boff;
      copy_h2d(bd.d_boff, off.data(), sizeof(int64_t) * (n + 1));
      copy_h2d(d_coff, coff, sizeof(int64_t) * n);
      copy_h2d(d_bps, bps, (size_t) nbytes);
    }
  { StageTimer tm("unpack_2bit", (double) nbytes + (double) bd.L, us);
    hipLaunchKernelGGL(k_unpack_seps, dim3(ceil_div(n, 256)), dim3(256), 0, us, (const int64_t *) bd.d_boff, n, bd.d_raw);
    hipLaunchKernelGGL(k_unpack_2bit, dim3((unsigned int) ceil_div(nbytes, UNPACK_SPAN)), dim3(UNPACK_THREADS), 0, us,
                       (const uint8_t *) d_bps, nbytes, (const int64_t *) d_coff, (const int64_t *) bd.d_boff, n, bd.d_raw);
    HIP_CHECK(hipGetLastError());
  }
  if (async)
    { // the alignment kernel's 2-bit words (both strands: this is how a READS block arrives) while we are at it,
      // off the main stream
      const int64_t nwords = (bd.L >> 5) + 1;
      bd.d_pkraw = (uint64_t *) dev_alloc(sizeof(uint64_t) * (nwords + 4), "packed bases");
      pack_bases(bd.d_bases(), bd.L, bd.d_pkraw, us);
      bd.d_rcraw = (uint64_t *) dev_alloc(sizeof(uint64_t) * (nwords + 4), "packed reverse complements");
      pack_revcomp(bd.d_bases(), bd.L, bd.d_boff, bd.nreads, bd.d_rcraw, us);
      HIP_CHECK(hipEventCreateWithFlags(&bd.uploaded, hipEventDisableTiming));
      HIP_CHECK(hipEventRecord(bd.uploaded, us));
      bd.d_tmp[0] = d_bps;              // freed once the main stream has been made to wait for the decode
      bd.d_tmp[1] = d_coff;
    }
  else
    { HIP_CHECK(hipStreamSynchronize(us));
      dev_free(d_bps);
      dev_free(d_coff);
    }
  st.blocks[block] = bd;
  return 0;
}

// ---- index registry -----------------------------------------------------------------------

void index_wait_ready(IndexDev &ix)
{ if (ix.pending) bg_flush();
  if (ix.built != nullptr)
    { HIP_CHECK(hipStreamWaitEvent(main_stream(), ix.built, 0));
      HIP_CHECK(hipEventDestroy(ix.built));
      ix.built = nullptr;
    }
}

static void free_index(IndexDev &ix)
{ index_wait_ready(ix);
  filter_forget_index(&ix);
  if (ix.borrowed)               // the buffers belong to a cache entry
    { ix.keys = nullptr; ix.vals = nullptr; ix.d_boff = nullptr; ix.d_chunk = nullptr; ix.d_bstart = nullptr;
      return;
    }
  dev_free(ix.keys);
  dev_free(ix.vals);
  dev_free(ix.d_boff);
  dev_free(ix.d_chunk);
  dev_free(ix.d_bstart);
  ix.keys = nullptr; ix.vals = nullptr; ix.d_boff = nullptr; ix.d_chunk = nullptr; ix.d_bstart = nullptr;
}

IndexDev *index_of(void *table, const char *who)
{ if (table == nullptr) return nullptr;
  IndexStub *stub = (IndexStub *) table;
  if (stub->magic != INDEX_MAGIC)
    fatal("%s: table argument is not a handle returned by Sort_Kmers", who);
  auto it = S().indices.find(stub->id);
  if (it == S().indices.end())
    fatal("%s: index %d was already consumed or released", who, stub->id);
  return &it->second;
}

__global__ void k_chunk_table(const int64_t *__restrict__ boff, int nreads, int64_t nchunk,
                              uint32_t *__restrict__ chunk)
{ int64_t c = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchunk) return;
  int64_t q = c << CHUNK_SHIFT;
  int lo = 0, hi = nreads;                 // largest i with boff[i] <= q
  while (lo < hi)
    { int m = (lo + hi + 1) >> 1;
      if (boff[m] > q) hi = m - 1; else lo = m;
    }
  chunk[c] = (uint32_t) lo;
}

// The positions of a finished index as (read << posbits) | rpos: the merge then builds a hit from one load per side
// instead of looking the read up through the chunk table and the offsets for every pair (repeat k-mers: hundreds of
// pairs per look-up).  Only when both fit 32 bits (read blocks and genome blocks do).
__global__ void k_pack_positions(uint32_t *__restrict__ vals, int64_t n, const int64_t *__restrict__ boff,
                                 const uint32_t *__restrict__ chunk, int posbits)
{ int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t q = vals[i];
  uint32_t r = chunk[q >> CHUNK_SHIFT];
  while (boff[r + 1] <= (int64_t) q) r += 1;
  vals[i] = (r << posbits) | (uint32_t) ((int64_t) q - boff[r]);
}

static int ceil_log2_i(int64_t x)
{ int b = 0;
  while (((int64_t) 1 << b) < x) b += 1;
  return b;
}

// Done by the merge for an index that will be merged against more than a couple of times: a retained or cached REF
// index, a READS index from its third pass on (the conversion costs about as much as two passes save).
void index_pack_positions(IndexDev &ix)
{ const int maxlen = ix.maxlen;
  const int pb = ceil_log2_i(maxlen > 1 ? (int64_t) maxlen + 1 : 2), rb = ceil_log2_i(ix.nreads > 1 ? ix.nreads : 1);
  static const bool off = getenv("DMH_NO_PACKED_POS") != nullptr;
  if (off || ix.posbits > 0 || ix.pending || maxlen <= 0 || pb + rb > 32 || ix.n <= 0) return;
  hipLaunchKernelGGL(k_pack_positions, dim3(ceil_div(ix.n, 256)), dim3(256), 0, stream(), ix.vals, ix.n,
                     (const int64_t *) ix.d_boff, (const uint32_t *) ix.d_chunk, pb);
  HIP_CHECK(hipGetLastError());
  ix.posbits = pb;
}

// Drop every code that occurs >= TooFrequent times (map.c:590-636, compsize/compress_thread).
__global__ void k_suppress_flag(const uint64_t *__restrict__ keys, int64_t n, int toofreq,
                                uint32_t *__restrict__ keep)
{ int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t s = run_start(keys, i);
  int64_t e = run_end(keys, n, i);
  keep[i] = ((e - s) < toofreq) ? 1u : 0u;
}

__global__ void k_compact(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                          const uint32_t *__restrict__ keep, const uint32_t *__restrict__ slot, int64_t n,
                          uint64_t *__restrict__ kout, uint32_t *__restrict__ vout)
{ int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (keep[i])
    { uint32_t o = slot[i];
      kout[o] = kin[i];
      vout[o] = vin[i];
    }
}

// -m masks (map.c:481-543): the k-mer ending at base offset q = vals[i] of read r survives iff no
// mask interval [b,e) of r has b <= rpos and e > rpos-(K-1), i.e. it lies inside one unmasked gap.
__global__ void k_mask_flag(const uint32_t *__restrict__ vals, int64_t n, const int64_t *__restrict__ boff,
                            int nreads, int K, const int64_t *__restrict__ anno, const int *__restrict__ data,
                            uint32_t *__restrict__ keep)
{ int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t q = vals[i];
  int lo = 0, hi = nreads;                 // largest r with boff[r] <= q
  while (lo < hi)
    { int m = (lo + hi + 1) >> 1;
      if (boff[m] > q) hi = m - 1; else lo = m;
    }
  const int     rpos  = (int) (q - boff[lo]);
  const int     start = rpos - (K - 1);
  const int64_t a0 = anno[lo] >> 1, a1 = anno[lo + 1] >> 1;      // interval numbers
  int64_t l = a0, h = a1;                  // first interval with end > start
  while (l < h)
    { int64_t m = (l + h) >> 1;
      if (data[2 * m + 1] > start) h = m; else l = m + 1;
    }
  keep[i] = (l < a1 && data[2 * l] <= rpos) ? 0u : 1u;
}

// ---- REF index cache ----------------------------------------------------------------------------
// Identity of a block's content for the cache: EVERY base and every read boundary go into two independent 64-bit
// hashes (four host threads, a few GB/s each: ~10 ms for a 375 Mbp block, against the upload and sort it saves),
// plus the parameters the index depends on.
static void hash_range(const uint8_t *p, int64_t n, uint64_t out[2])
{ uint64_t h1 = 1469598103934665603ull, h2 = 0x9e3779b97f4a7c15ull, h3 = 0xc2b2ae3d27d4eb4full, h4 = 0x165667b19e3779f9ull;
  int64_t i = 0;
  for (; i + 32 <= n; i += 32)           // four lanes, so that the multiplies overlap
    { uint64_t a, b, c, d;
      memcpy(&a, p + i, 8); memcpy(&b, p + i + 8, 8); memcpy(&c, p + i + 16, 8); memcpy(&d, p + i + 24, 8);
      h1 = (h1 ^ a) * 0x100000001b3ull; h2 = (h2 ^ b) * 0xff51afd7ed558ccdull;
      h3 = (h3 ^ c) * 0xc4ceb9fe1a85ec53ull; h4 = (h4 ^ d) * 0x9fb21c651e98df25ull;
      h1 ^= h1 >> 29; h2 ^= h2 >> 31; h3 ^= h3 >> 30; h4 ^= h4 >> 32;
    }
  for (; i < n; i++) h1 = (h1 ^ p[i]) * 0x100000001b3ull;
  out[0] = h1 ^ (h3 * 0x2545f4914f6cdd1dull);
  out[1] = h2 ^ (h4 * 0xd6e8feb86659fd93ull);
}

static void content_key(const DAZZ_DB *b, const Params &pm, uint64_t key[4])
{ const uint8_t *p = (const uint8_t *) b->bases;
  const int64_t  L = b->reads[b->nreads].boff;
  constexpr int NT = 4;
  uint64_t part[NT][2];
  std::vector<std::thread> th;
  for (int t = 0; t < NT; t++)
    th.emplace_back([&, t] { int64_t lo = L * t / NT, hi = L * (t + 1) / NT; hash_range(p + lo, hi - lo, part[t]); });
  for (auto &x : th) x.join();
  uint64_t h1 = 0x9e3779b97f4a7c15ull, h2 = 1469598103934665603ull;
  auto mix = [&](uint64_t v) { h1 = (h1 ^ v) * 1099511628211ull; h1 ^= h1 >> 32; h2 = (h2 + v) * 0xff51afd7ed558ccdull; h2 ^= h2 >> 33; };
  for (int t = 0; t < NT; t++) { mix(part[t][0]); mix(part[t][1]); }
  for (int i = 0; i <= b->nreads; i++) mix((uint64_t) b->reads[i].boff);
  key[0] = h1;
  key[1] = h2;
  key[2] = ((uint64_t) L << 24) ^ (uint64_t) b->nreads;
  key[3] = ((uint64_t) pm.Kmer << 40) ^ ((uint64_t) (uint32_t) pm.TooFrequent << 4) ^ 1ull;
}

static size_t index_bytes(const IndexDev &ix)
{ // (a bucket table attached later by a borrower is not counted: at most 1 GB per entry)
  return (size_t) ix.n * 12 + (size_t) (ix.nreads + 1) * 8 + (size_t) ix.nchunk * 4 + ix.bstart_bytes;
}

static IndexDev *cache_find(const uint64_t key[4])
{ State &st = S();
  for (auto it = st.ref_cached.begin(); it != st.ref_cached.end(); ++it)
    if (memcmp(it->ckey, key, sizeof(it->ckey)) == 0)
      { st.ref_cached.splice(st.ref_cached.begin(), st.ref_cached, it);     // most recently used first
        return &st.ref_cached.front();
      }
  return nullptr;
}

// Match_Filter is done with a REF index: keep it (cache on) or free it.  A borrowed record hands a bucket
// table it built back to its cache entry.
void release_ref_index(IndexDev &ix)
{ State &st = S();
  if (ix.borrowed)
    { IndexDev *c = cache_find(ix.ckey);
      if (c != nullptr)
        { c->borrows -= 1;
          if (c->d_bstart == nullptr)
            { c->d_bstart = ix.d_bstart; c->bshift = ix.bshift; c->bstart_bytes = ix.bstart_bytes; c->probe_keys = ix.probe_keys; c->low_off = ix.low_off; }
          else if (c->d_bstart != ix.d_bstart)
            dev_free(ix.d_bstart);         // two borrowers built a table each
        }
      free_index(ix);
      return;
    }
  if (!st.ref_cache || !ix.cacheable || ix.pending || cache_find(ix.ckey) != nullptr)
    { free_index(ix);
      return;
    }
  size_t bytes = index_bytes(ix);
  while (st.ref_cache_bytes + bytes > st.ref_cache_cap)
    { auto victim = st.ref_cached.end();
      for (auto it = st.ref_cached.end(); it != st.ref_cached.begin();)
        { --it;
          if (it->borrows == 0) { victim = it; break; }
        }
      if (victim == st.ref_cached.end()) break;
      st.ref_cache_bytes -= index_bytes(*victim);
      free_index(*victim);
      st.ref_cached.erase(victim);
    }
  if (st.ref_cache_bytes + bytes > st.ref_cache_cap)
    { free_index(ix);                        // no room (everything else is in use)
      return;
    }
  index_pack_positions(ix);                  // it will be merged against again
  st.ref_cached.push_front(ix);
  st.ref_cache_bytes += bytes;
  filter_forget_index(&ix);                  // (the handle's record goes away: nothing of the filter may keep its address)
  ix.keys = nullptr; ix.vals = nullptr; ix.d_boff = nullptr; ix.d_chunk = nullptr; ix.d_bstart = nullptr;
}

static void ref_cache_clear()
{ State &st = S();
  for (auto &ix : st.ref_cached) free_index(ix);
  st.ref_cached.clear();
  st.ref_cache_bytes = 0;
}

// ---- deferred index build: stages queued by bg_pump ------------------------------------------
struct SortJob
{ int       id;                 // key of the index in State::indices
  uint64_t *keys, *k0, *k1;     // buffers allocated for the index / ping-pong state
  uint32_t *vals, *v0, *v1;
  int64_t   kmers, kmers0;
  int       Kshift, npass;
  int       stage;              // index into the stage list (see bg_step)
  const DAZZ_DB *block;         // emit stage: the block whose device copy is read
  int       Kmer;
  int       posbits;            // > 0: positions emitted as (read << posbits) | rpos
  hipStream_t q = nullptr;      // the stream all its stages run on (nullptr: the main stream)
};
static std::deque<SortJob> g_bg;
// The deferred builds share ONE scratch (State::sort_aux: histogram, scan sums, ping-pong partners) but not always one
// stream: a job runs on the main stream or on the tail stream (bg_pump_tail).  An event behind the last stage of the
// job that finished last orders whoever touches the scratch next on ANOTHER stream -- the next job's first stage, or
// the main-stream scope under which Sort_Kmers grows (frees) the partner buffers.
static hipEvent_t  g_bg_done_ev = nullptr;
static hipStream_t g_bg_done_q  = nullptr;
static void bg_order_behind_last_job(hipStream_t q)
{ if (g_bg_done_q != nullptr && g_bg_done_q != q) HIP_CHECK(hipStreamWaitEvent(q, g_bg_done_ev, 0)); }

// the sorted data may sit in the scratch partner: the index then takes over the partner's buffers and
// the sort scratch keeps the ones allocated for the index (no device copy)
static void adopt_sorted(IndexDev &ix, SortScratch &scr, uint64_t *keys, uint32_t *vals, uint64_t *k0, uint32_t *v0,
                         int64_t kmers0, int64_t kmers)
{ if (k0 != keys)
    { scr.keys_alt.p   = keys;
      scr.keys_alt.cap = (size_t) kmers0;
      scr.vals_alt.p   = vals;
      scr.vals_alt.cap = (size_t) kmers0;
      keys = k0;
      vals = v0;
    }
  ix.keys = keys;
  ix.vals = vals;
  ix.n    = kmers;
}

// Stage list of a deferred build: EMIT_PARTS pieces of the k-mer emit, then per radix pass HIST_PARTS
// histogram pieces (the scan rides on the last) and SCAT_PARTS scatter pieces -- small enough
// (0.25-0.5 ms on a 0.26 Gbp block) to slot into the bubbles between the host's round trips.
// Measured on the bench workload, final kernels (ms per block / roofline fraction of the scatter launches):
// 1/1/1 36.8 / .73, 1/1/2 34.7 / .73, 1/2/4 35.0 / .70, 1/2/8 36.9 / .69.
static int EMIT_PARTS = 1, HIST_PARTS = 1, SCAT_PARTS = 2, PASS_PARTS = 3;
static void bg_parts_init()
{ static bool done = false;
  if (done) return;
  done = true;
  const char *e = getenv("DMH_BG_PARTS");          // "emit,hist,scatter" pieces, for experiments
  if (e != nullptr)
    { int a = 0, b = 0, c = 0;
      if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3 && a >= 1 && b >= 1 && c >= 1 && a <= 8 && b <= 8 && c <= 16)
        { EMIT_PARTS = a; HIST_PARTS = b; SCAT_PARTS = c; }
    }
  PASS_PARTS = HIST_PARTS + SCAT_PARTS;
}

static bool g_bg_whole = false;

static void bg_step()
{ bg_parts_init();
  SortJob &J = g_bg.front();
  State   &st = S();
  StreamScope on_its_stream(J.q != nullptr ? J.q : main_stream());
  if (J.stage < EMIT_PARTS)
    { auto b = st.blocks.find(J.block);
      if (b == st.blocks.end()) fatal("deferred index build lost its block");
      BlockDev &bd = b->second;
      auto it = st.indices.find(J.id);
      if (it == st.indices.end()) fatal("deferred index build lost its index");
      IndexDev &ix = it->second;
      if (J.stage == 0)
        { bg_order_behind_last_job(stream());      // (the previous job's last passes may still be reading the scratch on another stream)
          block_ready(bd);
          HIP_CHECK(hipMemcpyAsync(ix.d_boff, bd.d_boff, sizeof(int64_t) * (bd.nreads + 1), hipMemcpyDeviceToDevice, stream()));
          hipLaunchKernelGGL(k_chunk_table, dim3(ceil_div(ix.nchunk, 256)), dim3(256), 0, stream(),
                             (const int64_t *) ix.d_boff, bd.nreads, ix.nchunk, ix.d_chunk);
          HIP_CHECK(hipGetLastError());
        }
      kmer_emit(bd.d_bases(), bd.L, bd.d_boff, bd.nreads, J.Kmer, J.keys, J.vals, J.kmers, J.stage, EMIT_PARTS, J.posbits);
      J.stage += 1;
      return;
    }
  const int q = J.stage - EMIT_PARTS;
  const int p = q / PASS_PARTS, r = q % PASS_PARTS;
  const int w = J.Kshift - 8 * p;
  if (r < HIST_PARTS)
    radix_pass_hist(J.k0, J.kmers, 8 * p, w < 8 ? w : 8, st.sort_aux, r, HIST_PARTS);
  else
    { int  sp = r - HIST_PARTS, np = SCAT_PARTS;
      bool last = (r == PASS_PARTS - 1);
      if (g_bg_whole && sp == 0)              // a flush has no gaps to fill: the whole pass in one launch
        { np = 1; last = true; J.stage += SCAT_PARTS - 1; }
      radix_pass_scatter(J.k0, J.v0, J.k1, J.v1, J.kmers, 8 * p, w < 8 ? w : 8, st.sort_aux, sp, np);
      if (last)
        { std::swap(J.k0, J.k1);
          std::swap(J.v0, J.v1);
        }
    }
  J.stage += 1;
  if (J.stage == EMIT_PARTS + PASS_PARTS * J.npass)
    { auto it = st.indices.find(J.id);
      if (it == st.indices.end()) fatal("deferred index build lost its index");
      adopt_sorted(it->second, st.sort_aux, J.keys, J.vals, J.k0, J.v0, J.kmers0, J.kmers);
      it->second.pending = false;
      if (J.q != nullptr && J.q != main_stream())
        { if (it->second.built == nullptr) HIP_CHECK(hipEventCreateWithFlags(&it->second.built, hipEventDisableTiming));
          HIP_CHECK(hipEventRecord(it->second.built, J.q));
        }
      if (g_bg_done_ev == nullptr) HIP_CHECK(hipEventCreateWithFlags(&g_bg_done_ev, hipEventDisableTiming));
      HIP_CHECK(hipEventRecord(g_bg_done_ev, stream()));
      g_bg_done_q = stream();
      g_bg.pop_front();
    }
}

void bg_pump(int nstages)
{ StreamScope on_main(main_stream());        // (also when called from inside chain_finish: the index build lives on the main stream)
  // With the alignment kernel deferred (dmh_async_report(2)) nothing is pumped: Reporter queues the whole build
  // at the block boundary, where the main stream is idle for some 35 ms (the last batches finish one after the
  // other, then Reporter's host work); between the passes it has no gaps to fill (DMH_PUMP=1: as before).
  // Default workload 214.7 -> 211 ms per read block, one-REF-block chr1 674 -> 641 ms per step, ecoli unchanged.
  static const bool pump_anyway = getenv("DMH_PUMP") != nullptr;
  if (report_deferred_mode() && !pump_anyway) return;
  if (g_bg.empty() || report_in_flight()) return;
  for (int i = 0; i < nstages && !g_bg.empty(); i++) bg_step();
}

// One stage per call of the deferred build that is waiting, on the TAIL stream: the filter calls this at the end of a
// dense pass, so that the index of the NEXT read block is sorted -- HBM-bound -- next to the LDS- and latency-bound
// second halves of this block's passes instead of alone at the block boundary (DMH_INDEX_ON_TAIL=0: off).
void bg_pump_tail(int nstages)
{ static const bool on = getenv("DMH_INDEX_ON_TAIL") == nullptr || atoi(getenv("DMH_INDEX_ON_TAIL")) == 1;
  if (!on || g_bg.empty() || tail_stream() == main_stream()) return;
  SortJob &J = g_bg.front();
  if (J.stage == 0 && J.q == nullptr) J.q = tail_stream();
  if (J.q == nullptr) return;                   // (started on the main stream: it stays there)
  for (int i = 0; i < nstages && !g_bg.empty(); i++) bg_step();
}

// Memory ran out: the ping-pong partners of the k-mer sorts (12 GB behind a GRCh38-sized reference) are idle unless a
// deferred index build is under way; they come back with the next sort.
bool sort_scratch_release()
{ if (!g_bg.empty()) return false;
  State &st = S();
  bool got = false;
  for (SortScratch *sc : {&st.sort, &st.sort_aux})
    { if (sc->keys_alt.p != nullptr || sc->vals_alt.p != nullptr) got = true;
      sc->keys_alt.release();
      sc->vals_alt.release();
    }
  return got;
}

bool bg_on_tail() { return !g_bg.empty() && g_bg.front().q != nullptr && g_bg.front().q != main_stream(); }

void bg_flush()
{ StreamScope on_main(main_stream());
  g_bg_whole = true;
  while (!g_bg.empty()) bg_step();
  g_bg_whole = false;
}

static void *sort_kmers_impl(DAZZ_DB *block, int *len, bool deferred)
{ State &st = S();
  Params &pm = st.params;
  if (!pm.set) fatal("Sort_Kmers called before Set_Filter_Params");
  HostTimer ht("Sort_Kmers");
  int     nreads = block->nreads;
  int64_t L      = block->reads[nreads].boff;
  int64_t kmers  = L - (int64_t) pm.Kmer * nreads;        // map.c:676
  *len = 0;
  if (kmers <= 0) return nullptr;
  if (kmers > INT32_MAX) fatal("Sort_Kmers: block has %lld k-mers, limit is 2^31-1 (map.c:663)", (long long) kmers);
  int maxrl = 0;
  for (int i = 0; i < nreads; i++)
    { if (block->reads[i].rlen < pm.Kmer - 1)
        fatal("Sort_Kmers: read %d is shorter than k-1 (%d < %d)", i, block->reads[i].rlen, pm.Kmer - 1);
      maxrl = std::max(maxrl, block->reads[i].rlen);
    }

  const DAZZ_TRACK *mask = (const DAZZ_TRACK *) block->tracks;
  uint64_t ckey[4] = {0, 0, 0, 0};
  if (st.ref_cache && mask == nullptr && !deferred && block->bases != nullptr)
    { content_key(block, pm, ckey);
      IndexDev *c = cache_find(ckey);
      if (c != nullptr)
        { // the index of this very content is still in HBM: no upload, no sort
          IndexDev ix = *c;
          c->borrows += 1;
          ix.block    = block;
          ix.borrowed = true;
          ix.borrows  = 0;
          IndexStub *stub = (IndexStub *) malloc(sizeof(IndexStub));
          if (stub == nullptr) fatal("out of memory allocating index handle");
          stub->magic = INDEX_MAGIC;
          stub->id    = st.next_index_id++;
          stub->len   = (int32_t) ix.n;
          st.indices[stub->id] = ix;
          *len = (int) ix.n;
          if (VERBOSE)
            { printf("\n   Kmer count = %lld\n   Index found in HBM (%.2fGb)\n", (long long) kmers, (12. * ix.n) / 1073741824.);
              fflush(stdout);
            }
          return stub;
        }
    }
  // a build can only be deferred when the length of the index is known up front
  const bool defer = deferred && mask == nullptr && pm.TooFrequent == INT32_MAX;
  if (defer)
    { bg_flush();                                // one deferred build at a time (it owns sort_aux)
      // the flush only QUEUES what was left of the previous job, on that job's stream: the partner buffers grown (freed)
      // below under the main stream's scope must not change hands while its last scatter pass reads them on the tail stream
      bg_order_behind_last_job(main_stream());
    }
  SortScratch &scr = defer ? st.sort_aux : st.sort;

  if (VERBOSE)
    { printf("\n   Kmer count = %lld\n", (long long) kmers);
      printf("   Using %.2fGb of HBM\n", (24. * kmers) / 1073741824.);
      fflush(stdout);
    }

  HostTimer *hu = new HostTimer("sk_upload");
  BlockDev  &bd   = upload_block(block, false);
  delete hu;
  HostTimer *ha = new HostTimer("sk_alloc");

  IndexDev ix;
  ix.block  = block;
  ix.nreads = nreads;
  ix.L      = L;
  memcpy(ix.ckey, ckey, sizeof(ckey));
  ix.cacheable = (ckey[3] != 0);
  ix.maxlen = maxrl;
  ix.h_boff.resize(nreads + 1);
  for (int i = 0; i <= nreads; i++) ix.h_boff[i] = block->reads[i].boff;

  // same slack as the sort scratch: the two buffer pairs may swap roles below
  const int64_t kmers0 = kmers + kmers / 8 + 256;
  uint64_t *keys = (uint64_t *) dev_alloc(sizeof(uint64_t) * kmers0, "k-mer codes");
  uint32_t *vals = (uint32_t *) dev_alloc(sizeof(uint32_t) * kmers0, "k-mer positions");
  scr.keys_alt.ensure(kmers, "k-mer sort buffer");
  scr.vals_alt.ensure(kmers, "k-mer sort buffer");

  delete ha;
  HostTimer *he = new HostTimer("sk_enqueue");
  // positions straight as (read, rpos) words when they fit 32 bits and no mask wants block offsets (k_mask_flag)
  { const int pb = ceil_log2_i(maxrl > 1 ? (int64_t) maxrl + 1 : 2), rb = ceil_log2_i(nreads > 1 ? nreads : 1);
    static const bool off = getenv("DMH_NO_PACKED_POS") != nullptr;
    if (!off && mask == nullptr && maxrl > 0 && pb + rb <= 32) ix.posbits = pb;
  }
  if (!defer)
    { block_ready(bd);
      kmer_emit(bd.d_bases(), L, bd.d_boff, nreads, pm.Kmer, keys, vals, kmers, 0, 1, ix.posbits);
    }
  if (getenv("DMH_DEBUG_SYNC") != nullptr)
    { HostTimer hs("sk_emit_sync");
      HIP_CHECK(hipStreamSynchronize(stream()));
    }

  uint64_t *k0 = keys, *k1 = scr.keys_alt.p;
  uint32_t *v0 = vals, *v1 = scr.vals_alt.p;

  if (mask != nullptr)
    { // masked k-mers are dropped before the sort: the stable sort then leaves exactly the list the
      // reference obtains by sorting them to the end as all-ones sentinels and cutting them off
      // (map.c:536-542,722-724; identical unless a T^k k-mer ties with the sentinels under -T > 1)
      StageTimer tm("kmer_mask", 32.0 * kmers);
      const int64_t *anno = (const int64_t *) mask->anno;
      const int64_t  nint = anno[nreads];
      if (mask->data == nullptr && nint > 0) fatal("Sort_Kmers: mask track '%s' has no data loaded", mask->name);
      if (nint & 1) fatal("Sort_Kmers: mask track '%s' holds an odd number of interval ends", mask->name);
      DevBuf<int64_t>  d_anno;
      DevBuf<int>      d_data;
      DevBuf<uint32_t> keep, slot;
      d_anno.ensure(nreads + 1, "mask offsets");
      d_data.ensure(nint + 2, "mask intervals");
      keep.ensure(kmers, "mask flags");
      slot.ensure(kmers + 1, "mask slots");
      copy_h2d(d_anno.p, anno, sizeof(int64_t) * (nreads + 1));
      if (nint > 0) copy_h2d(d_data.p, mask->data, sizeof(int) * nint);
      int nb = ceil_div(kmers, 256);
      hipLaunchKernelGGL(k_mask_flag, dim3(nb), dim3(256), 0, stream(), (const uint32_t *) v0, kmers,
                         (const int64_t *) bd.d_boff, nreads, pm.Kmer, (const int64_t *) d_anno.p,
                         (const int *) d_data.p, keep.p);
      HIP_CHECK(hipMemcpyAsync(slot.p, keep.p, sizeof(uint32_t) * kmers, hipMemcpyDeviceToDevice, stream()));
      exclusive_scan_u32(slot.p, kmers, scr.bsum, slot.p + kmers);
      hipLaunchKernelGGL(k_compact, dim3(nb), dim3(256), 0, stream(), (const uint64_t *) k0,
                         (const uint32_t *) v0, (const uint32_t *) keep.p, (const uint32_t *) slot.p, kmers, k1, v1);
      uint32_t kept = 0;
      copy_d2h(&kept, slot.p + kmers, sizeof(uint32_t));
      HIP_CHECK(hipStreamSynchronize(stream()));
      std::swap(k0, k1);
      std::swap(v0, v1);
      kmers = kept;
      d_anno.release(); d_data.release(); keep.release(); slot.release();
    }

  if (kmers > 0 && !defer)
    radix_sort_pairs(&k0, &v0, &k1, &v1, kmers, 0, pm.Kshift, scr, "kmer_sort");

  if (pm.TooFrequent < INT32_MAX && kmers > 0)
    { StageTimer tm("kmer_suppress", 32.0 * kmers);
      DevBuf<uint32_t> keep, slot;
      keep.ensure(kmers, "suppress flags");
      slot.ensure(kmers + 1, "suppress slots");
      int nb = ceil_div(kmers, 256);
      hipLaunchKernelGGL(k_suppress_flag, dim3(nb), dim3(256), 0, stream(), (const uint64_t *) k0, kmers,
                         pm.TooFrequent, keep.p);
      HIP_CHECK(hipMemcpyAsync(slot.p, keep.p, sizeof(uint32_t) * kmers, hipMemcpyDeviceToDevice, stream()));
      exclusive_scan_u32(slot.p, kmers, scr.bsum, slot.p + kmers);
      hipLaunchKernelGGL(k_compact, dim3(nb), dim3(256), 0, stream(), (const uint64_t *) k0,
                         (const uint32_t *) v0, (const uint32_t *) keep.p, (const uint32_t *) slot.p, kmers, k1, v1);
      uint32_t kept = 0;
      copy_d2h(&kept, slot.p + kmers, sizeof(uint32_t));
      HIP_CHECK(hipStreamSynchronize(stream()));
      std::swap(k0, k1);
      std::swap(v0, v1);
      kmers = kept;
      keep.release();
      slot.release();
    }
  if (VERBOSE && (pm.TooFrequent < INT32_MAX || mask != nullptr))        // map.c:793-797
    printf("   Revised kmer count = %lld\n", (long long) kmers);

  delete he;
  HostTimer hf("sk_finish");
  if (defer)
    { ix.keys = nullptr;       // set when the last stage has been queued
      ix.vals = nullptr;
      ix.n    = kmers;
      ix.pending = true;
    }
  else
    adopt_sorted(ix, scr, keys, vals, k0, v0, kmers0, kmers);

  ix.d_boff = (int64_t *) dev_alloc(sizeof(int64_t) * (nreads + 1), "index read offsets");
  ix.nchunk = (L >> CHUNK_SHIFT) + 1;
  ix.d_chunk = (uint32_t *) dev_alloc(sizeof(uint32_t) * ix.nchunk, "index chunk table");
  if (!defer)            // (a deferred build fills both in its first stage, once the block has arrived)
    { HIP_CHECK(hipMemcpyAsync(ix.d_boff, bd.d_boff, sizeof(int64_t) * (nreads + 1), hipMemcpyDeviceToDevice, stream()));
      hipLaunchKernelGGL(k_chunk_table, dim3(ceil_div(ix.nchunk, 256)), dim3(256), 0, stream(),
                         (const int64_t *) ix.d_boff, nreads, ix.nchunk, ix.d_chunk);
      HIP_CHECK(hipGetLastError());
    }
  // no synchronisation here: every later use of the index is ordered on the same stream, and a
  // driver may queue more work (e.g. the next block's Reporter) behind the sort right away

  if (VERBOSE)
    { printf("   Index occupies %.2fGb\n", (12. * kmers) / 1073741824.);
      fflush(stdout);
    }

  if (kmers <= 0)
    { ix.pending = false;
      if (ix.keys == nullptr) { ix.keys = keys; ix.vals = vals; }
      free_index(ix);
      return nullptr;
    }

  IndexStub *stub = (IndexStub *) malloc(sizeof(IndexStub));
  if (stub == nullptr) fatal("out of memory allocating index handle");
  stub->magic = INDEX_MAGIC;
  stub->id    = st.next_index_id++;
  stub->len   = (int32_t) kmers;
  st.indices[stub->id] = ix;
  if (defer)
    { SortJob J;
      J.id = stub->id;
      J.keys = keys; J.vals = vals;
      J.k0 = k0; J.k1 = k1; J.v0 = v0; J.v1 = v1;
      J.kmers = kmers; J.kmers0 = kmers0;
      J.Kshift = pm.Kshift;
      J.npass = (pm.Kshift + 7) / 8;
      J.stage = 0;
      J.block = block;
      J.Kmer  = pm.Kmer;
      J.posbits = ix.posbits;
      g_bg.push_back(J);
      if (kmers <= 1) bg_flush();
    }
  *len = (int) kmers;
  return stub;
}


}  // namespace dmh

using namespace dmh;

// =========================================================================================
//  C-ABI
// =========================================================================================

extern "C" {

int dmh_device_count(void)
{ int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *dmh_last_error(void) { return g_err; }

void dmh_sync(void)
{ report_finish_pending();
  sync_all_streams();
}

void dmh_timing_enable(int on) { g_timing = (on != 0); g_timing_roofline_only = (on == 2); }

}  // extern "C" (interrupted)
static void dmh::timing_dump_at_exit()
{ const int n = dmh_timing_count();
  std::vector<std::pair<double, int>> order;
  for (int i = 0; i < n; i++) order.push_back({-g_slots[i].ms, i});
  std::sort(order.begin(), order.end());
  fprintf(stderr, "dmh: timers at exit (ms, calls)\n");
  for (auto &o : order)
    fprintf(stderr, "dmh:   %-28s %12.2f %9lld\n", g_slots[o.second].name.c_str(), g_slots[o.second].ms,
            (long long) g_slots[o.second].launches);
}
extern "C" {

void dmh_timing_reset(void)
{ drain_timers();
  std::lock_guard<std::mutex> lk(g_tmu);
  g_slots.clear();
}

int dmh_timing_count(void)
{ drain_timers();
  return (int) g_slots.size();
}

int dmh_timing_get(int idx, char *name, int namelen, double *ms, int64_t *launches, double *bytes)
{ drain_timers();
  if (idx < 0 || idx >= (int) g_slots.size()) return -1;
  snprintf(name, namelen, "%s", g_slots[idx].name.c_str());
  *ms = g_slots[idx].ms;
  *launches = g_slots[idx].launches;
  *bytes = g_slots[idx].bytes;
  return 0;
}

void dmh_reset(void)
{ State &st = S();
  report_finish_pending();
  if (g_init) sync_all_streams();
  for (auto &kv : st.indices) free_index(kv.second);
  st.indices.clear();
  ref_cache_clear();
  for (auto &kv : st.blocks) free_blockdev(kv.second);
  st.blocks.clear();
  st.reads.clear();
  st.cur_reads = nullptr;
  filter_discard();
  for (CandStore &cs : st.cand_slot)
    { for (int i = 0; i < 2; i++) { cs.pool[i].release(); cs.seg_off[i].release(); cs.seg_len[i].release(); }
      cs.jumps.release(); cs.cover.release(); cs.cov_off.release(); cs.ctl.release();
      cs.block = nullptr; cs.nreads = 0; cs.total_len = 0; cs.jtop = 0; cs.mirror_valid = false;
    }
  st.cand_cur = 0;
  st.tfilt = 0;
  st.last_seeds.clear();
  st.sort.hist.release();
  st.sort.bsum.release();
  st.sort.keys_alt.release();
  st.sort.vals_alt.release();
  st.sort_aux.hist.release();
  st.sort_aux.bsum.release();
  st.sort_aux.keys_alt.release();
  st.sort_aux.vals_alt.release();
  scan_state_reset();
  pool_release();
}

int dmh_preload_block(DAZZ_DB *block)
{ upload_block(block, true);
  return 0;
}

int dmh_device_memory(unsigned long long *free_bytes, unsigned long long *total_bytes)
{ S();                                   // (selects the device)
  size_t f = 0, t = 0;
  if (hipMemGetInfo(&f, &t) != hipSuccess) { (void) hipGetLastError(); return -1; }
  if (free_bytes)  *free_bytes  = f;
  if (total_bytes) *total_bytes = t;
  return 0;
}

int dmh_prepare_block(DAZZ_DB *block, int reads_side)
{ auto it = S().blocks.find(block);
  if (it == S().blocks.end()) return -1;
  ensure_packed(it->second, reads_side != 0);
  return 0;
}

int dmh_preload_block_2bit(DAZZ_DB *block, const unsigned char *bps, int64_t nbytes, const int64_t *coff)
{ return preload_block_2bit(block, bps, nbytes, coff); }

int dmh_download_block(DAZZ_DB *block, unsigned char *out)
{ State &st = S();
  auto it = st.blocks.find(block);
  if (it == st.blocks.end()) return -1;
  block_ready(it->second);
  copy_d2h(out, it->second.d_raw, (size_t) it->second.L + 1);
  HIP_CHECK(hipStreamSynchronize(stream()));
  return 0;
}

void dmh_release_block(DAZZ_DB *block)
{ if (S().blocks.count(block) == 0) return;   // (a deferred Reporter holds the copy it reads by itself)
  if (report_adopt_block(block)) return;     // its kernel still reads this resident copy: freed when it is collected
  drop_block(block);
}

void dmh_retain_ref_index(int on) { S().retain_ref = (on != 0); }

void dmh_ref_cache(int on, double max_gb)
{ State &st = S();
  st.ref_cache = (on != 0);
  if (max_gb > 0) st.ref_cache_cap = (size_t) (max_gb * 1073741824.);
  if (!st.ref_cache) ref_cache_clear();
}

void dmh_free_index(void *table)
{ if (table == nullptr) return;
  IndexStub *stub = (IndexStub *) table;
  if (stub->magic != INDEX_MAGIC) return;
  auto it = S().indices.find(stub->id);
  if (it != S().indices.end())
    { if (it->second.borrowed) release_ref_index(it->second);     // gives the cache entry back
      else free_index(it->second);
      S().indices.erase(it);
    }
  stub->magic = 0;
  free(stub);
}

void dmh_capture_seeds(int on) { S().capture_seeds = (on != 0); }

int dmh_test_option(const char *key, int value)
{ State &st = S();
  if (strcmp(key, "align_level") == 0)      st.opt_align_level = value;
  else if (strcmp(key, "chain_g") == 0)     st.opt_chain_g = (value == 16 || value == 32 || value == 64) ? value : 0;
  else if (strcmp(key, "pool_div") == 0)    st.opt_pool_div = value > 1 ? value : 1;
  else if (strcmp(key, "chain_pile") == 0)  st.opt_chain_pile = value;
  else if (strcmp(key, "pile_cap") == 0)    st.opt_pile_cap = value;
  else if (strcmp(key, "merge_dense") == 0) st.opt_merge_dense = value;
  else if (strcmp(key, "cluster") == 0)     st.opt_cluster = value;
  else if (strcmp(key, "chain_small") == 0) st.opt_chain_small = value;
  else if (strcmp(key, "chain_lane") == 0)  st.opt_chain_lane = value;
  else if (strcmp(key, "batch") == 0)       st.opt_batch = value;
  else if (strcmp(key, "expand") == 0)      st.opt_expand = value;
  else if (strcmp(key, "in_flight") == 0)   st.opt_in_flight = value;
  else if (strcmp(key, "hit_limit") == 0)   st.opt_hit_limit = value;
  else if (strcmp(key, "tails") == 0)       st.opt_tails = value;
  else if (strcmp(key, "defer_close") == 0) st.opt_defer_close = value;
  else if (strcmp(key, "front_early") == 0) st.opt_front_early = value;
  else if (strcmp(key, "seed_word") == 0)   st.opt_seed_word = value;
  else if (strcmp(key, "fin_stream") == 0)  st.opt_fin_stream = value;
  else if (strcmp(key, "alloc_fail") == 0)  g_alloc_fail = value > 0 ? value : 0;
  else if (strcmp(key, "align_g") == 0)     st.opt_align_g = value;
  else if (strcmp(key, "cluster_sort") == 0) st.opt_cluster_sort = value;
  else if (strcmp(key, "oom_finish") == 0)  st.opt_oom_finish = value > 0 ? value : 0;
  else if (strcmp(key, "chain_simple") == 0) st.opt_chain_simple = value;
  else if (strcmp(key, "read_sort") == 0)   st.opt_read_sort = value;
  else if (strcmp(key, "fuse") == 0)        st.opt_fuse = value;
  else if (strcmp(key, "chain_classes") == 0) st.opt_chain_classes = value;
  else if (strcmp(key, "pebble_spill") == 0) st.opt_pebble_spill = value;
  else if (strcmp(key, "front_ahead") == 0) st.opt_front_ahead = value;
  else if (strcmp(key, "match_compact") == 0) st.opt_match_compact = value;
  else if (strcmp(key, "sort_ballot") == 0) st.opt_sort_ballot = value;
  else return -1;
  return 0;
}

void *dmh_host_alloc(size_t bytes)
{ void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess)
    { (void) hipGetLastError();
      return nullptr;
    }
  return p;
}

void dmh_host_free(void *p)
{ if (p != nullptr) (void) hipHostFree(p); }

// ---- map.c:124-150 -------------------------------------------------------------------------
int Set_Filter_Params(int kmer, int suppress, int nthreads)
{ if (kmer <= 1) return 1;
  { static bool env_read = false;
    if (!env_read)
      { env_read = true;
        const char *e = getenv("DMH_REF_CACHE");
        if (e != nullptr && atoi(e) != 0) S().ref_cache = true;
      }
  }
  Params &p = S().params;
  p.Kmer = kmer;
  p.Suppress = suppress;
  p.Kshift = 2 * kmer;
  p.Kmask = (kmer >= 32) ? ~0ull : ((1ull << p.Kshift) - 1ull);
  p.TooFrequent = (suppress == 0) ? INT32_MAX : suppress;
  p.NTHREADS = 1;
  p.NSHIFT = 0;
  while (2 * p.NTHREADS <= nthreads)
    { p.NTHREADS *= 2;
      p.NSHIFT += 1;
    }
  p.set = true;
  return 0;
}

// ---- map.c:655-822 -------------------------------------------------------------------------
void *Sort_Kmers(DAZZ_DB *block, int *len)
{ return sort_kmers_impl(block, len, false); }

// Sort_Kmers whose radix passes are queued later, in the gaps of the work on the current READS block
// (bg_pump); whatever is left is queued when Match_Filter first uses the index.  Same result.
void *dmh_sort_kmers_async(DAZZ_DB *block, int *len)
{ return sort_kmers_impl(block, len, true); }

// KmerPos layout of the reference (map.c:78-82, little endian)
struct KmerPosRec { uint64_t code; int32_t rpos; int32_t read; };

int dmh_index_download(void *table, void *out)
{ IndexDev *ix = index_of(table, "dmh_index_download");
  if (ix == nullptr) return 0;
  index_wait_ready(*ix);
  std::vector<uint64_t> k(ix->n);
  std::vector<uint32_t> v(ix->n);
  copy_d2h(k.data(), ix->keys, sizeof(uint64_t) * ix->n);
  copy_d2h(v.data(), ix->vals, sizeof(uint32_t) * ix->n);
  HIP_CHECK(hipStreamSynchronize(stream()));
  KmerPosRec *o = (KmerPosRec *) out;
  const std::vector<int64_t> &boff = ix->h_boff;
  for (int64_t i = 0; i < ix->n; i++)
    { o[i].code = k[i];
      if (ix->posbits > 0)
        { o[i].read = (int32_t) (v[i] >> ix->posbits);
          o[i].rpos = (int32_t) (v[i] & ((1u << ix->posbits) - 1u));
        }
      else
        { int64_t q = v[i];
          int r = (int) (std::upper_bound(boff.begin(), boff.end(), q) - boff.begin()) - 1;
          o[i].read = r;
          o[i].rpos = (int32_t) (q - boff[r]);
        }
    }
  return (int) ix->n;
}

int dmh_sort_pairs(uint64_t *keys, uint32_t *vals, int64_t n, int nbits)
{ if (n <= 0) return 0;
  State &st = S();
  uint64_t *dk = (uint64_t *) dev_alloc(sizeof(uint64_t) * n, "sort keys");
  uint32_t *dv = (uint32_t *) dev_alloc(sizeof(uint32_t) * n, "sort vals");
  st.sort.keys_alt.ensure(n, "sort buffer");
  st.sort.vals_alt.ensure(n, "sort buffer");
  copy_h2d(dk, keys, sizeof(uint64_t) * n);
  copy_h2d(dv, vals, sizeof(uint32_t) * n);
  uint64_t *k0 = dk, *k1 = st.sort.keys_alt.p;
  uint32_t *v0 = dv, *v1 = st.sort.vals_alt.p;
  radix_sort_pairs(&k0, &v0, &k1, &v1, n, 0, nbits, st.sort, "sort_pairs");
  copy_d2h(keys, k0, sizeof(uint64_t) * n);
  copy_d2h(vals, v0, sizeof(uint32_t) * n);
  HIP_CHECK(hipStreamSynchronize(stream()));
  dev_free(dk);
  dev_free(dv);
  return 0;
}

// Stage-level: exclusive scan of n uint32 in place (the device-wide scan the sorts and the filter use), repeated `reps`
// times on the device copy of the input; *total = the sum; *ms (may be null) = mean time of one scan.
int dmh_exclusive_scan(uint32_t *data, int64_t n, int reps, uint32_t *total, double *ms)
{ if (n <= 0) return 0;
  State &st = S();
  uint32_t *din = (uint32_t *) dev_alloc(sizeof(uint32_t) * n, "scan input");
  uint32_t *dout = (uint32_t *) dev_alloc(sizeof(uint32_t) * (n + 1), "scan output");
  copy_h2d(din, data, sizeof(uint32_t) * n);
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0));
  HIP_CHECK(hipEventCreate(&e1));
  exclusive_scan_u32((const uint32_t *) din, dout, n, st.sort.bsum, dout + n);         // (allocations, first-use set-up)
  HIP_CHECK(hipEventRecord(e0, stream()));
  for (int r = 0; r < (reps > 0 ? reps : 1); r++)
    exclusive_scan_u32((const uint32_t *) din, dout, n, st.sort.bsum, dout + n);
  HIP_CHECK(hipEventRecord(e1, stream()));
  HIP_CHECK(hipEventSynchronize(e1));
  float t = 0;
  HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
  if (ms != nullptr) *ms = (double) t / (reps > 0 ? reps : 1);
  copy_d2h(data, dout, sizeof(uint32_t) * n);
  if (total != nullptr) copy_d2h(total, dout + n, sizeof(uint32_t));
  HIP_CHECK(hipStreamSynchronize(stream()));
  HIP_CHECK(hipEventDestroy(e0));
  HIP_CHECK(hipEventDestroy(e1));
  dev_free(din);
  dev_free(dout);
  return 0;
}

}  // extern "C"
