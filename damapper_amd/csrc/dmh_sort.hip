// dmh_sort.hip -- k-mer emit, stable LSD radix sort of (uint64 key, uint32 value) pairs and the
// device-wide exclusive scan they share.  gfx950 only: 64-lane wavefronts, 256-thread blocks.
//
// Replaces the reference's tuple_thread (map.c:466-579) and lex_sort/lex_thread
// (map.c:181-444).  The reference moves 16-byte records through 8-bit LSD passes with one
// pthread per slice; here a k-mer is a (code, base offset) pair of 12 bytes, a pass is
// histogram -> scan -> scatter, and stability inside a tile comes from wavefront match masks
// (ballot), so the result is the same total order (code, read, rpos) the reference produces.

#include <algorithm>
#include "dmh_common.h"
#include "dmh_state.h"
#include "dmh_device.h"

namespace dmh {

// ---------------------------------------------------------------------------------------
//  Device-wide exclusive scan (uint32)
// ---------------------------------------------------------------------------------------

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS   = 8;
constexpr int SCAN_CHUNK   = SCAN_THREADS * SCAN_ITEMS;

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const uint32_t *data, int64_t n, uint32_t *bsum)
{ __shared__ uint32_t ws[4];
  int64_t base = (int64_t) blockIdx.x * SCAN_CHUNK;
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++)
    { int64_t i = base + j * SCAN_THREADS + threadIdx.x;
      if (i < n) s += data[i];
    }
  uint32_t tot;
  block_excl_scan(s, ws, &tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_top(uint32_t *bsum, int nb, uint32_t *total)
{ __shared__ uint32_t ws[4];
  uint32_t carry = 0;
  for (int base = 0; base < nb; base += SCAN_THREADS)
    { int i = base + threadIdx.x;
      uint32_t v = (i < nb) ? bsum[i] : 0;
      uint32_t tot;
      uint32_t e = block_excl_scan(v, ws, &tot);
      if (i < nb) bsum[i] = carry + e;
      carry += tot;
    }
  if (threadIdx.x == 0 && total != nullptr) *total = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(uint32_t *data, int64_t n, const uint32_t *bsum)
{ __shared__ uint32_t ws[4];
  int64_t base = (int64_t) blockIdx.x * SCAN_CHUNK + (int64_t) threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++)
    { v[j] = (base + j < n) ? data[base + j] : 0;
      s += v[j];
    }
  uint32_t tot;
  uint32_t e = block_excl_scan(s, ws, &tot) + bsum[blockIdx.x];
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++)
    { if (base + j < n) data[base + j] = e;
      e += v[j];
    }
}

// One kernel instead of the three above: tiles take tickets (so that a tile's predecessors are always running or
// done), publish their sum, and find their prefix by looking back over the status words of the tiles before them
// (aggregate or inclusive prefix, value and state in ONE 64-bit word written by a device-scope atomic store -- no
// fence, nothing else is communicated).  The status words are never cleared: they carry the number of the scan that
// wrote them (epoch), and tickets count on from scan to scan.  15,000 three-kernel scans per 24 read blocks of the
// bench workload (11-28 us per launch) were 14 ms per read block.
constexpr int LB_ITEMS = 16;
constexpr int LB_TILE  = SCAN_THREADS * LB_ITEMS;
constexpr int LB_LOOK  = 4;                          // status words a lane of the look-back reads per round (256 tiles a round)
constexpr int64_t LB_MAX = (int64_t) 1 << 20;        // longer arrays take the three kernels (measured: 13.5 us either way at 1 M values, 55 against 31 us at 3.6 M; 6.7 against 10.8 us at 25 k)

__device__ __forceinline__ uint64_t lb_pack(uint32_t epoch, uint32_t flag, uint32_t value)
{ return ((uint64_t) epoch << 34) | ((uint64_t) flag << 32) | (uint64_t) value; }

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_lookback(const uint32_t *src, uint32_t *dst, int64_t n,
                                                                 unsigned long long *status, uint32_t *ticket, uint32_t tbase,
                                                                 uint32_t epoch, uint32_t *total)
{ __shared__ uint32_t ws[4];
  __shared__ uint32_t s_tile, s_prefix;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) s_tile = atomicAdd(ticket, 1u) - tbase;
  __syncthreads();
  const uint32_t tile = s_tile;
  const int64_t base = (int64_t) tile * LB_TILE + (int64_t) tid * LB_ITEMS;
  uint32_t v[LB_ITEMS], sum = 0;
#pragma unroll
  for (int j = 0; j < LB_ITEMS; j++)
    { v[j] = (base + j < n) ? src[base + j] : 0;
      sum += v[j];
    }
  uint32_t tot;
  uint32_t e = block_excl_scan(sum, ws, &tot);
  if (tid == 0)
    { __hip_atomic_store(status + tile, lb_pack(epoch, tile == 0 ? 2u : 1u, tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tile == 0) s_prefix = 0;
    }
  if (tile > 0 && tid < 64)
    { // All tiles of a scan are resident at once and start together, so a tile finds its predecessors' SUMS soon but
      // their prefixes late: a round takes in 256 predecessors (four per lane, nearest first), adds up the sums in front
      // of the nearest finished prefix, and goes further back only when there is none among them.
      uint32_t prefix = 0;
      int64_t  p = (int64_t) tile - 1;                 // the nearest tile not yet accounted for
      while (true)
        { unsigned long long w[LB_LOOK];
          bool wait;
          do
            { wait = false;
#pragma unroll
              for (int u = 0; u < LB_LOOK; u++)
                { const int64_t q = p - (int64_t) lane * LB_LOOK - u;      // (a tile before the first: a finished prefix of zero)
                  w[u] = (q >= 0) ? __hip_atomic_load(status + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : lb_pack(epoch, 2u, 0u);
                  wait = wait || (uint32_t) (w[u] >> 34) != epoch || ((w[u] >> 32) & 3ull) == 0;
                }
            }
          while (__any(wait));
          // inside the lane: the sums up to and including its nearest finished prefix
          uint32_t x = 0;
          bool     fin = false;
#pragma unroll
          for (int u = 0; u < LB_LOOK; u++)
            if (!fin)
              { x += (uint32_t) w[u];
                fin = ((w[u] >> 32) & 3ull) == 2ull;
              }
          const uint64_t done = __ballot(fin);
          const int f = done ? __ffsll((long long) done) - 1 : 64;      // the lane with the nearest finished prefix
          uint32_t y = (lane <= f) ? x : 0u;
          for (int o = 32; o > 0; o >>= 1) y += __shfl_down(y, o);
          prefix += __shfl(y, 0);
          if (f < 64) break;
          p -= 64 * LB_LOOK;
        }
      if (lane == 0)
        { s_prefix = prefix;
          __hip_atomic_store(status + tile, lb_pack(epoch, 2u, prefix + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
  __syncthreads();
  e += s_prefix;
#pragma unroll
  for (int j = 0; j < LB_ITEMS; j++)
    { if (base + j < n) dst[base + j] = e;
      e += v[j];
    }
  if (total != nullptr && tid == SCAN_THREADS - 1 && (int64_t) (tile + 1) * LB_TILE >= n) *total = e;
}

// Scratch buffer of a stream that scans: [0] ticket counter, [8 ...] status words of the look-back (LB_MAX / LB_TILE
// tiles), behind them the block sums of the three-kernel scan.  Per buffer: the allocation the words live in, the number
// of scans so far and of tiles.
constexpr size_t LB_WORDS = 8 + 2 * (size_t) (LB_MAX / LB_TILE) + 8;
struct LookbackState { const void *buf; unsigned gen; uint32_t epoch, tiles; };
static LookbackState g_lb[6] = {{nullptr, 0, 0, 0}, {nullptr, 0, 0, 0}, {nullptr, 0, 0, 0}, {nullptr, 0, 0, 0}, {nullptr, 0, 0, 0}, {nullptr, 0, 0, 0}};

void exclusive_scan_u32(const uint32_t *d_src, uint32_t *d_data, int64_t n, DevBuf<uint32_t> &bsum, uint32_t *d_total)
{ if (n <= 0)
    { if (d_total) HIP_CHECK(hipMemsetAsync(d_total, 0, sizeof(uint32_t), stream()));
      return;
    }
  static const bool three = getenv("DMH_SCAN3") != nullptr;         // (the three-kernel scan, for comparison)
  if (three || n > LB_MAX)
    { if (d_src != d_data) HIP_CHECK(hipMemcpyAsync(d_data, d_src, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, stream()));
      int nb = ceil_div(n, SCAN_CHUNK);
      bsum.ensure(LB_WORDS + (size_t) nb, "scan block sums");
      uint32_t *bs = bsum.p + LB_WORDS;
      hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(SCAN_THREADS), 0, stream(), d_data, n, bs);
      hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(SCAN_THREADS), 0, stream(), bs, nb, d_total);
      hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(SCAN_THREADS), 0, stream(), d_data, n, bs);
      HIP_CHECK(hipGetLastError());
      return;
    }
  const int nt = ceil_div(n, LB_TILE);
  bsum.ensure(LB_WORDS, "scan tile states");
  LookbackState *L = nullptr;
  for (auto &x : g_lb) if (x.buf == (const void *) &bsum) L = &x;
  if (L == nullptr || L->gen != bsum.gen || L->epoch >= 0x3ffffff0u)
    { // a new (or re-allocated, or very old) scratch buffer: its words start out as "never written"
      if (L == nullptr)
        { L = &g_lb[0];
          for (auto &x : g_lb) if (x.buf == nullptr) { L = &x; break; }
        }
      L->buf = (const void *) &bsum; L->gen = bsum.gen; L->epoch = 0; L->tiles = 0;
      HIP_CHECK(hipMemsetAsync(bsum.p, 0, sizeof(uint32_t) * LB_WORDS, stream()));
    }
  L->epoch += 1;
  unsigned long long *status = (unsigned long long *) (bsum.p + 8);
  hipLaunchKernelGGL(k_scan_lookback, dim3(nt), dim3(SCAN_THREADS), 0, stream(), d_src, d_data, n, status, bsum.p, L->tiles,
                     L->epoch, d_total);
  L->tiles += (uint32_t) nt;
  HIP_CHECK(hipGetLastError());
}

void exclusive_scan_u32(uint32_t *d_data, int64_t n, DevBuf<uint32_t> &bsum, uint32_t *d_total)
{ exclusive_scan_u32(d_data, d_data, n, bsum, d_total); }

// the scratch buffers have been released (dmh_reset): whatever they are when they come back, they start afresh
void scan_state_reset()
{ for (auto &x : g_lb) { x.buf = nullptr; x.gen = 0; x.epoch = 0; x.tiles = 0; } }

// ---------------------------------------------------------------------------------------
//  Radix sort pass: histogram -> scan -> stable scatter
// ---------------------------------------------------------------------------------------

constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS   = 16;
constexpr int SORT_TILE    = SORT_THREADS * SORT_ITEMS;    // 4096 pairs per workgroup
constexpr int SORT_WTILE   = 64 * SORT_ITEMS;              // 1024 pairs per wavefront

__device__ __forceinline__ int xcd_tile(int b, int ntiles)
{ const int per = ntiles >> 3;                 // tiles per XCD (the remainder keeps its own index)
  if (b >= (per << 3)) return b;
  return (b & 7) * per + (b >> 3);
}

// Digit histogram of every tile, laid out digit-major (hist[d*ntiles + tile]) so that one
// exclusive scan over the whole array yields each (digit, tile)'s first output slot.
__global__ __launch_bounds__(SORT_THREADS) void k_radix_hist(const uint64_t *__restrict__ keys, int64_t n,
                                                              int shift, uint32_t mask,
                                                              uint32_t *__restrict__ hist, int ntiles, int tile_base)
{ __shared__ uint32_t h[256];
  int tid = threadIdx.x;
  h[tid] = 0;
  __syncthreads();
  // (XCD-aware like the scatter: sixteen neighbouring tiles share each 64-byte line of hist[], and only tiles that run
  // on the same XCD merge their 4-byte counts in its L2 before the line is written back)
  const int tile = tile_base + xcd_tile(blockIdx.x, gridDim.x);
  int64_t base = (int64_t) tile * SORT_TILE;
  // all loads first, from clamped addresses (a guarded load ahead of its LDS atomic compiles to
  // load - wait - atomic, SORT_ITEMS round trips in a row)
  uint64_t k[SORT_ITEMS];
#pragma unroll
  for (int j = 0; j < SORT_ITEMS; j++)
    { int64_t i = base + j * SORT_THREADS + tid;
      k[j] = __builtin_nontemporal_load(keys + (i < n ? i : n - 1));
    }
#pragma unroll
  for (int j = 0; j < SORT_ITEMS; j++)
    { int64_t i = base + j * SORT_THREADS + tid;
      if (i < n)
        atomicAdd(&h[(uint32_t) (k[j] >> shift) & mask], 1u);
    }
  __syncthreads();
  hist[(size_t) tid * ntiles + tile] = h[tid];
}

// Stable scatter.  Each wavefront owns 1024 consecutive pairs (item j of lane l is element
// j*64+l of that run, so loads are coalesced); ranks inside the wavefront come from ballot
// match masks, ranks across wavefronts from a 4 x 256 LDS table.  The tile is then reordered
// through LDS into digit order, so that consecutive threads write consecutive slots of each
// digit's output run (coalesced 8-byte / 4-byte stores), the run's global start coming from the
// scanned histogram.
// ORD: the rank of a pair among its wavefront's pairs with the same digit is what an LDS atomic add on the digit's counter
// returns -- the adds of a wavefront are served in lane order on gfx950 (lds_atomics_ordered(): checked on the device) and
// one instruction after the other, so the ranks are stable; else eight ballots per pair (round 1).  The kernel moves its
// bytes at 0.7 of the HBM peak alone, but runs next to kernels that are bound by vector issue: the ballots were two thirds
// of its vector instructions.
template <bool VALS, bool ORD>
__global__ __launch_bounds__(SORT_THREADS, 4) void k_radix_scatter(const uint64_t *__restrict__ kin,
                                                                 const uint32_t *__restrict__ vin,
                                                                 uint64_t *__restrict__ kout,
                                                                 uint32_t *__restrict__ vout, int64_t n,
                                                                 int shift, uint32_t mask,
                                                                 const uint32_t *__restrict__ offs, int ntiles,
                                                                 int tile_base)
{ __shared__ uint64_t skey[SORT_TILE];        // keys in digit order; afterwards its first half holds the values
  __shared__ uint32_t wh[4][256];
  __shared__ uint32_t gdelta[256];
  __shared__ uint32_t ws[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int x = tid; x < 1024; x += SORT_THREADS) ((uint32_t *) wh)[x] = 0;
  __syncthreads();

  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an
  // L2), so give every XCD one contiguous range of tiles.  Neighbouring tiles write neighbouring
  // pieces of each digit's output run; processed back to back on one XCD they merge into whole
  // cache lines in that L2 before being written back.
  const int tile  = tile_base + xcd_tile(blockIdx.x, gridDim.x);     // (a launch may cover a range of tiles)
  const int64_t tile0 = (int64_t) tile * SORT_TILE;
  const int64_t base  = tile0 + (int64_t) w * SORT_WTILE + lane;
  uint64_t key[SORT_ITEMS];
  uint32_t val[SORT_ITEMS];
  uint32_t rk[SORT_ITEMS];
#pragma unroll
  for (int j = 0; j < SORT_ITEMS; j++)
    { int64_t i = base + j * 64;
      bool ok = i < n;
      key[j] = ok ? __builtin_nontemporal_load(kin + i) : 0;     // read once: keep it out of the caches' way
      val[j] = (VALS && ok) ? __builtin_nontemporal_load(vin + i) : 0;
    }

  volatile uint32_t *mywh = wh[w];
  const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int j = 0; j < SORT_ITEMS; j++)
    { bool     ok = (base + j * 64) < n;
      uint32_t d  = (uint32_t) (key[j] >> shift) & mask;
      if (ORD)
        { rk[j] = ok ? atomicAdd(&wh[w][d], 1u) : 0u;
          continue;
        }
      uint64_t peers = __ballot(ok);
#pragma unroll
      for (int b = 0; b < 8; b++)
        { bool     bit = (d >> b) & 1;
          uint64_t bal = __ballot(bit);
          peers &= bit ? bal : ~bal;
        }
      uint32_t below = __popcll(peers & lt);
      uint32_t cnt   = __popcll(peers);
      uint32_t b0    = ok ? mywh[d] : 0;
      __builtin_amdgcn_wave_barrier();
      if (ok && below == 0) mywh[d] = b0 + cnt;
      __builtin_amdgcn_wave_barrier();
      rk[j] = b0 + below;
    }
  __syncthreads();

  // digit tid: counts per wavefront -> tile-local start of the digit's run and per-wavefront bases
  { uint32_t c0 = wh[0][tid], c1 = wh[1][tid], c2 = wh[2][tid], c3 = wh[3][tid];
    uint32_t tot;
    uint32_t ls = block_excl_scan(c0 + c1 + c2 + c3, ws, &tot);
    wh[0][tid] = ls;
    wh[1][tid] = ls + c0;
    wh[2][tid] = ls + c0 + c1;
    wh[3][tid] = ls + c0 + c1 + c2;
    gdelta[tid] = offs[(size_t) tid * ntiles + tile] - ls;
  }
  __syncthreads();

  // keys and values go through the SAME 32 KB of LDS one after the other (37 KB per workgroup = four
  // workgroups per CU instead of three); the output slot of each position is kept in a register
#pragma unroll
  for (int j = 0; j < SORT_ITEMS; j++)
    { if ((base + j * 64) < n)
        { uint32_t d = (uint32_t) (key[j] >> shift) & mask;
          rk[j] += wh[w][d];                    // position inside the tile
          skey[rk[j]] = key[j];
        }
    }
  __syncthreads();

  const int count = (int) ((n - tile0 < (int64_t) SORT_TILE) ? (n - tile0) : (int64_t) SORT_TILE);
  uint32_t slot[SORT_ITEMS];
#pragma unroll
  for (int k2 = 0; k2 < SORT_ITEMS; k2++)
    { int i = tid + k2 * SORT_THREADS;
      if (i < count)
        { uint64_t k = skey[i];
          uint32_t d = (uint32_t) (k >> shift) & mask;
          slot[k2] = gdelta[d] + (uint32_t) i;
          kout[slot[k2]] = k;                   // (plain stores: neighbouring tiles' pieces merge in L2)
        }
    }
  if (!VALS) return;                            // (keys only: the words of the seed sort carry everything)
  __syncthreads();
  uint32_t *sval = (uint32_t *) skey;
#pragma unroll
  for (int j = 0; j < SORT_ITEMS; j++)
    if ((base + j * 64) < n) sval[rk[j]] = val[j];
  __syncthreads();
#pragma unroll
  for (int k2 = 0; k2 < SORT_ITEMS; k2++)
    { int i = tid + k2 * SORT_THREADS;
      if (i < count) vout[slot[k2]] = sval[i];
    }
}

// "DMH_SORT_BALLOT=1": the ranks from ballots whatever the device does
static bool scatter_ordered()
{ static const bool env_ballot = getenv("DMH_SORT_BALLOT") != nullptr && atoi(getenv("DMH_SORT_BALLOT")) != 0;
  const int o = S().opt_sort_ballot;                 // ("sort_ballot" test option)
  return !(o >= 0 ? o != 0 : env_ballot) && lds_atomics_ordered();
}

void radix_sort_pairs(uint64_t **keys, uint32_t **vals, uint64_t **okeys, uint32_t **ovals,
                      int64_t n, int lo_bit, int nbits, SortScratch &scr, const char *stage)
{ if (n <= 1 || nbits <= 0) return;
  if (n >= (int64_t) 0xffffffffll) fatal("radix_sort_pairs: %lld pairs exceed the 32-bit slot range", (long long) n);
  int ntiles = ceil_div(n, SORT_TILE);
  scr.hist.ensure((size_t) 256 * ntiles, "radix histogram");
  int npass = (nbits + 7) / 8;
  // algorithmic traffic of one pass with the reference's 16-byte records: read + write
  StageTimer tm(stage, (double) npass * (vals != nullptr ? 32.0 : 16.0) * (double) n);
  for (int p = 0; p < npass; p++)
    { int      shift = lo_bit + 8 * p;
      int      w     = nbits - 8 * p;
      uint32_t mask  = (w >= 8) ? 0xffu : ((1u << w) - 1u);
      { StageTimer th("k_radix_hist", 0.0);
        hipLaunchKernelGGL(k_radix_hist, dim3(ntiles), dim3(SORT_THREADS), 0, stream(),
                           (const uint64_t *) *keys, n, shift, mask, scr.hist.p, ntiles, 0);
      }
      { StageTimer ts("radix_scan", 0.0);
        exclusive_scan_u32(scr.hist.p, (int64_t) 256 * ntiles, scr.bsum, nullptr);
      }
      { // one LSD pass over 16-byte records in the reference: read 16 B + write 16 B per record; a keys-only pass
        // (the one-word hits of cluster mode) is charged what it moves, 8 B + 8 B
        StageTimer tc("k_radix_scatter", (vals != nullptr ? 32.0 : 16.0) * (double) n);
        const bool ord = scatter_ordered();
        if (vals != nullptr && ord)
          hipLaunchKernelGGL((k_radix_scatter<true, true>), dim3(ntiles), dim3(SORT_THREADS), 0, stream(),
                             (const uint64_t *) *keys, (const uint32_t *) *vals, *okeys, *ovals, n,
                             shift, mask, (const uint32_t *) scr.hist.p, ntiles, 0);
        else if (vals != nullptr)
          hipLaunchKernelGGL((k_radix_scatter<true, false>), dim3(ntiles), dim3(SORT_THREADS), 0, stream(),
                             (const uint64_t *) *keys, (const uint32_t *) *vals, *okeys, *ovals, n,
                             shift, mask, (const uint32_t *) scr.hist.p, ntiles, 0);
        else if (ord)
          hipLaunchKernelGGL((k_radix_scatter<false, true>), dim3(ntiles), dim3(SORT_THREADS), 0, stream(),
                             (const uint64_t *) *keys, (const uint32_t *) nullptr, *okeys, (uint32_t *) nullptr, n,
                             shift, mask, (const uint32_t *) scr.hist.p, ntiles, 0);
        else
          hipLaunchKernelGGL((k_radix_scatter<false, false>), dim3(ntiles), dim3(SORT_THREADS), 0, stream(),
                             (const uint64_t *) *keys, (const uint32_t *) nullptr, *okeys, (uint32_t *) nullptr, n,
                             shift, mask, (const uint32_t *) scr.hist.p, ntiles, 0);
      }
      HIP_CHECK(hipGetLastError());
      uint64_t *tk = *keys; *keys = *okeys; *okeys = tk;
      if (vals != nullptr) { uint32_t *tv = *vals; *vals = *ovals; *ovals = tv; }
    }
}

// One pass of the sort above cut into separately queued pieces (deferred index builds, bg_pump):
// the histogram in `nparts` launches over consecutive tile ranges (the scan follows the last one),
// the scatter likewise.
static void part_range(int ntiles, int part, int nparts, int &t0, int &t1)
{ int per = ((ntiles + nparts - 1) / nparts + 7) & ~7;       // whole XCD octets
  t0 = std::min(ntiles, part * per);
  t1 = (part == nparts - 1) ? ntiles : std::min(ntiles, t0 + per);
}

void radix_pass_hist(const uint64_t *keys, int64_t n, int shift, int w, SortScratch &scr, int part, int nparts)
{ int ntiles = ceil_div(n, SORT_TILE);
  if (part == 0) scr.hist.ensure((size_t) 256 * ntiles, "radix histogram");
  uint32_t mask = (w >= 8) ? 0xffu : ((1u << w) - 1u);
  int t0, t1;
  part_range(ntiles, part, nparts, t0, t1);
  if (t1 > t0)
    { StageTimer th("k_radix_hist", 0.0);
      hipLaunchKernelGGL(k_radix_hist, dim3(t1 - t0), dim3(SORT_THREADS), 0, stream(), keys, n, shift, mask, scr.hist.p,
                         ntiles, t0);
    }
  if (part == nparts - 1)
    { StageTimer ts("radix_scan", 0.0);
      exclusive_scan_u32(scr.hist.p, (int64_t) 256 * ntiles, scr.bsum, nullptr);
    }
  HIP_CHECK(hipGetLastError());
}

void radix_pass_scatter(const uint64_t *keys, const uint32_t *vals, uint64_t *okeys, uint32_t *ovals, int64_t n,
                        int shift, int w, SortScratch &scr, int part, int nparts)
{ int ntiles = ceil_div(n, SORT_TILE);
  uint32_t mask = (w >= 8) ? 0xffu : ((1u << w) - 1u);
  int t0, t1;
  part_range(ntiles, part, nparts, t0, t1);
  if (t1 <= t0) return;
  int64_t recs = std::min<int64_t>(n, (int64_t) t1 * SORT_TILE) - (int64_t) t0 * SORT_TILE;
  StageTimer tc("k_radix_scatter", 32.0 * (double) recs);
  if (scatter_ordered())
    hipLaunchKernelGGL((k_radix_scatter<true, true>), dim3(t1 - t0), dim3(SORT_THREADS), 0, stream(), keys, vals, okeys, ovals, n,
                       shift, mask, (const uint32_t *) scr.hist.p, ntiles, t0);
  else
    hipLaunchKernelGGL((k_radix_scatter<true, false>), dim3(t1 - t0), dim3(SORT_THREADS), 0, stream(), keys, vals, okeys, ovals, n,
                       shift, mask, (const uint32_t *) scr.hist.p, ntiles, t0);
  HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------
//  K-mer emit
// ---------------------------------------------------------------------------------------

constexpr int EMIT_THREADS = 256;
constexpr int EMIT_ITEMS   = 8;
constexpr int EMIT_TILE    = EMIT_THREADS * EMIT_ITEMS;   // base positions per workgroup

// One thread walks EMIT_ITEMS consecutive base positions with a rolling 2-bit code
// (map.c:560-571: first base most significant, rpos = index of the LAST base).  Position q of
// read i (i = number of separators before q) is the end of a k-mer iff the k bases ending at q
// are all < 4; its slot in the output is q - i*k - (k-1), i.e. (read, rpos) order, which is the
// order the reference's tuple_thread generates and the stable sort preserves.
__global__ __launch_bounds__(EMIT_THREADS) void k_kmer_emit(const uint8_t *__restrict__ bases, int64_t L,
                                                            const int64_t *__restrict__ boff, int nreads,
                                                            int K, uint64_t kmask,
                                                            uint64_t *__restrict__ keys,
                                                            uint32_t *__restrict__ vals, int64_t nk, int block_base,
                                                            int posbits)
{ __shared__ uint4    sraw[(EMIT_TILE + 64) / 16 + 1];   // the tile's bases + halo, as fetched (16-byte lines)
  __shared__ uint64_t scode[EMIT_TILE];
  __shared__ uint16_t sread[EMIT_TILE];     // read index relative to the tile's first read
  __shared__ uint8_t  sok[EMIT_TILE];
  __shared__ uint32_t ws[4];
  __shared__ int      s_i0;

  const int     tid  = threadIdx.x;
  const int64_t tile = (int64_t) (block_base + (int) blockIdx.x) * EMIT_TILE;
  const int     halo = K - 1;

  // The tile and its K-1 bases of halo arrive in one round of 16-byte loads (the block copy starts
  // 256-byte aligned one byte before bases[0] -- the leading separator -- and is padded behind).
  const uint8_t *raw = bases - 1;                          // raw[q + 1] = bases[q]
  const int64_t  r0  = tile - halo + 1;                    // first byte needed (negative only for the first tile)
  const int64_t  a0  = (r0 >> 4) << 4;
  const int      sh  = (int) (r0 - a0);
  const int      nvec = (sh + EMIT_TILE + halo + 15) >> 4;
  // The copy holds raw[0 .. L] and 64 bytes of padding: a 16-byte line that would reach past them is NOT fetched -- it
  // belongs to positions behind the block, which are turned into separators below anyway.  (The last tile used to
  // fetch up to EMIT_TILE + K bytes past the end of the allocation: harmless wherever the next page happened to be
  // mapped, a device fault where it was not -- seen on one box in five, with the smallest blocks of the test suite.)
  if (tid < nvec)
    { int64_t off = a0 + 16 * (int64_t) tid;
      sraw[tid] = (off >= 0 && off + 16 <= L + 1 + 64) ? *(const uint4 *) (raw + off)
                                                       : make_uint4(0x04040404u, 0x04040404u, 0x04040404u, 0x04040404u);
    }
  uint8_t *sb = (uint8_t *) sraw + sh;                     // sb[x] = bases[tile - halo + x]
  if (tile + EMIT_TILE > L)                                // the last tile: whatever lies behind the block is a separator
    { __syncthreads();
      for (int x = tid; x < EMIT_TILE + halo; x += EMIT_THREADS)
        if (tile - halo + x >= L) sb[x] = (uint8_t) 4;
    }
  if (tid == 0)
    { // largest i with boff[i] <= tile  (boff[0] = 0 <= tile always)
      int lo = 0, hi = nreads;
      while (lo < hi)
        { int m = (lo + hi + 1) >> 1;
          if (boff[m] > tile) hi = m - 1; else lo = m;
        }
      s_i0 = lo;
    }
  __syncthreads();

  const int x0 = tid * EMIT_ITEMS;
  uint32_t nsep = 0;
#pragma unroll
  for (int j = 0; j < EMIT_ITEMS; j++)
    nsep += (sb[halo + x0 + j] > 3);
  uint32_t tot;
  uint32_t sepb = block_excl_scan(nsep, ws, &tot);   // separators in tile positions [0, x0)

  uint64_t c   = 0;
  int      run = 0;
  for (int h = 0; h < halo; h++)
    { uint32_t b = sb[x0 + h];
      if (b > 3) { run = 0; c = 0; }
      else       { run += 1; c = ((c << 2) | b) & kmask; }
    }
#pragma unroll
  for (int j = 0; j < EMIT_ITEMS; j++)
    { uint32_t b = sb[halo + x0 + j];
      sread[x0 + j] = (uint16_t) sepb;
      if (b > 3) { run = 0; c = 0; sepb += 1; }
      else       { run += 1; c = ((c << 2) | b) & kmask; }
      scode[x0 + j] = c;
      sok[x0 + j]   = (run >= K);
    }
  __syncthreads();

  const int i0 = s_i0;
  for (int x = tid; x < EMIT_TILE; x += EMIT_THREADS)
    if (sok[x])
      { int64_t q = tile + x;
        int64_t i = i0 + sread[x];
        int64_t o = q - i * K - (K - 1);
        if (o >= 0 && o < nk)
          { keys[o] = scode[x];
            // posbits > 0: the position as (read << posbits) | rpos, what the merge makes hits of (index_pack_positions
            // would convert the sorted index later, with a table look-up per k-mer)
            vals[o] = (posbits > 0) ? (((uint32_t) i << posbits) | (uint32_t) (q - boff[i])) : (uint32_t) q;
          }
      }
}

void kmer_emit(const uint8_t *d_bases, int64_t L, const int64_t *d_boff, int nreads, int K,
               uint64_t *d_keys, uint32_t *d_vals, int64_t nk, int part, int nparts, int posbits)
{ if (L <= 0) return;
  uint64_t kmask = (K >= 32) ? ~0ull : ((1ull << (2 * K)) - 1ull);
  int nb = ceil_div(L, EMIT_TILE);
  int per = (nb + nparts - 1) / nparts;
  int b0 = std::min(nb, part * per), b1 = (part == nparts - 1) ? nb : std::min(nb, b0 + per);
  if (b1 <= b0) return;
  StageTimer tm("kmer_emit", ((double) L + 16.0 * (double) nk) * (double) (b1 - b0) / (double) nb);
  hipLaunchKernelGGL(k_kmer_emit, dim3(b1 - b0), dim3(EMIT_THREADS), 0, stream(), d_bases, L, d_boff, nreads,
                     K, kmask, d_keys, d_vals, nk, b0, posbits);
  HIP_CHECK(hipGetLastError());
}

}  // namespace dmh
