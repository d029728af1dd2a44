// dmh_merge.hip -- seed-hit merge of two sorted k-mer indices and the seed sort.
//
// Replaces count_thread / merge_thread (map.c:881-1002) and the pair sort of Match_Filter
// (map.c:2911-2937, 3110-3126).  The reference walks both sorted lists with one pthread per
// slice; here every READ k-mer looks its code up in the REF index through a bucket table on
// the code's top bits (both lists are sorted, so neighbouring lanes touch neighbouring table
// slots and REF entries), which gives the same multiset of (m_a x m_b) groups.  Hits are
// written in READ-index order with each READ k-mer's REF partners in REF-index order, then
// stably sorted on the packed key (aread, bread, apos): the resulting total order
// (aread, bread, apos, bpos) is the one the reference's chain_thread consumes.

#include "dmh_state.h"
#include "dmh_device.h"
#include "dmh_merge.h"
#include "dmh_chain.h"
#include <algorithm>
#include <deque>
#include <unordered_map>
#include <type_traits>

namespace dmh {

constexpr int MAXGRAM = 10000;     // map.c:32

// ---------------------------------------------------------------------------------------
//  Bucket table over the REF index: bstart[p] = first entry whose (code >> shift) >= p
// ---------------------------------------------------------------------------------------

__global__ void k_bucket_bounds(const uint64_t *__restrict__ keys, int64_t n, int shift, int64_t nbucket,
                                uint32_t *__restrict__ bstart)
{ int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  int64_t cur  = (i < n) ? (int64_t) (keys[i] >> shift) : nbucket;
  int64_t prev = (i > 0) ? (int64_t) (keys[i - 1] >> shift) : -1;
  for (int64_t p = prev + 1; p <= cur; p++)
    bstart[p] = (uint32_t) i;
}

struct BLookup
{ const uint64_t *keys;
  const uint32_t *bstart;
  int             shift;
  const uint16_t *klow;         // keys[i] & ((1 << shift) - 1) when shift <= 16 (big indices), else nullptr
};

__global__ void k_low16(const uint64_t *__restrict__ keys, int64_t n, uint32_t mask, uint16_t *__restrict__ klow)
{ int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) klow[i] = (uint16_t) ((uint32_t) keys[i] & mask);
}

// [lo, lo+cnt) = entries of the REF index with code c
__device__ __forceinline__ void b_range(const BLookup &B, uint64_t c, uint32_t &lo, uint32_t &cnt)
{ uint64_t p = c >> B.shift;
  uint32_t s = B.bstart[p], e = B.bstart[p + 1];
  if (B.klow != nullptr)
    { // the entries of a bucket differ in their low `shift` bits only, kept as a 2-byte array: a long bucket (a
      // repeat's k-mer with its copies) spans a few cache lines of it instead of dozens of the 8-byte keys, so
      // all but the first steps of the searches below stay in lines already fetched; the end of the run is
      // looked for by doubling from its start (runs are short next to their buckets' tails only rarely)
      const uint32_t cq = (uint32_t) (c & (((uint64_t) 1 << B.shift) - 1));
      uint32_t l = s, h = e;
      while (l < h)
        { uint32_t m = (l + h) >> 1;
          if ((uint32_t) B.klow[m] < cq) l = m + 1; else h = m;
        }
      lo = l;
      if (l >= e || (uint32_t) B.klow[l] != cq) { cnt = 0; return; }
      uint32_t a = l, step = 1, z;               // klow[a] == cq; find the first z > a with klow[z] != cq (or e)
      while (true)
        { uint32_t t = a + step;
          if (t >= e) { z = e; break; }
          if ((uint32_t) B.klow[t] == cq) { a = t; step <<= 1; }
          else { z = t; break; }
        }
      while (z - a > 1)
        { uint32_t m = (a + z) >> 1;
          if ((uint32_t) B.klow[m] == cq) a = m; else z = m;
        }
      cnt = z - l;
      return;
    }
  uint32_t l = s, h = e;
  while (l < h)
    { uint32_t m = (l + h) >> 1;
      if (B.keys[m] < c) l = m + 1; else h = m;
    }
  lo = l;
  if (l >= e || B.keys[l] != c) { cnt = 0; return; }
  uint32_t l2 = l + 1, h2 = e;
  while (l2 < h2)
    { uint32_t m = (l2 + h2) >> 1;
      if (B.keys[m] == c) l2 = m + 1; else h2 = m;
    }
  cnt = l2 - l;
}

// ---------------------------------------------------------------------------------------
//  Match: ONE pass over the READ index.  Every first element of a run of equal READ codes looks
//  its code up in the REF index; matched runs are appended to a compact entry list
//  {a_start, m_a, b_lo, m_b} (a few % of the k-mers) and counted into the product histogram
//  (map.c:881-934).  Entry order is arbitrary: the seed sort below is stable on
//  (aread, bread, apos) and ties can only come from one entry, whose pairs are expanded in
//  (a, b) order.
// ---------------------------------------------------------------------------------------

constexpr int MG_THREADS = 256;
constexpr int MG_ITEMS   = 8;
constexpr int MG_TILE    = MG_THREADS * MG_ITEMS;

struct MergeEntry { uint32_t a0, ma, b0, mb; };
constexpr uint32_t PENDING = 0xffffffffu;     // mb of an entry whose REF run has still to be searched for

// TWO: the same READ codes looked up in a second REF index in the same sweep -- the forward and the complement index of
// one REF block, which the driver merges against one after the other (damapper.c:851, 861).  The launch has two
// workgroups per tile, one per table, numbered so that the pair runs on the same XCD at the same time (workgroups b and
// b + 8 share an L2): the READ index's keys, 2.1 of the 4.9 GB a pass reads, come from HBM once for the two passes and
// from that L2 the second time.  (One workgroup probing both tables in turn holds the codes across the two rounds: 133
// registers against 74, three wavefronts per SIMD against six, and was slower than two launches.)  Each table has its own
// entry windows, counts and sums, so that everything behind the match stays per call (limit rule, candidate order,
// domination: map.c:2992-3055).
struct MatchOut
{ MergeEntry         *tile_entries;
  uint32_t           *tile_count;
  unsigned long long *tile_hits;
};

// COMPACT (round 4, last): the look-ups that pass the cheap tests -- a `first`, a bucket that is not empty, a first key
// that does not exceed the code: one in four -- are LISTED per wavefront (LDS), and the expensive part (the four probes,
// the open look-ups, the run test, the entry and its place in the tile's window) runs over the list with every lane busy:
// two or three rounds per wavefront instead of eight, in each of which one lane in fifty had something to do while the
// wavefront paid for all of it (the kernel is the largest consumer of vector instructions of a read block).
constexpr int MG_CAND = 192;        // candidates a wavefront may have listed: 63 waiting + two items' worth
template <int NK, bool LOW, bool TWO = false, bool COMPACT = true>
__global__ __launch_bounds__(MG_THREADS) void k_merge_match(const uint64_t *__restrict__ akeys, int64_t na,
                                                            BLookup B0, MatchOut O0, BLookup B1, MatchOut O1)
{ __shared__ unsigned long long s_hits[2][4];
  __shared__ uint32_t s_count;
  __shared__ uint32_t s_f0[4];       // per wavefront: the `first` flags of its lane 0 (bit j: item j)
  const int tid = threadIdx.x, lane = tid & 63;
  const int      side = TWO ? (int) ((blockIdx.x >> 3) & 1u) : 0;
  const uint32_t tile = TWO ? (((blockIdx.x >> 4) << 3) | (blockIdx.x & 7u)) : blockIdx.x;
  if ((int64_t) tile * MG_TILE >= na) return;                  // (TWO: the grid is rounded up to whole groups of eight tiles)

  // The look-ups of a thread's MG_ITEMS codes are independent: issue them phase by phase (codes,
  // bucket bounds, the first two keys of the bucket) so that every thread has MG_ITEMS loads in
  // flight instead of one dependent chain after the other.  The bucket table is sized for ~0.15
  // entries per bucket, so two keys settle all but a few look-ups per thousand; only those run the
  // bisection (a wavefront pays its dependent loads if ANY of its lanes needs it).
  const int64_t tile0 = (int64_t) tile * MG_TILE + tid;
  uint64_t c[MG_ITEMS];
  bool     first[MG_ITEMS];
  uint32_t bs[MG_ITEMS], be[MG_ITEMS];
  { uint64_t pv[MG_ITEMS];
    // (positions in 32 bits: an index has fewer than 2^32 k-mers, checked by the host -- the 64-bit clamps and compares of
    //  these sixteen loads were a tenth of the kernel's vector instructions; a tile in the middle needs no clamp at all)
    const uint32_t t0 = (uint32_t) tile * MG_TILE + (uint32_t) tid, n32 = (uint32_t) na;
    if (tile > 0 && (uint64_t) (tile + 1) * MG_TILE <= (uint64_t) na)         // (uniform)
      {
#pragma unroll
        for (int j = 0; j < MG_ITEMS; j++)
          { const uint32_t i = t0 + (uint32_t) (j * MG_THREADS);
            c[j]  = akeys[i];
            pv[j] = akeys[i - 1];
          }
#pragma unroll
        for (int j = 0; j < MG_ITEMS; j++) first[j] = pv[j] != c[j];
      }
    else
      {
#pragma unroll
        for (int j = 0; j < MG_ITEMS; j++)
          { // unconditional loads from clamped addresses: a guarded load compiles to a branch with a wait at
            // its join, which strings the 16 loads of this phase into 16 round trips
            const uint32_t i  = t0 + (uint32_t) (j * MG_THREADS);
            const uint32_t ic = i < n32 ? i : n32 - 1u;
            c[j]  = akeys[ic];
            pv[j] = akeys[ic > 0 ? ic - 1u : 0u];
          }
#pragma unroll
        for (int j = 0; j < MG_ITEMS; j++)
          { const uint32_t i = t0 + (uint32_t) (j * MG_THREADS);
            first[j] = (i < n32) && (i == 0 || pv[j] != c[j]);
          }
      }
  }
  uint64_t firstm[MG_ITEMS];                 // (wave-uniform: scalar registers)
#pragma unroll
  for (int j = 0; j < MG_ITEMS; j++) firstm[j] = __ballot(first[j]);
  if (COMPACT && lane == 0)
    { uint32_t f0 = 0;
#pragma unroll
      for (int j = 0; j < MG_ITEMS; j++) f0 |= (uint32_t) (firstm[j] & 1ull) << j;
      s_f0[tid >> 6] = f0;
    }
  const BLookup B = (TWO && side) ? B1 : B0;
  MergeEntry *__restrict__ tile_entries = (TWO && side) ? O1.tile_entries : O0.tile_entries;
  uint32_t *__restrict__ tile_count = (TWO && side) ? O1.tile_count : O0.tile_count;
  unsigned long long *__restrict__ tile_hits = (TWO && side) ? O1.tile_hits : O0.tile_hits;
  if (tid == 0) s_count = 0;
  __syncthreads();

  // hits of the tile: all of them, and those of the runs below MAXGRAM (what the limit rule can keep at most)
  unsigned long long local = 0, below = 0;
  MergeEntry *dst = tile_entries + (size_t) tile * MG_TILE;              // the tile's own window of the entry list
#pragma unroll
  for (int j = 0; j < MG_ITEMS; j++)
    { bs[j] = be[j] = 0;
      if (first[j])
        { uint64_t p = c[j] >> B.shift;
          bs[j] = B.bstart[p];
          be[j] = B.bstart[p + 1];
        }
    }
  // the first NK keys of the bucket (NK = 2 for the sparse table of a small index, 4 for the denser table of
  // a block of a genome): they settle the look-up unless the bucket is longer and its NK-th key does not
  // exceed the code -- only then the bisection runs
  // LOW: the entries of a bucket share the bits above `shift`, so the probe compares the low bits only, kept
  // in a 2-byte array of their own: a quarter of the REF key traffic of a pass (the bisection uses the keys)
  // LOW with four keys: the four 2-byte values are ONE (unaligned) 8-byte load -- a quarter of the load instructions and
  // half the registers of four separate ones; values past the bucket's end are never looked at (every use is
  // guarded by the bucket's size)
  using KT = typename std::conditional<LOW, uint32_t, uint64_t>::type;      // (registers: NK * MG_ITEMS of these)
  constexpr bool PACK4 = LOW && NK == 4;
  typedef uint64_t u64a2 __attribute__((aligned(2)));
  KT       kk[PACK4 ? 1 : NK][MG_ITEMS];
  uint64_t kq[PACK4 ? MG_ITEMS : 1];
  const uint64_t lowmask = LOW ? (((uint64_t) 1 << B.shift) - 1) : ~0ull;
#pragma unroll
  for (int j = 0; j < MG_ITEMS; j++)
    { if (PACK4)
        kq[j] = (bs[j] < be[j]) ? *(const u64a2 *) (B.klow + bs[j]) : 0;
      else
        {
#pragma unroll
          for (int t = 0; t < NK; t++)
            { if (LOW) kk[t][j] = (KT) ((bs[j] + t < be[j]) ? (uint32_t) B.klow[bs[j] + t] : 0xffffffffu);
              else     kk[t][j] = (KT) ((bs[j] + t < be[j]) ? B.keys[bs[j] + t] : ~0ull);
            }
        }
    }
  auto key_at = [&](int t, int j) -> KT
  { if (PACK4) return (KT) ((kq[j] >> (16 * t)) & 0xffffull);
    return kk[PACK4 ? 0 : t][j];
  };

  if constexpr (COMPACT)
    { __shared__ uint64_t s_cc[4][MG_CAND];
      __shared__ uint32_t s_bs[4][MG_CAND], s_sz[4][MG_CAND], s_id[4][MG_CAND];
      __shared__ uint64_t s_kq[PACK4 ? 4 : 1][PACK4 ? MG_CAND : 1];
      __shared__ KT       s_kk[PACK4 ? 1 : 4][PACK4 ? 1 : NK][PACK4 ? 1 : MG_CAND];
      const int      wv = tid >> 6;
      const uint64_t lt = (1ull << lane) - 1ull;
      uint32_t ncand = 0;                                   // (wave-uniform)
      // the code at i + 1 is the next lane's (same item), the next wavefront's lane 0 (lane 63), or the first thread's of
      // the item that follows (the tile's last thread): a `first` flag says whether it differs
      const uint32_t f0next = s_f0[wv < 3 ? wv + 1 : 0];     // (the barrier above stands between the writes and this)
#pragma unroll
      for (int j = 0; j < MG_ITEMS; j++)
        { const int64_t i = tile0 + j * MG_THREADS;
          const KT cq = (KT) (c[j] & lowmask);             // what the fetched keys are compared with
          bool pass;
          if (PACK4)
            { // the four probes here already, without a branch: what is listed is what becomes an entry (one look-up in
              // fifty; an empty bucket or a lane that is no `first` has size 0, its keys are never looked at)
              const uint32_t sz = be[j] - bs[j], q = (uint32_t) cq;
              const uint32_t k0 = (uint32_t) key_at(0, j), k1 = (uint32_t) key_at(1, j), k2 = (uint32_t) key_at(2, j), k3 = (uint32_t) key_at(3, j);
              pass = (sz > 0 && k0 == q) || (sz > 1 && k1 == q) || (sz > 2 && k2 == q) || (sz > 3 && k3 == q) || (sz > 4 && k3 <= q);
            }
          else
            pass = first[j] && bs[j] < be[j] && key_at(0, j) <= cq;
          const uint64_t m = __ballot(pass);
          if (m)
            { if (pass)
                { bool nfirst;                              // the k-mer at i + 1 is a `first` (or there is none)
                  if (lane < 63)   nfirst = ((firstm[j] >> (lane + 1)) & 1ull) != 0;
                  else if (wv < 3) nfirst = ((f0next >> j) & 1u) != 0;
                  else if (j < MG_ITEMS - 1) nfirst = ((f0next >> (j + 1)) & 1u) != 0;
                  else             nfirst = !(i + 1 < na && akeys[i + 1] == c[j]);       // (the next tile's)
                  const bool same = (i + 1 < na) && !nfirst;
                  const uint32_t pos = ncand + (uint32_t) __popcll(m & lt);
                  s_cc[wv][pos] = c[j];
                  s_bs[wv][pos] = bs[j];
                  s_sz[wv][pos] = be[j] - bs[j];
                  s_id[wv][pos] = (uint32_t) (j * MG_THREADS + tid) | (same ? 0x10000u : 0u);
                  if (PACK4) s_kq[PACK4 ? wv : 0][PACK4 ? pos : 0] = kq[PACK4 ? j : 0];
                  else
                    {
#pragma unroll
                      for (int t = 0; t < NK; t++) s_kk[PACK4 ? 0 : wv][PACK4 ? 0 : t][PACK4 ? 0 : pos] = kk[PACK4 ? 0 : t][j];
                    }
                }
              ncand += (uint32_t) __popcll(m);
            }
          if ((j & 1) == 1)
            { // every second item: rounds of 64 candidates off the top of the list (all of it behind the last item)
              __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
              __builtin_amdgcn_wave_barrier();
              while (ncand >= 64u || (j == MG_ITEMS - 1 && ncand > 0u))
                { const uint32_t take = ncand < 64u ? ncand : 64u, x = ncand - take + (uint32_t) lane;
                  const bool act = (uint32_t) lane < take;
                  const uint32_t xs = act ? x : 0u;
                  const uint64_t cc = s_cc[wv][xs];
                  const uint32_t bsv = s_bs[wv][xs], sz = s_sz[wv][xs], idw = s_id[wv][xs];
                  const uint64_t kqv = PACK4 ? s_kq[PACK4 ? wv : 0][PACK4 ? xs : 0] : 0ull;
                  KT kv[NK];
#pragma unroll
                  for (int t = 0; t < NK; t++)
                    kv[t] = PACK4 ? (KT) ((kqv >> (16 * t)) & 0xffffull) : s_kk[PACK4 ? 0 : wv][PACK4 ? 0 : t][PACK4 ? 0 : xs];
                  const KT      cq2 = (KT) (cc & lowmask);
                  const int64_t i2  = (int64_t) tile * MG_TILE + (int64_t) (idw & 0xffffu);
                  unsigned long long ct = 0;
                  uint32_t lo = 0, mb = 0;
                  int64_t  e = 0;
                  bool     hit = false;
                  if (act)
                    { if (sz > (uint32_t) NK && kv[NK - 1] <= cq2)
                        mb = PENDING;        // a long bucket: searched by k_merge_fixup, where every lane has such a look-up
                      else
                        {
#pragma unroll
                          for (int t = 0; t < NK; t++)
                            if ((uint32_t) t < sz && kv[t] == cq2)
                              { if (mb == 0) lo = bsv + (uint32_t) t;
                                mb += 1;
                              }
                        }
                      if (mb == PENDING)
                        { e = i2; hit = true; }             // (entry {i, 0, 0, PENDING}: no hits counted yet)
                      else if (mb != 0)
                        { e   = (idw & 0x10000u) ? run_end(akeys, na, i2) : i2 + 1;
                          ct  = (unsigned long long) (e - i2) * mb;
                          hit = true;
                        }
                    }
                  if (hit)
                    { local += ct;
                      if (ct < MAXGRAM) below += ct;
                    }
                  const uint64_t hm = __ballot(hit);
                  if (hm)
                    { uint32_t base = 0;
                      const int leader = __ffsll((long long) hm) - 1;
                      if (lane == leader) base = atomicAdd(&s_count, (uint32_t) __popcll(hm));
                      base = __shfl(base, leader);
                      if (hit)
                        { MergeEntry en;
                          en.a0 = (uint32_t) i2;
                          en.ma = (uint32_t) (e - i2);
                          en.b0 = lo;
                          en.mb = mb;
                          dst[base + __popcll(hm & lt)] = en;
                        }
                    }
                  ncand -= take;
                }
              __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
              __builtin_amdgcn_wave_barrier();
            }
        }
    }
  else
  {
#pragma unroll
  for (int j = 0; j < MG_ITEMS; j++)
    { int64_t i = tile0 + j * MG_THREADS;
      unsigned long long ct = 0;
      uint32_t lo = 0, mb = 0;
      int64_t  e = 0;
      bool     hit = false;
      const KT cq = (KT) (c[j] & lowmask);         // what the fetched keys are compared with
      if (first[j] && bs[j] < be[j] && key_at(0, j) <= cq)
        { const uint32_t sz = be[j] - bs[j];
          const uint64_t cc = c[j];
          if (sz > (uint32_t) NK && key_at(NK - 1, j) <= cq)
            mb = PENDING;            // a long bucket: searched by k_merge_fixup, where every lane has such a look-up
          else
            {
#pragma unroll
              for (int t = 0; t < NK; t++)
                if ((uint32_t) t < sz && key_at(t, j) == cq)
                  { if (mb == 0) lo = bs[j] + (uint32_t) t;
                    mb += 1;
                  }
            }
          if (mb == PENDING)
            { e = i; hit = true; }                 // (entry {i, 0, 0, PENDING}: no hits counted yet)
          else if (mb != 0)
            { // does the code go on at i + 1?  That k-mer is the next lane's (same item): it is no `first` then -- a bit of
              // the ballot instead of a dependent load in front of every entry (only the wavefront's last lane has to look)
              bool same;
              if (lane < 63) same = (i + 1 < na) && ((firstm[j] >> (lane + 1)) & 1ull) == 0;
              else           same = (i + 1 < na) && akeys[i + 1] == cc;
              e   = same ? run_end(akeys, na, i) : i + 1;
              ct  = (unsigned long long) (e - i) * mb;
              hit = true;
            }
        }
      if (hit)
        { local += ct;
          if (ct < MAXGRAM) below += ct;
        }
      uint64_t m = __ballot(hit);
      if (m)
        { uint32_t base = 0;
          int leader = __ffsll((long long) m) - 1;
          if (lane == leader) base = atomicAdd(&s_count, (uint32_t) __popcll(m));
          base = __shfl(base, leader);
          if (hit)
            { MergeEntry en;
              en.a0 = (uint32_t) i;
              en.ma = (uint32_t) (e - i);
              en.b0 = lo;
              en.mb = mb;
              dst[base + __popcll(m & ((1ull << lane) - 1ull))] = en;    // (a wavefront's entries are neighbours)
            }
        }
    }
  }
  for (int o = 32; o > 0; o >>= 1)
    { local += __shfl_down(local, o);
      below += __shfl_down(below, o);
    }
  if (lane == 0) { s_hits[0][tid >> 6] = local; s_hits[1][tid >> 6] = below; }
  __syncthreads();
  // Everything a tile found goes to the tile's OWN slots: its entries (the tile's MG_TILE-entry window
  // of tile_entries), their number and its two hit counts.  No global atomic: a device-scope atomic is
  // served at the memory side on this multi-die part at a few ns apiece.  The product histogram of
  // map.c:881-934 is only needed when the hits do not fit the memory limit, and is then made from the
  // entry list (k_entry_hist).  k_merge_gather / k_merge_reduce put the pieces together.
  if (tid == 0)
    { tile_hits[2 * (size_t) tile]     = s_hits[0][0] + s_hits[0][1] + s_hits[0][2] + s_hits[0][3];
      tile_hits[2 * (size_t) tile + 1] = s_hits[1][0] + s_hits[1][1] + s_hits[1][2] + s_hits[1][3];
      tile_count[tile] = s_count;
    }
}

// The look-ups k_merge_match left open: codes that fall into a LONG bucket of the REF index (a repeat's k-mer and its
// copies, or a neighbour of one).  Searching such a bucket is a chain of dependent loads; inside k_merge_match a
// wavefront paid for it whenever one of its 64 x 8 look-ups needed it (most wavefronts on a repeat-rich reference,
// three quarters of the kernel's cycles).  Here the open look-ups of a tile lie side by side in the tile's entry
// window, so the lanes that search are many.  An entry without partners stays in the list with zero size.
constexpr int FX_THREADS = 64;      // one wavefront per tile: what bounds the kernel is how many searches are in flight per CU
__global__ __launch_bounds__(FX_THREADS) void k_merge_fixup(const uint64_t *__restrict__ akeys, int64_t na, BLookup B,
                                                            MergeEntry *__restrict__ tile_entries,
                                                            const uint32_t *__restrict__ tile_count,
                                                            unsigned long long *__restrict__ tile_hits)
{ __shared__ uint16_t pend[MG_TILE];
  const int lane = threadIdx.x;
  const uint32_t cnt = tile_count[blockIdx.x];
  MergeEntry *ent = tile_entries + (size_t) blockIdx.x * MG_TILE;
  unsigned long long local = 0, below = 0;
  // the open look-ups are first listed (they are a minority of the tile's entries), so that the searches -- chains of
  // dependent loads -- run with every lane busy
  uint32_t np = 0;
  for (uint32_t x0 = 0; x0 < cnt; x0 += FX_THREADS)
    { const uint32_t x = x0 + lane;
      const bool p = x < cnt && ent[x].mb == PENDING;
      const uint64_t m = __ballot(p);
      if (p) pend[np + (uint32_t) __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t) x;
      np += (uint32_t) __popcll(m);
    }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (uint32_t y = lane; y < np; y += FX_THREADS)
    { const uint32_t x = pend[y];
      MergeEntry en = ent[x];
      const int64_t  i = en.a0;
      const uint64_t c = akeys[i];
      uint32_t lo = 0, mb = 0;
      b_range(B, c, lo, mb);
      if (mb != 0)
        { const int64_t e = (i + 1 < na && akeys[i + 1] == c) ? run_end(akeys, na, i) : i + 1;
          const unsigned long long ct = (unsigned long long) (e - i) * mb;
          en.ma = (uint32_t) (e - i); en.b0 = lo; en.mb = mb;
          local += ct;
          if (ct < (unsigned long long) MAXGRAM) below += ct;
        }
      else
        { en.ma = 0; en.b0 = 0; en.mb = 0; }
      ent[x] = en;
    }
  for (int o = 32; o > 0; o >>= 1)
    { local += __shfl_down(local, o);
      below += __shfl_down(below, o);
    }
  if (lane == 0)
    { if (local) tile_hits[2 * (size_t) blockIdx.x]     += local;
      if (below) tile_hits[2 * (size_t) blockIdx.x + 1] += below;
    }
}

// tile_count has been scanned in place (exclusive): copy every tile's entries to its place in the list
__global__ __launch_bounds__(256) void k_merge_gather(const MergeEntry *__restrict__ tile_entries,
                                                      const uint32_t *__restrict__ tile_off, int64_t ntiles,
                                                      uint32_t total, MergeEntry *__restrict__ entries)
{ const int wave = (int) (((int64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (wave >= ntiles) return;
  const uint32_t b = tile_off[wave];
  const uint32_t e = (wave + 1 < ntiles) ? tile_off[wave + 1] : total;
  const MergeEntry *src = tile_entries + (size_t) wave * MG_TILE;
  for (uint32_t x = lane; x < e - b; x += 64) entries[b + x] = src[x];
}

// sums the tiles' hit counts: sums[0] = all hits, sums[1] = hits of the runs below MAXGRAM
__global__ __launch_bounds__(256) void k_merge_reduce(const unsigned long long *__restrict__ tile_hits, int64_t ntiles,
                                                      unsigned long long *__restrict__ sums)
{ __shared__ unsigned long long sh[2][4];
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned long long a = 0, b = 0;
  for (int64_t t = (int64_t) blockIdx.x * 256 + tid; t < ntiles; t += (int64_t) gridDim.x * 256)
    { a += tile_hits[2 * t];
      b += tile_hits[2 * t + 1];
    }
  for (int o = 32; o > 0; o >>= 1)
    { a += __shfl_down(a, o);
      b += __shfl_down(b, o);
    }
  if (lane == 0) { sh[0][tid >> 6] = a; sh[1][tid >> 6] = b; }
  __syncthreads();
  if (tid == 0)
    { atomicAdd(&sums[0], sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
      atomicAdd(&sums[1], sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
    }
}

// the product histogram of the matched runs (map.c:926-927), from the entry list -- only when the limit rule needs it
__global__ void k_entry_hist(const MergeEntry *__restrict__ entries, uint32_t n, unsigned long long *__restrict__ gram)
{ uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long ct = (unsigned long long) entries[i].ma * entries[i].mb;
  if (ct < (unsigned long long) MAXGRAM) atomicAdd(&gram[ct], 1ull);
}

// hits each entry will produce under the limit (map.c:978-979)
__global__ void k_entry_sizes(const MergeEntry *__restrict__ entries, uint32_t n, int64_t limit,
                              uint32_t *__restrict__ sizes)
{ uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  MergeEntry e = entries[i];
  unsigned long long ct = (unsigned long long) e.ma * e.mb;
  sizes[i] = ((int64_t) ct < limit) ? (uint32_t) ct : 0u;
}

struct PosMap
{ const int64_t  *boff;
  const uint32_t *chunk;
  int             posbits;      // > 0: the index holds (read << posbits) | rpos already
};

__device__ __forceinline__ void locate(const PosMap &M, uint32_t q, uint32_t &read, uint32_t &rpos)
{ if (M.posbits > 0)
    { read = q >> M.posbits;
      rpos = q & ((1u << M.posbits) - 1u);
      return;
    }
  uint32_t r = M.chunk[q >> CHUNK_SHIFT];
  while (M.boff[r + 1] <= (int64_t) q) r += 1;
  read = r;
  rpos = (uint32_t) ((int64_t) q - M.boff[r]);
}

constexpr uint32_t BIG_ENTRY = 64;

// One hit into the batch.  mode 0: key aread | pass | bread | apos, value bpos (the reference's seed order);
// mode 1: bpos in the key and apos as the value (cluster mode, two arrays); mode >= 2: cluster mode with everything in
// ONE word, aread | bread | bpos | apos with apos in the low mode - 2 bits -- the seed sort then moves 8 bytes per hit.
__device__ __forceinline__ void put_hit(uint64_t *__restrict__ hkeys, uint32_t *__restrict__ hvals, uint32_t o,
                                        uint32_t ar, uint32_t ap, uint32_t br, uint32_t bp, int bshift, int pshift,
                                        int mode, uint64_t passor)
{ const uint64_t top = ((uint64_t) ar << bshift) | passor | ((uint64_t) br << pshift);
  if (mode >= 2)
    hkeys[o] = top | ((uint64_t) bp << (mode - 2)) | (uint64_t) ap;
  else
    { hkeys[o] = top | (uint64_t) (mode ? bp : ap);
      hvals[o] = mode ? ap : bp;
    }
}

// Expand entries into hits (map.c:984-994).  Small entries: one thread each; entries with more
// than BIG_ENTRY pairs are queued for the wavefront-per-entry kernel below.
__global__ void k_merge_expand(const MergeEntry *__restrict__ entries, uint32_t n,
                               const uint32_t *__restrict__ offs, const uint32_t *__restrict__ sizes,
                               const uint32_t *__restrict__ avals, const uint32_t *__restrict__ bvals,
                               PosMap MA, PosMap MB, int bshift, int pshift, int swap, uint64_t passor,
                               uint64_t *__restrict__ hkeys, uint32_t *__restrict__ hvals,
                               uint32_t *__restrict__ biglist, unsigned int *__restrict__ nbig)
{ uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t sz = sizes[i];
  if (sz == 0) return;
  if (sz > BIG_ENTRY)
    { biglist[atomicAdd(nbig, 1u)] = i;
      return;
    }
  MergeEntry e = entries[i];
  uint32_t off = offs[i];
  for (uint32_t a = 0; a < e.ma; a++)
    { uint32_t ar, ap;
      locate(MA, avals[e.a0 + a], ar, ap);
      for (uint32_t b = 0; b < e.mb; b++)
        { uint32_t br, bp;
          locate(MB, bvals[e.b0 + b], br, bp);
          put_hit(hkeys, hvals, off, ar, ap, br, bp, bshift, pshift, swap, passor);
          off += 1;
        }
    }
}

__global__ __launch_bounds__(256) void k_merge_expand_big(const MergeEntry *__restrict__ entries,
                                                          const uint32_t *__restrict__ biglist,
                                                          const unsigned int *__restrict__ nbig_p,
                                                          const uint32_t *__restrict__ offs,
                                                          const uint32_t *__restrict__ avals,
                                                          const uint32_t *__restrict__ bvals,
                                                          PosMap MA, PosMap MB, int bshift, int pshift, int swap,
                                                          uint64_t passor, uint64_t *__restrict__ hkeys,
                                                          uint32_t *__restrict__ hvals)
{ const int lane = threadIdx.x & 63;
  const uint32_t nbig = *nbig_p;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nbig; w += nwaves)
    { uint32_t   i = biglist[w];
      MergeEntry e = entries[i];
      uint32_t off = offs[i];
      uint32_t tot = e.ma * e.mb;
      for (uint32_t t = lane; t < tot; t += 64)
        { uint32_t a = t / e.mb, b = t - a * e.mb;
          uint32_t ar, ap, br, bp;
          locate(MA, avals[e.a0 + a], ar, ap);
          locate(MB, bvals[e.b0 + b], br, bp);
          put_hit(hkeys, hvals, off + t, ar, ap, br, bp, bshift, pshift, swap, passor);
        }
    }
}

// Expansion by OUTPUT tiles: a workgroup writes EX_TILE consecutive hits, consecutive threads consecutive slots
// (the entry-per-thread kernels above write every entry's pairs from one thread: 3 bytes of HBM traffic per byte of
// hits).  The tile's entries are found by two searches in the offsets, their offsets go to LDS, every slot then finds
// its entry there.  Zero-sized entries (runs at or above the limit) simply never contain a slot.
#ifndef DMH_EX_UNROLL
#define DMH_EX_UNROLL 2
#endif
constexpr int EX_TILE = 2048;
constexpr int EX_OFFS = 2 * EX_TILE + 2;

// tfirst[T] = the entry that holds slot T * EX_TILE (every slot lies in exactly one entry of non-zero size)
__global__ void k_tile_first(const uint32_t *__restrict__ eoff, uint32_t nent, uint32_t *__restrict__ tfirst)
{ const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nent) return;
  const uint32_t o0 = eoff[e], o1 = eoff[e + 1];
  if (o1 == o0) return;
  for (uint32_t T = (o0 + (uint32_t) EX_TILE - 1) / (uint32_t) EX_TILE; T <= (o1 - 1) / (uint32_t) EX_TILE; T++) tfirst[T] = e;
}

// rstart != nullptr: the hits go to their READ's segment instead of slot o -- row (entry, a) of the merge owns the
// mb slots from rstart[aread] + rowdst[first slot of the row] (k_row_place)
__global__ __launch_bounds__(256) void k_merge_expand_tiled(const MergeEntry *__restrict__ entries, uint32_t nent,
                                                            const uint32_t *__restrict__ eoff,
                                                            const uint32_t *__restrict__ tfirst,
                                                            const uint32_t *__restrict__ avals,
                                                            const uint32_t *__restrict__ bvals,
                                                            PosMap MA, PosMap MB, int bshift, int pshift, int swap,
                                                            uint64_t passor, uint64_t *__restrict__ hkeys,
                                                            uint32_t *__restrict__ hvals,
                                                            const uint32_t *__restrict__ rstart,
                                                            const uint32_t *__restrict__ rowdst)
{ __shared__ uint32_t s_off[EX_OFFS];
  __shared__ uint32_t s_e[2];
  const uint32_t total = eoff[nent];
  const uint32_t o0 = blockIdx.x * (uint32_t) EX_TILE;
  if (o0 >= total) return;
  const uint32_t o1 = (total - o0 < (uint32_t) EX_TILE) ? total : o0 + (uint32_t) EX_TILE;     // one past the tile's last slot
  if (threadIdx.x < 2)
    { // the entry of the tile's first slot, and an entry at or behind the one of its last slot (the first slot of the
      // next tile lies in it or in a later one)
      s_e[threadIdx.x] = (threadIdx.x == 0) ? tfirst[blockIdx.x] : (o1 < total ? tfirst[blockIdx.x + 1] : nent - 1);
    }
  __syncthreads();
  const uint32_t e_lo = s_e[0], ne = s_e[1] - s_e[0] + 1;
  const bool     inlds = ne + 1 <= (uint32_t) EX_OFFS;
  if (inlds)
    for (uint32_t i = threadIdx.x; i <= ne; i += 256) s_off[i] = eoff[e_lo + i];
  __syncthreads();
  // A thread's EX_TILE / 256 slots in PHASES -- all their entries, then all their positions, then all their places --
  // so that the loads of a phase are in flight together: slot by slot the kernel waited for four dependent round trips
  // per hit (86 % of its wave cycles, at full occupancy).
  constexpr int NK = EX_TILE / 256;
  uint32_t ee[NK], tt[NK];
  MergeEntry en[NK];
  bool ok[NK];
#pragma unroll
  for (int k = 0; k < NK; k++)
    { const uint32_t o = o0 + threadIdx.x + 256u * (uint32_t) k;
      ok[k] = o < o1;
      const uint32_t oc = ok[k] ? o : o1 - 1;
      uint32_t lo = 0, hi = ne;                // last i in [0, ne) with off[i] <= oc
      if (inlds)
        while (hi - lo > 1)
          { const uint32_t m = (lo + hi) >> 1;
            if (s_off[m] <= oc) lo = m; else hi = m;
          }
      else
        while (hi - lo > 1)
          { const uint32_t m = (lo + hi) >> 1;
            if (eoff[e_lo + m] <= oc) lo = m; else hi = m;
          }
      ee[k] = e_lo + lo;
      tt[k] = oc - (inlds ? s_off[lo] : eoff[e_lo + lo]);
    }
#pragma unroll
  for (int k = 0; k < NK; k++) en[k] = entries[ee[k]];
  uint32_t qa[NK], qb[NK], bb[NK];
#pragma unroll
  for (int k = 0; k < NK; k++)
    { const uint32_t a = tt[k] / en[k].mb;
      bb[k] = tt[k] - a * en[k].mb;
      qa[k] = avals[en[k].a0 + a];
      qb[k] = bvals[en[k].b0 + bb[k]];
    }
  uint32_t ar[NK], ap[NK], br[NK], bp[NK], dst[NK];
#pragma unroll
  for (int k = 0; k < NK; k++)
    { locate(MA, qa[k], ar[k], ap[k]);
      locate(MB, qb[k], br[k], bp[k]);
    }
#pragma unroll
  for (int k = 0; k < NK; k++)
    { const uint32_t o = o0 + threadIdx.x + 256u * (uint32_t) k;
      const uint32_t oc = ok[k] ? o : o1 - 1;
      dst[k] = (rstart != nullptr) ? rstart[ar[k]] + rowdst[oc - bb[k]] + bb[k] : oc;
    }
#pragma unroll
  for (int k = 0; k < NK; k++)
    if (ok[k]) put_hit(hkeys, hvals, dst[k], ar[k], ap[k], br[k], bp[k], bshift, pshift, swap, passor);
}

// Rows of the merge: READ k-mer a of an entry with its mb partners.  Every row takes mb consecutive slots of its
// read's segment: rcount[aread] counts the hits of the read, the value before the add is the row's place inside the
// segment, kept at the row's first slot in expansion order (rowdst is as long as the hit list, one word per row used).
// The order of the rows inside a segment is whatever the atomics make it -- the per-read sort does not care.
__global__ void k_row_place(const MergeEntry *__restrict__ entries, uint32_t n, const uint32_t *__restrict__ esz,
                            const uint32_t *__restrict__ eoff, const uint32_t *__restrict__ avals, PosMap MA,
                            uint32_t *__restrict__ rcount, uint32_t *__restrict__ rowdst)
{ const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || esz[i] == 0) return;
  const MergeEntry e = entries[i];
  const uint32_t o = eoff[i];
  for (uint32_t a = 0; a < e.ma; a++)
    { uint32_t ar, ap;
      locate(MA, avals[e.a0 + a], ar, ap);
      rowdst[o + a * e.mb] = atomicAdd(&rcount[ar], e.mb);
    }
}

// Fused route (dmh_readsort.h): the rows of the merge listed PER READ instead of expanded.  Row (entry, a) -- READ k-mer a
// of a matched run with its mb partners -- becomes one 8-byte record in its read's stretch of the row table: the stretch of
// read r starts at boff[r] (a read has fewer rows than bases, so the stretches cannot meet), the place inside it comes
// from an atomic count of the read's rows, and rcount[r] sums the read's hits as k_row_place does.  One wavefront per
// tile of k_merge_match, straight from the tiles' own windows of the entry list (toff = the scanned tile counts), so
// neither the compact entry list nor the entries' offsets are needed: k_merge_gather, k_entry_sizes, the scan over the
// entries, k_tile_first and k_merge_expand_tiled of round 3's route do not run.  [r0, r1): the reads whose rows are
// wanted (a pass of more hits than 32-bit slots is made range by range).
__global__ __launch_bounds__(256) void k_row_list(const MergeEntry *__restrict__ tile_entries,
                                                  const uint32_t *__restrict__ toff, int64_t ntiles, uint32_t total,
                                                  int64_t limit, const uint32_t *__restrict__ avals, int aposbits,
                                                  const int64_t *__restrict__ boff, uint32_t *__restrict__ rcount,
                                                  uint32_t *__restrict__ nrow, uint64_t *__restrict__ rows,
                                                  uint32_t r0, uint32_t r1)
{ const int64_t wave = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= ntiles) return;
  const uint32_t b = toff[wave];
  const uint32_t e = (wave + 1 < ntiles) ? toff[wave + 1] : total;
  const MergeEntry *src = tile_entries + (size_t) wave * MG_TILE;
  const uint32_t apm = (1u << aposbits) - 1u;
  for (uint32_t x = lane; x < e - b; x += 64)
    { const MergeEntry en = src[x];
      const unsigned long long ct = (unsigned long long) en.ma * en.mb;
      if (ct == 0 || (int64_t) ct >= limit) continue;          // (map.c:978-979; an open look-up without partners has mb = 0)
      for (uint32_t a = 0; a < en.ma; a++)
        { const uint32_t q  = avals[en.a0 + a];
          const uint32_t ar = q >> aposbits, ap = q & apm;
          if (ar < r0 || ar >= r1) continue;                  // (a pass split into ranges of reads: this range's rows only)
          const uint32_t at = atomicAdd(&nrow[ar], 1u);
          atomicAdd(&rcount[ar], en.mb);
          if (rows != nullptr)                                // (nullptr: the hits per read are only counted)
            rows[boff[ar] + at] = ((uint64_t) en.b0 << 32) | ((uint64_t) ap << RW_MB) | (uint64_t) en.mb;
        }
    }
}

// ... and the few reads whose hits do not fit the LDS of a workgroup (k_read_sort_big orders them through global
// memory): their rows expanded into words at the read's segment, one workgroup per read of the list.
__global__ __launch_bounds__(256) void k_rows_expand_big(const uint32_t *__restrict__ list, const uint32_t *__restrict__ count,
                                                         const uint64_t *__restrict__ rows, const int64_t *__restrict__ rowbase,
                                                         const uint32_t *__restrict__ nrow, const uint32_t *__restrict__ rstart,
                                                         const uint32_t *__restrict__ bvals, int bposbits, int pbits, int qbits,
                                                         int bshift, uint64_t *__restrict__ hkeys)
{ __shared__ uint32_t ws[4];
  const uint32_t bpm = (1u << bposbits) - 1u;
  for (uint32_t lb = blockIdx.x; lb < *count; lb += gridDim.x)
    { const uint32_t r = list[lb], nr = nrow[r], s0 = rstart[r];
      const uint64_t *rw = rows + rowbase[r];
      uint32_t done = 0;                                        // hits of the rows before this chunk
      for (uint32_t x0 = 0; x0 < nr; x0 += 256)
        { const uint32_t x = x0 + threadIdx.x;
          const uint64_t rc = (x < nr) ? rw[x] : 0ull;
          const uint32_t mb = (uint32_t) (rc & RW_MBMASK);
          uint32_t all;
          const uint32_t o = done + block_excl_scan(mb, ws, &all);
          const uint64_t lo = (uint64_t) r << bshift | ((rc >> RW_MB) & RW_APMASK);
          for (uint32_t t = 0; t < mb; t++)
            { const uint32_t q = bvals[(uint32_t) (rc >> 32) + t];
              hkeys[s0 + o + t] = lo | ((uint64_t) (q >> bposbits) << (pbits + qbits)) | ((uint64_t) (q & bpm) << pbits);
            }
          done += all;
        }
      __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
//  Group boundaries of the sorted hits: one group per (aread, bread)
// ---------------------------------------------------------------------------------------

__global__ void k_group_flag(const uint64_t *__restrict__ hkeys, int64_t n, int pshift, uint32_t *__restrict__ flag)
{ int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flag[i] = (i == 0 || (hkeys[i] >> pshift) != (hkeys[i - 1] >> pshift)) ? 1u : 0u;
}

__global__ void k_group_write(const uint64_t *__restrict__ hkeys, int64_t n, int pshift,
                              const uint32_t *__restrict__ slot, uint32_t *__restrict__ gstart, uint32_t ngroups,
                              uint64_t *__restrict__ gkey)
{ int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (i == 0) gstart[ngroups] = (uint32_t) n;           // the end of the last group
  if (i == 0 || (hkeys[i] >> pshift) != (hkeys[i - 1] >> pshift))
    { gstart[slot[i]] = (uint32_t) i;
      gkey[slot[i]]   = hkeys[i] >> pshift;
    }
}

// ---------------------------------------------------------------------------------------
//  B-clusters (cluster mode): the hits arrive sorted (aread, bread, bpos) with bpos in the low qbits of the key
//  and apos as the value.  A cluster starts where the pair changes or the gap to the previous bpos exceeds
//  MAX_GAP.  k_cluster_rekey then gives every hit the key (cluster << pbits) | apos and the value bpos: the
//  second sort puts each cluster into (apos, bpos) order without moving a hit out of its cluster's range.
// ---------------------------------------------------------------------------------------

// The first sort only orders the hits of a pair by bpos >> cb (cb = CLUSTER_CB: buckets of 4096 bases, which saves
// two radix passes): a cluster is a maximal run of hits whose buckets are equal or adjacent -- a union of the
// MAX_GAP-connected sets, which is all the chain kernels need.  Inside a bucket the hits keep the order the merge
// emitted them in, i.e. for equal (aread, apos) ascending bpos (one READ k-mer's partners in REF-index order); the
// second sort, stable on (cluster, apos), therefore ends in (apos, bpos) order.
constexpr int CLUSTER_CB = 12;

__device__ __forceinline__ bool cluster_start(uint64_t a, uint64_t b, int qbits, int cb)
{ const uint64_t qm = (1ull << qbits) - 1ull;
  return (a >> qbits) != (b >> qbits) || ((b & qm) >> cb) - ((a & qm) >> cb) > 1ull;
}

}  // namespace dmh
#include "dmh_readsort.h"
namespace dmh {

__global__ void k_cluster_flag(const uint64_t *__restrict__ hkeys, int64_t n, int qbits, int cb, uint32_t *__restrict__ flag)
{ int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flag[i] = (i == 0 || cluster_start(hkeys[i - 1], hkeys[i], qbits, cb)) ? 1u : 0u;
}

// slot[] = exclusive scan of the flags: hit i belongs to cluster slot[i] + flag[i] - 1.  Out of place (the
// neighbour's key is read): (hkeys, hvals) -> (okeys, ovals).
__global__ void k_cluster_rekey(const uint64_t *__restrict__ hkeys, const uint32_t *__restrict__ hvals, int64_t n,
                                int qbits, int cb, int pbits, int packed, const uint32_t *__restrict__ slot,
                                uint64_t *__restrict__ okeys, uint32_t *__restrict__ ovals,
                                uint32_t *__restrict__ gstart, uint32_t ncl, uint64_t *__restrict__ gkey)
{ int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = hkeys[i];
  const uint64_t qm = (1ull << qbits) - 1ull;
  const bool start = (i == 0) || cluster_start(hkeys[i - 1], k, qbits, cb);
  const uint32_t c = slot[i] + (start ? 1u : 0u) - 1u;
  if (packed)          // the chain kernels' word: apos | bpos << 32 (the cluster is known from gstart)
    okeys[i] = (uint64_t) hvals[i] | ((k & qm) << 32);
  else                 // for the global cluster sort: (cluster, apos) key, bpos value
    { okeys[i] = ((uint64_t) c << pbits) | (uint64_t) hvals[i];
      ovals[i] = (uint32_t) (k & qm);
    }
  if (start)
    { gstart[c] = (uint32_t) i;
      gkey[c]   = k >> qbits;
    }
  if (i == 0) gstart[ncl] = (uint32_t) n;
}

// ---------------------------------------------------------------------------------------
//  One-word hits (put_hit mode >= 2): w = aread | bread | bpos | apos, sorted on the bits from pbits + cb up.  w >> pbits
//  is the key the kernels above look at.  Two passes over the words instead of flag / scan / rekey over keys, values
//  and a flag per hit: the cluster starts are counted per tile of WC_TILE hits, the tile counts are scanned (a few ten
//  thousand numbers), and the second pass numbers the clusters inside its tile and writes the chain kernels' word.
// ---------------------------------------------------------------------------------------
constexpr int WC_CHUNKS = 4;                       // a wavefront takes WC_CHUNKS x 128 consecutive words, two per lane and chunk
constexpr int WC_WAVE   = WC_CHUNKS * 128;
constexpr int WC_TILE   = 4 * WC_WAVE;
typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));

// The wavefront's words (lane l holds words 2l, 2l+1 of each chunk: every load instruction covers 1 KB without gaps;
// the arrays start on a 16-byte boundary) and the start flags of the lane's two words per chunk (bit 0, bit 1).
__device__ __forceinline__ void wc_load(const uint64_t *__restrict__ w, int64_t n, int64_t wbase, int lane, int pbits,
                                        int qbits, int cb, uint64_t (&x)[WC_CHUNKS][2], uint32_t (&f)[WC_CHUNKS])
{
#pragma unroll
  for (int c = 0; c < WC_CHUNKS; c++)
    { const int64_t i = wbase + c * 128 + 2 * lane;
      if (i + 2 <= n)
        { const ull2 t = __builtin_nontemporal_load((const ull2 *) (w + i));
          x[c][0] = t.x; x[c][1] = t.y;
        }
      else
        { x[c][0] = (i < n) ? w[i] : 0;
          x[c][1] = 0;
        }
    }
  uint64_t carry = (wbase > 0 && wbase < n) ? w[wbase - 1] : 0;        // the word before the chunk's first
#pragma unroll
  for (int c = 0; c < WC_CHUNKS; c++)
    { const int64_t i = wbase + c * 128 + 2 * lane;
      uint64_t prev = ((uint64_t) (uint32_t) __shfl_up((int) (uint32_t) (x[c][1] >> 32), 1) << 32)
                      | (uint64_t) (uint32_t) __shfl_up((int) (uint32_t) x[c][1], 1);
      if (lane == 0) prev = carry;
      uint32_t fl = 0;
      if (i < n && (i == 0 || cluster_start(prev >> pbits, x[c][0] >> pbits, qbits, cb))) fl |= 1u;
      if (i + 1 < n && cluster_start(x[c][0] >> pbits, x[c][1] >> pbits, qbits, cb)) fl |= 2u;
      f[c] = fl;
      carry = ((uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (x[c][1] >> 32), 63) << 32)
              | (uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) x[c][1], 63);
    }
}

__global__ __launch_bounds__(256) void k_wcluster_count(const uint64_t *__restrict__ w, int64_t n, int pbits, int qbits,
                                                        int cb, uint32_t *__restrict__ tcount)
{ __shared__ uint32_t ws[4];
  uint64_t x[WC_CHUNKS][2];
  uint32_t f[WC_CHUNKS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  wc_load(w, n, (int64_t) blockIdx.x * WC_TILE + (int64_t) wv * WC_WAVE, lane, pbits, qbits, cb, x, f);
  uint32_t c = 0;
#pragma unroll
  for (int k = 0; k < WC_CHUNKS; k++) c += (uint32_t) __popc(f[k]);
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if (lane == 0) ws[wv] = c;
  __syncthreads();
  if (threadIdx.x == 0) tcount[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// toff = exclusive scan of the tile counts.  sab[i] = apos | bpos << 32; gstart / gkey per cluster.
__global__ __launch_bounds__(256) void k_wcluster_rekey(const uint64_t *__restrict__ w, int64_t n, int pbits, int qbits,
                                                        int cb, const uint32_t *__restrict__ toff,
                                                        uint64_t *__restrict__ sab, uint32_t *__restrict__ gstart,
                                                        uint32_t ncl, uint64_t *__restrict__ gkey)
{ __shared__ uint32_t ws[4];
  uint64_t x[WC_CHUNKS][2];
  uint32_t f[WC_CHUNKS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t lt = (1ull << lane) - 1ull;
  const int64_t wbase = (int64_t) blockIdx.x * WC_TILE + (int64_t) wv * WC_WAVE;
  wc_load(w, n, wbase, lane, pbits, qbits, cb, x, f);
  // starts before the lane's first word of each chunk, inside the wavefront
  uint32_t before[WC_CHUNKS], run = 0;
#pragma unroll
  for (int k = 0; k < WC_CHUNKS; k++)
    { const uint64_t b0 = __ballot(f[k] & 1u), b1 = __ballot(f[k] & 2u);
      before[k] = run + (uint32_t) __popcll(b0 & lt) + (uint32_t) __popcll(b1 & lt);
      run += (uint32_t) __popcll(b0) + (uint32_t) __popcll(b1);
    }
  if (lane == 0) ws[wv] = run;
  __syncthreads();
  uint32_t cbase = toff[blockIdx.x];
  for (int i = 0; i < wv; i++) cbase += ws[i];
  const uint64_t pm = (1ull << pbits) - 1ull, qm = (1ull << qbits) - 1ull;
  const int top = pbits + qbits;
#pragma unroll
  for (int k = 0; k < WC_CHUNKS; k++)
    { const int64_t i = wbase + k * 128 + 2 * lane;
      uint32_t c = cbase + before[k];
      if (f[k] & 1u)
        { gstart[c] = (uint32_t) i;
          gkey[c]   = x[k][0] >> top;
          c += 1;
        }
      if (f[k] & 2u)
        { gstart[c] = (uint32_t) (i + 1);
          gkey[c]   = x[k][1] >> top;
        }
      ull2 t;
      t.x = (x[k][0] & pm) | (((x[k][0] >> pbits) & qm) << 32);
      t.y = (x[k][1] & pm) | (((x[k][1] >> pbits) & qm) << 32);
      if (i + 2 <= n) *(ull2 *) (sab + i) = t;
      else if (i < n) sab[i] = t.x;
    }
  if (blockIdx.x == 0 && threadIdx.x == 0) gstart[ncl] = (uint32_t) n;
}

// ---------------------------------------------------------------------------------------
//  Every cluster into (apos, bpos) order without another trip of ALL hits through the global radix sort: the hits
//  of a cluster are contiguous after the first sort, and clusters are small (some twenty hits; a read's true locus a
//  few hundred), so each is sorted where it lies -- by rank counting in a 16-lane group (<= CS_SMALL hits), by a
//  bitonic sort of (apos, position) words in the LDS of a workgroup (<= CS_MEDIUM), and the few larger ones (repeat
//  piles) by the radix sort on their own range.  Stable in apos, so equal apos keep ascending bpos.
// ---------------------------------------------------------------------------------------
constexpr int CS_SMALL  = 64;
constexpr int CS_MEDIUM = 4096;

// ranks of the group's T x 16 words among the n the group holds: word = apos << 6 | position (distinct), the LDS copy
// is padded with all-ones words to a multiple of four and read four words at a time
template <int T>
__device__ __forceinline__ void csort_ranks(const uint32_t *vv, uint32_t n, const uint32_t (&v)[4], uint32_t (&r)[4])
{ const uint4 *v4 = (const uint4 *) vv;
  const uint32_t n4 = (n + 3) >> 2;
#pragma unroll
  for (int t = 0; t < 4; t++) r[t] = 0;
  for (uint32_t jj = 0; jj < n4; jj++)
    { const uint4 b = v4[jj];
#pragma unroll
      for (int t = 0; t < T; t++)
        r[t] += (b.x < v[t] ? 1u : 0u) + (b.y < v[t] ? 1u : 0u) + (b.z < v[t] ? 1u : 0u) + (b.w < v[t] ? 1u : 0u);
    }
}

// A 16-lane group per cluster, four clusters per wavefront, walking the clusters with a fixed stride.  The walk is a
// software pipeline: while a cluster is ranked, the hits of the group's next cluster and the bounds of the one after
// are already on their way (a cluster is some 160 bytes: without that every step waits a full memory round trip).
__global__ __launch_bounds__(256) void k_csort_small(const uint64_t *__restrict__ kin,
                                                     const uint32_t *__restrict__ gstart, uint32_t ncl,
                                                     uint64_t *__restrict__ kout)
{ constexpr int G = 16, GPB = 256 / G, PER = CS_SMALL / G;
  static_assert(PER == 4, "four words per lane");
  __shared__ __attribute__((aligned(16))) uint32_t s_v[GPB][CS_SMALL];
  const uint32_t lane = threadIdx.x & (G - 1);
  uint32_t *vv = s_v[threadIdx.x / G];
  const uint64_t stride = (uint64_t) gridDim.x * GPB;
  uint64_t c = (uint64_t) blockIdx.x * GPB + threadIdx.x / G;

  uint32_t g0 = 0, n = 0, g1 = 0, n1 = 0;          // bounds of cluster c and of c + stride (n = 0: none / not ours)
  if (c < ncl)          { g0 = gstart[c];          n  = gstart[c + 1] - g0; }
  if (c + stride < ncl) { g1 = gstart[c + stride]; n1 = gstart[c + stride + 1] - g1; }
  if (n  > (uint32_t) CS_SMALL) n  = 0;
  if (n1 > (uint32_t) CS_SMALL) n1 = 0;
  uint64_t k[PER], kn[PER];
#pragma unroll
  for (int t = 0; t < PER; t++) k[t] = (lane + G * t < n) ? kin[g0 + lane + G * t] : 0;

  // (a wavefront waits for ALL its outstanding memory operations at once, stores included: the sorted hits of a cluster
  // are written at the head of the NEXT step, ahead of that step's loads, so that the wait for those loads at the
  // step's end finds everything a whole ranking old)
  uint64_t pk[PER];
  uint32_t pr[PER], pg0 = 0, pn = 0;
#pragma unroll
  for (int t = 0; t < PER; t++) { pk[t] = 0; pr[t] = 0; }
  // (the loads above have arrived before the loop is entered: left pending, they would make the compiler wait at the
  // first use of k, g1 and n1 INSIDE the loop, i.e. behind the loads of every step)
  __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
  while (__any(c < ncl))
    {
#pragma unroll
      for (int t = 0; t < PER; t++)
        if (lane + G * t < pn) kout[pg0 + pr[t]] = pk[t];
      // hits one step ahead, bounds two steps ahead (their difference is taken at the step's end: here it would wait)
#pragma unroll
      for (int t = 0; t < PER; t++) kn[t] = (lane + G * t < n1) ? kin[g1 + lane + G * t] : 0;
      uint32_t g2 = 0, e2 = 0;
      if (c + 2 * stride < ncl) { g2 = gstart[c + 2 * stride]; e2 = gstart[c + 2 * stride + 1]; }
      asm volatile("" ::: "memory");

      uint32_t v[PER], r[PER];
#pragma unroll
      for (int t = 0; t < PER; t++)
        { const uint32_t i = lane + G * t;
          v[t] = (i < n) ? ((((uint32_t) k[t]) << 6) | i) : 0xffffffffu;
          vv[i] = v[t];
        }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (!__any(n > 16u))      csort_ranks<1>(vv, n, v, r);
      else if (!__any(n > 32u)) csort_ranks<2>(vv, n, v, r);
      else                      csort_ranks<4>(vv, n, v, r);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < PER; t++) { pk[t] = k[t]; pr[t] = r[t]; }
      pg0 = g0; pn = n;

      c += stride;
      g0 = g1; n = n1;
      g1 = g2; n1 = (e2 - g2 > (uint32_t) CS_SMALL) ? 0 : e2 - g2;
#pragma unroll
      for (int t = 0; t < PER; t++) k[t] = kn[t];
    }
#pragma unroll
  for (int t = 0; t < PER; t++)
    if (lane + G * t < pn) kout[pg0 + pr[t]] = pk[t];
}

// clusters of CS_SMALL+1 .. CS_MEDIUM hits -> mlist, larger ones -> blist
__global__ __launch_bounds__(1024) void k_csort_classify(const uint32_t *__restrict__ gstart, uint32_t ncl,
                                                         uint32_t *__restrict__ mlist, uint32_t *__restrict__ blist,
                                                         unsigned int *__restrict__ counts)
{ // one pair of device atomics per 1024 clusters (they are served one after the other, a few ns apiece)
  __shared__ uint32_t wm[16], wb[16], base[2];
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  const int      lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint32_t n = (c < ncl) ? gstart[c + 1] - gstart[c] : 0;
  const bool ism = n > (uint32_t) CS_SMALL && n <= (uint32_t) CS_MEDIUM, isb = n > (uint32_t) CS_MEDIUM;
  const uint64_t mm = __ballot(ism), mb = __ballot(isb);
  if (lane == 0) { wm[wv] = (uint32_t) __popcll(mm); wb[wv] = (uint32_t) __popcll(mb); }
  __syncthreads();
  if (threadIdx.x == 0)
    { uint32_t tm = 0, tb = 0;
      for (int i = 0; i < 16; i++)
        { const uint32_t a = wm[i], b2 = wb[i];
          wm[i] = tm; wb[i] = tb;
          tm += a; tb += b2;
        }
      base[0] = tm ? atomicAdd(counts, tm) : 0;
      base[1] = tb ? atomicAdd(counts + 1, tb) : 0;
    }
  __syncthreads();
  if (ism) mlist[base[0] + wm[wv] + (uint32_t) __popcll(mm & lt)] = c;
  if (isb) blist[base[1] + wb[wv] + (uint32_t) __popcll(mb & lt)] = c;
}

// one workgroup per listed cluster: bitonic sort of the words (apos << 12 | position) in LDS (apos < 2^20), then the
// hits are moved in that order.  Launched twice over the same list: single wavefronts take the clusters of up to
// CS_WAVE hits (a true locus of a read: a few hundred hits; the barriers of the network then cost nothing), workgroups
// of 256 the rest.
constexpr int CS_WAVE = 512;

__global__ __launch_bounds__(256) void k_csort_medium(const uint64_t *__restrict__ kin,
                                                      const uint32_t *__restrict__ gstart, const uint32_t *__restrict__ list,
                                                      const unsigned int *__restrict__ nlist_p,
                                                      uint64_t *__restrict__ kout, uint32_t nlo, uint32_t nhi)
{ extern __shared__ uint32_t it[];              // nhi words (a power of two)
  const uint32_t nlist = *nlist_p, T = blockDim.x;
  for (uint32_t x = blockIdx.x; x < nlist; x += gridDim.x)
    { const uint32_t c = list[x], g0 = gstart[c], n = gstart[c + 1] - g0;
      if (n <= nlo || n > nhi) continue;        // (uniform in the workgroup)
      uint32_t m = 1;
      while (m < n) m <<= 1;
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < m; i += T)
        it[i] = (i < n) ? ((((uint32_t) kin[g0 + i]) << 12) | i) : 0xffffffffu;
      __syncthreads();
      for (uint32_t k = 2; k <= m; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1)
          { for (uint32_t i = threadIdx.x; i < m; i += T)
              { const uint32_t l = i ^ j;
                if (l > i)
                  { const uint32_t a = it[i], b = it[l];
                    const bool up = ((i & k) == 0);
                    if ((a > b) == up) { it[i] = b; it[l] = a; }
                  }
              }
            __syncthreads();
          }
      for (uint32_t r = threadIdx.x; r < n; r += T)
        { const uint32_t i = it[r] & 0xfffu;
          kout[g0 + r] = kin[g0 + i];
        }
    }
}

// The read-sorted route (dmh_readsort.h) leaves the clusters of more than RS_IN hits in the order of their coarse
// buckets: here each is put into (apos, bpos) order where it lies -- a bitonic sort of the words with their halves
// swapped (apos << 32 | bpos) in LDS, in place.  Launched like k_csort_medium: wavefronts for the clusters of up to
// CS_WAVE hits, workgroups for the rest up to CS_MEDIUM.
__global__ __launch_bounds__(256) void k_csort_wide(uint64_t *__restrict__ ab, const uint32_t *__restrict__ gstart,
                                                    const uint32_t *__restrict__ list, const unsigned int *__restrict__ nlist_p,
                                                    uint32_t nlo, uint32_t nhi)
{ extern __shared__ __attribute__((aligned(16))) uint64_t wit[];       // nhi words (a power of two)
  const uint32_t nlist = *nlist_p, T = blockDim.x;
  for (uint32_t x = blockIdx.x; x < nlist; x += gridDim.x)
    { const uint32_t c = list[x], g0 = gstart[c], n = gstart[c + 1] - g0;
      if (n <= nlo || n > nhi) continue;        // (uniform in the workgroup)
      uint32_t m = 1;
      while (m < n) m <<= 1;
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < m; i += T)
        { const uint64_t v = (i < n) ? ab[g0 + i] : 0;
          wit[i] = (i < n) ? ((v << 32) | (v >> 32)) : ~0ull;
        }
      __syncthreads();
      for (uint32_t k = 2; k <= m; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1)
          { for (uint32_t i = threadIdx.x; i < m; i += T)
              { const uint32_t l = i ^ j;
                if (l > i)
                  { const uint64_t a = wit[i], b = wit[l];
                    const bool up = ((i & k) == 0);
                    if ((a > b) == up) { wit[i] = b; wit[l] = a; }
                  }
              }
            __syncthreads();
          }
      for (uint32_t r = threadIdx.x; r < n; r += T)
        { const uint64_t v = wit[r];
          ab[g0 + r] = (v << 32) | (v >> 32);
        }
    }
}

// (key with apos in its low pbits, value bpos) -> the chain kernels' word
__global__ void k_pack_ab(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, int64_t n, uint64_t pmask,
                          uint64_t *__restrict__ ab)
{ int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ab[i] = (keys[i] & pmask) | ((uint64_t) vals[i] << 32);
}

// pairs of the clusters: flag[c] = cluster c opens a new (aread, bread) pair
__global__ void k_pair_flag(const uint64_t *__restrict__ gkey, uint32_t ncl, uint32_t *__restrict__ flag,
                            uint32_t *__restrict__ pmax)
{ uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncl) return;
  flag[c] = (c == 0 || gkey[c] != gkey[c - 1]) ? 1u : 0u;
  pmax[c] = 0;
}

// cog[] = exclusive scan of those flags on entry; on exit cog[c] = pair of cluster c, ocfirst[p] = first cluster of
// pair p (ocfirst[npairs] = ncl), pmax[p] = largest apos + 1 among the hits of pair p (the apos of the pair's last hit
// in the reference's order; every cluster is in apos order by now, so its last hit holds its largest).  The clusters of
// a pair are neighbours: their maximum is taken inside the wavefront first, one atomic per pair and wavefront.
__global__ __launch_bounds__(256) void k_pair_max(const uint64_t *__restrict__ gkey, uint32_t ncl, uint32_t *__restrict__ cog,
                                                  const uint32_t *__restrict__ npairs_p, uint32_t *__restrict__ ocfirst,
                                                  const uint64_t *__restrict__ sab, const uint32_t *__restrict__ gstart,
                                                  uint32_t *__restrict__ pmax)
{ const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool ok = c < ncl;
  uint32_t p = 0xffffffffu, a = 0;
  if (ok)
    { const bool start = (c == 0 || gkey[c] != gkey[c - 1]);
      p = cog[c] + (start ? 1u : 0u) - 1u;
      a = (uint32_t) sab[gstart[c + 1] - 1] + 1u;
      cog[c] = p;
      if (start) ocfirst[p] = c;
      if (c == 0) ocfirst[*npairs_p] = ncl;
    }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1)
    { const uint32_t pa = __shfl_up(a, o), pp = __shfl_up(p, o);
      if (lane >= o && pp == p && pa > a) a = pa;
    }
  const uint32_t pn = __shfl_down(p, 1);
  if (ok && (lane == 63 || pn != p)) atomicMax(&pmax[p], a);
}

__global__ void k_pair_glast(const uint32_t *__restrict__ cog, uint32_t ncl, const uint32_t *__restrict__ pmax,
                             uint32_t *__restrict__ glast)
{ uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < ncl) glast[c] = pmax[cog[c]];
}

// ---------------------------------------------------------------------------------------
//  Host orchestration
// ---------------------------------------------------------------------------------------

static int ceil_log2(int64_t x)      // smallest b with 2^b >= x
{ int b = 0;
  while (((int64_t) 1 << b) < x) b += 1;
  return b;
}

// DB.c:1044-1076 restated (no QV pseudo-track: damapper never loads one).  A track whose size field
// says 4 has its int64 offsets read as ints there, exactly as the reference does for the merged
// mask (damapper.c:262 sets size = sizeof(int) on an int64 vector).
static int64_t sizeof_db(const DAZZ_DB *db)
{ int64_t s = 112 + 40 * ((int64_t) db->nreads + 2) + (int64_t) strlen(db->path) + 1
              + (db->totlen + db->nreads + 4);
  for (const DAZZ_TRACK *t = (const DAZZ_TRACK *) db->tracks; t != nullptr; t = t->next)
    { s += (int64_t) sizeof(DAZZ_TRACK) + (int64_t) strlen(t->name) + 1 + (int64_t) t->size * (db->nreads + 1);
      if (t->data != nullptr)
        { if (t->size == 8)
            s += 4 * ((const int64_t *) t->anno)[db->nreads];
          else
            s += 4 * (int64_t) ((const int *) t->anno)[db->nreads];
        }
    }
  return s;
}

// The batches: one being collected (ring[cur]) and up to IN_FLIGHT submitted ones whose chain kernels -- repeat piles
// above all -- are still running while later passes are merged, sorted and chained.  They are finished in order.
constexpr int IN_FLIGHT = 8;
constexpr int NRING = IN_FLIGHT + 3;       // (+ the one being collected, the one being finished, and the one a pass has
                                           //  just submitted while the oldest waits for the next pass's match kernels)
static MergeResult g_ring[NRING];
static int         g_cur = 0;
static std::deque<int> g_inflight;

MergeResult &filter_batch() { return g_ring[g_cur]; }

// k_read_sort ranks by the values LDS atomic adds return: true when this device serves them in lane order (it does on
// gfx950; checked once, some 200,000 adds over conflict patterns from one to 127 counters in play)
bool lds_atomics_ordered()
{ static int known = -1;
  if (known < 0)
    { static const bool off = getenv("DMH_NO_LDS_ORDER") != nullptr;
      unsigned int bad = 1;
      if (!off)
        { DevBuf<uint32_t> d;
          d.ensure(4, "LDS order test");
          HIP_CHECK(hipMemsetAsync(d.p, 0, 4 * sizeof(uint32_t), stream()));
          for (uint32_t seed = 1; seed <= 3; seed++)
            hipLaunchKernelGGL(k_lds_order_test, dim3(28), dim3(256), 0, stream(), seed, d.p);
          HIP_CHECK(hipGetLastError());
          copy_d2h(&bad, d.p, sizeof(unsigned int));
          d.release();
        }
      known = (bad == 0) ? 1 : 0;
      if (getenv("DMH_STATS")) fprintf(stderr, "dmh: LDS atomics served in lane order: %s\n", known ? "yes" : "no (ballot ranks)");
    }
  return known == 1;
}

// the per-read route's coarse buckets: 2^RS_CB bases (DMH_RS_CB: experiments)
static int rs_coarse_bits()
{ static const int v = getenv("DMH_RS_CB") ? std::max(8, std::min(24, atoi(getenv("DMH_RS_CB")))) : RS_CB;
  return v;
}

// per-read route: segment starts of the pass being collected (main stream only; a dense pass is closed at once)
static DevBuf<uint64_t> g_rows;          // fused route: the row table of the pass being collected (read by its per-read kernel,
                                         //   which is queued before the next pass lists its rows)
static DevBuf<uint32_t> g_rowdst;        // (the other per-read-sort buffers belong to the batch: MergeResult::rs_*, its tail runs
                                         //  on a stream of its own next to the following pass)
static uint32_t g_known[5] = {0, 0, 0, 0, 0};     // clusters, clusters for the radix sort, big groups, piles, hits in piles (k_read_sort)
static bool     g_have_known = false;
static uint32_t g_rcnt[5] = {0, 0, 0, 0, 0}, g_rlim = 0;   // reads per k_read_sort workgroup size, largest segment, reads above the
                                                            // limit (fetched for statistics and experiments only)
static bool     g_rs_small = false;                        // this pass: no 1024-thread workgroups (see k_read_classes' limits)
static uint32_t g_rgrid[4] = {0, 0, 0, 0};                  // the most reads each size class can hold: the grids
constexpr int RS_T0 = 256, RS_T1 = 512, RS_T2 = 1024;      // k_read_sort workgroups: segments of up to 4,096 / 8,192 / 16,384 hits

// hits a batch may collect before it is closed (68 B of HBM per hit through the sort and the chain kernels)
constexpr int64_t BATCH_CAP = (int64_t) 160 << 20;

static void    close_batch(MergeResult &R, int phase = 0);
static void    close_pending();
static int     g_pending_close = -1;       // ring slot whose close stopped at the cluster-count copy (close_batch phase 1)
static bool    g_closing = false;          // inside the second half of such a close
static int     g_closing_slot = -1;        // ... and the ring slot it works on (busy for filter_reclaim: it is in no list meanwhile)
static int64_t finish_oldest();
static void    finish_tick_if_due(int64_t *nfilt);
static bool    finish_tick(int64_t *nfilt);
static int     flight_keep();
static void    finish_pipeline(int64_t *nfilt);
static bool    filter_reclaim();
static bool    g_finish_in_close = false;    // this pass's share of finishing is done behind its per-read kernels (close_batch)
static int     g_flight_cap = IN_FLIGHT;     // batches that may be in flight: fewer after every time memory ran out, one more
                                             //   again after every READS block that went through without (filter_flush_block)
static bool    g_reclaimed_in_block = false; // memory ran out since the last READS block was flushed

// cluster mode: every cluster sorted where it lies (k_csort_*) rather than all hits through the radix sort again
static bool local_cluster_sort(int pbits)
{ static const bool global_csort = getenv("DMH_GLOBAL_CLUSTER_SORT") != nullptr;
  return !(global_csort || S().opt_cluster_sort == 0 || pbits > 20);
}

// The front of a pass -- the REF index's bucket table (first use), match + count (map.c:2984-2988), the open look-ups,
// the hit sums and the scan over the tiles' entry counts: everything up to the first round trip.  It depends on the two
// indices only and leaves its results in buffers no later stage of the PREVIOUS pass reads, so the filter may queue
// it early: merge_pass does at its start, or -- when the driver has said which pass comes next (dmh_next_pass) --
// close_batch does for the FOLLOWING pass, in front of its wait for the cluster counters: the main stream then goes
// straight on while the host queues the second half of the pass on the tail stream (0.5 ms).
// Two sets of the front's results: when the driver has said which pass comes next (dmh_next_pass) and that pass meets the
// same READ index -- the complement of the REF block at hand --, ONE sweep of the READ keys probes both REF indices
// (k_merge_match<.., TWO>), the second table's results wait in set 1 for the call that follows.
struct FrontBufs
{ DevBuf<unsigned long long> gram, thits;
  DevBuf<MergeEntry> tentries;
  DevBuf<uint32_t>   tcount;
  DevBuf<unsigned int> mctl;
  const IndexDev *a = nullptr, *b = nullptr;                            // the pass whose front is queued in this set
  unsigned long long hsum[3] = {0, 0, 0};                               // its hit sums and run count, when the round trip of the
  bool have_sums = false;                                               //   sweep's own pass has fetched them along
  // a front made AHEAD, on the front stream (round 4): `made` is recorded there behind the front and the copy of its sums
  // to h_sums; the main stream and the host wait for it when the pass comes.  `consumed` is recorded on the main stream
  // behind the last kernel of a pass that reads the set: whoever fills the set again waits for it.
  hipEvent_t made = nullptr, consumed = nullptr;
  bool ahead = false, posted = false, was_made = false, was_used = false;
  unsigned long long *h_sums = nullptr;                                 // page-locked: {all hits, hits below MAXGRAM, runs}
};
// Two banks of two sets: the passes of a REF block (forward, complement) use the sets of one bank, the front of the REF
// block that follows is made in the other
static FrontBufs g_fb[4];
static std::unordered_map<const IndexDev *, IndexDev *> g_next_of;       // REF index -> the REF index the driver named after it last time
static DevBuf<MergeEntry> g_entries;
static IndexDev *g_hint_a = nullptr, *g_hint_b = nullptr;               // the pass the driver says comes next
static int g_hint_kshift = 0;

static void ensure_bucket_table(IndexDev &Bx, int Kshift);

static BLookup lookup_of(const IndexDev &Bx)
{ return BLookup{Bx.keys, Bx.d_bstart, Bx.bshift, Bx.low_off ? (const uint16_t *) (Bx.d_bstart + Bx.low_off) : nullptr}; }

// Fused per-read route: a pass returns with its per-read kernels queued on the main stream and their counters on the way
// to page-locked memory; its second half (cluster numbering, pairs, chain stage: the tail stream's) is queued by the NEXT
// pass once that has given the main stream its own work -- the host then waits for the counters of kernels that have
// long finished instead of standing between the main stream and its next launch twice per pass (round 4: 26 % of the
// main stream's time were such gaps).  "defer_close" test option = 0 / DMH_DEFER_CLOSE=0: every pass closed before
// Match_Filter returns.
static bool defer_fused_close()
{ static const int env = getenv("DMH_DEFER_CLOSE") ? atoi(getenv("DMH_DEFER_CLOSE")) : 4;
  const int o = S().opt_defer_close;
  return (o >= 0 ? o : env) == 4;
}

// B2 != nullptr: the same sweep also probes B2 (results in set 1)
static void match_front(IndexDev &A, IndexDev &Bx, int Kshift, IndexDev *B2 = nullptr, int bank = 0, hipStream_t on = nullptr)
{ State &st = S();
  if (on == nullptr) on = main_stream();
  StreamScope on_stream(on);
  const bool ahead = (on != main_stream());
  FrontBufs *const fb = g_fb + 2 * bank;
  for (int q = 0; q < 2; q++)
    { // whoever used the set last -- the stream that filled it, the main stream that read it -- is done before it is filled again
      FrontBufs &F = fb[q];
      if (F.made == nullptr)
        { HIP_CHECK(hipEventCreateWithFlags(&F.made, hipEventDisableTiming));
          HIP_CHECK(hipEventCreateWithFlags(&F.consumed, hipEventDisableTiming));
          HIP_CHECK(hipHostMalloc((void **) &F.h_sums, 4 * sizeof(unsigned long long), hipHostMallocDefault));
        }
      if (F.was_made) HIP_CHECK(hipStreamWaitEvent(on, F.made, 0));
      if (F.was_used) HIP_CHECK(hipStreamWaitEvent(on, F.consumed, 0));
      F.ahead = false; F.posted = false;
    }
  ensure_bucket_table(Bx, Kshift);
  if (B2 != nullptr)
    { ensure_bucket_table(*B2, Kshift);
      if (B2->bshift != Bx.bshift || B2->probe_keys != Bx.probe_keys || (B2->low_off != 0) != (Bx.low_off != 0)) B2 = nullptr;
    }
  const BLookup BL = lookup_of(Bx), BL2 = (B2 != nullptr) ? lookup_of(*B2) : BL;
  if (A.n >= ((int64_t) 1 << 32)) fatal("Match_Filter: a READS index of %lld k-mers (positions are kept in 32 bits)", (long long) A.n);
  const int64_t ntiles = ceil_div(A.n, MG_TILE);
  MatchOut MO[2];
  for (int q = 0; q < (B2 != nullptr ? 2 : 1); q++)
    { FrontBufs &F = fb[q];
      F.gram.ensure(MAXGRAM + 8, "hit histogram");
      F.mctl.ensure(4, "merge counters");
      F.tentries.ensure((size_t) ntiles * MG_TILE, "matched runs (per tile)");
      F.tcount.ensure(ntiles + 1, "matched runs per tile");
      F.thits.ensure(2 * (size_t) ntiles, "hits per tile");
      HIP_CHECK(hipMemsetAsync(F.gram.p + MAXGRAM + 1, 0, sizeof(unsigned long long) * 3, stream()));     // [0] all hits, [1] hits of the runs below MAXGRAM
      HIP_CHECK(hipMemsetAsync(F.mctl.p, 0, 4 * sizeof(unsigned int), stream()));
      MO[q] = MatchOut{F.tentries.p, F.tcount.p, F.thits.p};
      F.have_sums = false;
    }
  if (B2 == nullptr) MO[1] = MO[0];
  { // bytes the kernel has to read in this layout: 8 per READ k-mer, 4 per bucket and 2 (the low bits) or 8 (the key) per
    // REF k-mer -- SURVEY 8(d) prices the merge at the reference's 16-byte records, 16 (A.n + B.n), which is more than
    // this kernel moves
    const double per_table = 4.0 * ((double) ((int64_t) 1 << (Kshift - Bx.bshift)) + 2) + (BL.klow != nullptr ? 2.0 : 8.0) * Bx.n;
    StageTimer tm(B2 != nullptr ? "merge_match2" : "merge_match", 8.0 * A.n + (B2 != nullptr ? 2.0 : 1.0) * per_table);
    const dim3 grid(B2 != nullptr ? (uint32_t) (((ntiles + 7) / 8) * 16) : (uint32_t) ntiles), blk(MG_THREADS);
    const uint64_t *ak = (const uint64_t *) A.keys;
    const bool four = Bx.probe_keys == 4, low = BL.klow != nullptr;
    // "DMH_MATCH_COMPACT=0": the look-ups item by item, as until round 4 (comparisons)
    static const bool env_compact = getenv("DMH_MATCH_COMPACT") == nullptr || atoi(getenv("DMH_MATCH_COMPACT")) != 0;
    const bool compact = st.opt_match_compact >= 0 ? st.opt_match_compact != 0 : env_compact;      // ("match_compact" test option)
#define MM_LAUNCH(NK_, LOW_, TWO_) \
    do { if (compact) hipLaunchKernelGGL((k_merge_match<NK_, LOW_, TWO_, true>), grid, blk, 0, stream(), ak, A.n, BL, MO[0], BL2, MO[1]); \
         else         hipLaunchKernelGGL((k_merge_match<NK_, LOW_, TWO_, false>), grid, blk, 0, stream(), ak, A.n, BL, MO[0], BL2, MO[1]); } while (0)
    if (B2 != nullptr)
      { if (four && low)  MM_LAUNCH(4, true, true);
        else if (four)    MM_LAUNCH(4, false, true);
        else if (low)     MM_LAUNCH(2, true, true);
        else              MM_LAUNCH(2, false, true);
      }
    else
      { if (four && low)  MM_LAUNCH(4, true, false);
        else if (four)    MM_LAUNCH(4, false, false);
        else if (low)     MM_LAUNCH(2, true, false);
        else              MM_LAUNCH(2, false, false);
      }
#undef MM_LAUNCH
  }
  { StageTimer tm("merge_fixup", 0.0);
    for (int q = 0; q < (B2 != nullptr ? 2 : 1); q++)
      { FrontBufs &F = fb[q];
        hipLaunchKernelGGL(k_merge_fixup, dim3(ntiles), dim3(FX_THREADS), 0, stream(), (const uint64_t *) A.keys, A.n, q ? BL2 : BL,
                           F.tentries.p, (const uint32_t *) F.tcount.p, F.thits.p);
        hipLaunchKernelGGL(k_merge_reduce, dim3(64), dim3(256), 0, stream(), (const unsigned long long *) F.thits.p, ntiles,
                           F.gram.p + MAXGRAM + 1);
        exclusive_scan_u32(F.tcount.p, ntiles, (ahead ? st.sort_front : st.sort).bsum, F.tcount.p + ntiles);
      }
    HIP_CHECK(hipGetLastError());
  }
  fb[0].a = &A; fb[0].b = &Bx;
  if (B2 != nullptr) { fb[1].a = &A; fb[1].b = B2; }
  else               { fb[1].a = nullptr; fb[1].b = nullptr; }
  if (ahead)
    { // the sums of the front travel to page-locked memory behind it: no round trip when the pass comes
      for (int q = 0; q < (B2 != nullptr ? 2 : 1); q++)
        { FrontBufs &F = fb[q];
          unsigned long long *sums = F.gram.p + MAXGRAM + 1;
          HIP_CHECK(hipMemcpyAsync(sums + 2, F.tcount.p + ntiles, sizeof(unsigned int), hipMemcpyDeviceToDevice, stream()));
          HIP_CHECK(hipMemcpyAsync(F.h_sums, sums, sizeof(unsigned long long) * 3, hipMemcpyDeviceToHost, stream()));
          F.posted = true;
        }
    }
  for (int q = 0; q < (B2 != nullptr ? 2 : 1); q++)
    { FrontBufs &F = fb[q];
      HIP_CHECK(hipEventRecord(F.made, stream()));
      F.was_made = true;
      F.ahead = ahead;
    }
}

static void ensure_bucket_table(IndexDev &Bx, int Kshift)
{ State &st = S();
  // --- bucket table over the REF index: part of the index, built the first time it is merged against
  //     (a retained REF index meets every READS block)
  if (Bx.d_bstart == nullptr)
    { // small index: ~0.15-0.25 entries per bucket, two-key probes (the table stays in the caches).  A block of a
      // genome (10^8 k-mers and more): the table is streamed from HBM by every pass, so at most 2^29 buckets
      // (2 GB; 0.65 entries per bucket for a 375 Mbp block) and four-key probes, which leave the bisection to
      // a look-up in a thousand ("merge_dense" test option: one bucket per entry + four keys on any index)
      const int dense = st.opt_merge_dense;
      const bool big = (dense >= 0) ? (dense != 0) : (Bx.n >= ((int64_t) 1 << 26));
      int P = ceil_log2(Bx.n) + 2;
      if (P < 8) P = 8;
      static const int env_p = getenv("DMH_BUCKET_BITS") ? atoi(getenv("DMH_BUCKET_BITS")) : 0;       // (experiments)
      const int pmax = env_p > 0 ? env_p : 29;
      if (big) { if (P > pmax) P = pmax; } else { if (P > 28) P = 28; }
      if (dense == 1) P = std::max(4, ceil_log2(Bx.n));
      if (dense == 2) P = std::max(4, Kshift - 16);          // (test: the 2-byte probe array on any index)
      if (P > Kshift) P = Kshift;
      Bx.probe_keys = big ? 4 : 2;
      Bx.bshift = Kshift - P;
      int64_t nbucket = (int64_t) 1 << P;
      // an index whose buckets leave at most 16 bits undecided also gets those bits as a 2-byte array behind the
      // table (same allocation): the probes of a pass then read a quarter of the REF key bytes and hold half the
      // registers (k_merge_match<4> with 8-byte keys runs at two waves per SIMD, with the low bits at four)
      const bool low16 = Bx.bshift <= 16;
      size_t words = (size_t) (nbucket + 2);
      Bx.low_off = low16 ? words : 0;
      if (low16) words += ((size_t) Bx.n + 1) / 2 + 3;          // (+ room for the 8-byte probe of the last entry)
      Bx.d_bstart = (uint32_t *) dev_alloc(sizeof(uint32_t) * words, "bucket table");
      Bx.bstart_bytes = sizeof(uint32_t) * words;
      StageTimer tm("merge_buckets", 8.0 * Bx.n);
      hipLaunchKernelGGL(k_bucket_bounds, dim3(ceil_div(Bx.n + 1, 256)), dim3(256), 0, stream(),
                         (const uint64_t *) Bx.keys, Bx.n, Bx.bshift, nbucket, Bx.d_bstart);
      if (low16)
        hipLaunchKernelGGL(k_low16, dim3(ceil_div(Bx.n, 256)), dim3(256), 0, stream(), (const uint64_t *) Bx.keys, Bx.n,
                           (uint32_t) (((uint64_t) 1 << Bx.bshift) - 1), (uint16_t *) (Bx.d_bstart + Bx.low_off));
    }
}

void filter_hint_next_pass(IndexDev *A, IndexDev *B, int Kshift)
{ g_hint_a = A; g_hint_b = B; g_hint_kshift = Kshift; }

void filter_forget_index(const IndexDev *ix)
{ if (g_hint_a == ix || g_hint_b == ix) { g_hint_a = nullptr; g_hint_b = nullptr; }
  for (int q = 0; q < 4; q++)
    if (g_fb[q].a == ix || g_fb[q].b == ix)
      { // (a front made ahead for a pass that did not come may still be reading the index: whatever frees it comes after)
        if (g_fb[q].ahead && g_fb[q].was_made) (void) hipStreamWaitEvent(main_stream(), g_fb[q].made, 0);
        g_fb[q].a = nullptr; g_fb[q].b = nullptr;
      }
  g_next_of.erase(ix);
  for (auto it = g_next_of.begin(); it != g_next_of.end(); )
    if (it->second == ix) it = g_next_of.erase(it); else ++it;
}

int64_t merge_pass(const DAZZ_DB *ablock, const DAZZ_DB *bblock, IndexDev &A, IndexDev &Bx,
                   int Kshift, int comp, int64_t *nfilt)
{ State &st = S();
  MergeResult *Rp = &filter_batch();
#define R (*Rp)
  HostTimer ht("merge_and_sort");
  static const bool hooked = (set_reclaim_hook(filter_reclaim), true);
  (void) hooked;

  // --- the front of the pass (match_front above), unless the sweep of the pass before has made it already.  When the
  //     driver has named the pass that follows (dmh_next_pass) and it meets the same READ index -- the complement of this
  //     REF block, resident and complete --, this sweep probes both indices ("front_early" test option = 0: never)
  int fs = -1;
  for (int q = 0; q < 4; q++)
    if (g_fb[q].a == &A && g_fb[q].b == &Bx) fs = q;
  const bool front_was_made = fs >= 0;
  static const bool env_two = getenv("DMH_MATCH_TWO") == nullptr || atoi(getenv("DMH_MATCH_TWO")) != 0;
  auto usable = [&](const IndexDev *x) -> bool
  { return x != nullptr && x->keys != nullptr && !x->pending && x->built == nullptr && x->n > 0; };
  if (g_hint_a == &A && g_hint_b != nullptr && g_hint_b != &Bx) g_next_of[&Bx] = g_hint_b;       // (the order of the REF indices, learnt)
  if (fs < 0)
    { IndexDev *B2 = nullptr;
      if (env_two && st.opt_front_early != 0 && g_hint_a == &A && g_hint_b != &Bx && usable(g_hint_b) && g_hint_kshift == Kshift)
        B2 = g_hint_b;
      // into a bank that holds nobody's front (one that was made ahead for a pass that did not come is given up)
      int bank = 0;
      if (g_fb[0].a != nullptr || g_fb[1].a != nullptr) bank = 1;
      for (int q = 0; q < 2; q++) { g_fb[2 * bank + q].a = nullptr; g_fb[2 * bank + q].b = nullptr; }
      match_front(A, Bx, Kshift, B2, bank);
      fs = 2 * bank;
    }
  FrontBufs &FS = g_fb[fs];
  if (FS.ahead) HIP_CHECK(hipStreamWaitEvent(main_stream(), FS.made, 0));      // (a front made on the front stream)
  // ... and the front of the REF block that FOLLOWS, on the front stream, next to this pass's per-read kernels: the driver has
  // named the next pass (dmh_next_pass), and the pass after that is the one that followed it for the READS block before
  if (front_stream() != main_stream() && st.opt_front_early != 0 && env_two && g_hint_a == &A && g_hint_b != &Bx && usable(g_hint_b) &&
      g_hint_kshift == Kshift && stream() == main_stream() && !st.capture_seeds && !VERBOSE)
    { auto has_front = [&](const IndexDev *x) -> bool
      { for (int q = 0; q < 4; q++) if (g_fb[q].a == &A && g_fb[q].b == x) return true;
        return false;
      };
      auto after = [&](IndexDev *x) -> IndexDev *
      { auto it = g_next_of.find(x);
        return (it != g_next_of.end() && it->second != x && it->second != &Bx && usable(it->second)) ? it->second : nullptr;
      };
      // the first REF index to come that has no front yet: the next pass's, or -- when the sweep of this REF block has made
      // that one -- the first of the REF block that follows: queued at a REF block's FIRST pass, its front has two passes'
      // time (queued at the second pass it was late one time in three: 0.4 ms of main stream waiting per REF block)
      IndexDev *t = g_hint_b;
      for (int hop = 0; hop < 2 && t != nullptr && has_front(t); hop++) t = after(t);
      const int other = (fs >> 1) ^ 1;
      if (t != nullptr && !has_front(t) && g_fb[2 * other].a == nullptr && g_fb[2 * other + 1].a == nullptr)
        { IndexDev *B2 = after(t);
          if (B2 != nullptr && has_front(B2)) B2 = nullptr;
          // (once per READS index: the front stream starts behind what the main stream had queued when the index was
          //  first met -- the index's own build, the REF indices')
          static const IndexDev *seen_a = nullptr;
          static hipEvent_t ev_a = nullptr;
          if (seen_a != &A)
            { if (ev_a == nullptr) HIP_CHECK(hipEventCreateWithFlags(&ev_a, hipEventDisableTiming));
              HIP_CHECK(hipEventRecord(ev_a, main_stream()));
              HIP_CHECK(hipStreamWaitEvent(front_stream(), ev_a, 0));
              seen_a = &A;
            }
          match_front(A, *t, Kshift, B2, other, front_stream());
        }
    }
  g_fb[fs].a = nullptr; g_fb[fs].b = nullptr;
  g_hint_a = nullptr; g_hint_b = nullptr;
  struct ConsumedMark        // (every way out of this call: the set may be filled again behind what has been queued here)
  { FrontBufs &F;
    ~ConsumedMark() { if (F.consumed != nullptr) { (void) hipEventRecord(F.consumed, main_stream()); F.was_used = true; } }
  } consumed_mark{FS};
  DevBuf<MergeEntry> &entries = g_entries, &tentries = g_fb[fs].tentries;
  DevBuf<uint32_t>   &tcount = g_fb[fs].tcount;
  DevBuf<unsigned long long> &gram = g_fb[fs].gram;
  DevBuf<unsigned int> &mctl = g_fb[fs].mctl;
  const int64_t ntiles = ceil_div(A.n, MG_TILE);
  unsigned long long *sums = gram.p + MAXGRAM + 1;         // [0] all hits, [1] hits of the runs below MAXGRAM
  // one round trip for the pass: the number of matched runs and the two hit sums (the count is parked behind the sums) --
  // of BOTH tables when this sweep has filled the second set for the pass that follows.  The copies are queued right behind
  // the front, ahead of everything the host does next for OTHER passes (the close that is pending: 0.3 ms of launches; the
  // oldest batch's finish: a millisecond and more of waiting for the finish stream): the trace of round 4 showed the main
  // stream standing still for 1.4 ms per REF block between the front's last kernel and these copies
  unsigned int nent = 0;
  unsigned long long hsum[3] = {0, 0, 0};
  if (front_was_made && FS.posted)
    { // made ahead: its sums have been on their way since (the event is long done when the GPU is the bottleneck)
      HostTimer hw("wait_front_ahead");
      HIP_CHECK(hipEventSynchronize(FS.made));
      for (int q = 0; q < 3; q++) FS.hsum[q] = FS.h_sums[q];
      FS.have_sums = true;
      FS.posted = false;
    }
  const bool cached_sums = front_was_made && g_fb[fs].have_sums;
  FrontBufs *F2 = nullptr;                 // the other set, when this sweep has filled it for the pass that follows
  static unsigned long long *h_sums = nullptr;           // page-locked
  static hipEvent_t ev_sums = nullptr;
  if (!cached_sums)
    { static const bool env_sc = getenv("DMH_SUMS_CACHE") == nullptr || atoi(getenv("DMH_SUMS_CACHE")) != 0;
      if (env_sc && !front_was_made && (fs & 1) == 0 && g_fb[fs + 1].a != nullptr && g_fb[fs + 1].b != nullptr) F2 = &g_fb[fs + 1];
      if (h_sums == nullptr)
        { HIP_CHECK(hipHostMalloc((void **) &h_sums, 8 * sizeof(unsigned long long), hipHostMallocDefault));
          HIP_CHECK(hipEventCreateWithFlags(&ev_sums, hipEventDisableTiming));
        }
      HIP_CHECK(hipMemcpyAsync(sums + 2, tcount.p + ntiles, sizeof(unsigned int), hipMemcpyDeviceToDevice, stream()));
      if (F2 != nullptr)
        { HIP_CHECK(hipMemcpyAsync(F2->gram.p + MAXGRAM + 1 + 2, F2->tcount.p + ntiles, sizeof(unsigned int), hipMemcpyDeviceToDevice, stream()));
          HIP_CHECK(hipMemcpyAsync(sums + 3, F2->gram.p + MAXGRAM + 1, sizeof(unsigned long long) * 3, hipMemcpyDeviceToDevice, stream()));
        }
      HIP_CHECK(hipMemcpyAsync(h_sums, sums, sizeof(unsigned long long) * (F2 != nullptr ? 6 : 3), hipMemcpyDeviceToHost, stream()));
      HIP_CHECK(hipEventRecord(ev_sums, stream()));
    }
  // the batch that the previous pass left over the limit is finished now: the host waits for the finish stream, the
  // main stream has the match kernels (and the previous pass's chain kernels) to run meanwhile
  // (... unless this pass's front was made by the sweep before: the main stream has nothing of this pass yet -- the close
  //  that is pending is then taken up behind this pass's per-read kernels, close_batch phase 4, or by whatever closes next)
  if (!(front_was_made && defer_fused_close())) close_pending();
  // (When the sweep of the pass before has made this pass's front already, the main stream has NOTHING queued here: the
  //  host's wait for the finish stream -- a millisecond -- would be the main stream's too, on every second pass.  The
  //  oldest batch is then finished a little later, behind this pass's per-read kernels: close_batch.  With the back half
  //  of a pass deferred EVERY pass finishes there: the front of a sweep is 3.5 ms of main-stream work, the pending close
  //  and the finish together are 5 ms of host time.)
  static const bool env_fic = getenv("DMH_FINISH_IN_CLOSE") == nullptr || atoi(getenv("DMH_FINISH_IN_CLOSE")) != 0;      // (measured: 111.2 against 113.1 ms per read block, 44.7 against 44.8 on chr1 -- the main stream's gap per REF block 0.3 ms instead of 1.4, the block time the same within the noise: the GPU is full)
  g_finish_in_close = (front_was_made || (env_fic && defer_fused_close())) && flight_keep() > 0;
  if (!g_finish_in_close)
  { static const bool nosteps = getenv("DMH_FINISH_STEPS") == nullptr;
    static const bool pipe = getenv("DMH_FINISH_PIPE") != nullptr;      // (opt-in: measured 178.5 against 173.1 ms per read block)
    const int keep2 = flight_keep();
    while ((int) g_inflight.size() > (nosteps || keep2 == 0 ? keep2 : keep2 + 1)) *nfilt += finish_oldest();
    if (pipe && nosteps && keep2 > 0) finish_pipeline(nfilt);
    finish_tick_if_due(nfilt);              // (step 1 of 3; the others behind the expansion and behind the seed sort)
  }
  if (cached_sums)
    { for (int q = 0; q < 3; q++) hsum[q] = g_fb[fs].hsum[q]; }         // (fetched with the sweep's own pass: no round trip)
  else
    { HostTimer hw("wait_hit_sums");
      bg_pump(1);                            // (a stage of the next block's index behind the copies, as copy_d2h does)
      HIP_CHECK(hipEventSynchronize(ev_sums));
      for (int q = 0; q < 3; q++) hsum[q] = h_sums[q];                  // the limit rule next
      if (F2 != nullptr)
        { for (int q = 0; q < 3; q++) F2->hsum[q] = h_sums[3 + q];
          F2->have_sums = true;
        }
    }
  g_fb[fs].have_sums = false;
  nent = (unsigned int) (hsum[2] & 0xffffffffull);
  // the compact entry list, made when somebody needs it (the fused per-read route reads the tiles' own windows)
  bool gathered = false;
  auto gather_entries = [&]()
  { if (gathered) return;
    gathered = true;
    entries.ensure((size_t) nent + 64, "matched runs");
    if (nent > 0)
      hipLaunchKernelGGL(k_merge_gather, dim3(ceil_div(ntiles * 64, 256)), dim3(256), 0, stream(),
                         (const MergeEntry *) tentries.p, (const uint32_t *) tcount.p, ntiles, nent, entries.p);
  };

  // --- limit (map.c:2992-3055)
  int64_t limit, nhits;
  if (VERBOSE) printf("\n");
  if (MEM_LIMIT > 0)
    { int64_t alen = A.n, blen = Bx.n;
      int64_t avail = (int64_t) (MEM_LIMIT - (uint64_t) (sizeof_db(ablock) + sizeof_db(bblock))) / 16;
      if (avail > alen + 2 * blen)
        avail = (avail - alen) / 2;
      else
        avail = avail - (alen + blen);
      avail *= .98;
      int j;
      if ((int64_t) hsum[1] <= avail)
        { // every run below MAXGRAM fits: the scan over the histogram would run to its end (map.c:3030-3036)
          j = MAXGRAM;
          nhits = (int64_t) hsum[1];
        }
      else
        { std::vector<unsigned long long> histo(MAXGRAM + 1, 0);
          gather_entries();
          HIP_CHECK(hipMemsetAsync(gram.p, 0, sizeof(unsigned long long) * (MAXGRAM + 1), stream()));
          if (nent > 0)
            hipLaunchKernelGGL(k_entry_hist, dim3(ceil_div(nent, 256)), dim3(256), 0, stream(),
                               (const MergeEntry *) entries.p, nent, gram.p);
          copy_d2h(histo.data(), gram.p, sizeof(unsigned long long) * (MAXGRAM + 1));
          int64_t tom = 0;
          for (j = 0; j < MAXGRAM; j++)
            { tom += j * (int64_t) histo[j];
              if (tom > avail) break;
            }
          nhits = 0;
          for (int x = 1; x < j; x++) nhits += x * (int64_t) histo[x];
        }
      limit = j;
      if (limit <= 1)
        { fprintf(stderr, "\nError: Insufficient memory allocation (%.1fGb), reduce block size\n",
                  (1. * MEM_LIMIT) / 0x40000000ll);
          fflush(stderr);
          Clean_Exit(1);
        }
      if (limit < 10)
        { fprintf(stderr, "\nWarning: Sensitivity hampered by low memory allocation (%.1fGb), reduce block size\n",
                  (1. * MEM_LIMIT) / 0x40000000ll);
          fflush(stderr);
        }
      if (VERBOSE)
        { printf("   Capping mutual k-mer matches over %d (effectively -t%d)\n", (int) limit, (int) sqrt(1. * limit));
          fflush(stdout);
        }
    }
  else
    { limit = INT32_MAX;
      nhits = (int64_t) hsum[0];
    }
  st.last_limit = (int) limit;
  if (VERBOSE)
    { printf("   Hit count = %lld\n", (long long) nhits);
      fflush(stdout);
    }
  if (nhits == 0) return 0;
  const int64_t pass_hits = nhits;

  // Everything behind the limit rule, for the hits of the reads [r0, r1) -- the whole pass, or (below) one range of a
  // pass that has more hits than 32-bit slots
  auto submit = [&](const int r0, const int r1, const int64_t nhits) -> int64_t
  { const bool ranged = (r0 > 0 || r1 < ablock->nreads);

  // --- which batch the pass goes to.  Cluster mode (dmh_merge.h): when a read meets thousands of hits per pass
  //     (repeat families), the hits of an (aread, bread) pair are cut into B-clusters, which chain independently -- a
  //     pair's live window would hold hundreds of hits of unrelated loci, a cluster's holds a few dozen; such a pass
  //     is processed by itself ("cluster" test option: 1 always, 0 never).  Sparse passes are collected: the sort,
  //     the chain kernels and the domination pass then run once per batch instead of once per pass.
  static const int env_cl = getenv("DMH_CLUSTER") ? atoi(getenv("DMH_CLUSTER")) : -1;
  static const int env_batch = getenv("DMH_BATCH") ? atoi(getenv("DMH_BATCH")) : 1;
  const int optcl = st.opt_cluster >= 0 ? st.opt_cluster : env_cl;
  const int pbits = ceil_log2(ablock->maxlen > 1 ? ablock->maxlen : 1);
  const int abits = ceil_log2(ablock->nreads), bbits = ceil_log2(bblock->nreads);
  const int qbits = ceil_log2(bblock->maxlen > 1 ? (int64_t) bblock->maxlen + 1 : 2);
  bool cluster = (optcl >= 0) ? (optcl != 0) : (nhits >= (int64_t) 1024 * ablock->nreads);
  if (ranged) cluster = true;                 // (a range of an oversize pass: dense by definition, and only the per-read route takes ranges)
  if (qbits + bbits + abits > 64 || nhits >= (int64_t) 0x7fffffff) cluster = false;
  const bool batching = !cluster && !VERBOSE && !st.capture_seeds && (st.opt_batch >= 0 ? st.opt_batch != 0 : env_batch != 0);
  if (R.open && (cluster || !batching || R.ablock != ablock || R.npass >= MAXPASS || R.nhits + nhits > BATCH_CAP ||
                 bbits > R.bbits || pbits != R.pbits || abits != R.abits))
    { close_batch(R);
      Rp = &filter_batch();                  // (the next slot of the ring)
    }
  if (!R.open)
    { R.open = true;
      R.ablock = ablock;
      R.cslot = st.cand_cur;
      R.npass = 0;
      R.nhits = 0;
      R.clustered = cluster;
      R.pbits = pbits; R.abits = abits; R.bbits = bbits; R.qbits = qbits;
      R.passbits = batching ? 5 : 0;
      // cluster mode: one word per hit through the seed sort when aread | bread | bpos | apos fit it and the clusters
      // are sorted where they lie ("seed_word" test option: 0 = always keys and values)
      R.word = cluster && local_cluster_sort(pbits) && st.opt_seed_word != 0 && abits + bbits + qbits + pbits <= 64;
      if ((cluster ? qbits : pbits) + R.bbits + R.passbits + R.abits > 64)
        { if (pbits + R.bbits + R.abits > 64) fatal("Match_Filter: seed key needs %d bits", pbits + R.bbits + R.abits);
          R.passbits = 0;
        }
      const int64_t cap = std::max<int64_t>(R.passbits ? BATCH_CAP : 0, nhits);
      R.keys.ensure(cap, "seed keys");
      if (!R.word) R.vals.ensure(cap, "seed positions");
    }
  const int lowbits = R.word ? R.qbits + R.pbits : (R.clustered ? R.qbits : R.pbits);      // bits below bread in the first sort's key
  const int pshift = lowbits, bshift = lowbits + R.bbits + R.passbits;
  const int hitmode = R.word ? 2 + R.pbits : (R.clustered ? 1 : 0);                        // (put_hit)
  const uint64_t passor = (uint64_t) R.npass << (lowbits + R.bbits);

  // positions as (read, rpos) words for indices that keep being merged against (dmh_core.hip index_pack_positions)
  if (st.retain_ref && !Bx.borrowed) index_pack_positions(Bx);          // (a cached index is packed when it enters the cache)
  if (st.mf_calls >= 3) index_pack_positions(A);

  // --- the per-read route (dmh_readsort.h): a dense pass in one-word mode writes its hits read by read
  static const int env_rs = getenv("DMH_READ_SORT") ? atoi(getenv("DMH_READ_SORT")) : 1;
  static const bool env_per_entry = getenv("DMH_EXPAND_PER_ENTRY") != nullptr;
  const int  rs_cb = std::min(rs_coarse_bits(), R.qbits);
  const bool by_read = R.word && R.npass == 0 && !env_per_entry && st.opt_expand != 0 &&
                       (st.opt_read_sort >= 0 ? st.opt_read_sort != 0 : env_rs != 0) &&
                       R.pbits + R.qbits - rs_cb + R.bbits <= 32 &&        // (the sort key of k_read_sort is one 32-bit word)
                       // (a pass in which the AVERAGE read has more hits than the LDS kernels take -- a one-family reference:
                       //  28,000 per read -- would send every read through k_read_sort_big, one workgroup each through global
                       //  memory: 4.5 s per read block against 1.9 with the global sort, which is made for that; test options
                       //  that force a route are obeyed)
                       (st.opt_read_sort > 0 || nhits <= (int64_t) ablock->nreads * (RS_T2 * RS_E / 2));
  R.read_sorted = false;
  R.fused = false;
  // ... and FUSED (round 4, dmh_readsort.h): the rows of the merge are listed per read, and the per-read kernel expands,
  // orders and -- plain chains -- scores them without a trip of the hits through HBM.  Needs what the row records
  // assume: partner counts below 2^14 (the -M limit rule's MAXGRAM), apos in 18 bits, (read, rpos) words in both indices;
  // "fuse" test option / DMH_FUSE = 0: round 3's route
  static const int env_fuse = getenv("DMH_FUSE") ? atoi(getenv("DMH_FUSE")) : 1;
  static const bool env_dc0 = getenv("DMH_DEFER_CLOSE") != nullptr && atoi(getenv("DMH_DEFER_CLOSE")) == 1;
  const bool fuse = by_read && (st.opt_fuse >= 0 ? st.opt_fuse != 0 : env_fuse != 0) && limit <= (int64_t) MAXGRAM &&
                    R.pbits <= RW_AP && A.posbits > 0 && Bx.posbits > 0 && !(env_dc0 || (st.opt_defer_close > 0 && st.opt_defer_close != 4)) &&
                    A.L + 1 < ((int64_t) 1 << 32);
  if (ranged && !fuse)
    fatal("Match_Filter: %lld seed hits in one block pair exceed the 32-bit slot range and the pass cannot be split (it takes "
          "the -M limit rule, reads of up to 2^18 bases and a seed word of 64 bits); use smaller blocks or -M", (long long) pass_hits);
  if (fuse)
    { StageTimer tm("merge_rows", 16.0 * nhits);
      const int nr = ablock->nreads;
      R.rs_start.ensure(2 * ((size_t) nr + 2), "first hit of each read, rows of each read");
      R.rs_stats.ensure(8, "read segment statistics");
      g_rows.ensure((size_t) A.L + 2, "rows of the merge, read by read");
      uint32_t *nrow = R.rs_start.p + nr + 2;
      HIP_CHECK(hipMemsetAsync(R.rs_start.p, 0, sizeof(uint32_t) * 2 * ((size_t) nr + 2), stream()));
      HIP_CHECK(hipMemsetAsync(R.rs_stats.p, 0, sizeof(uint32_t) * 8, stream()));
      hipLaunchKernelGGL(k_row_list, dim3(ceil_div(ntiles * 64, 256)), dim3(256), 0, stream(), (const MergeEntry *) tentries.p,
                         (const uint32_t *) tcount.p, ntiles, nent, limit, (const uint32_t *) A.vals, A.posbits,
                         (const int64_t *) A.d_boff, R.rs_start.p, nrow, g_rows.p, (uint32_t) r0, (uint32_t) r1);
      exclusive_scan_u32(R.rs_start.p, nr, st.sort.bsum, R.rs_start.p + nr);
      R.rs_lists.ensure(4 * (size_t) nr + 16, "reads by segment size");
      g_rs_small = report_in_flight() || st.opt_read_sort == 6 || st.opt_read_sort == 7;      // (see the unfused route below)
      const uint32_t lim2 = (st.opt_read_sort == 3 || st.opt_read_sort == 7) ? 512u
                            : (g_rs_small ? (uint32_t) RS_T1 * RS_E : (uint32_t) RS_T2 * RS_E);
      const uint32_t lim0 = std::min(lim2, (st.opt_read_sort == 2) ? 128u : (uint32_t) RS_T0 * RS_E);
      const uint32_t lim1 = std::min(lim2, (st.opt_read_sort == 2) ? 128u : (uint32_t) RS_T1 * RS_E);
      hipLaunchKernelGGL(k_read_classes, dim3(ceil_div(nr, 256)), dim3(256), 0, stream(), (const uint32_t *) R.rs_start.p, nr,
                         lim0, lim1, lim2, R.rs_lists.p, R.rs_stats.p);
      g_rlim = lim2;
      g_rgrid[0] = (uint32_t) nr;
      g_rgrid[1] = (uint32_t) std::min<int64_t>(nr, nhits / ((int64_t) lim0 + 1) + 1);
      g_rgrid[2] = (uint32_t) std::min<int64_t>(nr, nhits / ((int64_t) lim1 + 1) + 1);
      g_rgrid[3] = (uint32_t) std::min<int64_t>(nr, nhits / ((int64_t) lim2 + 1) + 1);
      // the reads beyond the LDS kernels (a tandem array's: one or two per block) get their words the old way
      hipLaunchKernelGGL(k_rows_expand_big, dim3(std::min<uint32_t>(g_rgrid[3], 256u)), dim3(256), 0, stream(),
                         (const uint32_t *) (R.rs_lists.p + 3 * (size_t) nr), (const uint32_t *) (R.rs_stats.p + 4),
                         (const uint64_t *) g_rows.p, (const int64_t *) A.d_boff, (const uint32_t *) nrow,
                         (const uint32_t *) R.rs_start.p, (const uint32_t *) Bx.vals, Bx.posbits, R.pbits, R.qbits, bshift,
                         R.keys.p);
      HIP_CHECK(hipGetLastError());
      finish_tick_if_due(nfilt);
      static const bool want = getenv("DMH_STATS") != nullptr;
      if (want)
        { HostTimer hw("wait_read_classes");
          copy_d2h(g_rcnt, R.rs_stats.p, 5 * sizeof(uint32_t), 1);
        }
      else
        { for (int q = 0; q < 4; q++) g_rcnt[q == 3 ? 4 : q] = g_rgrid[q];
          g_rcnt[2] = std::min(g_rcnt[2], 2048u);
        }
      R.read_sorted = true;
      R.fused = true;
      R.f_rows = g_rows.p; R.f_rowbase = A.d_boff; R.f_bvals = Bx.vals; R.f_bposbits = Bx.posbits;
    }

  // --- expand the entries under the limit into hits (map.c:3095-3099), behind the batch's earlier passes
  static DevBuf<uint32_t> esz, eoff, ebig;
  uint32_t total = 0, nbig = 0;
  uint64_t *hk = R.keys.p + R.nhits;
  uint32_t *hv = R.vals.p + R.nhits;
  if (!fuse)
  { StageTimer tm("merge_emit", 16.0 * nhits);
    gather_entries();
    esz.ensure((size_t) nent + 1, "entry sizes");
    eoff.ensure((size_t) nent + 2, "entry offsets");
    ebig.ensure((size_t) nent + 1, "big entries");
    int nb = ceil_div(nent, 256);
    hipLaunchKernelGGL(k_entry_sizes, dim3(nb), dim3(256), 0, stream(), (const MergeEntry *) entries.p, nent, limit, esz.p);
    exclusive_scan_u32((const uint32_t *) esz.p, eoff.p, nent, st.sort.bsum, eoff.p + nent);
    PosMap MA{A.d_boff, A.d_chunk, A.posbits}, MB{Bx.d_boff, Bx.d_chunk, Bx.posbits};
    const bool per_entry = env_per_entry;
    if (!per_entry && st.opt_expand != 0)
      { static DevBuf<uint32_t> tfirst;
        tfirst.ensure((size_t) ceil_div(nhits, EX_TILE) + 2, "first entry of each hit tile");
        hipLaunchKernelGGL(k_tile_first, dim3(nb), dim3(256), 0, stream(), (const uint32_t *) eoff.p, nent, tfirst.p);
        const uint32_t *rstart = nullptr, *rowdst = nullptr;
        if (by_read)
          { // segment of every read: hits per read counted row by row, then scanned over the reads
            const int nr = ablock->nreads;
            R.rs_start.ensure((size_t) nr + 2, "first hit of each read");
            g_rowdst.ensure((size_t) nhits + 1, "place of each merge row in its read's segment");
            R.rs_stats.ensure(8, "read segment statistics");
            HIP_CHECK(hipMemsetAsync(R.rs_start.p, 0, sizeof(uint32_t) * ((size_t) nr + 2), stream()));
            HIP_CHECK(hipMemsetAsync(R.rs_stats.p, 0, sizeof(uint32_t) * 8, stream()));
            hipLaunchKernelGGL(k_row_place, dim3(nb), dim3(256), 0, stream(), (const MergeEntry *) entries.p, nent,
                               (const uint32_t *) esz.p, (const uint32_t *) eoff.p, (const uint32_t *) A.vals, MA,
                               R.rs_start.p, g_rowdst.p);
            exclusive_scan_u32(R.rs_start.p, nr, st.sort.bsum, R.rs_start.p + nr);
            // the reads by segment size: one list per workgroup size of k_read_sort ("read_sort" test option = 2: the
            // largest workgroups take everything above 128 hits; = 3: segments above 512 hits count as too large)
            R.rs_lists.ensure(4 * (size_t) nr + 16, "reads by segment size");
            // While the deferred alignment kernel of the previous block runs (20 ms, its workgroups persistent, LDS pebbles
            // on every CU) a 1024-thread workgroup with 140 KB of LDS finds no CU to run on until that kernel ENDS, and the
            // main stream stands still behind it (the trace: 10-17 ms for a launch that takes 0.06): the reads of that
            // class then go through global memory with small workgroups, which fit anywhere ("read_sort" test option = 6:
            // always)
            g_rs_small = report_in_flight() || st.opt_read_sort == 6 || st.opt_read_sort == 7;      // (7 = 3 + 6)
            const uint32_t lim2 = (st.opt_read_sort == 3 || st.opt_read_sort == 7) ? 512u
                                  : (g_rs_small ? (uint32_t) RS_T1 * RS_E : (uint32_t) RS_T2 * RS_E);
            const uint32_t lim0 = std::min(lim2, (st.opt_read_sort == 2) ? 128u : (uint32_t) RS_T0 * RS_E);
            const uint32_t lim1 = std::min(lim2, (st.opt_read_sort == 2) ? 128u : (uint32_t) RS_T1 * RS_E);
            hipLaunchKernelGGL(k_read_classes, dim3(ceil_div(nr, 256)), dim3(256), 0, stream(), (const uint32_t *) R.rs_start.p, nr,
                               lim0, lim1, lim2, R.rs_lists.p, R.rs_stats.p);
            g_rlim = lim2;
            // a class of more than L hits per read holds fewer than nhits / L reads
            g_rgrid[0] = (uint32_t) nr;
            g_rgrid[1] = (uint32_t) std::min<int64_t>(nr, nhits / ((int64_t) lim0 + 1) + 1);
            g_rgrid[2] = (uint32_t) std::min<int64_t>(nr, nhits / ((int64_t) lim1 + 1) + 1);
            g_rgrid[3] = (uint32_t) std::min<int64_t>(nr, nhits / ((int64_t) lim2 + 1) + 1);
            rstart = R.rs_start.p; rowdst = g_rowdst.p;
          }
        hipLaunchKernelGGL(k_merge_expand_tiled, dim3(ceil_div(nhits, EX_TILE)), dim3(256), 0, stream(),
                         (const MergeEntry *) entries.p, nent, (const uint32_t *) eoff.p, (const uint32_t *) tfirst.p,
                         (const uint32_t *) A.vals,
                         (const uint32_t *) Bx.vals, MA, MB, bshift, pshift, hitmode, passor, hk, hv, rstart, rowdst);
        if (by_read)
          { // every read is ordered by a workgroup of its own: in LDS, or -- a read with more hits than the LDS of a
            // workgroup takes: a tandem array -- through global memory (k_read_sort_big).  The workgroups find the
            // length of their list on the device; the host only asks for statistics and experiments
            finish_tick_if_due(nfilt);      // (the expansion is queued: the oldest batch's finish moves on a step)
            static const bool want = getenv("DMH_STATS") != nullptr || getenv("DMH_READ_SORT_FORK") != nullptr;
            if (want)
              { HostTimer hw("wait_read_classes");
                copy_d2h(g_rcnt, R.rs_stats.p, 5 * sizeof(uint32_t), 1);
              }
            else
              { for (int q = 0; q < 4; q++) g_rcnt[q == 3 ? 4 : q] = g_rgrid[q];
                g_rcnt[2] = std::min(g_rcnt[2], 2048u);
              }
            R.read_sorted = true;
          }
      }
    else
      { hipLaunchKernelGGL(k_merge_expand, dim3(nb), dim3(256), 0, stream(), (const MergeEntry *) entries.p, nent,
                           (const uint32_t *) eoff.p, (const uint32_t *) esz.p, (const uint32_t *) A.vals,
                           (const uint32_t *) Bx.vals, MA, MB, bshift, pshift, hitmode, passor, hk, hv, ebig.p, mctl.p + 1);
        // the entries of more than BIG_ENTRY pairs, a wavefront each: their number stays on the device (the launch is
        // sized for the most there can be of them: nhits / BIG_ENTRY)
        const int64_t maxbig = std::min<int64_t>(nent, nhits / (int64_t) BIG_ENTRY + 1);
        hipLaunchKernelGGL(k_merge_expand_big, dim3(std::min<int64_t>(ceil_div(maxbig * 64, 256), 65536)), dim3(256), 0, stream(),
                           (const MergeEntry *) entries.p, (const uint32_t *) ebig.p, (const unsigned int *) (mctl.p + 1),
                           (const uint32_t *) eoff.p, (const uint32_t *) A.vals, (const uint32_t *) Bx.vals, MA, MB, bshift, pshift,
                           hitmode, passor, hk, hv);
      }
    HIP_CHECK(hipGetLastError());
    if (!by_read) finish_tick_if_due(nfilt);
    static const bool check = getenv("DMH_CHECK_EMIT") != nullptr;
    if (check)
      { copy_d2h(&total, eoff.p + nent, sizeof(uint32_t));
        if ((int64_t) total != nhits)
          fatal("Match_Filter: emitted %u hits but the histogram promised %lld", total, (long long) nhits);
      }
  }
  (void) nbig;
  R.pass[R.npass].bbase = bblock->tfirst;
  R.pass[R.npass].comp  = comp;
  R.npass += 1;
  R.nhits += nhits;
  if (!R.passbits)
    { // (Experiment, DMH_DEFER_CLOSE=1 or the "defer_close" test option = 1.  Per-read route: the close stops behind the
      //  per-read sort -- queued on the tail stream, its counters on their way to page-locked memory -- and is taken up by
      //  the NEXT pass once its match kernels are queued, so that the sort runs next to them.  Measured: 202 against 146 ms
      //  per read block.  The sort takes twice as long next to the match kernels (LDS-bound workgroups of 35-70 KB among
      //  the match kernel's waves), and the host -- which then waits for the sort, for the oldest batch's finish and for
      //  the hit sums one after the other with only the match kernels queued -- leaves the main stream idle most of a pass.)
      static const bool env_dc = getenv("DMH_DEFER_CLOSE") != nullptr && atoi(getenv("DMH_DEFER_CLOSE")) != 0;
      const bool defer_close = (env_dc || st.opt_defer_close > 0) && R.word && R.read_sorted && !st.capture_seeds && !VERBOSE &&
                               flight_keep() > 0 && st.opt_defer_close != 0 && tail_stream() != main_stream() && R.nhits > 0;
      // (The light version, DMH_DEFER_CLOSE=2: the close stops BEHIND the round trip for the cluster counters --
      //  everything up to there on the main stream, as ever -- and the next pass queues the second half on the tail stream
      //  right after its own match kernels, so that the main stream gets those 0.5 ms earlier (the host takes that long to
      //  queue the forty launches of a second half).  Measured: 158 against 146 ms per read block -- the second half then
      //  runs next to the expansion and the per-read sort instead of next to the match kernels.)
      static const bool env_dl = getenv("DMH_DEFER_CLOSE") != nullptr && atoi(getenv("DMH_DEFER_CLOSE")) == 2;
      const bool defer_light = !defer_close && env_dl && st.opt_defer_close != 0 && R.word && R.read_sorted &&
                               !st.capture_seeds && !VERBOSE && flight_keep() > 0 && tail_stream() != main_stream() && R.nhits > 0;
      const bool defer_fused = !defer_close && !defer_light && R.fused && R.word && R.read_sorted && !st.capture_seeds && !VERBOSE &&
                               flight_keep() > 0 && tail_stream() != main_stream() && R.nhits > 0 && defer_fused_close();
      close_batch(R, defer_close ? 1 : (defer_light ? 3 : (defer_fused ? 4 : 0)));
      if (R.word) bg_pump_tail(1);      // (a stage of the next read block's index, behind this pass's second half)
    }
  else
    close_pending();                    // (a pass that is collected, not closed: nothing would take the pending close up)
  if (g_finish_in_close)
    { // (the pass did not go through the per-read route's close, which takes this up behind its kernels: here then,
      //  behind whatever this pass has queued)
      g_finish_in_close = false;
      const int keep2 = flight_keep();
      while ((int) g_inflight.size() > keep2) *nfilt += finish_oldest();
    }
  // batches in flight: under -v every call reports its own candidate counts, and the stage-level API looks at the
  // lists after every call -- then nothing stays in flight
  static const int env_fl = getenv("DMH_IN_FLIGHT") ? atoi(getenv("DMH_IN_FLIGHT")) : IN_FLIGHT;
  const int keep = (VERBOSE || st.capture_seeds || st.opt_in_flight == 0) ? 0 : std::min(g_flight_cap, std::max(0, env_fl));
  while ((int) g_inflight.size() > (keep > 0 ? keep + 1 : 0)) *nfilt += finish_oldest();
  // nothing in flight, nothing being collected: the next pass goes where buffers of its size lie already (with every batch
  // finished at once the ring otherwise alternates between two slots -- 57 GB each against a one-family reference, and
  // the idle one is what the memory reclaim then frees and the next pass allocates again, every pass)
  if (g_inflight.empty() && g_pending_close < 0 && !g_ring[g_cur].open)
    { int warm = g_cur;
      for (int i = 0; i < NRING; i++)
        if (g_ring[i].keys.cap > g_ring[warm].keys.cap) warm = i;
      g_cur = warm;
    }
  return nhits;
  };        // submit

  // Hits are numbered in 32 bits behind the merge (group starts, links, chain ends); the reference counts them in 64
  // (map.c:876, 2899).  Its -M rule keeps a pass under (M - DBs) / 32 hits -- below 2^32 for -M up to 137 GB --, and a
  // 375 Mbp block pair reaches 10^9 only against a one-family reference; a pass beyond the range is made in RANGES OF
  // READS, each a batch of its own: the hits of different reads never meet before the domination rule, which is per read
  // (map.c:1668-1766), so the candidate lists come out as from one pass ("hit_limit" test option: the split at a smaller
  // number).
  const int64_t hlim = st.opt_hit_limit > 0 ? (int64_t) st.opt_hit_limit : (int64_t) 0xfffffff0ll;
  if (pass_hits < hlim) return submit(0, ablock->nreads, pass_hits);
  { const int nr = ablock->nreads;
    if (!(A.posbits > 0 && limit <= (int64_t) MAXGRAM))
      fatal("Match_Filter: %lld seed hits in one block pair exceed the 32-bit slot range; use smaller blocks or -M", (long long) pass_hits);
    // hits per read: the rows counted, not listed
    static DevBuf<uint32_t> cnt_all;
    cnt_all.ensure(2 * ((size_t) nr + 2), "hits per read of an oversize pass");
    HIP_CHECK(hipMemsetAsync(cnt_all.p, 0, sizeof(uint32_t) * 2 * ((size_t) nr + 2), stream()));
    hipLaunchKernelGGL(k_row_list, dim3(ceil_div(ntiles * 64, 256)), dim3(256), 0, stream(), (const MergeEntry *) tentries.p,
                       (const uint32_t *) tcount.p, ntiles, nent, limit, (const uint32_t *) A.vals, A.posbits,
                       (const int64_t *) A.d_boff, cnt_all.p, cnt_all.p + nr + 2, (uint64_t *) nullptr, 0u, (uint32_t) nr);
    std::vector<uint32_t> hc((size_t) nr);
    copy_d2h(hc.data(), cnt_all.p, sizeof(uint32_t) * (size_t) nr);
    const int64_t lim = std::max<int64_t>(1, std::min<int64_t>(hlim - 1, (int64_t) 0x7ffffff0ll));
    int64_t total = 0, acc = 0;
    int     first = 0, nranges = 0;
    for (int r = 0; r < nr; r++)
      { if ((int64_t) hc[r] > lim)
          fatal("Match_Filter: read %d alone meets %u seed hits in one block pair, more than a batch takes (%lld)", r, hc[r], (long long) lim);
        if (acc + (int64_t) hc[r] > lim)
          { st.capture_append = nranges > 0;
            total += submit(first, r, acc);
            nranges += 1;
            first = r; acc = 0;
          }
        acc += hc[r];
      }
    if (acc > 0) { st.capture_append = nranges > 0; total += submit(first, nr, acc); nranges += 1; }
    st.capture_append = false;
    if (total != pass_hits) fatal("Match_Filter: the ranges of an oversize pass hold %lld hits, the pass %lld", (long long) total, (long long) pass_hits);
    if (getenv("DMH_STATS")) fprintf(stderr, "dmh: a pass of %lld hits made in %d ranges of reads\n", (long long) pass_hits, nranges);
    return total;
  }
#undef R
}

static bool    g_finishing = false;          // inside chain_finish (its own allocations must not finish the next batch)
static int     g_finishing_slot = -1;        // ... and the ring slot it works on: busy for filter_reclaim although it has left g_inflight

// One step of finishing the oldest batch (chain_finish_step): called where a pass has just queued main-stream work, so
// that the finish stream's small kernels run while the host is busy elsewhere.  Returns true when that batch is done.
static void finish_retire(int x);
static bool finish_tick(int64_t *nfilt)
{ if (g_inflight.empty()) return false;
  const int x = g_inflight.front();
  MergeResult &R = g_ring[x];
  g_finishing = true;
  g_finishing_slot = x;
  bool done;
  { HostTimer hf("filter_finish");
    done = chain_finish_step(R, nfilt);
  }
  g_finishing = false;
  g_finishing_slot = -1;
  if (done)
    { g_inflight.pop_front();
      finish_retire(x);
    }
  return done;
}

// The finishes of the batches in flight as a pipeline: every batch takes ONE step per pass -- the step whose input
// (the pile walks; the count of chain ends; the candidate counters) arrived during the pass before -- so the host
// queues three steps of three batches on the finish stream and waits for none.  Steps b and c work on the candidate
// lists of the READS block and are taken by the oldest batch only (step b of a batch reads what step c of the one
// before it has put in place); step a (chain ends per group) has no such order.  A batch whose piles are still being
// walked stays where it is -- and everything younger behind it -- until the bound on the batches in flight forces it
// (finish_oldest, which waits).
static void finish_pipeline(int64_t *nfilt)
{ size_t i = 0;
  g_finishing = true;
  while (i < g_inflight.size())
    { const int x = g_inflight[i];
      MergeResult &R = g_ring[x];
      const int stage = chain_finish_stage(R);
      if ((stage == 0 || i == 0) && chain_finish_ready(R))
        { bool done;
          { HostTimer hf("filter_finish");
            done = chain_finish_step(R, nfilt);
          }
          if (done)
            { g_inflight.erase(g_inflight.begin() + (long) i);      // (i == 0: only steps b and c end a finish)
              finish_retire(x);
              continue;
            }
        }
      i += 1;
    }
  g_finishing = false;
}

// batches the filter keeps in flight before it starts finishing the oldest
static int flight_keep()
{ State &st = S();
  static const int env_fl = getenv("DMH_IN_FLIGHT") ? atoi(getenv("DMH_IN_FLIGHT")) : IN_FLIGHT;
  return (VERBOSE || st.capture_seeds || st.opt_in_flight == 0) ? 0 : std::min(g_flight_cap, std::max(0, env_fl));
}

// a step of the oldest batch's finish when one is due: more batches in flight than are kept, or that batch under way
static void finish_tick_if_due(int64_t *nfilt)
{ static const bool off = getenv("DMH_FINISH_STEPS") == nullptr;      // (opt-in: measured 187.2 against 182.6 ms per read block with one whole finish behind the match kernels)
  if (off || g_inflight.empty()) return;
  if ((int) g_inflight.size() > flight_keep() || chain_finish_started(g_ring[g_inflight.front()])) (void) finish_tick(nfilt);
}

static int64_t finish_oldest()
{ const int x = g_inflight.front();
  g_inflight.pop_front();
  MergeResult &R = g_ring[x];
  g_finishing = true;
  g_finishing_slot = x;
  int64_t nfilt;
  { HostTimer hf("filter_finish");
    nfilt = chain_finish(R, R.ablock);
  }
  g_finishing = false;
  g_finishing_slot = -1;
  finish_retire(x);
  return nfilt;
}

static void finish_retire(int x)
{ MergeResult &R = g_ring[x];
  if (finish_stream() != main_stream())
    { // whatever the main stream does from here on (the slot's buffers are about to be used again, Reporter reads the
      // candidate lists) comes after the finish stream's last kernel
      static hipEvent_t fin_done = nullptr;
      if (fin_done == nullptr) HIP_CHECK(hipEventCreateWithFlags(&fin_done, hipEventDisableTiming));
      HIP_CHECK(hipEventRecord(fin_done, finish_stream()));
      HIP_CHECK(hipStreamWaitEvent(main_stream(), fin_done, 0));
    }
  R.nhits = 0;
  R.npass = 0;
}

// Device memory ran out (dev_alloc): the batches in flight are finished -- in order, as always -- and every ring slot
// that is neither being collected nor in flight gives its buffers back.  A pass of 10^9 hits needs some 80 bytes
// per hit; three such batches in flight next to 120 GB of indices do not fit 288 GB, one at a time does.
static void release_idle_slots(bool &got)
{ for (int i = 0; i < NRING; i++)
    { bool busy = (i == g_cur) || (i == g_finishing_slot) || (i == g_pending_close) || (i == g_closing_slot);       // (chain_finish still reads the slot it is finishing)
      for (int y : g_inflight) busy = busy || (y == i);
      if (busy) continue;
      MergeResult &R = g_ring[i];
      if (R.keys.p || R.alt_keys.p || R.vals.p || R.alt_vals.p) got = true;
      R.keys.release(); R.alt_keys.release(); R.vals.release(); R.alt_vals.release();
      R.gstart.release(); R.gkey.release(); R.cog.release(); R.ocfirst.release(); R.glast.release();
      R.biglist.release(); R.pilelist.release(); R.medlist.release();
      R.rs_start.release(); R.rs_stats.release(); R.rs_lists.release(); R.rs_ncl.release(); R.rs_lcl.release();
      R.rs_wcnt.release(); R.rs_wlist[0].release(); R.rs_wlist[1].release(); R.pmax.release(); R.rs_lcnt.release();
      chain_release(R);
    }
}

static bool filter_reclaim()
{ bool got = false;
  // first what nobody is using -- ring slots that are neither being collected nor in flight, the k-mer sorts' partner
  // buffers --: finishing the batches in flight takes memory itself (a gigabyte of scan results per 2 * 10^8 groups), and
  // the allocation that failed may have left none
  const int flying = (int) g_inflight.size();
  if (!g_finishing && !g_closing)
    { HIP_CHECK(hipDeviceSynchronize());
      release_idle_slots(got);
      if (sort_scratch_release()) got = true;
      if (chain_drop_unused()) got = true;
    }
  if (!g_finishing && !g_closing) close_pending();
  if (!g_finishing)
    while (!g_inflight.empty())
      { S().tfilt += finish_oldest();
        got = true;
      }
  // (what was in flight next to the batch that did not fit is one batch too many)
  if (g_flight_cap > 0 && !S().simulated_oom) g_flight_cap = std::max(0, std::min(g_flight_cap - 1, flying - 1));
  g_reclaimed_in_block = true;
  HIP_CHECK(hipDeviceSynchronize());          // (buffers are about to go back to the driver)
  const double live_before = dev_live_gb();
  release_idle_slots(got);
  static const bool stats = getenv("DMH_STATS") != nullptr;
  if (stats)
    { int nbusy = 0;
      for (int i = 0; i < NRING; i++)
        { bool busy = (i == g_cur) || (i == g_finishing_slot) || (i == g_pending_close) || (i == g_closing_slot);
          for (int y : g_inflight) busy = busy || (y == i);
          nbusy += busy ? 1 : 0;
        }
      fprintf(stderr, "dmh: out of device memory: batches in flight finished (%d left, finishing %d, closing %d), idle ring slots "
              "released (%d busy, collecting slot %d, pending %d), in-flight cap now %d; %.1f -> %.1f GB held\n", (int) g_inflight.size(), (int) g_finishing,
              (int) g_closing, nbusy, g_cur, g_pending_close, g_flight_cap, live_before, dev_live_gb());
    }
  return got;
}

int64_t filter_flush()
{ close_pending();
  MergeResult &R = filter_batch();
  if (R.open) close_batch(R);
  int64_t nfilt = 0;
  while (!g_inflight.empty()) nfilt += finish_oldest();
  return nfilt;
}

// Reporter of one READS block: its open batch is closed, its batches in flight are finished (they are the oldest: the
// filter finishes in order); what the following block has queued meanwhile stays as it is
int64_t filter_flush_block(int cslot)
{ int64_t nfilt = 0;
  bool did = false;
  close_pending();
  { MergeResult &R = filter_batch();
    if (R.open && R.cslot == cslot) { close_batch(R); did = true; }
  }
  while (true)
    { bool any = false;
      for (int y : g_inflight) any = any || (g_ring[y].cslot == cslot);
      if (!any) break;
      nfilt += finish_oldest();
      did = true;
    }
  // one transient shortage must not cost every later block its batches in flight: a block that went through without one
  // gives a batch back (a cap that is too high again only costs another reclaim)
  if (did)
    { if (!g_reclaimed_in_block && g_flight_cap < IN_FLIGHT) g_flight_cap += 1;
      g_reclaimed_in_block = false;
    }
  return nfilt;
}

// the batches of ONE block are dropped (its store is about to be given to another block)
void filter_discard_block(int cslot)
{ close_pending();
  bool any = false;
  for (int y : g_inflight) any = any || (g_ring[y].cslot == cslot);
  MergeResult &R = filter_batch();
  if (R.open && R.cslot == cslot) { R.open = false; R.nhits = 0; R.npass = 0; }
  if (!any) return;
  join_side_streams();
  std::deque<int> keep;
  for (int y : g_inflight)
    if (g_ring[y].cslot == cslot) { chain_abandon(g_ring[y]); g_ring[y].nhits = 0; g_ring[y].npass = 0; }
    else keep.push_back(y);
  g_inflight.swap(keep);
}

void filter_discard()
{ close_pending();
  join_side_streams();             // (their kernels may still read the batches that are dropped here)
  for (int y : g_inflight) chain_abandon(g_ring[y]);       // ... and the pile stream's, which the join does not cover
  g_inflight.clear();
  for (int i = 0; i < NRING; i++) { g_ring[i].open = false; g_ring[i].nhits = 0; g_ring[i].npass = 0; }
  g_flight_cap = IN_FLIGHT;                    // (dmh_reset / a new run: what memory allowed before says nothing any more)
  g_reclaimed_in_block = false;
}

// sort and groups for everything the batch holds, then its chain kernels are queued (chain_submit) and the batch
// joins the ones in flight
static void close_pending()
{ if (g_pending_close < 0 || g_closing) return;
  const int x = g_pending_close;
  g_pending_close = -1;
  g_closing = true;
  g_closing_slot = x;
  close_batch(g_ring[x], 2);
  g_closing = false;
  g_closing_slot = -1;
}

// phase 0: everything.  Per-read route only: phase 1 = up to the per-read sort and the copy of its counters (queued on
// the tail stream; the batch becomes g_pending_close and the next one is collected in another slot), phase 2 = the rest.
static void close_batch(MergeResult &R, int phase)
{ State &st = S();
  std::unique_ptr<StreamScope> tail_scope;       // set where the pass moves to the tail stream (per-read route)
  std::unique_ptr<HostTimer> post_sync;          // host time from the counter round trip to the end of the close
  if (phase != 2 && phase != 4) close_pending(); // (one close at a time; phase 4 takes the pending one up behind its own per-read kernels)
  const int64_t nhits = R.nhits;
  const DAZZ_DB *ablock = R.ablock;
  R.open = false;
  if (nhits == 0) { R.npass = 0; return; }
  HostTimer ht("filter_close");
  const bool cluster = R.clustered;
  const int  qbits = R.qbits;
  const int  lowbits = cluster ? qbits : R.pbits;
  const int  pshift = lowbits;
  R.alt_keys.ensure(nhits, "seed sort buffer");
  if (!R.word) R.alt_vals.ensure(nhits, "seed sort buffer");

  // --- seed sort (map.c:3110-3126)
  uint64_t *k0 = R.keys.p, *k1 = R.alt_keys.p;
  uint32_t *v0 = R.vals.p, *v1 = R.alt_vals.p;
  static const int env_cb = getenv("DMH_CLUSTER_CB") ? atoi(getenv("DMH_CLUSTER_CB")) : CLUSTER_CB;
  const int cb = cluster ? std::min(env_cb, qbits) : 0;              // (the bits below cb of bpos are not sorted on)
  const bool rs = R.word && R.read_sorted;             // hits lie read by read: ordered per read in LDS (dmh_readsort.h)
  if (rs)
    ;
  else if (R.word)
    radix_sort_pairs(&k0, nullptr, &k1, nullptr, nhits, R.pbits + cb, qbits - cb + R.bbits + R.abits, st.sort, "seed_sort");
  else
    radix_sort_pairs(&k0, &v0, &k1, &v1, nhits, cb, lowbits + R.bbits + R.passbits + R.abits - cb, st.sort, "seed_sort");

  // --- groups
  static DevBuf<uint32_t> flag;
  if (!R.word) flag.ensure(nhits + 1, "group flags");
  if (!cluster)
    { StageTimer tm("seed_groups", 24.0 * nhits);
      int nb = ceil_div(nhits, 256);
      hipLaunchKernelGGL(k_group_flag, dim3(nb), dim3(256), 0, stream(), (const uint64_t *) k0, nhits, pshift, flag.p);
      exclusive_scan_u32(flag.p, nhits, st.sort.bsum, flag.p + nhits);
      uint32_t ng = 0;
      copy_d2h(&ng, flag.p + nhits, sizeof(uint32_t), 1);
      R.ngroups = ng;
      R.gstart.ensure((size_t) ng + 1, "group starts");
      R.gkey.ensure((size_t) ng + 1, "group keys");
      hipLaunchKernelGGL(k_group_write, dim3(nb), dim3(256), 0, stream(), (const uint64_t *) k0, nhits, pshift,
                         (const uint32_t *) flag.p, R.gstart.p, ng, R.gkey.p);
      hipLaunchKernelGGL(k_pack_ab, dim3(nb), dim3(256), 0, stream(), (const uint64_t *) k0, (const uint32_t *) v0, nhits,
                         ((uint64_t) 1 << R.pbits) - 1, k1);
      std::swap(k0, k1);
    }
  else
    { uint32_t ncl = 0;
      const bool local_sort = R.word || local_cluster_sort(R.pbits);
      if (rs)
        { const int nr = ablock->nreads;
          DevBuf<uint64_t> &lcl = R.rs_lcl;
          DevBuf<uint32_t> &rncl = R.rs_ncl;
          if (phase == 1)
            { // the sort and the copy of its counters go to the tail stream, behind the expansion just queued
              static hipEvent_t ev_fork1 = nullptr;
              if (ev_fork1 == nullptr) HIP_CHECK(hipEventCreateWithFlags(&ev_fork1, hipEventDisableTiming));
              HIP_CHECK(hipEventRecord(ev_fork1, stream()));
              HIP_CHECK(hipStreamWaitEvent(tail_stream(), ev_fork1, 0));
              tail_scope.reset(new StreamScope(tail_stream()));
            }
          else if (phase == 2)
            { if (R.h_known[7] != 0u) HIP_CHECK(hipStreamWaitEvent(tail_stream(), R.ev_known, 0));      // (the light version, phase 4)
              tail_scope.reset(new StreamScope(tail_stream()));       // (else: where phase 1 left off)
            }
          if (phase != 2)
          {
          lcl.ensure((size_t) nhits + 1, "cluster records per read");
          rncl.ensure((size_t) nr + 2, "clusters per read");
          HIP_CHECK(hipMemsetAsync(rncl.p, 0, sizeof(uint32_t) * ((size_t) nr + 2), stream()));
          ReadSortArgs RA;
          RA.in = k0; RA.out = k1; RA.rstart = R.rs_start.p;
          // the coarse buckets of this route (64 kb: a locus rarely lies across two) and its radix passes: the sort key
          // apos | coarse key << pbits cut into equal digits of at most RS_DB bits
          const int rcb = std::min(rs_coarse_bits(), qbits), kbits = qbits - rcb + R.bbits;
          RA.pbits = R.pbits; RA.qbits = qbits; RA.bbits = R.bbits; RA.cb = rcb;
          RA.kbits = kbits;
          RA.npass = 0;
          { const int bits = R.pbits + kbits, nd = (bits + RS_DB - 1) / RS_DB;
            for (int k = 0, at = 0; k < nd; k++)
              { const int wdt = (bits - at + (nd - k) - 1) / (nd - k);
                RA.sh[RA.npass] = (uint8_t) at; RA.wd[RA.npass] = (uint8_t) wdt;
                RA.npass += 1; at += wdt;
              }
          }
          // ("read_sort" test option = 5: ranks from ballots, as on a device whose LDS does not serve atomics in lane order)
          static const bool env_ballot = getenv("DMH_RS_BALLOT") != nullptr && atoi(getenv("DMH_RS_BALLOT")) != 0;      // (experiments)
          RA.ordered = (lds_atomics_ordered() && st.opt_read_sort != 5 && !env_ballot) ? 1 : 0;
          RA.lcl = lcl.p; RA.rncl = rncl.p;
          // the clusters are counted by size while they are found, in the chain kernels' classes and the cluster sorts'
          chain_group_limits(R);
          R.rs_wcnt.ensure(8, "cluster counters");
          HIP_CHECK(hipMemsetAsync(R.rs_wcnt.p, 0, 8 * sizeof(unsigned int), stream()));
          RA.counts = R.rs_wcnt.p;
          RA.medium = (uint32_t) CS_MEDIUM;
          RA.big_min = R.small_hi;
          RA.pile_min = R.pile_min > 0 ? R.pile_min : 0xffffffffu;
          RA.rows = nullptr; RA.rowbase = nullptr; RA.nrow = nullptr; RA.bvals = nullptr; RA.bposbits = 0; RA.rowsort = 0;
          RA.from = nullptr; RA.sums = nullptr; RA.lcnt = nullptr; RA.K = 0; RA.hithr = 0; RA.simple_max = 0;
          if (R.fused)
            { RA.rows = R.f_rows; RA.rowbase = R.f_rowbase; RA.nrow = R.rs_start.p + nr + 2;
              RA.bvals = R.f_bvals; RA.bposbits = R.f_bposbits;
              // plain chains are scored by the same kernel when k_chain_simple would run for this batch: the batch's link
              // and chain-end buffers are made now, the per-cluster results go through lcnt to k_cluster_compact
              chain_prepare_hits(R, &RA.from, &RA.sums);
              if (chain_simple_wanted(R))
                { R.rs_lcnt.ensure((size_t) nhits + 1, "chain ends per cluster of a read");
                  RA.lcnt = R.rs_lcnt.p;
                  RA.K = st.params.Kmer; RA.hithr = HITMIN * st.params.Kmer;
                  RA.simple_max = R.lane_max;
                }
            }
          // Fused route: with the rows in apos order the hits start out ordered by (apos, contig, bpos), and the radix passes
          // cover the coarse key alone -- when the bit map over the apos values fits the digit counters' place of the
          // workgroup size at hand (10 bytes per 64 values in 16 T bytes: reads of up to 16 k / 32 k / 64 k bases)
          ReadSortArgs RB = RA;                      // the launch arguments with that configuration
          auto rowsort_fits = [&](int T) { return R.fused && ((((size_t) 1 << R.pbits) + 63) / 64) * 10 + 16 <= (size_t) 16 * T; };
          if (R.fused)
            { static const bool env_rs2 = getenv("DMH_ROW_SORT") == nullptr || atoi(getenv("DMH_ROW_SORT")) != 0;
              RB.rowsort = env_rs2 ? 1 : 0;
              RB.npass = 0;
              const int nd = (kbits + RS_DB - 1) / RS_DB;
              for (int k = 0, at = 0; k < nd; k++)
                { const int wdt = (kbits - at + (nd - k) - 1) / (nd - k);
                  RB.sh[RB.npass] = (uint8_t) (R.pbits + at); RB.wd[RB.npass] = (uint8_t) wdt;
                  RB.npass += 1; at += wdt;
                }
            }
          auto args_for = [&](int T, const ReadSortArgs &base) -> ReadSortArgs
          { ReadSortArgs X = (RB.rowsort && rowsort_fits(T)) ? RB : RA;
            X.list = base.list; X.count = base.count;
            return X;
          };
          static bool attr = false;
          if (!attr)
            { attr = true;
              HIP_CHECK(hipFuncSetAttribute((const void *) k_read_sort<RS_T1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int) read_sort_lds<RS_T1>()));
              HIP_CHECK(hipFuncSetAttribute((const void *) k_read_sort<RS_T1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int) read_sort_lds<RS_T1>()));
              HIP_CHECK(hipFuncSetAttribute((const void *) k_read_sort<RS_T2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int) read_sort_lds<RS_T2>()));
              HIP_CHECK(hipFuncSetAttribute((const void *) k_read_sort<RS_T2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int) read_sort_lds<RS_T2>()));
              HIP_CHECK(hipFuncSetAttribute((const void *) k_read_sort<RS_T1, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int) read_sort_lds<RS_T1>()));
              HIP_CHECK(hipFuncSetAttribute((const void *) k_read_sort<RS_T1, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int) read_sort_lds<RS_T1>()));
              HIP_CHECK(hipFuncSetAttribute((const void *) k_read_sort<RS_T2, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int) read_sort_lds<RS_T2>()));
              HIP_CHECK(hipFuncSetAttribute((const void *) k_read_sort<RS_T2, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int) read_sort_lds<RS_T2>()));
            }
          // clusters ordered on 32-bit words when apos leaves the room ("read_sort" test option = 4: on 64-bit words)
          const bool k32 = R.pbits + rcb + 6 <= 32 && st.opt_read_sort != 4;
          { // (8 B read + 8 B written per hit).  The three workgroup sizes run one after the other, the largest segments
            // first.  Next to each other (DMH_READ_SORT_FORK: the two larger sizes on streams of their own, to fill the
            // tail of half-empty CUs each launch ends in) they take TWICE as long -- 75.7 against 36.7 ms per read block:
            // a CU holds four small workgroups or one large one, and mixed they keep each other's LDS from being used
            StageTimer tm("read_sort", 16.0 * nhits);
            static hipStream_t side[2] = {nullptr, nullptr};
            static hipEvent_t  ev_go = nullptr, ev_back[2] = {nullptr, nullptr};
            if (side[0] == nullptr)
              { for (int i = 0; i < 2; i++)
                  { HIP_CHECK(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
                    HIP_CHECK(hipEventCreateWithFlags(&ev_back[i], hipEventDisableTiming));
                  }
                HIP_CHECK(hipEventCreateWithFlags(&ev_go, hipEventDisableTiming));
              }
            static const bool serial = getenv("DMH_READ_SORT_FORK") == nullptr;
            const bool fork = !serial && (g_rcnt[2] > 0 || g_rcnt[1] > 0) && g_rcnt[0] > 0;
            if (fork) HIP_CHECK(hipEventRecord(ev_go, stream()));
            bool used[2] = {false, false};
            if (g_rcnt[4] > 0)
              { // (first: the few workgroups of this launch run longest)
                RA.list = R.rs_lists.p + 3 * (size_t) nr; RA.count = R.rs_stats.p + 4;
                if (g_rs_small)
                  hipLaunchKernelGGL((k_read_sort_big<RS_T0>), dim3(std::min<uint32_t>(g_rcnt[4], 2048u)), dim3(RS_T0), 0, stream(), RA);
                else
                  hipLaunchKernelGGL((k_read_sort_big<RS_T2>), dim3(std::min<uint32_t>(g_rcnt[4], 1024u)), dim3(RS_T2), 0, stream(), RA);
              }
            if (g_rcnt[2] > 0 && !g_rs_small)
              { hipStream_t q = fork ? side[0] : stream();
                if (fork) { HIP_CHECK(hipStreamWaitEvent(q, ev_go, 0)); used[0] = true; }
                RA.list = R.rs_lists.p + 2 * (size_t) nr; RA.count = R.rs_stats.p + 2;
                if (R.fused && k32) hipLaunchKernelGGL((k_read_sort<RS_T2, true, true>), dim3(g_rcnt[2]), dim3(RS_T2), read_sort_lds<RS_T2>(), q, args_for(RS_T2, RA));
                else if (R.fused)   hipLaunchKernelGGL((k_read_sort<RS_T2, false, true>), dim3(g_rcnt[2]), dim3(RS_T2), read_sort_lds<RS_T2>(), q, args_for(RS_T2, RA));
                else if (k32) hipLaunchKernelGGL((k_read_sort<RS_T2, true>), dim3(g_rcnt[2]), dim3(RS_T2), read_sort_lds<RS_T2>(), q, RA);
                else     hipLaunchKernelGGL((k_read_sort<RS_T2, false>), dim3(g_rcnt[2]), dim3(RS_T2), read_sort_lds<RS_T2>(), q, RA);
              }
            if (g_rcnt[1] > 0)
              { hipStream_t q = fork ? side[1] : stream();
                if (fork) { HIP_CHECK(hipStreamWaitEvent(q, ev_go, 0)); used[1] = true; }
                RA.list = R.rs_lists.p + (size_t) nr; RA.count = R.rs_stats.p + 1;
                if (R.fused && k32) hipLaunchKernelGGL((k_read_sort<RS_T1, true, true>), dim3(g_rcnt[1]), dim3(RS_T1), read_sort_lds<RS_T1>(), q, args_for(RS_T1, RA));
                else if (R.fused)   hipLaunchKernelGGL((k_read_sort<RS_T1, false, true>), dim3(g_rcnt[1]), dim3(RS_T1), read_sort_lds<RS_T1>(), q, args_for(RS_T1, RA));
                else if (k32) hipLaunchKernelGGL((k_read_sort<RS_T1, true>), dim3(g_rcnt[1]), dim3(RS_T1), read_sort_lds<RS_T1>(), q, RA);
                else     hipLaunchKernelGGL((k_read_sort<RS_T1, false>), dim3(g_rcnt[1]), dim3(RS_T1), read_sort_lds<RS_T1>(), q, RA);
              }
            if (g_rcnt[0] > 0)
              { RA.list = R.rs_lists.p; RA.count = R.rs_stats.p;
                if (R.fused && k32) hipLaunchKernelGGL((k_read_sort<RS_T0, true, true>), dim3(g_rcnt[0]), dim3(RS_T0), read_sort_lds<RS_T0>(), stream(), args_for(RS_T0, RA));
                else if (R.fused)   hipLaunchKernelGGL((k_read_sort<RS_T0, false, true>), dim3(g_rcnt[0]), dim3(RS_T0), read_sort_lds<RS_T0>(), stream(), args_for(RS_T0, RA));
                else if (k32) hipLaunchKernelGGL((k_read_sort<RS_T0, true>), dim3(g_rcnt[0]), dim3(RS_T0), read_sort_lds<RS_T0>(), stream(), RA);
                else     hipLaunchKernelGGL((k_read_sort<RS_T0, false>), dim3(g_rcnt[0]), dim3(RS_T0), read_sort_lds<RS_T0>(), stream(), RA);
              }
            for (int i = 0; i < 2; i++)
              if (used[i])
                { HIP_CHECK(hipEventRecord(ev_back[i], side[i]));
                  HIP_CHECK(hipStreamWaitEvent(stream(), ev_back[i], 0));
                }
            HIP_CHECK(hipGetLastError());
          }
          StageTimer tm("seed_clusters", 0.0);
          exclusive_scan_u32(rncl.p, nr, (tail_scope ? st.sort_tail : st.sort).bsum, rncl.p + nr);
          HIP_CHECK(hipMemcpyAsync(R.rs_wcnt.p + 2, rncl.p + nr, sizeof(uint32_t), hipMemcpyDeviceToDevice, stream()));
          }
          if (phase == 1)
            { // the counters travel to page-locked memory; the next pass picks the close up (close_pending)
              if (R.h_known == nullptr)
                { HIP_CHECK(hipHostMalloc((void **) &R.h_known, 8 * sizeof(uint32_t), hipHostMallocDefault));
                  HIP_CHECK(hipEventCreateWithFlags(&R.ev_known, hipEventDisableTiming));
                }
              R.h_known[7] = 0u;
              HIP_CHECK(hipMemcpyAsync(R.h_known, R.rs_wcnt.p + 2, 5 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream()));
              HIP_CHECK(hipEventRecord(R.ev_known, stream()));
              tail_scope.reset();
              int slot = -1;
              for (int i = 0; i < NRING; i++)
                if (&g_ring[i] == &R) slot = i;
              g_pending_close = slot;
              for (int i = 0; i < NRING; i++)
                { bool busy = (i == slot);
                  for (int y : g_inflight) busy = busy || (y == i);
                  if (!busy) { g_cur = i; break; }
                }
              return;
            }
          { int64_t nf = 0;                   // (the per-read sort is queued: the third step of the oldest batch's finish)
            finish_tick_if_due(&nf);
            if (g_finish_in_close && (phase == 0 || phase == 4))
              { // the oldest batch's finish that merge_pass has left for here (its front was made by the pass before: the
                // per-read kernels just queued are what the main stream runs while the host waits for the finish stream)
                g_finish_in_close = false;
                const int keep2 = flight_keep();
                while ((int) g_inflight.size() > keep2) nf += finish_oldest();
              }
            st.tfilt += nf;
          }
          if (phase == 4)
            { // the counters travel to page-locked memory behind the per-read kernels (main stream); the pass before is
              // closed now -- its counters have long arrived --, this one by the pass that follows (close_pending)
              if (R.h_known == nullptr)
                { HIP_CHECK(hipHostMalloc((void **) &R.h_known, 8 * sizeof(uint32_t), hipHostMallocDefault));
                  HIP_CHECK(hipEventCreateWithFlags(&R.ev_known, hipEventDisableTiming));
                }
              R.h_known[7] = 2u;
              HIP_CHECK(hipMemcpyAsync(R.h_known, R.rs_wcnt.p + 2, 5 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream()));
              HIP_CHECK(hipEventRecord(R.ev_known, stream()));
              int slot = -1;
              for (int i = 0; i < NRING; i++)
                if (&g_ring[i] == &R) slot = i;
              // (this batch's slot is in no list while the pending close runs: it stays g_cur until then)
              close_pending();
              g_pending_close = slot;
              for (int i = 0; i < NRING; i++)
                { bool busy = (i == slot);
                  for (int y : g_inflight) busy = busy || (y == i);
                  if (!busy) { g_cur = i; break; }
                }
              return;
            }
          // one round trip: clusters, clusters the radix sort has to finish, the chain kernels' big groups and piles
          { HostTimer hw("wait_cluster_counts");
            if (phase == 2 && R.h_known[7] == 1u)
              for (int q = 0; q < 5; q++) g_known[q] = R.h_known[q];
            else if (phase == 2)
              { HIP_CHECK(hipEventSynchronize(R.ev_known));
                for (int q = 0; q < 5; q++) g_known[q] = R.h_known[q];
              }
            else
              copy_d2h(g_known, R.rs_wcnt.p + 2, 5 * sizeof(uint32_t), 1);
          }
          if (phase == 3 && g_known[1] == 0)
            { // stop here: the counters are kept with the batch, the tail stream will start behind what the main stream
              // has done up to now (the event), the next pass takes the close up (close_pending)
              if (R.h_known == nullptr)
                { HIP_CHECK(hipHostMalloc((void **) &R.h_known, 8 * sizeof(uint32_t), hipHostMallocDefault));
                  HIP_CHECK(hipEventCreateWithFlags(&R.ev_known, hipEventDisableTiming));
                }
              for (int q = 0; q < 5; q++) R.h_known[q] = g_known[q];
              R.h_known[7] = 1u;            // (phase 2: nothing to wait for, the tail stream waits for ev_known)
              HIP_CHECK(hipEventRecord(R.ev_known, stream()));
              int slot = -1;
              for (int i = 0; i < NRING; i++)
                if (&g_ring[i] == &R) slot = i;
              g_pending_close = slot;
              for (int i = 0; i < NRING; i++)
                { bool busy = (i == slot);
                  for (int y : g_inflight) busy = busy || (y == i);
                  if (!busy) { g_cur = i; break; }
                }
              return;
            }
          ncl = g_known[0];
          g_have_known = true;
          post_sync.reset(new HostTimer("close_behind_counters"));
          if (phase == 2 && g_known[1] > 0)
            { // a pile across buckets: the radix sort's scratch is the main stream's -- the rest of this close goes there,
              // behind what the tail stream has done so far
              static hipEvent_t ev_join = nullptr;
              if (ev_join == nullptr) HIP_CHECK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
              HIP_CHECK(hipEventRecord(ev_join, stream()));
              tail_scope.reset();
              HIP_CHECK(hipStreamWaitEvent(stream(), ev_join, 0));
            }
          // The rest of the pass -- numbering the clusters, the sorts of the few that span buckets, the pairs, the chain
          // stage -- needs no number from the device any more and nothing the following pass touches (its buffers are
          // the batch's own): it runs on the tail stream, next to that pass's merge kernels.  (Not when a pile across
          // buckets needs the radix sort, whose scratch is the main stream's; DMH_BACK_TAIL=0: off.)
          { static const bool env_bt = getenv("DMH_BACK_TAIL") == nullptr || atoi(getenv("DMH_BACK_TAIL")) != 0;
            if (env_bt && g_known[1] == 0 && !st.capture_seeds && !VERBOSE && tail_stream() != main_stream() &&
                stream() == main_stream())
              { static hipEvent_t ev_fork = nullptr;
                if (ev_fork == nullptr) HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
                HIP_CHECK(hipEventRecord(ev_fork, stream()));
                HIP_CHECK(hipStreamWaitEvent(tail_stream(), ev_fork, 0));
                tail_scope.reset(new StreamScope(tail_stream()));
              }
          }
          { static const bool stats = getenv("DMH_STATS") != nullptr;
            if (stats)
              fprintf(stderr, "dmh: read sort: %lld hits, reads by workgroup size %u %u %u, through global memory %u, largest segment %u, clusters %u\n",
                      (long long) nhits, g_rcnt[0], g_rcnt[1], g_rcnt[2], g_rcnt[4], g_rcnt[3], ncl);
          }
          R.ngroups = ncl;
          StageTimer tmc("seed_clusters", 0.0);
          R.gstart.ensure((size_t) ncl + 1, "cluster starts");
          R.gkey.ensure((size_t) ncl + 1, "cluster keys");
          R.rs_wlist[0].ensure((size_t) ncl + 64, "clusters for the workgroup sort");
          R.rs_wlist[1].ensure((size_t) ncl + 64, "clusters for the radix sort");
          uint32_t *f_gcount = nullptr, *f_gbase = nullptr, *f_gdone = nullptr;
          if (R.fused) chain_prepare_groups(R, &f_gcount, &f_gbase, &f_gdone);
          hipLaunchKernelGGL(k_cluster_compact, dim3(ceil_div((int64_t) nr * 64, 256)), dim3(256), 0, stream(),
                             (const uint64_t *) lcl.p, (const uint32_t *) R.rs_start.p, (const uint32_t *) rncl.p, nr, R.bbits,
                             (uint32_t) nhits, R.gstart.p, R.gkey.p, (uint32_t) RS_IN, (uint32_t) CS_MEDIUM,
                             R.rs_wlist[0].p, R.rs_wlist[1].p, R.rs_wcnt.p,
                             (const uint32_t *) ((R.fused && chain_simple_wanted(R)) ? R.rs_lcnt.p : nullptr), f_gcount, f_gbase, f_gdone);
          std::swap(k0, k1);
        }
      else if (R.word)
        { StageTimer tm("seed_clusters", 40.0 * nhits);
          static DevBuf<uint32_t> tcount;
          const int nt = ceil_div(nhits, WC_TILE);
          tcount.ensure((size_t) nt + 1, "cluster starts per tile");
          hipLaunchKernelGGL(k_wcluster_count, dim3(nt), dim3(256), 0, stream(), (const uint64_t *) k0, nhits, R.pbits, qbits,
                             cb, tcount.p);
          exclusive_scan_u32(tcount.p, nt, st.sort.bsum, tcount.p + nt);
          copy_d2h(&ncl, tcount.p + nt, sizeof(uint32_t), 1);
          R.ngroups = ncl;
          R.gstart.ensure((size_t) ncl + 1, "cluster starts");
          R.gkey.ensure((size_t) ncl + 1, "cluster keys");
          hipLaunchKernelGGL(k_wcluster_rekey, dim3(nt), dim3(256), 0, stream(), (const uint64_t *) k0, nhits, R.pbits, qbits,
                             cb, (const uint32_t *) tcount.p, k1, R.gstart.p, ncl, R.gkey.p);
          std::swap(k0, k1);
        }
      else
        { StageTimer tm("seed_clusters", 40.0 * nhits);
          int nb = ceil_div(nhits, 256);
          hipLaunchKernelGGL(k_cluster_flag, dim3(nb), dim3(256), 0, stream(), (const uint64_t *) k0, nhits, qbits, cb, flag.p);
          exclusive_scan_u32(flag.p, nhits, st.sort.bsum, flag.p + nhits);
          copy_d2h(&ncl, flag.p + nhits, sizeof(uint32_t), 1);
          R.ngroups = ncl;
          R.gstart.ensure((size_t) ncl + 1, "cluster starts");
          R.gkey.ensure((size_t) ncl + 1, "cluster keys");
          hipLaunchKernelGGL(k_cluster_rekey, dim3(nb), dim3(256), 0, stream(), (const uint64_t *) k0, (const uint32_t *) v0,
                             nhits, qbits, cb, R.pbits, local_sort ? 1 : 0, (const uint32_t *) flag.p, k1, v1, R.gstart.p, ncl,
                             R.gkey.p);
          std::swap(k0, k1);
          std::swap(v0, v1);
        }
      // every cluster into (apos, bpos) order; nothing leaves its cluster's range
      if (!local_sort)
        { radix_sort_pairs(&k0, &v0, &k1, &v1, nhits, 0, R.pbits + ceil_log2((int64_t) ncl + 1), st.sort, "cluster_sort");
          hipLaunchKernelGGL(k_pack_ab, dim3(ceil_div(nhits, 256)), dim3(256), 0, stream(), (const uint64_t *) k0,
                             (const uint32_t *) v0, nhits, ((uint64_t) 1 << R.pbits) - 1, k1);
          std::swap(k0, k1);
        }
      else
        { StageTimer tm("cluster_sort", 16.0 * nhits);
          static DevBuf<uint32_t> mlist, blist;
          static DevBuf<unsigned int> cnt;
          if (!rs)
            { mlist.ensure((size_t) ncl + 64, "clusters for the workgroup sort");
              blist.ensure((size_t) ncl + 64, "clusters for the radix sort");
              cnt.ensure(4, "cluster sort counters");
              HIP_CHECK(hipMemsetAsync(cnt.p, 0, 4 * sizeof(unsigned int), stream()));
            }
          if (!rs)
            hipLaunchKernelGGL(k_csort_classify, dim3(ceil_div(ncl, 1024)), dim3(1024), 0, stream(), (const uint32_t *) R.gstart.p,
                               ncl, mlist.p, blist.p, cnt.p);
          if (rs)
            { // only the clusters that span coarse buckets and were too large for k_read_sort are left (k_cluster_compact
              // has listed them): sorted where they lie, on both positions
              hipLaunchKernelGGL(k_csort_wide, dim3(4096), dim3(64), CS_WAVE * sizeof(uint64_t), stream(), k0,
                                 (const uint32_t *) R.gstart.p, (const uint32_t *) R.rs_wlist[0].p, (const unsigned int *) R.rs_wcnt.p,
                                 1u, (uint32_t) CS_WAVE);        // (k_read_sort_big lists small clusters too)
              hipLaunchKernelGGL(k_csort_wide, dim3(1024), dim3(256), CS_MEDIUM * sizeof(uint64_t), stream(), k0,
                                 (const uint32_t *) R.gstart.p, (const uint32_t *) R.rs_wlist[0].p, (const unsigned int *) R.rs_wcnt.p,
                                 (uint32_t) CS_WAVE, (uint32_t) CS_MEDIUM);
              HIP_CHECK(hipGetLastError());
              unsigned int hc[2] = {0, 0};
              if (g_known[1] > 0)             // (k_read_sort counted them: no round trip when there are none)
                copy_d2h(hc, R.rs_wcnt.p, 2 * sizeof(unsigned int), 1);
              if (hc[1] > 0)
                { // repeat piles across buckets: stable radix passes on bpos (the high half of the word), then on apos
                  std::vector<uint32_t> big(hc[1]), gs((size_t) ncl + 1);
                  copy_d2h(big.data(), R.rs_wlist[1].p, sizeof(uint32_t) * hc[1]);
                  copy_d2h(gs.data(), R.gstart.p, sizeof(uint32_t) * ((size_t) ncl + 1));
                  for (uint32_t c : big)
                    { const int64_t b0 = gs[c], bn = (int64_t) gs[c + 1] - b0;
                      uint64_t *sk = k0 + b0, *ok = k1 + b0;
                      radix_sort_pairs(&sk, nullptr, &ok, nullptr, bn, 32, qbits, st.sort, "pile_sort");
                      radix_sort_pairs(&sk, nullptr, &ok, nullptr, bn, 0, R.pbits, st.sort, "pile_sort");
                      if (sk != k0 + b0)
                        HIP_CHECK(hipMemcpyAsync(k0 + b0, sk, sizeof(uint64_t) * bn, hipMemcpyDeviceToDevice, stream()));
                    }
                }
              std::swap(k0, k1);             // (undone below: the sorted hits are on the k0 side on this route)
            }
          else
            {
          hipLaunchKernelGGL(k_csort_small, dim3(std::min<int64_t>(((int64_t) ncl + 15) / 16, 256 * 8)), dim3(256), 0, stream(),
                             (const uint64_t *) k0, (const uint32_t *) R.gstart.p, ncl, k1);
          hipLaunchKernelGGL(k_csort_medium, dim3(8192), dim3(64), CS_WAVE * sizeof(uint32_t), stream(), (const uint64_t *) k0,
                             (const uint32_t *) R.gstart.p, (const uint32_t *) mlist.p, (const unsigned int *) cnt.p, k1,
                             (uint32_t) CS_SMALL, (uint32_t) CS_WAVE);
          hipLaunchKernelGGL(k_csort_medium, dim3(2048), dim3(256), CS_MEDIUM * sizeof(uint32_t), stream(), (const uint64_t *) k0,
                             (const uint32_t *) R.gstart.p, (const uint32_t *) mlist.p, (const unsigned int *) cnt.p, k1,
                             (uint32_t) CS_WAVE, (uint32_t) CS_MEDIUM);
          HIP_CHECK(hipGetLastError());
          unsigned int hc[2] = {0, 0};
          copy_d2h(hc, cnt.p, 2 * sizeof(unsigned int), 1);
          if (hc[1] > 0)
            { // repeat piles: the radix sort (on apos, the low bits of the word; the value arrays only ride along) on
              // each one's own range; sorted data must end in the k1 side
              std::vector<uint32_t> big(hc[1]), gs((size_t) ncl + 1);
              copy_d2h(big.data(), blist.p, sizeof(uint32_t) * hc[1]);
              copy_d2h(gs.data(), R.gstart.p, sizeof(uint32_t) * ((size_t) ncl + 1));
              for (uint32_t c : big)
                { const int64_t b0 = gs[c], bn = (int64_t) gs[c + 1] - b0;
                  uint64_t *sk = k0 + b0, *ok = k1 + b0;
                  radix_sort_pairs(&sk, nullptr, &ok, nullptr, bn, 0, R.pbits, st.sort, "pile_sort");
                  if (sk != k1 + b0)
                    HIP_CHECK(hipMemcpyAsync(k1 + b0, sk, sizeof(uint64_t) * bn, hipMemcpyDeviceToDevice, stream()));
                }
            }
            }
          std::swap(k0, k1);
        }
      { StageTimer tm("seed_pairs", 0.0);
        R.cog.ensure((size_t) ncl + 2, "pair of cluster");
        R.ocfirst.ensure((size_t) ncl + 2, "first cluster of pair");
        R.glast.ensure((size_t) ncl + 1, "last apos of pair");
        DevBuf<uint32_t> &pmax = R.pmax;
        pmax.ensure((size_t) ncl + 2, "largest apos of pair");
        int nb = ceil_div(ncl, 256);
        hipLaunchKernelGGL(k_pair_flag, dim3(nb), dim3(256), 0, stream(), (const uint64_t *) R.gkey.p, ncl, R.cog.p, pmax.p);
        exclusive_scan_u32(R.cog.p, ncl, (tail_scope ? st.sort_tail : st.sort).bsum, R.cog.p + ncl);
        hipLaunchKernelGGL(k_pair_max, dim3(nb), dim3(256), 0, stream(), (const uint64_t *) R.gkey.p, ncl, R.cog.p,
                           (const uint32_t *) (R.cog.p + ncl), R.ocfirst.p, (const uint64_t *) k0,
                           (const uint32_t *) R.gstart.p, pmax.p);
        hipLaunchKernelGGL(k_pair_glast, dim3(nb), dim3(256), 0, stream(), (const uint32_t *) R.cog.p, ncl,
                           (const uint32_t *) pmax.p, R.glast.p);
      }
    }
  R.sab = k0;
  // (the lists of big groups and piles; their numbers come back with the copy -- or are known from k_read_sort)
  chain_classify_groups(R, (rs && g_have_known) ? g_known + 2 : nullptr);
  g_have_known = false;
  HIP_CHECK(hipGetLastError());

  if (st.capture_seeds)
    { // stage-level API only: the hits in the reference's (aread, bread, apos, bpos) order
      std::vector<uint64_t> hk(nhits), gk((size_t) R.ngroups);
      std::vector<uint32_t> gs((size_t) R.ngroups + 1);
      copy_d2h(hk.data(), R.sab, sizeof(uint64_t) * nhits);
      copy_d2h(gk.data(), R.gkey.p, sizeof(uint64_t) * (size_t) R.ngroups);
      copy_d2h(gs.data(), R.gstart.p, sizeof(uint32_t) * ((size_t) R.ngroups + 1));
      HIP_CHECK(hipStreamSynchronize(stream()));
      const uint64_t bm = ((uint64_t) 1 << R.bbits) - 1;
      struct Hit { uint64_t pair; uint32_t apos, bpos; };
      std::vector<Hit> hs((size_t) nhits);
      for (int64_t g = 0; g < R.ngroups; g++)
        for (uint32_t i = gs[g]; i < gs[g + 1]; i++)
          hs[i] = Hit{gk[g], (uint32_t) hk[i], (uint32_t) (hk[i] >> 32)};
      if (cluster)
        std::sort(hs.begin(), hs.end(), [](const Hit &x, const Hit &y)
                  { if (x.pair != y.pair) return x.pair < y.pair;
                    if (x.apos != y.apos) return x.apos < y.apos;
                    return x.bpos < y.bpos;
                  });
      const size_t sbase = st.capture_append ? st.last_seeds.size() : 0;      // (a later range of a split pass: behind the earlier ones)
      st.last_seeds.resize(sbase + (size_t) nhits * 4);
      for (int64_t i = 0; i < nhits; i++)
        { st.last_seeds[sbase + 4 * i + 0] = (int) hs[i].apos - (int) hs[i].bpos;
          st.last_seeds[sbase + 4 * i + 1] = (int) hs[i].apos;
          st.last_seeds[sbase + 4 * i + 2] = (int) (hs[i].pair & bm);
          st.last_seeds[sbase + 4 * i + 3] = (int) (hs[i].pair >> R.bbits);
        }
    }
  { HostTimer hs("chain_submit_host");
    chain_submit(R, ablock);
  }
  tail_scope.reset();            // (what follows is the main stream's business again)
  // the batch is in flight; collect the next one in a free slot of the ring
  int slot = -1;
  for (int i = 0; i < NRING; i++)
    if (&g_ring[i] == &R) slot = i;
  g_inflight.push_back(slot);
  report_poll();                 // (the chain kernels just queued keep the GPU busy meanwhile)
  // (tried: the first step of the oldest batch's finish queued here, so that the next pass finds its chain-end count
  //  in -- slower: the finish stream's kernels then run next to the chain kernels instead of next to the memory-bound
  //  match kernels)
  { static const bool early = getenv("DMH_EARLY_FINISH") != nullptr;       // (opt-in: measured 185.4 against 180.7 ms per read block)
    if (early && (int) g_inflight.size() > flight_keep() && flight_keep() > 0 &&
        !chain_finish_started(g_ring[g_inflight.front()]))
      { int64_t nf = 0;
        (void) finish_tick(&nf);
        st.tfilt += nf;
      }
  }
  // (the oldest batch is finished by the NEXT pass, behind its match kernels: finishing blocks the host for the
  //  finish stream's round trips, and here the main stream would run dry meanwhile -- 0.57 ms per pass in the trace)
  while ((int) g_inflight.size() > g_flight_cap + 1) st.tfilt += finish_oldest();
  if (phase == 2) return;            // (the next batch is being collected already)
  for (int i = 0; i < NRING; i++)
    { bool busy = false;
      for (int y : g_inflight) busy = busy || (y == i);
      if (!busy) { g_cur = i; break; }
    }
}

#ifdef DMH_RS_PROFILE
}  // namespace dmh
extern "C" void dmh_rs_profile(unsigned long long *out, int reset)
{ HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(dmh::g_rsprof), sizeof(unsigned long long) * 16));
  if (reset)
    { unsigned long long z[16] = {0};
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(dmh::g_rsprof), z, sizeof(z)));
    }
}
namespace dmh {
#endif

}  // namespace dmh
