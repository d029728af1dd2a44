// dmh_readsort.h -- seed hits of a dense pass put into chain order PER READ, in LDS (included by dmh_merge.hip).
//
// The reference sorts all hits of a Match_Filter call on (aread, bread, apos) bytes (map.c:2911-2937, 3110-3126); what
// the chain stage needs from that is only the resulting order.  In cluster mode (dmh_merge.h) round 2 got there by
// four global radix passes over one word per hit, two numbering passes and a sort of every cluster where it lies:
// ~150 bytes of HBM traffic per hit.  A read, however, meets a few thousand hits per pass -- 30 KB: its whole order
// fits the LDS of one workgroup.  So the merge writes every hit straight into its READ's segment (the segments are
// laid out by an atomic count per row of the merge and a scan over the reads: dmh_merge.hip k_row_place), and one
// workgroup per read
//   1. sorts the segment by (coarse key, apos), the coarse key being (bread, bpos >> cb): stable LSD radix passes,
//      the words held in registers between reading and re-writing the one LDS copy.  A word's rank among the words of
//      its wavefront with the same digit is what its add to the wavefront's LDS counter of the digit RETURNS: the LDS
//      serves the lanes of one ds_add_rtn in ascending lane order and a wavefront's instructions in program order
//      (checked when the library starts, lds_atomics_ordered(); wavefront ballots -- nine per word and pass instead of
//      one instruction -- take over where that should not hold).  A row of the merge (one READ k-mer and its
//      partners) lies in the segment in bpos order, so equal (bucket, apos) end in bpos order too;
//   2. cuts it into B-clusters: a pair change, or a gap of more than one coarse bucket (a union of gap-connected
//      sets, which is all the chain kernels need);
//   3. a cluster inside ONE bucket is in (apos, bpos) order by now.  The others (a locus across a bucket boundary;
//      marked in a bit per cluster) are put into that order by rank counting in LDS (up to RS_IN hits), or left to
//      k_csort_wide / the radix sort (dmh_merge.hip), for which k_cluster_compact lists them;
//   4. writes the chain kernels' words (apos | bpos << 32) to their final places and leaves one record per cluster
//      (first hit, contig, "spans buckets") for k_cluster_compact, which numbers the clusters of all reads (scan
//      over the reads' counts) into gstart / gkey.
// The order of the rows inside a segment as the merge leaves it is arbitrary (atomic placement), which is harmless:
// the hits of a read are distinct in (bread, bpos, apos).  HBM traffic per hit: 8 B written by the merge, 8 B read and
// 8 B written here.
//
// FUSED (round 4; k_read_sort<T, K32, true>): the hits of a read never exist in HBM before they are in chain order.
//   0. the workgroup takes the read's ROWS of the merge -- one 8-byte record {first REF partner, apos, partners} per READ
//      k-mer with hits, left in the read's own stretch of a row table by k_row_list (dmh_merge.hip) -- into LDS, turns
//      the partner counts into offsets (a scan), marks the first slot of every row in a bit map, and every thread makes
//      the words of ITS slots in registers: row = bits set at or below the slot (one popcount on a word the whole
//      wavefront reads), partner = slot - offset, one 4-byte gather from the REF index's position array
//      (map.c:984-994: the pairs of a row in REF-index order).  Steps 1-4 as above.
//   5. the words in final order go back to LDS and the groups that are PLAIN CHAINS are scored where they lie -- what
//      k_chain_simple (dmh_chain.hip) does from HBM: every hit a break or a step, scores a segmented prefix sum
//      (map.c:1781-1872) -- by a 16-lane group per cluster; links and chain ends go out, the cluster's record says how
//      many ends it reported and that it is done, and k_cluster_compact hands that to the chain stage (gcount / gbase /
//      gdone), which walks what is left.
// HBM traffic per hit: 4 B gathered (REF position), 8 + 4 B written (word, link).  k_merge_expand_tiled, k_row_place,
// k_read_sort and k_chain_simple of round 3 are this one kernel.
#pragma once

namespace dmh {

constexpr int RS_E  = 16;        // words per thread: a workgroup of T threads takes segments of up to 16 T hits
constexpr int RS_IN = 64;        // clusters of up to this many hits are ordered inside k_read_sort (= CS_SMALL)
constexpr int RS_DB = 9;         // most bits of a radix digit
constexpr int RS_MAXPASS = 8;
constexpr int RS_CB = 16;        // coarse buckets of 64 kb: one locus in thirty lies across a boundary

struct ReadSortArgs
{ const uint64_t *in;            // the pass's words, read by read (w = aread | bread | bpos | apos)
  uint64_t       *out;           // the chain kernels' words, same positions
  const uint32_t *rstart;        // first word of every read (+ end)
  const uint32_t *list;          // the reads this launch takes, one per workgroup (k_read_classes)
  const uint32_t *count;         // ... and how many they are (on the device: the grid is sized for the most there can be)
  int             pbits, qbits, bbits, cb;
  int             kbits;         // bits of the coarse key (bread, bpos >> cb)
  int             npass;         // radix passes over the sort key apos | coarse key << pbits, least significant digit first
  uint8_t         sh[RS_MAXPASS], wd[RS_MAXPASS];        // first bit and width of each digit
  int             ordered;       // the LDS serves atomics in lane order (else: ballots)
  uint64_t       *lcl;           // per cluster, at rstart[r] + j: spans buckets << 63 | contig << 32 | first word
  uint32_t       *rncl;          // clusters per read
  // clusters by size, counted as they are found (the host reads them with the number of clusters: one round trip):
  // counts[3] += clusters across buckets of more than `medium` hits (the radix sort finishes those), [4] += clusters of
  // big_min .. pile_min - 1 hits, [5] += piles, [6] += hits in piles (the chain kernels' size classes)
  uint32_t       *counts;
  uint32_t        medium, big_min, pile_min;
  // fused route, front: the rows of the merge instead of expanded words
  const uint64_t *rows;          // row records (RW_*), read r's at rowbase[r] + [0, nrow[r])
  const int64_t  *rowbase;       //   (the READS block's read offsets: a read has fewer rows than bases)
  const uint32_t *nrow;
  const uint32_t *bvals;         // the REF index's positions, (contig << bposbits) | bpos
  int             bposbits;
  int             rowsort;       // the rows are put into apos order first (the passes then cover the coarse key alone)
  // fused route, back: plain chains scored in place (simple_max = 0: not at all)
  uint32_t       *from;          // per hit: predecessor (NIL for a chain's first hit)
  ChainSummary   *sums;          // chain ends, at slot (first hit of the cluster + HITMIN - 1) / HITMIN onwards
  uint32_t       *lcnt;          // per cluster, at rstart[r] + j: chain ends reported | done << 31 (0: left to the chain kernels)
  int             K, hithr;
  uint32_t        simple_max;    // clusters of up to this many hits are tried
};


template <int T>
__device__ __forceinline__ uint32_t block_excl_scan_t(uint32_t v, uint32_t *ws, uint32_t *total)
{ constexpr int W = T / 64;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));      // (a scalar: tests against it are no lane masks)
  const uint32_t x = wave_incl_scan_dpp(v);
  if (lane == 63) ws[w] = x;
  __syncthreads();
  uint32_t wbase = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < W; i++)
    { const uint32_t c = ws[i];
      if (i < w) wbase += c;
      tot += c;
    }
  *total = tot;
  __syncthreads();
  return wbase + x - v;
}

// -DDMH_RS_PROFILE (make profile): cycles per phase, summed over the workgroups (thread 0's clock)
#ifdef DMH_RS_PROFILE
__device__ unsigned long long g_rsprof[16];
#define RSPROF_T0() long long _t0 = clock64()
#define RSPROF_LAP(i) do { long long _t1 = clock64(); if (threadIdx.x == 0) atomicAdd(&g_rsprof[i], (unsigned long long) (_t1 - _t0)); _t0 = _t1; } while (0)
#define RSPROF_ADD(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_rsprof[i], (unsigned long long) (v)); } while (0)
#else
#define RSPROF_T0()
#define RSPROF_LAP(i)
#define RSPROF_ADD(i, v)
#endif

// What the ranks of k_read_sort rest on: 64 lanes add to a handful of LDS words (two 16-bit counters each) in several
// rounds; every add must return the number of earlier adds to its counter -- lower lanes of its round, all lanes of
// earlier rounds -- as counted with ballots.  bad[0] = number of adds for which it did not.
__global__ __launch_bounds__(256) void k_lds_order_test(uint32_t seed, uint32_t *bad)
{ __shared__ uint32_t cnt[4][64];
  __shared__ uint32_t ref[4][128];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int x = threadIdx.x; x < 4 * 64; x += 256) (&cnt[0][0])[x] = 0;
  for (int x = threadIdx.x; x < 4 * 128; x += 256) (&ref[0][0])[x] = 0;
  __syncthreads();
  uint32_t s = seed * 2654435761u + (blockIdx.x * 256u + threadIdx.x) * 40503u + 1u, nbad = 0;
  const int distinct = 1 + (int) (blockIdx.x % 7) * 21;          // 1 .. 127 different digits in play
  for (int r = 0; r < 12; r++)
    { s = s * 1664525u + 1013904223u;
      const uint32_t d = ((s >> 11) % (uint32_t) distinct) & 127u, hs = (d & 1u) * 16u;
      const uint32_t got = (atomicAdd(&cnt[w][d >> 1], 1u << hs) >> hs) & 0xffffu;
      uint64_t peers = ~0ull;
      for (int b = 0; b < 7; b++)
        { const bool bit = (d >> b) & 1u;
          const uint64_t bal = __ballot(bit);
          peers &= bit ? bal : ~bal;
        }
      const uint32_t below = (uint32_t) __popcll(peers & ((1ull << lane) - 1ull));
      const uint32_t before = ref[w][d];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (below == 0) ref[w][d] = before + (uint32_t) __popcll(peers);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (got != before + below) nbad += 1;
    }
  if (nbad) atomicAdd(bad, nbad);
}

template <int T> constexpr size_t read_sort_lds_words()
{ return ((sizeof(uint64_t) * (size_t) T * RS_E + sizeof(uint16_t) * (size_t) (T / 64) * (1 << RS_DB)
           + sizeof(uint64_t) * (T * RS_E / 64 + 2) + sizeof(uint32_t) * (T * RS_E / 32 + 2)
           + sizeof(uint16_t) * (T * RS_E / 64 + 2) + 64) + 7) / 8;
}

// The kernel tests `position < n` and `slot < rounds` some four hundred times; left alone the compiler computes the sixteen
// lane masks of each once and keeps them -- 64 scalar registers of masks, most of them spilled to lanes of a vector register
// and read back with two v_readlane each (a tenth of the kernel's vector instructions, round 4).  Behind every barrier the
// two numbers are made opaque, so that each phase computes the masks it needs (one v_cmp each) and lets them go.
#define RS_FORGET() asm volatile("" : "+s"(n), "+s"(rounds))

template <int T, bool K32, bool FUSE = false>
__global__ __launch_bounds__(T, 4) void k_read_sort(ReadSortArgs A)
{ constexpr int W = T / 64, D = 1 << RS_DB, E = RS_E;
  extern __shared__ __attribute__((aligned(16))) uint64_t rs_lds[];
  uint64_t *buf   = rs_lds;                                   // T * E words
  uint16_t *wh    = (uint16_t *) (buf + (size_t) T * E);      // W x D digit counters, then offsets
  uint64_t *flagw = (uint64_t *) (wh + (size_t) W * D);       // cluster-start bits, one word per 64 hits (+ sentinel)
  uint32_t *mbits = (uint32_t *) (flagw + (T * E / 64 + 2));  // per cluster: spans buckets (T * E / 32 + 2 words)
  uint16_t *wpref = (uint16_t *) (mbits + (T * E / 32 + 2));  // starts before each flag word
  // (fused route, step 5) per cluster: not a sequence of plain chains / not to be tried; (step 3) per position: a run starts
  // here -- a cluster, or the next coarse bucket inside one
  uint32_t *cbad  = (uint32_t *) (rs_lds + read_sort_lds_words<T>());
  uint64_t *runw  = (uint64_t *) (cbad + (T * E / 32 + 2));
  __shared__ uint32_t ws[W];
  __shared__ uint32_t ws2[W];

  // (the grid is sized for the most reads the class can hold -- at most 2048 of the largest workgroups, which then take
  //  several reads each)
  constexpr bool LOOP = (T >= 1024);
  uint32_t rb = blockIdx.x;
  if (rb >= *A.count) return;                                 // (uniform)
  do
  {
  const uint32_t r  = A.list[rb];
  const uint32_t s0 = A.rstart[r];
  uint32_t n  = (uint32_t) __builtin_amdgcn_readfirstlane((int) (A.rstart[r + 1] - s0));      // 0 < n <= T * E (k_read_classes); a scalar (RS_FORGET)

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);      // (a scalar: `v < w` is a scalar test, not a lane mask kept in two registers)
  RSPROF_T0();
  RSPROF_ADD(8, 1); RSPROF_ADD(9, n);
  uint32_t rounds = (n + 64 * W - 1) / (64 * W);        // <= E
  const uint32_t wbase  = (uint32_t) w * rounds * 64;         // the wavefront's words: [wbase, wbase + 64 rounds)
  const uint64_t lt = (1ull << lane) - 1ull;

  uint64_t key[E];
  if (!FUSE)
    {
#pragma unroll
      for (int j = 0; j < E; j++)
        { const uint32_t i = wbase + 64u * j + lane;
          const uint64_t v = __builtin_nontemporal_load(A.in + s0 + (i < n ? i : n - 1));
          key[j] = (i < n) ? v : ~0ull;
        }
    }
  else
    { // ---- 0. the read's rows -> LDS, partner counts -> offsets, first slot of every row -> a bit; then every slot's word
      const uint32_t  nr = A.nrow[r];                          // 0 < nr <= n
      const uint64_t *rw = A.rows + A.rowbase[r];
      uint64_t *rflag = flagw;                                 // (the cluster flags' place: free until step 2)
      uint16_t *rpref = wpref;
      const uint32_t nfw = n / 64 + 1;
      for (uint32_t x = tid; x < nfw; x += T) rflag[x] = 0;
      const uint32_t rrounds = (nr + 64 * W - 1) / (64 * W);   // <= E
      const uint32_t rbase   = (uint32_t) w * rrounds * 64;    // the wavefront's rows: [rbase, rbase + 64 rrounds)
      uint64_t rec[E];
      uint32_t off[E];
      uint32_t carry = 0;
#pragma unroll
      for (int j = 0; j < E; j++)
        { rec[j] = 0; off[j] = 0;
          if ((uint32_t) j < rrounds)
            { const uint32_t x = rbase + 64u * j + lane;
              rec[j] = (x < nr) ? __builtin_nontemporal_load(rw + x) : 0ull;
            }
        }
      if (A.rowsort)
        { // The rows into apos order (a read has one row per apos): a bit per apos value in the digit counters' place, the
          // rank of a row = bits below its own.  The words made below then start out in (apos, contig, bpos) order -- a row's
          // partners are in REF-index order --, so the radix passes only have to order the COARSE key: two passes, not three.
          uint64_t *abm = (uint64_t *) wh;
          const uint32_t aw = ((1u << A.pbits) + 63u) / 64u;
          uint16_t *apf = (uint16_t *) (abm + aw);
          for (uint32_t x = tid; x < aw; x += T) abm[x] = 0;
          __syncthreads(); RS_FORGET();
#pragma unroll
          for (int j = 0; j < E; j++)
            if ((uint32_t) j < rrounds)
              { const uint32_t x = rbase + 64u * j + lane, ap = (uint32_t) ((rec[j] >> RW_MB) & RW_APMASK);
                if (x < nr) atomicOr((unsigned long long *) &abm[ap >> 6], 1ull << (ap & 63u));
              }
          __syncthreads(); RS_FORGET();
          { const uint32_t per = (aw + T - 1) / T, lo = min(aw, (uint32_t) tid * per), hi = min(aw, lo + per);
            uint32_t c = 0, all;
            for (uint32_t x = lo; x < hi; x++) c += (uint32_t) __popcll(abm[x]);
            uint32_t e = block_excl_scan_t<T>(c, ws, &all);
            for (uint32_t x = lo; x < hi; x++) { apf[x] = (uint16_t) e; e += (uint32_t) __popcll(abm[x]); }
          }
          __syncthreads(); RS_FORGET();
#pragma unroll
          for (int j = 0; j < E; j++)
            if ((uint32_t) j < rrounds)
              { const uint32_t x = rbase + 64u * j + lane, ap = (uint32_t) ((rec[j] >> RW_MB) & RW_APMASK);
                if (x < nr) buf[(uint32_t) apf[ap >> 6] + (uint32_t) __popcll(abm[ap >> 6] & ((1ull << (ap & 63u)) - 1ull))] = rec[j];
              }
          __syncthreads(); RS_FORGET();
#pragma unroll
          for (int j = 0; j < E; j++)
            if ((uint32_t) j < rrounds)
              { const uint32_t x = rbase + 64u * j + lane;
                rec[j] = (x < nr) ? buf[x] : 0ull;
              }
        }
#pragma unroll
      for (int j = 0; j < E; j++)
        if ((uint32_t) j < rrounds)
          { const uint32_t mb  = (uint32_t) (rec[j] & RW_MBMASK);
            const uint32_t inc = wave_incl_scan_dpp(mb);
            off[j] = carry + inc - mb;
            carry += (uint32_t) __builtin_amdgcn_readlane((int) inc, 63);
          }
      if (lane == 0) ws[w] = carry;
      __syncthreads(); RS_FORGET();
      uint32_t wv0 = 0;
#pragma unroll
      for (int v = 0; v < W; v++)
        if (v < w) wv0 += ws[v];
#pragma unroll
      for (int j = 0; j < E; j++)
        if ((uint32_t) j < rrounds)
          { const uint32_t x = rbase + 64u * j + lane;
            if (x < nr)
              { const uint32_t o = wv0 + off[j];                // < n <= T * E <= 2^14
                buf[x] = (rec[j] & ~RW_MBMASK) | (uint64_t) o;
                atomicOr((unsigned long long *) &rflag[o >> 6], 1ull << (o & 63u));
              }
          }
      __syncthreads(); RS_FORGET();
      { uint32_t c = ((uint32_t) tid < nfw) ? (uint32_t) __popcll(rflag[tid]) : 0u, all;
        const uint32_t e = block_excl_scan_t<T>(c, ws, &all);
        if ((uint32_t) tid < nfw) rpref[tid] = (uint16_t) e;
      }
      __syncthreads(); RS_FORGET();
      uint32_t q[E];
#pragma unroll
      for (int j = 0; j < E; j++)
        { key[j] = 0; q[j] = 0;
          if ((uint32_t) j < rounds)                             // (uniform)
            { // (unconditional loads from clamped slots: a guarded load is a branch with a wait at its join)
              const uint32_t i = wbase + 64u * j + lane, ic = (i < n) ? i : n - 1;
              const uint32_t wi = ic >> 6, bi = ic & 63u;
              const uint32_t row = (uint32_t) rpref[wi] + (uint32_t) __popcll(rflag[wi] & (~0ull >> (63u - bi))) - 1u;
              const uint64_t rc = buf[row];
              key[j] = (rc >> RW_MB) & RW_APMASK;                // apos for now
              q[j]   = A.bvals[(uint32_t) (rc >> 32) + (ic - (uint32_t) (rc & RW_MBMASK))];
            }
        }
      const uint32_t bpm = (1u << A.bposbits) - 1u;
      const int      top = A.pbits + A.qbits;
#pragma unroll
      for (int j = 0; j < E; j++)
        { const uint32_t i = wbase + 64u * j + lane;
          const uint64_t v = ((uint64_t) (q[j] >> A.bposbits) << top) | ((uint64_t) (q[j] & bpm) << A.pbits) | key[j];
          key[j] = (i < n) ? v : ~0ull;
        }
      __syncthreads(); RS_FORGET();                                           // (the rows have been read: buf and the flag words are free)
    }
#ifdef DMH_RS_PROFILE
  __builtin_amdgcn_s_waitcnt(0x0F70); __syncthreads(); RS_FORGET();
#endif
  RSPROF_LAP(0);

  // ---- 1. stable LSD passes over the sort key: apos in its low bits, the coarse key above (one word of at most 32 bits,
  // so that a digit may lie across the two fields: 27 bits here, three passes of nine)
  const uint64_t apm = (1ull << A.pbits) - 1ull;
  const uint32_t ckm = (A.kbits >= 32) ? 0xffffffffu : ((1u << A.kbits) - 1u);
  const int      csh = A.pbits + A.cb;
  auto rs_key = [&](uint64_t x) -> uint32_t { return (uint32_t) (x & apm) | (((uint32_t) (x >> csh) & ckm) << A.pbits); };
  uint16_t *mywh = wh + (size_t) w * D;
  uint32_t *my32 = (uint32_t *) mywh;
  for (int p = 0; p < A.npass; p++)
    { const int sh = A.sh[p], wd = A.wd[p];
      const uint32_t dmask = (1u << wd) - 1u;
      for (int x = tid; x < W * D / 2; x += T) ((uint32_t *) wh)[x] = 0;
      __syncthreads(); RS_FORGET();
      uint32_t rk[E];
      if (A.ordered)
        { // the counter before the add = words of this wavefront with the digit in earlier rounds and lower lanes
#pragma unroll
          for (int j = 0; j < E; j++)
            { rk[j] = 0;
              if ((uint32_t) j < rounds)
                { const uint32_t i = wbase + 64u * j + lane;
                  const uint32_t d = (rs_key(key[j]) >> sh) & dmask;
                  if (i < n) rk[j] = atomicAdd(&my32[d >> 1], 1u << ((d & 1u) * 16u));
                }
            }
#pragma unroll
          for (int j = 0; j < E; j++)
            if ((uint32_t) j < rounds)
              rk[j] = (rk[j] >> ((((rs_key(key[j]) >> sh) & dmask) & 1u) * 16u)) & 0xffffu;
        }
      else
        {
#pragma unroll
          for (int j = 0; j < E; j++)
            { rk[j] = 0;
              if ((uint32_t) j < rounds)                       // (uniform)
                { const uint32_t i  = wbase + 64u * j + lane;
                  const bool     ok = i < n;
                  const uint32_t d  = (rs_key(key[j]) >> sh) & dmask;
                  uint64_t peers = __ballot(ok);
                  for (int b = 0; b < wd; b++)
                    { const bool     bit = (d >> b) & 1u;
                      const uint64_t bal = __ballot(bit);
                      peers &= bit ? bal : ~bal;
                    }
                  const uint32_t below = (uint32_t) __popcll(peers & lt);
                  const uint32_t cnt   = (uint32_t) __popcll(peers);
                  const uint32_t b0    = ok ? (uint32_t) mywh[d] : 0u;
                  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                  __builtin_amdgcn_wave_barrier();
                  if (ok && below == 0) mywh[d] = (uint16_t) (b0 + cnt);
                  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                  __builtin_amdgcn_wave_barrier();
                  rk[j] = b0 + below;
                }
              __builtin_amdgcn_sched_barrier(0);               // (one item at a time: interleaved, their ballots spill)
            }
        }
      __syncthreads(); RS_FORGET();
      RSPROF_LAP(1);
      // digit d: counts per wavefront -> first slot of (d, wavefront) in the order digit, wavefront, position
      { constexpr int PER = (D + T - 1) / T;                   // digits per thread (consecutive)
        uint32_t c[PER][W], tot = 0;
#pragma unroll
        for (int q = 0; q < PER; q++)
          { const int d = tid * PER + q;
#pragma unroll
            for (int v = 0; v < W; v++)
              { c[q][v] = (d < D) ? (uint32_t) wh[(size_t) v * D + d] : 0u;
                tot += c[q][v];
              }
          }
        uint32_t all;
        uint32_t run = block_excl_scan_t<T>(tot, ws, &all);
#pragma unroll
        for (int q = 0; q < PER; q++)
          { const int d = tid * PER + q;
#pragma unroll
            for (int v = 0; v < W; v++)
              { if (d < D) wh[(size_t) v * D + d] = (uint16_t) run;
                run += c[q][v];
              }
          }
      }
      __syncthreads(); RS_FORGET();
      RSPROF_LAP(2);
#pragma unroll
      for (int j = 0; j < E; j++)
        if ((uint32_t) j < rounds)
          { const uint32_t i = wbase + 64u * j + lane;
            if (i < n)
              { const uint32_t d = (rs_key(key[j]) >> sh) & dmask;
                buf[rk[j] + (uint32_t) mywh[d]] = key[j];
              }
          }
      __syncthreads(); RS_FORGET();
#pragma unroll
      for (int j = 0; j < E; j++)
        if ((uint32_t) j < rounds)
          { const uint32_t i = wbase + 64u * j + lane;
            key[j] = (i < n) ? buf[i] : ~0ull;
          }
      __syncthreads(); RS_FORGET();                                         // (everybody has its words: buf and wh are free again)
      RSPROF_LAP(3);
    }
  if (A.npass == 0)
    {
#pragma unroll
      for (int j = 0; j < E; j++)
        if ((uint32_t) j < rounds)
          { const uint32_t i = wbase + 64u * j + lane;
            if (i < n) buf[i] = key[j];
          }
      __syncthreads(); RS_FORGET();
    }

  // ---- 2. cluster starts: bit i of the flag words (the words of a round are consecutive, so a ballot is a flag word),
  // and the positions inside a cluster at which the bucket changes
  const uint32_t nw = n / 64 + 1;                              // flag words, the sentinel bit n included
  uint32_t chg = 0;                                            // bit j: my word of round j opens a new bucket inside its cluster
  for (int x = tid; x < (int) (n / 32 + 1); x += T) mbits[x] = 0;
  if (FUSE)
    for (int x = tid; x < (int) (n / 32 + 1); x += T) cbad[x] = 0;
#pragma unroll
  for (int j = 0; j < E; j++)
    if ((uint32_t) j < rounds)
      { const uint32_t i = wbase + 64u * j + lane;
        bool f = false;
        // (the predecessor: the lane below, or -- lane 0 -- the last lane of the wavefront's round before; the first
        //  round's comes from the wavefront before, through LDS)
        uint64_t pk = wave_from_below(key[j]);
        if (lane == 0) pk = (j == 0) ? buf[wbase > 0 && wbase - 1 < n ? wbase - 1 : 0] : wave_lane_u64(key[j > 0 ? j - 1 : 0], 63);
        if (i < n)
          { const uint64_t pv = (i > 0) ? pk >> A.pbits : 0, cv = key[j] >> A.pbits;
            f = (i == 0) || cluster_start(pv, cv, A.qbits, A.cb);
            if (!f && (pv >> A.cb) != (cv >> A.cb)) chg |= 1u << j;
          }
        else if (i == n)
          f = true;                                            // sentinel: the end of the last cluster
        const uint64_t m = __ballot(f);
        if (lane == 0 && (wbase + 64u * j) / 64 < nw) flagw[(wbase + 64u * j) / 64] = m;
        if (FUSE)
          { const uint64_t mr = __ballot(f || ((chg >> j) & 1u));
            if (lane == 0 && (wbase + 64u * j) / 64 < nw) runw[(wbase + 64u * j) / 64] = mr;
          }
      }
  // (the sentinel lies in the word after the last wavefront's last round when n is a multiple of 64 W)
  if (tid == 0 && n == rounds * 64 * W) { flagw[n / 64] = 1ull; if (FUSE) runw[n / 64] = 1ull; }
  __syncthreads(); RS_FORGET();
  uint32_t ncl;
  { uint32_t c = 0;
    if ((uint32_t) tid < nw)
      { uint64_t m = flagw[tid];
        if ((uint32_t) tid == n / 64) m &= ~(1ull << (n & 63));           // (the sentinel is not a cluster)
        c = (uint32_t) __popcll(m);
      }
    const uint32_t e = block_excl_scan_t<T>(c, ws, &ncl);
    if ((uint32_t) tid < nw) wpref[tid] = (uint16_t) e;
  }
  __syncthreads(); RS_FORGET();
  // the cluster of every word (starts at or below it, minus one); a bucket change marks its cluster
  uint16_t cid[E];
#pragma unroll
  for (int j = 0; j < E; j++)
    { cid[j] = 0;
      if ((uint32_t) j < rounds)
        { const uint32_t i = wbase + 64u * j + lane, wi = i >> 6, bi = i & 63;
          if (i < n)
            { const uint32_t c = (uint32_t) wpref[wi] + (uint32_t) __popcll(flagw[wi] & (~0ull >> (63 - bi))) - 1u;
              cid[j] = (uint16_t) c;
              if ((chg >> j) & 1u) atomicOr(&mbits[c >> 5], 1u << (c & 31u));
            }
        }
    }
  if (tid == 0) A.rncl[r] = ncl;
  __syncthreads(); RS_FORGET();
  RSPROF_LAP(4);

  // ---- 3. every start leaves its cluster's record; the words of clusters that span buckets look for their cluster's
  // bounds: a round is 64 consecutive, aligned positions, i.e. ONE flag word; the start of a lane's cluster is the highest
  // flag at or below its position (in this word or the one before: a cluster that begins earlier is longer than RS_IN
  // anyway), its end the lowest flag above (this word or the next).
  // What such a cluster is ordered by: (apos, bpos).  K32: as ONE 32-bit word, apos above bpos minus the first base of the
  // cluster's first coarse bucket -- RS_IN hits or fewer, so RS_IN adjacent buckets at most, and the difference fits the
  // bits apos leaves free (pbits + cb + 6 <= 32); else as apos << 32 | bpos.
  const uint64_t pm = (1ull << A.pbits) - 1ull, qm = (1ull << A.qbits) - 1ull, bm = (1ull << A.bbits) - 1ull;
  uint32_t se[E];                                              // start | length << 16 (length 0: the hit keeps its position)
  uint32_t k32[K32 ? E : 1];
  uint32_t anym = 0;                                           // bit j: some lane of round j has a cluster to sort within
#pragma unroll
  for (int j = 0; j < E; j++)
    { se[j] = 0;
      if (K32) k32[j] = 0;
      if ((uint32_t) j < rounds)
        { const uint32_t i = wbase + 64u * j + lane, wi = i >> 6, bi = i & 63;
          bool multi = false;
          if (i < n)
            { const uint32_t c = cid[j];
              multi = (mbits[c >> 5] >> (c & 31u)) & 1u;
              if ((flagw[wi] >> bi) & 1ull)
                { // (fused route: every cluster leaves this kernel in chain order -- nothing for k_csort_wide)
                  A.lcl[s0 + c] = ((uint64_t) (multi && !FUSE) << 63) | (((key[j] >> (A.pbits + A.qbits)) & bm) << 32) | (uint64_t) (s0 + i);
                  // the cluster's size: up to the next start (the sentinel ends the search)
                  uint64_t m = (bi == 63) ? 0ull : (flagw[wi] & (~0ull << (bi + 1)));
                  uint32_t ww = wi;
                  while (m == 0) { ww += 1; m = flagw[ww]; }
                  const uint32_t size = ww * 64 + (uint32_t) __ffsll((long long) m) - 1 - i;
                  // (step 5 leaves alone: a cluster that is too large)
                  if (FUSE && size > A.simple_max) atomicOr(&cbad[c >> 5], 1u << (c & 31u));
                  if (size >= A.big_min)
                    { if (size >= A.pile_min) { atomicAdd(A.counts + 5, 1u); atomicAdd(A.counts + 6, size); }
                      else atomicAdd(A.counts + 4, 1u);
                    }
                  if (multi && !FUSE && size > A.medium) atomicAdd(A.counts + 3, 1u);
                }
            }
          if (FUSE)
            { // A cluster that spans coarse buckets is a sequence of RUNS -- one per bucket, each in (apos, bpos) order
              // (the passes are stable and the words started out that way) --, so its order is a MERGE: the place of a hit
              // is its place in its own run plus, for every other run of its cluster, the number of that run's hits that
              // come before it (a binary search).  Any size; nothing is left to k_csort_wide.
              se[j] = i;
              if (__any(multi))
                { if (multi)
                    { const uint64_t x = key[j];
                      const uint64_t kk = ((x & pm) << 32) | ((x >> A.pbits) & qm);
                      auto at_or_below = [&](const uint64_t *mw, uint32_t p) -> uint32_t
                      { uint32_t ww = p >> 6;
                        uint64_t y = mw[ww] & (~0ull >> (63u - (p & 63u)));
                        while (y == 0) { ww -= 1; y = mw[ww]; }
                        return ww * 64 + 63u - (uint32_t) __clzll((long long) y);
                      };
                      auto above = [&](const uint64_t *mw, uint32_t p) -> uint32_t          // (the sentinel n ends every search)
                      { uint32_t ww = p >> 6;
                        uint64_t y = ((p & 63u) == 63u) ? 0ull : (mw[ww] & (~0ull << ((p & 63u) + 1)));
                        while (y == 0) { ww += 1; y = mw[ww]; }
                        return ww * 64 + (uint32_t) __ffsll((long long) y) - 1;
                      };
                      const uint32_t cs = at_or_below(flagw, i), ce = above(flagw, i), rs = at_or_below(runw, i);
                      uint32_t place = cs + (i - rs);
                      for (uint32_t a = cs; a < ce; )
                        { const uint32_t b = above(runw, a);                 // the run [a, b)
                          if (a != rs)
                            { uint32_t lo = a, hi = b;                       // first word of the run that does not come before mine
                              while (lo < hi)
                                { const uint32_t md = (lo + hi) >> 1;
                                  const uint64_t y = buf[md];
                                  if ((((y & pm) << 32) | ((y >> A.pbits) & qm)) < kk) lo = md + 1; else hi = md;
                                }
                              place += lo - a;
                            }
                          a = b;
                        }
                      se[j] = place;
                    }
                }
            }
          else if (__any(multi))                               // (one locus in fifteen: most rounds skip this)
            { const uint64_t f0 = flagw[wi > 0 ? wi - 1 : 0], f1 = flagw[wi < nw ? wi : 0], f2 = flagw[wi + 1 < nw ? wi + 1 : 0];
              if (multi)
                { const uint64_t x = key[j];
                  uint32_t s = 0xffffffffu, e = 0xffffffffu;
                  const uint64_t ms = f1 & (~0ull >> (63 - bi));
                  if (ms) s = wi * 64 + 63 - (uint32_t) __clzll((long long) ms);
                  else if (wi > 0 && f0) s = (wi - 1) * 64 + 63 - (uint32_t) __clzll((long long) f0);
                  const uint64_t me = (bi == 63) ? 0ull : (f1 & (~0ull << (bi + 1)));
                  if (me) e = wi * 64 + (uint32_t) __ffsll((long long) me) - 1;
                  else if (wi + 1 < nw && f2) e = (wi + 1) * 64 + (uint32_t) __ffsll((long long) f2) - 1;
                  if (s != 0xffffffffu && e != 0xffffffffu && e - s <= (uint32_t) RS_IN)
                    { se[j] = s | ((e - s) << 16);
                      if (K32)
                        { const uint32_t base = (uint32_t) ((buf[s] >> A.pbits) & qm) >> A.cb << A.cb;
                          k32[j] = ((uint32_t) (x & pm) << (32 - A.pbits)) | ((uint32_t) ((x >> A.pbits) & qm) - base);
                        }
                    }
                }
              if (__any(se[j] != 0)) anym |= 1u << j;
            }
        }
    }
  __syncthreads(); RS_FORGET();                                             // (nobody reads the sorted words any more)
  uint32_t *b32 = (uint32_t *) buf;
#pragma unroll
  for (int j = 0; j < E; j++)
    if ((uint32_t) j < rounds)
      { const uint32_t i = wbase + 64u * j + lane;
        const uint64_t x = key[j];
        key[j] = (x & pm) | (((x >> A.pbits) & qm) << 32);     // the chain kernels' word: apos | bpos << 32
        if (!FUSE && se[j] != 0)                               // (only the words of clusters that are ranked are read back)
          { if (K32) b32[i] = k32[j];
            else     buf[i] = (key[j] << 32) | (key[j] >> 32);
          }
      }
  __syncthreads(); RS_FORGET();
  RSPROF_LAP(5);

  // ---- 4. clusters across buckets of up to RS_IN hits: rank by counting; everything else keeps its position
#pragma unroll
  for (int j = 0; j < E; j++)
    if ((uint32_t) j < rounds)
      { const uint32_t i = wbase + 64u * j + lane;
        uint32_t dst = FUSE ? se[j] : i;
        if (!FUSE && ((anym >> j) & 1u))                       // (uniform)
          { const uint32_t s = se[j] & 0xffffu, m = se[j] >> 16;
            uint32_t rank = 0;
            if (K32)
              { const uint32_t kk = k32[j];
                for (uint32_t t = 0; t < m; t += 4)
                  { uint32_t v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = b32[s + (t + u < m ? t + u : 0)];
#pragma unroll
                    for (int u = 0; u < 4; u++) rank += (t + u < m && v[u] < kk) ? 1u : 0u;
                  }
              }
            else
              { const uint64_t kk = (key[j] << 32) | (key[j] >> 32);
                for (uint32_t t = 0; t < m; t += 4)
                  { uint64_t v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = buf[s + (t + u < m ? t + u : 0)];
#pragma unroll
                    for (int u = 0; u < 4; u++) rank += (t + u < m && v[u] < kk) ? 1u : 0u;
                  }
              }
            if (m) dst = s + rank;
          }
        if (i < n) A.out[s0 + dst] = key[j];
        if (FUSE) se[j] = dst;
      }
#ifdef DMH_RS_PROFILE
  __builtin_amdgcn_s_waitcnt(0x0F70); __syncthreads(); RS_FORGET();
#endif
  RSPROF_LAP(6);

  // ---- 5. plain chains scored where they lie (k_chain_simple's rules, dmh_chain.hip; map.c:1781-1872) -- hit by hit, not
  // cluster by cluster: a position is a BREAK (first hit of its cluster, or more than MAX_GAP behind its predecessor in A)
  // or must be a step (the predecessor is its only candidate and is removed by it), else its cluster is left to the chain
  // kernels ("bad", a bit per cluster).  Chains are the runs between breaks: the score of a chain's end is the difference of
  // two prefix sums of the per-hit increments (one scan over the read; the sums at the chain ends are kept, by chain number
  // = breaks at or below the position), and the ends that score are counted per cluster in a bit map.
  if (FUSE && A.simple_max > 0)
    { __syncthreads(); RS_FORGET();                                           // (the ranking above has read buf for the last time)
#pragma unroll
      for (int j = 0; j < E; j++)
        if ((uint32_t) j < rounds)
          { const uint32_t i = wbase + 64u * j + lane;
            if (i < n) buf[se[j]] = key[j];
          }
      uint64_t *bkw    = (uint64_t *) mbits;                     // breaks, a bit per position (+ the sentinel n)
      uint16_t *bkpref = wh;                                     // breaks before each word of them
      uint32_t *pe     = (uint32_t *) (wh + (T * E / 64 + 2));   // prefix sum at every chain's end, by chain number
      constexpr uint32_t PECAP = (uint32_t) ((W * D * 2 - 2 * (T * E / 64 + 2)) / 4);
      __syncthreads(); RS_FORGET();
      const int K = A.K;
      uint32_t P[E];
      uint32_t endm = 0, carry = 0;
      // the wavefront's words in registers (its positions are consecutive: the neighbours of a position are the lanes
      // beside it, the last lane of the round before, the first of the round after -- one LDS read per position)
      uint64_t wq[E];
#pragma unroll
      for (int j = 0; j < E; j++)
        { wq[j] = 0;
          if ((uint32_t) j < rounds)
            { const uint32_t i = wbase + 64u * j + lane;
              wq[j] = buf[i < n ? i : n - 1];
            }
        }
      const uint32_t wend = wbase + 64u * rounds;                // the position behind the wavefront's last
      uint64_t below = buf[wbase > 0 ? (wbase - 1 < n ? wbase - 1 : n - 1) : 0];
      const uint64_t after = buf[wend < n ? wend : n - 1];
#pragma unroll
      for (int j = 0; j < E; j++)
        { P[j] = 0;
          if ((uint32_t) j < rounds)
            { const uint32_t i = wbase + 64u * j + lane, wi = i >> 6, bi = i & 63u;
              const bool in = i < n;
              const uint64_t wv = wq[j];
              uint64_t pw = wave_from_below(wv), xw = wave_from_above(wv);
              if (lane == 0) pw = below;
              below = wave_lane_u64(wv, 63);
              { const uint64_t nx0 = ((uint32_t) (j + 1) < rounds && j + 1 < E) ? wave_lane_u64(wq[j + 1 < E ? j + 1 : j], 0) : after;
                if (lane == 63) xw = nx0;
              }
              const uint64_t fw = flagw[wi < nw ? wi : 0];       // (bit n is the sentinel: a start)
              const bool cs  = in && ((fw >> bi) & 1ull);
              const bool csn = in && (bi < 63 ? ((fw >> (bi + 1)) & 1ull) != 0 : (flagw[wi + 1] & 1ull) != 0);
              const int a = (int) (uint32_t) wv + 1, b = (int) (wv >> 32) + 1, pa = (int) (uint32_t) pw + 1, pb = (int) (pw >> 32) + 1;
              const int d = a - b, pd = pa - pb;
              const bool brk = in && (cs || pa < a - MAX_GAP);
              bool bad = false;
              { const bool cand = (pd > d) ? (pb >= b - MAX_GAP) : (pb <= b);
                int dd = pd - d;
                if (dd < 0) dd = -dd;
                bad = in && !brk && !(cand && 5 * dd <= a - pa);
              }
              const int adv = (pd > d) ? a - pa : b - pb;          // left neighbour: advance in A; right: in B
              const uint32_t inc = !in ? 0u : (uint32_t) (brk ? K : (adv >= K ? K : adv));
              const int  na  = (int) (uint32_t) xw + 1;
              const bool end = in && (csn || a < na - MAX_GAP);
              if (end) endm |= 1u << j;
              if (in) A.from[s0 + i] = brk ? NIL : s0 + i - 1;
              const uint64_t bm = __ballot(brk || i == n);
              if (lane == 0 && wi < nw) bkw[wi] = bm;
              if (bad)
                { const uint32_t c = (uint32_t) wpref[wi] + (uint32_t) __popcll(fw & (~0ull >> (63u - bi))) - 1u;
                  atomicOr(&cbad[c >> 5], 1u << (c & 31u));
                }
              const uint32_t sc = wave_incl_scan_dpp(inc);
              P[j] = carry + sc;
              carry += (uint32_t) __builtin_amdgcn_readlane((int) sc, 63);
            }
        }
      if (tid == 0 && n == rounds * 64 * W) bkw[n / 64] = 1ull;
      if (lane == 0) ws2[w] = carry;
      __syncthreads(); RS_FORGET();
      uint32_t wv0 = 0;
#pragma unroll
      for (int v = 0; v < W; v++)
        if (v < w) wv0 += ws2[v];
      uint32_t nchain;
      { uint32_t c = 0;
        if ((uint32_t) tid < nw)
          { uint64_t m = bkw[tid];
            if ((uint32_t) tid == n / 64) m &= ~(1ull << (n & 63));           // (the sentinel is not a chain)
            c = (uint32_t) __popcll(m);
          }
        const uint32_t e = block_excl_scan_t<T>(c, ws, &nchain);
        if ((uint32_t) tid < nw) bkpref[tid] = (uint16_t) e;
      }
      __syncthreads(); RS_FORGET();
      const bool fits = nchain <= PECAP;                         // (else: a read of singletons -- all of it left to the chain kernels)
#pragma unroll
      for (int j = 0; j < E; j++)
        if (fits && ((endm >> j) & 1u))
          { const uint32_t i = wbase + 64u * j + lane, wi = i >> 6, bi = i & 63u;
            pe[(uint32_t) bkpref[wi] + (uint32_t) __popcll(bkw[wi] & (~0ull >> (63u - bi))) - 1u] = wv0 + P[j];
          }
      __syncthreads(); RS_FORGET();
      RSPROF_LAP(7);
      // A thread per CLUSTER (a read has some 150): its first hit is the c-th start bit -- the word by a search in the
      // prefix counts, the bit by clearing c' lower ones --, its chains are the breaks inside it, the score of each the
      // difference of two neighbouring prefix sums; the chains that score are reported (in any order: k_order_summaries
      // ranks them), their number and "done" is the cluster's record for the chain stage
      for (uint32_t c = (uint32_t) tid; c < ncl; c += T)
        { uint32_t lo = 0, hi = nw;                              // last word with wpref[] <= c
          while (hi - lo > 1)
            { const uint32_t m = (lo + hi) >> 1;
              if ((uint32_t) wpref[m] <= c) lo = m; else hi = m;
            }
          uint64_t fm = flagw[lo];
          if (lo == n / 64) fm &= ~(1ull << (n & 63));
          for (uint32_t t = c - (uint32_t) wpref[lo]; t > 0; t--) fm &= fm - 1;
          const uint32_t cs = lo * 64 + (uint32_t) __ffsll((long long) fm) - 1;
          uint32_t ce;
          { const uint32_t bi = cs & 63u;
            uint64_t m = (bi == 63) ? 0ull : (flagw[lo] & (~0ull << (bi + 1)));
            uint32_t ww = lo;
            while (m == 0) { ww += 1; m = flagw[ww]; }
            ce = ww * 64 + (uint32_t) __ffsll((long long) m) - 1;
          }
          uint32_t v = 0;
          if (fits && !((cbad[c >> 5] >> (c & 31u)) & 1u))
            { const uint32_t obase = (s0 + cs + HITMIN - 1) / HITMIN;
              uint32_t k = (uint32_t) bkpref[cs >> 6] + (uint32_t) __popcll(bkw[cs >> 6] & (~0ull >> (63u - (cs & 63u)))) - 1u;
              uint32_t prev = (k > 0) ? pe[k - 1] : 0u, En = 0;
              uint32_t st = cs;                                  // first hit of the chain at hand
              while (st < ce)
                { // its last hit: the position before the next break (the sentinel n is one)
                  uint32_t ww = st >> 6;
                  uint64_t m = ((st & 63u) == 63u) ? 0ull : (bkw[ww] & (~0ull << ((st & 63u) + 1)));
                  while (m == 0) { ww += 1; m = bkw[ww]; }
                  const uint32_t nx = ww * 64 + (uint32_t) __ffsll((long long) m) - 1;
                  const uint32_t cur = pe[k];
                  if ((int) (cur - prev) >= A.hithr)
                    { const uint64_t ow = buf[st], ev = buf[nx - 1];
                      ChainSummary sm;
                      sm.group = 0;                              // (its number in the pass, whether the end is still in the
                      sm.live  = LIVE_LATER;                     //  window when its PAIR ends, and the order key that follows
                      sm.okey  = 0;                              //  from that: k_pack_summaries)
                      sm.cost  = (int) (cur - prev);
                      sm.ab    = (int) (uint32_t) ow + 1 - K;
                      sm.bb    = (int) (ow >> 32) + 1 - K;
                      sm.ae    = (int) (uint32_t) ev + 1;
                      sm.be    = (int) (ev >> 32) + 1;
                      sm.end   = s0 + nx - 1;
                      A.sums[obase + En] = sm;
                      En += 1;
                    }
                  prev = cur; k += 1; st = nx;
                }
              v = En | 0x80000000u;
            }
#ifdef DMH_RS_PROFILE_STATS                                     // (not with the phase clock: these loops are a third of the phase they sit in)
          else
            { // hits in the clusters left to the chain kernels, by cluster size (<= 16, 32, 64, 128, 256, more)
              const uint32_t sz = ce - cs;
              const int bk = sz <= 16 ? 0 : (sz <= 32 ? 1 : (sz <= 64 ? 2 : 3));            // ([13]: everything above 64)
              atomicAdd(&g_rsprof[10 + bk], (unsigned long long) sz);
              // ... and how many of them lie in clusters that fall apart by DIAGONAL (bins of 1024, classes = runs of
              // occupied bins with an empty bin between them): [14] hits in clusters of two classes and more, [15] of one
              if (sz <= 1023)
                { int dmin = 0x7fffffff, dmax = (int) 0x80000000;
                  for (uint32_t t = cs; t < ce; t++)
                    { const uint64_t y = buf[t]; const int dg = (int) (uint32_t) y - (int) (y >> 32);
                      dmin = min(dmin, dg); dmax = max(dmax, dg);
                    }
                  unsigned long long bm[4] = {0, 0, 0, 0};
                  if (((dmax - dmin) >> 10) < 256)
                    { for (uint32_t t = cs; t < ce; t++)
                        { const uint64_t y = buf[t]; const int b = (((int) (uint32_t) y - (int) (y >> 32)) - dmin) >> 10;
                          bm[b >> 6] |= 1ull << (b & 63);
                        }
                      int classes = 0, prev = 0, gap = 2;
                      for (int b = 0; b < 256; b++)
                        { const int on = (int) ((bm[b >> 6] >> (b & 63)) & 1ull);
                          if (on) { if (gap >= 2) classes += 1; gap = 0; } else gap += 1;
                          prev = on;
                        }
                      (void) prev;
                      atomicAdd(&g_rsprof[14], (unsigned long long) (classes >= 2 ? sz : 0));
                    }
                }
            }
#endif
          A.lcnt[s0 + c] = v;
        }
#ifdef DMH_RS_PROFILE
      __builtin_amdgcn_s_waitcnt(0x0F70); __syncthreads(); RS_FORGET();
#endif
      RSPROF_LAP(15);                                            // (the thread-per-cluster part of step 5)
    }
  if (!LOOP) break;
  __syncthreads(); RS_FORGET();                                             // (the next read of this workgroup starts with empty LDS)
  rb += gridDim.x;
  } while (rb < *A.count);
}

// dynamic LDS of k_read_sort<T>: the unfused kernel's arrays, in 8-byte words (read_sort_lds_words, declared ahead of the
// kernel), and behind them the two bit maps of the fused route's steps 3 and 5
template <int T> constexpr size_t read_sort_lds()
{ return sizeof(uint64_t) * read_sort_lds_words<T>()
         + sizeof(uint64_t) * (T * RS_E / 64 + 2) + sizeof(uint32_t) * (T * RS_E / 32 + 2);
}

// A read with more hits than the largest workgroup's LDS takes (a tandem array: a repeat pile and what surrounds it): the
// same steps through GLOBAL memory, one workgroup per read -- the segment in A.in and its place in A.out are the two
// buffers of the radix passes (ranks from ballots; a wavefront owns a contiguous stretch of the words), the clusters
// are cut and counted thread by thread over contiguous stretches, and no cluster is put in order here: every one that
// spans buckets is marked "spans buckets" AND "not ordered" (bits 63, 62 of its record), whatever its size, for
// k_cluster_compact to list.  A handful of reads per pass at most, and only opposite the repeat they come from.
template <int T>
__global__ __launch_bounds__(T) void k_read_sort_big(ReadSortArgs A)
{ constexpr int W = T / 64, D = 1 << RS_DB;
  __shared__ uint32_t wh[W * D];
  __shared__ uint32_t ws[W];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint64_t apm = (1ull << A.pbits) - 1ull;
  const uint32_t ckm = (A.kbits >= 32) ? 0xffffffffu : ((1u << A.kbits) - 1u);
  const int      csh = A.pbits + A.cb;
  auto rs_key = [&](uint64_t x) -> uint32_t { return (uint32_t) (x & apm) | (((uint32_t) (x >> csh) & ckm) << A.pbits); };
  const uint64_t pm = (1ull << A.pbits) - 1ull, qm = (1ull << A.qbits) - 1ull, bm = (1ull << A.bbits) - 1ull;

  for (uint32_t b = blockIdx.x; b < *A.count; b += gridDim.x)
    { const uint32_t r  = A.list[b];
      const uint32_t s0 = A.rstart[r];
      const uint32_t n  = A.rstart[r + 1] - s0;
      uint64_t *src = const_cast<uint64_t *>(A.in) + s0, *dst = A.out + s0;
      const uint32_t per = (n + 64 * W - 1) / (64 * W) * 64;          // words of a wavefront
      const uint32_t wb = min(n, (uint32_t) w * per), we = min(n, wb + per);

      // ---- 1. stable LSD passes, src -> dst
      for (int p = 0; p < A.npass; p++)
        { const int sh = A.sh[p], wd = A.wd[p];
          const uint32_t dmask = (1u << wd) - 1u;
          for (int x = tid; x < W * D; x += T) wh[x] = 0;
          __syncthreads();
          for (uint32_t i0 = wb; i0 < we; i0 += 64)
            { const uint32_t i = i0 + lane;
              if (i < we) atomicAdd(&wh[w * D + ((rs_key(src[i]) >> sh) & dmask)], 1u);
            }
          __syncthreads();
          { constexpr int PER = (D + T - 1) / T;
            uint32_t c[PER][W], tot = 0;
#pragma unroll
            for (int q = 0; q < PER; q++)
              { const int d = tid * PER + q;
#pragma unroll
                for (int v = 0; v < W; v++)
                  { c[q][v] = (d < D) ? wh[v * D + d] : 0u;
                    tot += c[q][v];
                  }
              }
            uint32_t all;
            uint32_t run = block_excl_scan_t<T>(tot, ws, &all);
#pragma unroll
            for (int q = 0; q < PER; q++)
              { const int d = tid * PER + q;
#pragma unroll
                for (int v = 0; v < W; v++)
                  { if (d < D) wh[v * D + d] = run;
                    run += c[q][v];
                  }
              }
          }
          __syncthreads();
          for (uint32_t i0 = wb; i0 < we; i0 += 64)                    // (uniform per wavefront)
            { const uint32_t i = i0 + lane;
              const bool ok = i < we;
              const uint64_t x = ok ? src[i] : 0;
              const uint32_t d = (rs_key(x) >> sh) & dmask;
              uint64_t peers = __ballot(ok);
              for (int bb = 0; bb < wd; bb++)
                { const bool     bit = (d >> bb) & 1u;
                  const uint64_t bal = __ballot(bit);
                  peers &= bit ? bal : ~bal;
                }
              const uint32_t below = (uint32_t) __popcll(peers & lt);
              const uint32_t cnt   = (uint32_t) __popcll(peers);
              const uint32_t b0    = ok ? wh[w * D + d] : 0u;
              __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
              __builtin_amdgcn_wave_barrier();
              if (ok && below == 0) wh[w * D + d] = b0 + cnt;
              __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
              __builtin_amdgcn_wave_barrier();
              if (ok) dst[b0 + below] = x;
            }
          __threadfence_block();
          __syncthreads();
          uint64_t *t = src; src = dst; dst = t;
        }

      // ---- 2. clusters: thread t looks at the words [t c, (t + 1) c)
      const uint32_t c = (n + T - 1) / T;
      const uint32_t lo = min(n, (uint32_t) tid * c), hi = min(n, lo + c);
      uint32_t nf = 0;
      for (uint32_t i = lo; i < hi; i++)
        { const uint64_t pv = (i > 0) ? src[i - 1] >> A.pbits : 0, cv = src[i] >> A.pbits;
          if (i == 0 || cluster_start(pv, cv, A.qbits, A.cb)) nf += 1;
        }
      uint32_t ncl;
      const uint32_t base = block_excl_scan_t<T>(nf, ws, &ncl);
      { uint32_t cid = base;
        for (uint32_t i = lo; i < hi; i++)
          { const uint64_t x = src[i];
            const uint64_t pv = (i > 0) ? src[i - 1] >> A.pbits : 0, cv = x >> A.pbits;
            if (i == 0 || cluster_start(pv, cv, A.qbits, A.cb))
              { A.lcl[s0 + cid] = (1ull << 62) | (((x >> (A.pbits + A.qbits)) & bm) << 32) | (uint64_t) (s0 + i);
                cid += 1;
              }
          }
      }
      __threadfence_block();
      __syncthreads();
      { uint32_t cid = base;                                   // (clusters started before word i, this thread's included)
        for (uint32_t i = lo; i < hi; i++)
          { const uint64_t pv = (i > 0) ? src[i - 1] >> A.pbits : 0, cv = src[i] >> A.pbits;
            if (i == 0 || cluster_start(pv, cv, A.qbits, A.cb)) cid += 1;
            else if ((pv >> A.cb) != (cv >> A.cb))
              atomicOr((unsigned long long *) (A.lcl + s0 + cid - 1), 1ull << 63);
          }
      }
      __threadfence_block();
      __syncthreads();
      for (uint32_t j = tid; j < ncl; j += T)
        { const uint64_t v = A.lcl[s0 + j];
          const uint32_t st = (uint32_t) v;
          const uint32_t nx = (j + 1 < ncl) ? (uint32_t) A.lcl[s0 + j + 1] : s0 + n;
          const uint32_t size = nx - st;
          if (size >= A.big_min)
            { if (size >= A.pile_min) { atomicAdd(A.counts + 5, 1u); atomicAdd(A.counts + 6, size); }
              else atomicAdd(A.counts + 4, 1u);
            }
          if ((v >> 63) && size > A.medium) atomicAdd(A.counts + 3, 1u);
        }
      if (tid == 0) A.rncl[r] = ncl;
      if (A.lcnt != nullptr)                                   // (fused route: nothing of this read is chained here)
        for (uint32_t j = tid; j < ncl; j += T) A.lcnt[s0 + j] = 0;
      // ---- 3. the chain kernels' words (apos | bpos << 32), in A.out
      for (uint32_t i = tid; i < n; i += T)
        { const uint64_t x = src[i];
          A.out[s0 + i] = (x & pm) | (((x >> A.pbits) & qm) << 32);
        }
      __threadfence_block();
      __syncthreads();
    }
}

// The reads by segment size, one list per workgroup size of k_read_sort (lists[c] at c * nreads): class 0 up to lim[0]
// hits, class 1 up to lim[1], class 2 up to lim[2], class 3 (k_read_sort_big) everything larger; reads without hits
// are in no list; stats[0..2], stats[4] = list lengths, stats[3] = the largest segment.
__global__ __launch_bounds__(256) void k_read_classes(const uint32_t *__restrict__ rstart, int nreads, uint32_t lim0,
                                                      uint32_t lim1, uint32_t lim2, uint32_t *__restrict__ lists,
                                                      uint32_t *__restrict__ stats)
{ const int r = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint32_t n = (r < nreads) ? rstart[r + 1] - rstart[r] : 0;
  const int c = (n == 0) ? -1 : (n <= lim0 ? 0 : (n <= lim1 ? 1 : (n <= lim2 ? 2 : 3)));
#pragma unroll
  for (int k = 0; k < 4; k++)
    { const uint64_t m = __ballot(c == k);
      if (m == 0) continue;
      uint32_t base = 0;
      if (lane == __ffsll((long long) m) - 1) base = atomicAdd(stats + (k < 3 ? k : 4), (uint32_t) __popcll(m));
      base = __shfl(base, __ffsll((long long) m) - 1);
      if (c == k) lists[(size_t) k * nreads + base + (uint32_t) __popcll(m & lt)] = (uint32_t) r;
    }
  uint32_t v = n;
  for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t) __shfl_down(v, o));
  if (lane == 0 && v > 0) atomicMax(stats + 3, v);
}

// cbase = exclusive scan of rncl: the clusters of read r become clusters cbase[r] ... of the pass.  One wavefront per read.
// Clusters that span coarse buckets and were too large for k_read_sort are listed for the sorts that finish them:
// mlist (up to `medium` hits, k_csort_wide) and blist (larger: the radix sort); counts[0], counts[1] = list lengths.
// gcount != nullptr (fused route): the chain stage's per-group arrays are filled here from lcnt (nullptr: all zero).
__global__ __launch_bounds__(256) void k_cluster_compact(const uint64_t *__restrict__ lcl, const uint32_t *__restrict__ rstart,
                                                         const uint32_t *__restrict__ cbase, int nreads, int bbits,
                                                         uint32_t nhits, uint32_t *__restrict__ gstart,
                                                         uint64_t *__restrict__ gkey, uint32_t small, uint32_t medium,
                                                         uint32_t *__restrict__ mlist, uint32_t *__restrict__ blist,
                                                         unsigned int *__restrict__ counts,
                                                         const uint32_t *__restrict__ lcnt, uint32_t *__restrict__ gcount,
                                                         uint32_t *__restrict__ gbase, uint32_t *__restrict__ gdone)
{ const int r = (int) (((int64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (r >= nreads) return;
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint32_t c0 = cbase[r], nc = cbase[r + 1] - c0, s0 = rstart[r], s1 = rstart[r + 1];
  for (uint32_t j0 = 0; j0 < nc; j0 += 64)
    { const uint32_t j = j0 + lane;
      const bool ok = j < nc;
      const uint64_t v = ok ? lcl[s0 + j] : 0;
      const uint32_t st = (uint32_t) v;
      uint32_t nx = (uint32_t) __shfl_down((int) st, 1);             // the next cluster's first hit
      if (ok && (lane == 63 || j + 1 >= nc)) nx = (j + 1 < nc) ? (uint32_t) lcl[s0 + j + 1] : s1;
      if (ok)
        { gstart[c0 + j] = st;
          gkey[c0 + j]   = ((uint64_t) r << bbits) | ((v >> 32) & 0x3fffffffull);
          if (gcount != nullptr)
            { // fused route: what k_read_sort made of the cluster (chain ends reported, done) as the chain stage keeps it
              const uint32_t lc = (lcnt != nullptr) ? lcnt[s0 + j] : 0u;
              gcount[c0 + j] = lc & 0x7fffffffu;
              gbase[c0 + j]  = (st + (uint32_t) HITMIN - 1u) / (uint32_t) HITMIN;
              gdone[c0 + j]  = lc >> 31;
            }
        }
      const uint32_t sz = ok ? nx - st : 0;
      const bool wide = ok && (v >> 63) && (sz > small || ((v >> 62) & 1ull));      // (bit 62: not ordered where it was found)
      const bool ism = wide && sz <= medium, isb = wide && sz > medium;
      const uint64_t mm = __ballot(ism), mb = __ballot(isb);
      if (mm)
        { uint32_t base = 0;
          if (lane == __ffsll((long long) mm) - 1) base = atomicAdd(counts, (uint32_t) __popcll(mm));
          base = __shfl(base, __ffsll((long long) mm) - 1);
          if (ism) mlist[base + (uint32_t) __popcll(mm & lt)] = c0 + j;
        }
      if (mb)
        { uint32_t base = 0;
          if (lane == __ffsll((long long) mb) - 1) base = atomicAdd(counts + 1, (uint32_t) __popcll(mb));
          base = __shfl(base, __ffsll((long long) mb) - 1);
          if (isb) blist[base + (uint32_t) __popcll(mb & lt)] = c0 + j;
        }
    }
  if (r == nreads - 1 && lane == 0) gstart[cbase[nreads]] = nhits;
}

}  // namespace dmh
