// dmh_readsort.h -- seed hits of a dense pass put into chain order PER READ, in LDS (included by dmh_merge.hip).
//
// The reference sorts all hits of a Match_Filter call on (aread, bread, apos) bytes (map.c:2911-2937, 3110-3126); what
// the chain stage needs from that is only the resulting order.  In cluster mode (dmh_merge.h) round 2 got there by
// four global radix passes over one word per hit, two numbering passes and a sort of every cluster where it lies:
// ~150 bytes of HBM traffic per hit.  A read, however, meets a few thousand hits per pass -- 30 KB: its whole order
// fits the LDS of one workgroup.  So the merge writes every hit straight into its READ's segment (the segments are
// laid out by an atomic count per row of the merge and a scan over the reads: dmh_merge.hip k_row_place), and one
// workgroup per read
//   1. sorts the segment by the coarse key (bread, bpos >> cb) -- stable LSD radix passes of up to 9 bits, ranks from
//      wavefront ballots, the words held in registers between reading and re-writing the one LDS copy;
//   2. cuts it into B-clusters (pair changes, or a gap of more than one coarse bucket: the definition of round 2's
//      k_wcluster_count, so both routes give the same clusters);
//   3. puts every cluster of up to RS_IN hits into (apos, bpos) order by rank counting in LDS and writes the chain
//      kernels' words (apos | bpos << 32) to their final places; larger clusters are written as they are and finished
//      by k_csort_wide (dmh_merge.hip), the rare repeat piles by the radix sort;
//   4. leaves one record per cluster (first hit, contig) for k_cluster_compact, which numbers the clusters of all
//      reads (scan over the reads' counts) into gstart / gkey.
// The order inside a segment as the merge leaves it is arbitrary (atomic placement), which is harmless: the hits of a
// read are distinct in (bread, bpos, apos), and every step above ends in a total order on those.
// HBM traffic per hit: 8 B written by the merge, 8 B read and 8 B written here.
#pragma once

namespace dmh {

constexpr int RS_E  = 16;        // words per thread: a workgroup of T threads takes segments of up to 16 T hits
constexpr int RS_IN = 64;        // clusters of up to this many hits are ordered inside k_read_sort (= CS_SMALL)
constexpr int RS_DB = 9;         // most bits of a radix digit

struct ReadSortArgs
{ const uint64_t *in;            // the pass's words, read by read (w = aread | bread | bpos | apos)
  uint64_t       *out;           // the chain kernels' words, same positions
  const uint32_t *rstart;        // first word of every read (+ end)
  uint32_t        nlo, nhi;      // this launch takes the reads with nlo < hits <= nhi
  int             pbits, qbits, bbits, cb;
  int             npass, dbits, kbits;       // radix passes of dbits over the kbits of the coarse key
  uint64_t       *lcl;           // per cluster, at rstart[r] + j: contig << 32 | first word
  uint32_t       *rncl;          // clusters per read
};

template <int T>
__device__ __forceinline__ uint32_t block_excl_scan_t(uint32_t v, uint32_t *ws, uint32_t *total)
{ constexpr int W = T / 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t x = wave_incl_scan(v, lane);
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

template <int T>
__global__ __launch_bounds__(T) void k_read_sort(ReadSortArgs A)
{ constexpr int W = T / 64, D = 1 << RS_DB, E = RS_E;
  extern __shared__ __attribute__((aligned(16))) uint64_t rs_lds[];
  uint64_t *buf   = rs_lds;                                   // T * E words
  uint16_t *wh    = (uint16_t *) (buf + (size_t) T * E);      // W x D digit counters, then offsets
  uint64_t *flagw = (uint64_t *) (wh + (size_t) W * D);       // cluster-start bits, one word per 64 hits (+ sentinel)
  uint16_t *wpref = (uint16_t *) (flagw + (T * E / 64 + 2));  // starts before each flag word
  __shared__ uint32_t ws[W];

  const uint32_t r  = blockIdx.x;
  const uint32_t s0 = A.rstart[r];
  const uint32_t n  = A.rstart[r + 1] - s0;
  if (n <= A.nlo || n > A.nhi) return;                        // (uniform: another launch's read, or none)

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t rounds = (n + 64 * W - 1) / (64 * W);        // <= E
  const uint32_t wbase  = (uint32_t) w * rounds * 64;         // the wavefront's words: [wbase, wbase + 64 rounds)
  const uint64_t lt = (1ull << lane) - 1ull;

  uint64_t key[E];
#pragma unroll
  for (int j = 0; j < E; j++)
    { const uint32_t i = wbase + 64u * j + lane;
      key[j] = ((uint32_t) j < rounds && i < n) ? __builtin_nontemporal_load(A.in + s0 + i) : ~0ull;
    }

  // ---- 1. stable LSD passes over the coarse key
  uint16_t *mywh = wh + (size_t) w * D;
  for (int p = 0; p < A.npass; p++)
    { const int sh = A.pbits + A.cb + p * A.dbits;
      const int wd = (A.kbits - p * A.dbits < A.dbits) ? A.kbits - p * A.dbits : A.dbits;
      const uint32_t dmask = (1u << wd) - 1u;
      for (int x = tid; x < W * D; x += T) wh[x] = 0;
      __syncthreads();
      uint32_t rk[E];
#pragma unroll
      for (int j = 0; j < E; j++)
        { if ((uint32_t) j < rounds)                           // (uniform)
            { const uint32_t i  = wbase + 64u * j + lane;
              const bool     ok = i < n;
              const uint32_t d  = (uint32_t) (key[j] >> sh) & dmask;
              uint64_t peers = __ballot(ok);
#pragma unroll
              for (int b = 0; b < RS_DB; b++)
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
        }
      __syncthreads();
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
      __syncthreads();
#pragma unroll
      for (int j = 0; j < E; j++)
        if ((uint32_t) j < rounds)
          { const uint32_t i = wbase + 64u * j + lane;
            if (i < n)
              { const uint32_t d = (uint32_t) (key[j] >> sh) & dmask;
                buf[rk[j] + (uint32_t) mywh[d]] = key[j];
              }
          }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < E; j++)
        if ((uint32_t) j < rounds)
          { const uint32_t i = wbase + 64u * j + lane;
            key[j] = (i < n) ? buf[i] : ~0ull;
          }
      __syncthreads();                                         // (everybody has its words: buf and wh are free again)
    }
  if (A.npass == 0)
    { // (a key of no bits: one contig and one bucket -- the words only have to be in LDS)
#pragma unroll
      for (int j = 0; j < E; j++)
        if ((uint32_t) j < rounds)
          { const uint32_t i = wbase + 64u * j + lane;
            if (i < n) buf[i] = key[j];
          }
      __syncthreads();
    }

  // ---- 2. cluster starts: bit i of the flag words; the words of a wavefront are consecutive, so a ballot is a flag word
  const uint32_t nw = n / 64 + 1;                              // flag words, the sentinel bit n included
#pragma unroll
  for (int j = 0; j < E; j++)
    if ((uint32_t) j < rounds)
      { const uint32_t i = wbase + 64u * j + lane;
        bool f = false;
        if (i < n)
          f = (i == 0) || cluster_start(buf[i - 1] >> A.pbits, key[j] >> A.pbits, A.qbits, A.cb);
        else if (i == n)
          f = true;                                            // sentinel: the end of the last cluster
        const uint64_t m = __ballot(f);
        if (lane == 0 && (wbase + 64u * j) / 64 < nw) flagw[(wbase + 64u * j) / 64] = m;
      }
  // (the sentinel lies in the word after the last wavefront's last round when n is a multiple of 64 W)
  if (tid == 0 && n == rounds * 64 * W) flagw[n / 64] = 1ull;
  __syncthreads();
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
  __syncthreads();
  // the words become (apos << 32 | bpos): what the clusters are ordered by, and the chain kernels' word with its
  // halves swapped; every start leaves its cluster's record
  const uint64_t pm = (1ull << A.pbits) - 1ull, qm = (1ull << A.qbits) - 1ull, bm = (1ull << A.bbits) - 1ull;
#pragma unroll
  for (int j = 0; j < E; j++)
    if ((uint32_t) j < rounds)
      { const uint32_t i = wbase + 64u * j + lane;
        if (i < n)
          { const uint64_t x = key[j];
            key[j] = ((x & pm) << 32) | ((x >> A.pbits) & qm);
            buf[i] = key[j];
            const uint64_t fw = flagw[i >> 6];
            if ((fw >> (i & 63)) & 1ull)
              { const uint32_t c = (uint32_t) wpref[i >> 6] + (uint32_t) __popcll(fw & ((1ull << (i & 63)) - 1ull));
                A.lcl[s0 + c] = (((x >> (A.pbits + A.qbits)) & bm) << 32) | (uint64_t) (s0 + i);
              }
          }
      }
  if (tid == 0) A.rncl[r] = ncl;
  __syncthreads();

  // ---- 3. clusters of up to RS_IN hits: rank by counting; the others keep the order they have
#pragma unroll
  for (int j = 0; j < E; j++)
    if ((uint32_t) j < rounds)
      { const uint32_t i = wbase + 64u * j + lane;
        if (i < n)
          { const uint64_t k = key[j];
            const uint32_t wi = i >> 6, bi = i & 63;
            // start of my cluster: the highest flag at or below i (in this flag word or the one before -- a cluster
            // that begins earlier is longer than RS_IN anyway); its end: the lowest flag above i (this word or the
            // next, the same way)
            uint32_t s = 0xffffffffu, e = 0xffffffffu;
            { const uint64_t m = flagw[wi] & (~0ull >> (63 - bi));
              if (m) s = wi * 64 + 63 - (uint32_t) __clzll((long long) m);
              else if (wi > 0)
                { const uint64_t m2 = flagw[wi - 1];
                  if (m2) s = (wi - 1) * 64 + 63 - (uint32_t) __clzll((long long) m2);
                }
            }
            { const uint64_t m = (bi == 63) ? 0ull : (flagw[wi] & (~0ull << (bi + 1)));
              if (m) e = wi * 64 + (uint32_t) __ffsll((long long) m) - 1;
              else if (wi + 1 < nw)
                { const uint64_t m2 = flagw[wi + 1];
                  if (m2) e = (wi + 1) * 64 + (uint32_t) __ffsll((long long) m2) - 1;
                }
            }
            uint32_t dst = i;
            if (s != 0xffffffffu && e != 0xffffffffu && e - s <= (uint32_t) RS_IN)
              { uint32_t rank = 0;
                for (uint32_t q = s; q < e; q++) rank += (buf[q] < k) ? 1u : 0u;
                dst = s + rank;
              }
            A.out[s0 + dst] = (k << 32) | (k >> 32);
          }
      }
}

// dynamic LDS of k_read_sort<T>
template <int T> constexpr size_t read_sort_lds()
{ return sizeof(uint64_t) * (size_t) T * RS_E + sizeof(uint16_t) * (size_t) (T / 64) * (1 << RS_DB)
         + sizeof(uint64_t) * (T * RS_E / 64 + 2) + sizeof(uint16_t) * (T * RS_E / 64 + 2) + 64;
}

// cbase = exclusive scan of rncl: the clusters of read r become clusters cbase[r] ... of the pass.  One wavefront per read.
__global__ __launch_bounds__(256) void k_cluster_compact(const uint64_t *__restrict__ lcl, const uint32_t *__restrict__ rstart,
                                                         const uint32_t *__restrict__ cbase, int nreads, int bbits,
                                                         uint32_t nhits, uint32_t *__restrict__ gstart,
                                                         uint64_t *__restrict__ gkey)
{ const int r = (int) (((int64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (r >= nreads) return;
  const uint32_t c0 = cbase[r], nc = cbase[r + 1] - c0, s0 = rstart[r];
  for (uint32_t j = lane; j < nc; j += 64)
    { const uint64_t v = lcl[s0 + j];
      gstart[c0 + j] = (uint32_t) v;
      gkey[c0 + j]   = ((uint64_t) r << bbits) | (v >> 32);
    }
  if (r == nreads - 1 && lane == 0) gstart[cbase[nreads]] = nhits;
}

}  // namespace dmh
