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
};

// [lo, lo+cnt) = entries of the REF index with code c
__device__ __forceinline__ void b_range(const BLookup &B, uint64_t c, uint32_t &lo, uint32_t &cnt)
{ uint64_t p = c >> B.shift;
  uint32_t s = B.bstart[p], e = B.bstart[p + 1];
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
//  Count: histogram of group products (map.c:881-934)
// ---------------------------------------------------------------------------------------

constexpr int MG_THREADS = 256;
constexpr int MG_ITEMS   = 8;
constexpr int MG_TILE    = MG_THREADS * MG_ITEMS;

__global__ __launch_bounds__(MG_THREADS) void k_merge_count(const uint64_t *__restrict__ akeys, int64_t na,
                                                            BLookup B, unsigned long long *__restrict__ gram,
                                                            unsigned long long *__restrict__ nhits)
{ __shared__ uint32_t h[MAXGRAM];
  for (int x = threadIdx.x; x < MAXGRAM; x += MG_THREADS) h[x] = 0;
  __syncthreads();
  unsigned long long local = 0;
  int64_t base = (int64_t) blockIdx.x * MG_TILE;
  for (int j = 0; j < MG_ITEMS; j++)
    { int64_t i = base + j * MG_THREADS + threadIdx.x;
      if (i >= na) continue;
      uint64_t c = akeys[i];
      if (i > 0 && akeys[i - 1] == c) continue;          // not the first of its run
      int64_t  e = run_end(akeys, na, i);
      uint32_t lo, mb;
      b_range(B, c, lo, mb);
      if (mb == 0) continue;
      unsigned long long ct = (unsigned long long) (e - i) * mb;
      local += ct;
      if (ct < MAXGRAM) atomicAdd(&h[(int) ct], 1u);
    }
  // block reduce of local
  for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(nhits, local);
  __syncthreads();
  for (int x = threadIdx.x; x < MAXGRAM; x += MG_THREADS)
    if (h[x]) atomicAdd(&gram[x], (unsigned long long) h[x]);
}

// ---------------------------------------------------------------------------------------
//  Emit (map.c:939-1002): two passes over the READ index, sizes then records
// ---------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t hits_of(const uint64_t *__restrict__ akeys, int64_t na, int64_t i,
                                            const BLookup &B, int64_t limit, uint32_t &lo)
{ uint64_t c = akeys[i];
  uint32_t mb;
  b_range(B, c, lo, mb);
  if (mb == 0) return 0;
  int64_t ma;
  bool ls = (i == 0) || (akeys[i - 1] != c);
  bool le = (i + 1 >= na) || (akeys[i + 1] != c);
  if (ls && le) ma = 1;
  else ma = run_end(akeys, na, i) - run_start(akeys, i);
  if (ma * (int64_t) mb < limit) return mb;
  return 0;
}

__global__ __launch_bounds__(MG_THREADS) void k_merge_sizes(const uint64_t *__restrict__ akeys, int64_t na,
                                                            BLookup B, int64_t limit,
                                                            uint32_t *__restrict__ bsum)
{ __shared__ uint32_t ws[4];
  int64_t base = (int64_t) blockIdx.x * MG_TILE + (int64_t) threadIdx.x * MG_ITEMS;
  uint32_t s = 0;
  for (int j = 0; j < MG_ITEMS; j++)
    { int64_t i = base + j;
      uint32_t lo;
      if (i < na) s += hits_of(akeys, na, i, B, limit, lo);
    }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) bsum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

struct PosMap
{ const int64_t  *boff;
  const uint32_t *chunk;
};

__device__ __forceinline__ void locate(const PosMap &M, uint32_t q, uint32_t &read, uint32_t &rpos)
{ uint32_t r = M.chunk[q >> CHUNK_SHIFT];
  while (M.boff[r + 1] <= (int64_t) q) r += 1;
  read = r;
  rpos = (uint32_t) ((int64_t) q - M.boff[r]);
}

__global__ __launch_bounds__(MG_THREADS) void k_merge_emit(const uint64_t *__restrict__ akeys,
                                                           const uint32_t *__restrict__ avals, int64_t na,
                                                           BLookup B, const uint32_t *__restrict__ bvals,
                                                           int64_t limit, const uint32_t *__restrict__ bsum,
                                                           PosMap MA, PosMap MB, int bshift, int pshift,
                                                           uint64_t *__restrict__ hkeys,
                                                           uint32_t *__restrict__ hvals)
{ __shared__ uint32_t ws[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int64_t base = (int64_t) blockIdx.x * MG_TILE + (int64_t) threadIdx.x * MG_ITEMS;
  uint32_t cnt[MG_ITEMS], lo[MG_ITEMS];
  uint32_t s = 0;
  for (int j = 0; j < MG_ITEMS; j++)
    { int64_t i = base + j;
      cnt[j] = 0;
      lo[j]  = 0;
      if (i < na) cnt[j] = hits_of(akeys, na, i, B, limit, lo[j]);
      s += cnt[j];
    }
  // exclusive scan of s across the block
  uint32_t x = s;
  for (int o = 1; o < 64; o <<= 1)
    { uint32_t y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
  if (lane == 63) ws[w] = x;
  __syncthreads();
  uint32_t off = bsum[blockIdx.x] + x - s;
  for (int i = 0; i < w; i++) off += ws[i];

  for (int j = 0; j < MG_ITEMS; j++)
    { if (cnt[j] == 0) continue;
      int64_t  i = base + j;
      uint32_t ar, ap;
      locate(MA, avals[i], ar, ap);
      uint64_t kbase = ((uint64_t) ar << bshift) | (uint64_t) ap;
      for (uint32_t t = 0; t < cnt[j]; t++)
        { uint32_t br, bp;
          locate(MB, bvals[lo[j] + t], br, bp);
          hkeys[off] = kbase | ((uint64_t) br << pshift);
          hvals[off] = bp;
          off += 1;
        }
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
                              const uint32_t *__restrict__ slot, uint32_t *__restrict__ gstart)
{ int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (i == 0 || (hkeys[i] >> pshift) != (hkeys[i - 1] >> pshift))
    gstart[slot[i]] = (uint32_t) i;
}

// ---------------------------------------------------------------------------------------
//  Host orchestration
// ---------------------------------------------------------------------------------------

static int ceil_log2(int64_t x)      // smallest b with 2^b >= x
{ int b = 0;
  while (((int64_t) 1 << b) < x) b += 1;
  return b;
}

// DB.c:1044-1076 restated for a block without tracks
static int64_t sizeof_db(const DAZZ_DB *db)
{ return 112 + 40 * ((int64_t) db->nreads + 2) + (int64_t) strlen(db->path) + 1
         + (db->totlen + db->nreads + 4);
}

void merge_and_sort(MergeResult &R, const DAZZ_DB *ablock, const DAZZ_DB *bblock,
                    IndexDev &A, IndexDev &Bx, int Kshift)
{ State &st = S();
  R.nhits = 0;
  R.ngroups = 0;

  // --- key layout (map.c:2911-2937 uses whole bytes; bit-packing keeps the same order)
  R.pbits = ceil_log2(ablock->maxlen > 1 ? ablock->maxlen : 1);
  R.bbits = ceil_log2(bblock->nreads);
  R.abits = ceil_log2(ablock->nreads);
  if (R.pbits + R.bbits + R.abits > 64)
    fatal("Match_Filter: seed key needs %d bits", R.pbits + R.bbits + R.abits);
  int pshift = R.pbits, bshift = R.pbits + R.bbits;

  // --- bucket table over the REF index
  int P = ceil_log2(Bx.n);
  if (P < 8) P = 8;
  if (P > 28) P = 28;
  if (P > Kshift) P = Kshift;
  int     shift   = Kshift - P;
  int64_t nbucket = (int64_t) 1 << P;
  static DevBuf<uint32_t> bstart;
  bstart.ensure(nbucket + 2, "bucket table");
  { StageTimer tm("merge_buckets", 8.0 * Bx.n);
    hipLaunchKernelGGL(k_bucket_bounds, dim3(ceil_div(Bx.n + 1, 256)), dim3(256), 0, stream(),
                       (const uint64_t *) Bx.keys, Bx.n, shift, nbucket, bstart.p);
  }
  BLookup BL{Bx.keys, bstart.p, shift};

  // --- count (map.c:2984-2988)
  static DevBuf<unsigned long long> gram;
  gram.ensure(MAXGRAM + 1, "hit histogram");
  HIP_CHECK(hipMemsetAsync(gram.p, 0, sizeof(unsigned long long) * (MAXGRAM + 1), stream()));
  int nblk = ceil_div(A.n, MG_TILE);
  { StageTimer tm("merge_count", 16.0 * (A.n + Bx.n));
    hipLaunchKernelGGL(k_merge_count, dim3(nblk), dim3(MG_THREADS), 0, stream(), (const uint64_t *) A.keys, A.n,
                       BL, gram.p, gram.p + MAXGRAM);
  }
  std::vector<unsigned long long> histo(MAXGRAM + 1);
  HIP_CHECK(hipMemcpyAsync(histo.data(), gram.p, sizeof(unsigned long long) * (MAXGRAM + 1),
                           hipMemcpyDeviceToHost, stream()));
  HIP_CHECK(hipStreamSynchronize(stream()));

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
      int64_t tom = 0;
      int j;
      for (j = 0; j < MAXGRAM; j++)
        { tom += j * (int64_t) histo[j];
          if (tom > avail) break;
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
      nhits = 0;
      for (j = 1; j < limit; j++) nhits += j * (int64_t) histo[j];
    }
  else
    { limit = INT32_MAX;
      nhits = (int64_t) histo[MAXGRAM];
    }
  st.last_limit = (int) limit;
  R.limit = limit;
  if (VERBOSE)
    { printf("   Hit count = %lld\n", (long long) nhits);
      fflush(stdout);
    }
  if (nhits >= (int64_t) 0xfffffff0ll)
    fatal("Match_Filter: %lld seed hits in one block pair exceed the 32-bit slot range; use smaller blocks or -M",
          (long long) nhits);
  R.nhits = nhits;
  if (nhits == 0) return;

  // --- emit (map.c:3095-3099)
  R.keys.ensure(nhits, "seed keys");
  R.vals.ensure(nhits, "seed positions");
  st.sort.keys_alt.ensure(nhits, "seed sort buffer");
  st.sort.vals_alt.ensure(nhits, "seed sort buffer");
  static DevBuf<uint32_t> msum;
  msum.ensure(nblk + 1, "merge block sums");
  { StageTimer tm("merge_emit", 16.0 * (A.n + Bx.n) + 16.0 * nhits);
    hipLaunchKernelGGL(k_merge_sizes, dim3(nblk), dim3(MG_THREADS), 0, stream(), (const uint64_t *) A.keys, A.n,
                       BL, limit, msum.p);
    exclusive_scan_u32(msum.p, nblk, st.sort.bsum, msum.p + nblk);
    PosMap MA{A.d_boff, A.d_chunk}, MB{Bx.d_boff, Bx.d_chunk};
    hipLaunchKernelGGL(k_merge_emit, dim3(nblk), dim3(MG_THREADS), 0, stream(), (const uint64_t *) A.keys,
                       (const uint32_t *) A.vals, A.n, BL, (const uint32_t *) Bx.vals, limit,
                       (const uint32_t *) msum.p, MA, MB, bshift, pshift, R.keys.p, R.vals.p);
    HIP_CHECK(hipGetLastError());
  }
  uint32_t total = 0;
  HIP_CHECK(hipMemcpyAsync(&total, msum.p + nblk, sizeof(uint32_t), hipMemcpyDeviceToHost, stream()));
  HIP_CHECK(hipStreamSynchronize(stream()));
  if ((int64_t) total != nhits)
    fatal("Match_Filter: emitted %u hits but the histogram promised %lld", total, (long long) nhits);

  // --- seed sort (map.c:3110-3126)
  uint64_t *k0 = R.keys.p, *k1 = st.sort.keys_alt.p;
  uint32_t *v0 = R.vals.p, *v1 = st.sort.vals_alt.p;
  radix_sort_pairs(&k0, &v0, &k1, &v1, nhits, 0, R.pbits + R.bbits + R.abits, st.sort, "seed_sort");
  R.skeys = k0;
  R.svals = v0;

  // --- groups
  static DevBuf<uint32_t> flag;
  flag.ensure(nhits + 1, "group flags");
  { StageTimer tm("seed_groups", 24.0 * nhits);
    int nb = ceil_div(nhits, 256);
    hipLaunchKernelGGL(k_group_flag, dim3(nb), dim3(256), 0, stream(), (const uint64_t *) R.skeys, nhits, pshift, flag.p);
    exclusive_scan_u32(flag.p, nhits, st.sort.bsum, flag.p + nhits);
    uint32_t ng = 0;
    HIP_CHECK(hipMemcpyAsync(&ng, flag.p + nhits, sizeof(uint32_t), hipMemcpyDeviceToHost, stream()));
    HIP_CHECK(hipStreamSynchronize(stream()));
    R.ngroups = ng;
    R.gstart.ensure((size_t) ng + 1, "group starts");
    hipLaunchKernelGGL(k_group_write, dim3(nb), dim3(256), 0, stream(), (const uint64_t *) R.skeys, nhits, pshift,
                       (const uint32_t *) flag.p, R.gstart.p);
    uint32_t endv = (uint32_t) nhits;
    HIP_CHECK(hipMemcpyAsync(R.gstart.p + ng, &endv, sizeof(uint32_t), hipMemcpyHostToDevice, stream()));
    HIP_CHECK(hipStreamSynchronize(stream()));
  }

  if (st.capture_seeds)
    { std::vector<uint64_t> hk(nhits);
      std::vector<uint32_t> hv(nhits);
      HIP_CHECK(hipMemcpy(hk.data(), R.skeys, sizeof(uint64_t) * nhits, hipMemcpyDeviceToHost));
      HIP_CHECK(hipMemcpy(hv.data(), R.svals, sizeof(uint32_t) * nhits, hipMemcpyDeviceToHost));
      st.last_seeds.resize((size_t) nhits * 4);
      uint64_t pm = ((uint64_t) 1 << R.pbits) - 1, bm = ((uint64_t) 1 << R.bbits) - 1;
      for (int64_t i = 0; i < nhits; i++)
        { int apos  = (int) (hk[i] & pm);
          int bread = (int) ((hk[i] >> pshift) & bm);
          int aread = (int) (hk[i] >> bshift);
          st.last_seeds[4 * i + 0] = apos - (int) hv[i];
          st.last_seeds[4 * i + 1] = apos;
          st.last_seeds[4 * i + 2] = bread;
          st.last_seeds[4 * i + 3] = aread;
        }
    }
}

}  // namespace dmh
