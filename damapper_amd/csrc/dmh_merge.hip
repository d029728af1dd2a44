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
#include <algorithm>

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
constexpr int HREP       = 64;      // replicated histograms / counters to spread atomic traffic
constexpr int SMALLBIN   = 32;

struct MergeEntry { uint32_t a0, ma, b0, mb; };

__global__ __launch_bounds__(MG_THREADS) void k_merge_match(const uint64_t *__restrict__ akeys, int64_t na,
                                                            BLookup B, unsigned long long *__restrict__ gram,
                                                            unsigned long long *__restrict__ nhits,
                                                            MergeEntry *__restrict__ entries,
                                                            unsigned int *__restrict__ nentries)
{ __shared__ MergeEntry stage[MG_TILE];        // matched runs of this tile, flushed with one atomic
  __shared__ uint32_t h[SMALLBIN];
  __shared__ unsigned long long s_hits[4];
  __shared__ uint32_t s_count, s_gbase;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < SMALLBIN) h[tid] = 0;
  if (tid == 0) s_count = 0;
  __syncthreads();

  const int rep = blockIdx.x & (HREP - 1);
  unsigned long long local = 0;

  // The look-ups of a thread's MG_ITEMS codes are independent: issue them phase by phase (codes,
  // bucket bounds, the first two keys of the bucket) so that every thread has MG_ITEMS loads in
  // flight instead of one dependent chain after the other.  The bucket table is sized for ~0.15
  // entries per bucket, so two keys settle all but a few look-ups per thousand; only those run the
  // bisection (a wavefront pays its dependent loads if ANY of its lanes needs it).
  const int64_t tile0 = (int64_t) blockIdx.x * MG_TILE + tid;
  uint64_t c[MG_ITEMS];
  bool     first[MG_ITEMS];
  uint32_t bs[MG_ITEMS], be[MG_ITEMS];
#pragma unroll
  for (int j = 0; j < MG_ITEMS; j++)
    { int64_t i = tile0 + j * MG_THREADS;
      c[j] = (i < na) ? akeys[i] : 0;
      uint64_t prev = (i > 0 && i < na) ? akeys[i - 1] : 0;
      first[j] = (i < na) && (i == 0 || prev != c[j]);
    }
#pragma unroll
  for (int j = 0; j < MG_ITEMS; j++)
    { bs[j] = be[j] = 0;
      if (first[j])
        { uint64_t p = c[j] >> B.shift;
          bs[j] = B.bstart[p];
          be[j] = B.bstart[p + 1];
        }
    }
  uint64_t k0[MG_ITEMS], k1[MG_ITEMS];
#pragma unroll
  for (int j = 0; j < MG_ITEMS; j++)
    { k0[j] = (bs[j] < be[j]) ? B.keys[bs[j]] : ~0ull;
      k1[j] = (bs[j] + 1 < be[j]) ? B.keys[bs[j] + 1] : ~0ull;
    }

#pragma unroll
  for (int j = 0; j < MG_ITEMS; j++)
    { int64_t i = tile0 + j * MG_THREADS;
      unsigned long long ct = 0;
      uint32_t lo = 0, mb = 0;
      int64_t  e = 0;
      bool     hit = false;
      if (first[j] && bs[j] < be[j] && k0[j] <= c[j])
        { const uint32_t sz = be[j] - bs[j];
          const uint64_t cc = c[j];
          if (k0[j] == cc && (sz == 1 || k1[j] != cc)) { lo = bs[j]; mb = 1; }
          else if (k0[j] == cc && sz == 2)              { lo = bs[j]; mb = 2; }
          else if (k0[j] <  cc && sz == 1)              mb = 0;
          else if (k0[j] <  cc && k1[j] > cc)           mb = 0;
          else if (k0[j] <  cc && k1[j] == cc && sz == 2) { lo = bs[j] + 1; mb = 1; }
          else if (k0[j] <  cc && k1[j] <  cc && sz == 2) mb = 0;
          else b_range(B, cc, lo, mb);
          if (mb != 0)
            { e   = (i + 1 < na && akeys[i + 1] == cc) ? run_end(akeys, na, i) : i + 1;
              ct  = (unsigned long long) (e - i) * mb;
              hit = true;
            }
        }
      if (hit)
        { if (ct < SMALLBIN) atomicAdd(&h[(int) ct], 1u);
          else if (ct < MAXGRAM) atomicAdd(&gram[(size_t) rep * (MAXGRAM + 1) + ct], 1ull);
          local += ct;
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
              stage[base + __popcll(m & ((1ull << lane) - 1ull))] = en;
            }
        }
    }
  for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
  if (lane == 0) s_hits[tid >> 6] = local;
  __syncthreads();
  if (tid == 0)
    { unsigned long long t = s_hits[0] + s_hits[1] + s_hits[2] + s_hits[3];
      if (t) atomicAdd(&nhits[rep], t);
      s_gbase = s_count ? atomicAdd(nentries, s_count) : 0;
    }
  if (tid < SMALLBIN && h[tid])
    atomicAdd(&gram[(size_t) rep * (MAXGRAM + 1) + tid], (unsigned long long) h[tid]);
  __syncthreads();
  const uint32_t cnt = s_count, gb = s_gbase;
  for (uint32_t x = tid; x < cnt; x += MG_THREADS)
    entries[gb + x] = stage[x];
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
};

__device__ __forceinline__ void locate(const PosMap &M, uint32_t q, uint32_t &read, uint32_t &rpos)
{ uint32_t r = M.chunk[q >> CHUNK_SHIFT];
  while (M.boff[r + 1] <= (int64_t) q) r += 1;
  read = r;
  rpos = (uint32_t) ((int64_t) q - M.boff[r]);
}

constexpr uint32_t BIG_ENTRY = 64;

// Expand entries into hits (map.c:984-994).  Small entries: one thread each; entries with more
// than BIG_ENTRY pairs are queued for the wavefront-per-entry kernel below.
__global__ void k_merge_expand(const MergeEntry *__restrict__ entries, uint32_t n,
                               const uint32_t *__restrict__ offs, const uint32_t *__restrict__ sizes,
                               const uint32_t *__restrict__ avals, const uint32_t *__restrict__ bvals,
                               PosMap MA, PosMap MB, int bshift, int pshift,
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
      uint64_t kbase = ((uint64_t) ar << bshift) | (uint64_t) ap;
      for (uint32_t b = 0; b < e.mb; b++)
        { uint32_t br, bp;
          locate(MB, bvals[e.b0 + b], br, bp);
          hkeys[off] = kbase | ((uint64_t) br << pshift);
          hvals[off] = bp;
          off += 1;
        }
    }
}

__global__ __launch_bounds__(256) void k_merge_expand_big(const MergeEntry *__restrict__ entries,
                                                          const uint32_t *__restrict__ biglist, uint32_t nbig,
                                                          const uint32_t *__restrict__ offs,
                                                          const uint32_t *__restrict__ avals,
                                                          const uint32_t *__restrict__ bvals,
                                                          PosMap MA, PosMap MB, int bshift, int pshift,
                                                          uint64_t *__restrict__ hkeys, uint32_t *__restrict__ hvals)
{ const int lane = threadIdx.x & 63;
  uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (w >= nbig) return;
  uint32_t   i = biglist[w];
  MergeEntry e = entries[i];
  uint32_t off = offs[i];
  uint32_t tot = e.ma * e.mb;
  for (uint32_t t = lane; t < tot; t += 64)
    { uint32_t a = t / e.mb, b = t - a * e.mb;
      uint32_t ar, ap, br, bp;
      locate(MA, avals[e.a0 + a], ar, ap);
      locate(MB, bvals[e.b0 + b], br, bp);
      hkeys[off + t] = ((uint64_t) ar << bshift) | ((uint64_t) br << pshift) | (uint64_t) ap;
      hvals[off + t] = bp;
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

void merge_and_sort(MergeResult &R, const DAZZ_DB *ablock, const DAZZ_DB *bblock,
                    IndexDev &A, IndexDev &Bx, int Kshift)
{ State &st = S();
  HostTimer ht("merge_and_sort");
  R.nhits = 0;
  R.ngroups = 0;

  // --- key layout (map.c:2911-2937 uses whole bytes; bit-packing keeps the same order)
  R.pbits = ceil_log2(ablock->maxlen > 1 ? ablock->maxlen : 1);
  R.bbits = ceil_log2(bblock->nreads);
  R.abits = ceil_log2(ablock->nreads);
  if (R.pbits + R.bbits + R.abits > 64)
    fatal("Match_Filter: seed key needs %d bits", R.pbits + R.bbits + R.abits);
  int pshift = R.pbits, bshift = R.pbits + R.bbits;

  // --- bucket table over the REF index: part of the index, built the first time it is merged against
  //     (a retained REF index meets every READS block)
  if (Bx.d_bstart == nullptr)
    { int P = ceil_log2(Bx.n) + 2;         // ~0.15 entries per bucket (up to 1 GB of table)
      if (P < 8) P = 8;
      if (P > 28) P = 28;
      if (P > Kshift) P = Kshift;
      Bx.bshift = Kshift - P;
      int64_t nbucket = (int64_t) 1 << P;
      Bx.d_bstart = (uint32_t *) dev_alloc(sizeof(uint32_t) * (nbucket + 2), "bucket table");
      StageTimer tm("merge_buckets", 8.0 * Bx.n);
      hipLaunchKernelGGL(k_bucket_bounds, dim3(ceil_div(Bx.n + 1, 256)), dim3(256), 0, stream(),
                         (const uint64_t *) Bx.keys, Bx.n, Bx.bshift, nbucket, Bx.d_bstart);
    }
  BLookup BL{Bx.keys, Bx.d_bstart, Bx.bshift};

  // --- match + count (map.c:2984-2988)
  static DevBuf<unsigned long long> gram;
  static DevBuf<MergeEntry> entries;
  static DevBuf<unsigned int> mctl;
  const size_t GR = (size_t) HREP * (MAXGRAM + 1);
  gram.ensure(GR + HREP, "hit histogram");
  mctl.ensure(4, "merge counters");
  int64_t ecap = std::min<int64_t>(A.n, Bx.n) + 64;
  entries.ensure(ecap, "matched runs");
  HIP_CHECK(hipMemsetAsync(gram.p, 0, sizeof(unsigned long long) * (GR + HREP), stream()));
  HIP_CHECK(hipMemsetAsync(mctl.p, 0, 4 * sizeof(unsigned int), stream()));
  { StageTimer tm("merge_match", 16.0 * (A.n + Bx.n));
    hipLaunchKernelGGL(k_merge_match, dim3(ceil_div(A.n, MG_TILE)), dim3(MG_THREADS), 0, stream(),
                       (const uint64_t *) A.keys, A.n, BL, gram.p, gram.p + GR, entries.p, mctl.p);
    HIP_CHECK(hipGetLastError());
  }
  std::vector<unsigned long long> hrep(GR + HREP);
  unsigned int nent = 0;
  copy_d2h(&nent, mctl.p, sizeof(unsigned int));
  copy_d2h(hrep.data(), gram.p, sizeof(unsigned long long) * (GR + HREP), 2);   // the host reduces it next
  std::vector<unsigned long long> histo(MAXGRAM + 1, 0);
  for (int r = 0; r < HREP; r++)
    { for (int j = 0; j < MAXGRAM; j++) histo[j] += hrep[(size_t) r * (MAXGRAM + 1) + j];
      histo[MAXGRAM] += hrep[GR + r];
    }

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

  // --- expand the entries under the limit into hits (map.c:3095-3099)
  R.keys.ensure(nhits, "seed keys");
  R.vals.ensure(nhits, "seed positions");
  st.sort.keys_alt.ensure(nhits, "seed sort buffer");
  st.sort.vals_alt.ensure(nhits, "seed sort buffer");
  static DevBuf<uint32_t> esz, eoff, ebig;
  esz.ensure((size_t) nent + 1, "entry sizes");
  eoff.ensure((size_t) nent + 2, "entry offsets");
  ebig.ensure((size_t) nent + 1, "big entries");
  uint32_t total = 0, nbig = 0;
  { StageTimer tm("merge_emit", 16.0 * nhits);
    int nb = ceil_div(nent, 256);
    hipLaunchKernelGGL(k_entry_sizes, dim3(nb), dim3(256), 0, stream(), (const MergeEntry *) entries.p, nent, limit, esz.p);
    HIP_CHECK(hipMemcpyAsync(eoff.p, esz.p, sizeof(uint32_t) * nent, hipMemcpyDeviceToDevice, stream()));
    exclusive_scan_u32(eoff.p, nent, st.sort.bsum, eoff.p + nent);
    PosMap MA{A.d_boff, A.d_chunk}, MB{Bx.d_boff, Bx.d_chunk};
    hipLaunchKernelGGL(k_merge_expand, dim3(nb), dim3(256), 0, stream(), (const MergeEntry *) entries.p, nent,
                       (const uint32_t *) eoff.p, (const uint32_t *) esz.p, (const uint32_t *) A.vals,
                       (const uint32_t *) Bx.vals, MA, MB, bshift, pshift, R.keys.p, R.vals.p, ebig.p, mctl.p + 1);
    copy_d2h(&total, eoff.p + nent, sizeof(uint32_t));
    copy_d2h(&nbig, mctl.p + 1, sizeof(unsigned int), 1);
    if (nbig > 0)
      hipLaunchKernelGGL(k_merge_expand_big, dim3(ceil_div((int64_t) nbig * 64, 256)), dim3(256), 0, stream(),
                         (const MergeEntry *) entries.p, (const uint32_t *) ebig.p, nbig, (const uint32_t *) eoff.p,
                         (const uint32_t *) A.vals, (const uint32_t *) Bx.vals, MA, MB, bshift, pshift,
                         R.keys.p, R.vals.p);
    HIP_CHECK(hipGetLastError());
  }
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
    copy_d2h(&ng, flag.p + nhits, sizeof(uint32_t), 1);
    R.ngroups = ng;
    R.gstart.ensure((size_t) ng + 1, "group starts");
    hipLaunchKernelGGL(k_group_write, dim3(nb), dim3(256), 0, stream(), (const uint64_t *) R.skeys, nhits, pshift,
                       (const uint32_t *) flag.p, R.gstart.p);
    uint32_t endv = (uint32_t) nhits;
    copy_h2d(R.gstart.p + ng, &endv, sizeof(uint32_t));
    HIP_CHECK(hipStreamSynchronize(stream()));
  }

  if (st.capture_seeds)
    { std::vector<uint64_t> hk(nhits);
      std::vector<uint32_t> hv(nhits);
      copy_d2h(hk.data(), R.skeys, sizeof(uint64_t) * nhits);
      copy_d2h(hv.data(), R.svals, sizeof(uint32_t) * nhits);
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
