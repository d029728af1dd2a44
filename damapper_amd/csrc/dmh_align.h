// dmh_align.h -- device side of the local alignment (align.c:353-1946) for one wavefront.
//
// One wavefront runs one Local_Alignment: lane l of a 64-diagonal chunk owns diagonal base+l of
// the d-wave, the furthest-reaching state of the band (V, 60-column match vector T, match count
// M, trace-pebble heads HA/HB, next trace lines NA/NB) lives in a ring in LDS, double buffered so
// that every diagonal of a wave is computed from the frozen previous wave.  The order-dependent
// parts of the reference's sequential loop become wavefront scans/ballots:
//   * best / last / trim point  = record setters of a running max (min for the reverse wave),
//   * boundary clips            = min/max diagonal that ran into a sequence terminator,
//   * trace pebbles             = bump-allocated from a per-wavefront arena in global memory.
// Restricted to damapper's use: one seed diagonal, no diagonal borders (map.c:2513).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dmh {

constexpr int      AL_PATH_LEN  = 60;                      // align.c:169
constexpr uint64_t AL_PATH_TOP  = 0x1000000000000000ull;   // align.c:173
constexpr uint64_t AL_PATH_INT  = 0x0fffffffffffffffull;   // align.c:174
constexpr int      AL_TRIM_LEN  = 15;                      // align.c:163
constexpr int      AL_TRIM_MASK = 0x7fff;
constexpr int      AL_TRIM_MLAG = 250;                     // align.c:176
constexpr int      AL_WAVE_LAG  = 30;                      // align.c:177
constexpr int      AL_DUB_TRIM  = 45;                      // align.c:167
constexpr int      AL_BIG       = 0x7fffffff;

constexpr int AL_OK           = 0;
constexpr int AL_ERR_BAND     = 1;    // band wider than the ring
constexpr int AL_ERR_PEBBLES  = 2;    // pebble arena exhausted
constexpr int AL_ERR_TRACE    = 3;    // trace buffer exhausted
constexpr int AL_ERR_LOOP     = 4;    // more waves than the sequences can hold (guards against hangs)

// band ring field indices (ints); two buffers of the path-carried fields, one of NA/NB
enum { F_V = 0, F_M, F_HA, F_HB, F_HAM, F_HBM, F_TL, F_TH, F_COUNT };

struct AlignSpecDev
{ int            TS, PATH_AVE, REACH;
  int            mscore, dscore;     // align.c:253-254
  const int16_t *SCORE, *TABLE;      // device copies of the full tables (not used by the hot loop)
};

// Sequences are read through 2-bit packed copies of the blocks (32 bases per 64-bit word, base q of
// the block in bits 2*(q&31).. of word q>>5; separators pack as 0 and are never compared because
// every run is bounded by the sequence lengths).  pa points at the packed READS block -- its
// reverse-complemented twin for a complemented A -- and abase is the block offset of A[0].
struct SeqView
{ const uint64_t *pa;
  const uint64_t *pb;
  int64_t        abase, bbase;
  int            alen, blen;
  int            comp;   // A is used reverse-complemented (ACOMP_FLAG semantics)
};

struct PathRes
{ int abpos, bbpos, aepos, bepos, diffs;
  int atlen, btlen;      // number of uint16 trace values
  int aorg, borg;        // offset of the first value relative to the trace buffers' origin
};

// Per-wavefront working storage
template <int W> struct WaveWork
{ int      *band;        // ring: (F_COUNT*2 + 2) * W ints  (LDS or global)
  int4     *cells;       // pebble arena {ptr, diag, diff, mark}
  int       cellcap;
  int4     *cells_hi  = nullptr;       // lane-group kernels: the pebbles from number `cellsplit` on live here (global
  int       cellsplit = 0x7fffffff;    //   memory) -- the first ones in the group's LDS; cellcap counts both parts
  uint16_t *atrace;      // origin (element 0 of the forward part); room for tcap values either side
  uint16_t *btrace;
  int       tcap;
  int      *mbox;        // 16 ints of LDS: lane 0 hands the trace-back results to the wavefront
  const int2 *trimtab;   // 32 x {sum, max proper-prefix sum} of 5-column chunks (LDS, per workgroup)
};

// Pebble number i of a pass (see cells_hi)
template <class WK> __device__ __forceinline__ void cell_put(const WK &wk, int i, const int4 &v)
{ if (i < wk.cellsplit) wk.cells[i] = v;
  else                  wk.cells_hi[i - wk.cellsplit] = v;
}

struct CellsView
{ int4 *lo, *hi;
  int   split;
  __device__ __forceinline__ int4 at(int i) const
  { if (i < split) return lo[i];
    return hi[i - split];
  }
  __device__ __forceinline__ void setx(int i, int x) const
  { if (i < split) lo[i].x = x;
    else           hi[i - split].x = x;
  }
};

template <class WK> __device__ __forceinline__ CellsView cells_of(const WK &wk)
{ return CellsView{wk.cells, wk.cells_hi, wk.cellsplit}; }

// 32 bases starting at block position q (q may be as low as -64: the packed arrays carry two pad
// words in front and two behind)
__device__ __forceinline__ uint64_t load32(const uint64_t *__restrict__ p, int64_t q)
{ int64_t  w  = q >> 5;
  int      sh = (int) (q & 31) << 1;
  // both words always (the arrays are padded): a load guarded by sh != 0 would be a branch with a wait
  // at its join, and the A and B windows of a snake step would be fetched one after the other
  uint64_t lo = p[w];
  uint64_t hi = p[w + 1];
  return (lo >> sh) | ((hi << 1) << (63 - sh));
}

// XOR of the two 32-base windows A[qa..] and B[qb..]: all four words are requested before any is used,
// so the two sequences cost one memory round trip per snake step, not two
__device__ __forceinline__ uint64_t diff32(const uint64_t *__restrict__ pa, int64_t qa,
                                           const uint64_t *__restrict__ pb, int64_t qb)
{ const int64_t wa = qa >> 5, wb = qb >> 5;
  const int     sa = (int) (qa & 31) << 1, sb = (int) (qb & 31) << 1;
  const uint64_t alo = pa[wa], ahi = pa[wa + 1];
  const uint64_t blo = pb[wb], bhi = pb[wb + 1];
  const uint64_t a = (alo >> sa) | ((ahi << 1) << (63 - sa));
  const uint64_t b = (blo >> sb) | ((bhi << 1) << (63 - sb));
  return a ^ b;
}

// Length of the exact-match run A[x..] == B[y..] (forward snake, align.c:748-768), bounded by the
// sequence ends; *end is set when the run stopped at a sequence end rather than at a mismatch.
__device__ __forceinline__ int match_fwd(const SeqView &S, int x, int y, bool &end)
{ int rem = min(S.alen - x, S.blen - y);
  int n = 0;
  end = true;
  while (n < rem)
    { uint64_t d = diff32(S.pa, S.abase + x + n, S.pb, S.bbase + y + n);
      int m = d ? (__builtin_ctzll(d) >> 1) : 32;
      if (m >= rem - n) { n = rem; break; }
      n += m;
      if (m < 32) { end = false; break; }
    }
  return n;
}

// Same going down: A[x-1-i] == B[y-1-i] (reverse snake, align.c:1403-1423)
__device__ __forceinline__ int match_rev(const SeqView &S, int x, int y, bool &end)
{ int rem = min(x, y);
  int n = 0;
  end = true;
  while (n < rem)
    { uint64_t d = diff32(S.pa, S.abase + x - n - 32, S.pb, S.bbase + y - n - 32);
      int m = d ? (__builtin_clzll(d) >> 1) : 32;
      if (m >= rem - n) { n = rem; break; }
      n += m;
      if (m < 32) { end = false; break; }
    }
  return n;
}

// n matches appended to the 60-column window b with match count m (align.c:763-767 n times):
// each step counts a match if the column leaving the window (bit 60) was a difference.
__device__ __forceinline__ void push_matches(uint64_t &b, int &m, int n)
{ if (n <= 0) return;
  int nn = n < 61 ? n : 61;
  uint64_t leaving = (b >> (61 - nn)) & ((nn >= 64) ? ~0ull : ((1ull << nn) - 1ull));
  m += nn - __popcll(leaving);
  b = (n >= 64) ? ~0ull : ((b << n) | ((1ull << n) - 1ull));
}

template <int W> struct Ring
{ int *p;
  __device__ __forceinline__ int &f(int buf, int field, int k) const
  { return p[(field * 2 + buf) * W + (k & (W - 1))]; }
  __device__ __forceinline__ int &na(int k) const { return p[(F_COUNT * 2) * W + (k & (W - 1))]; }
  __device__ __forceinline__ int &nb(int k) const { return p[(F_COUNT * 2 + 1) * W + (k & (W - 1))]; }
};

// Lanes of one wavefront hand data to each other through LDS / global memory: make every earlier
// store of the wavefront complete (s_waitcnt) before any later load is issued.
__device__ __forceinline__ void wave_sync()
{ __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Inside the d-wave loop lanes only exchange data through LDS, which executes a wavefront's
// accesses in order: a compiler-level ordering point is enough (no wait for the pebble stores).
__device__ __forceinline__ void wave_lds_sync()
{ __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// TABLE[x] / SCORE[x] of align.c:207-218 computed on the fly (15 columns, most significant first):
// cheaper than two dependent global loads for the one or two record-setting lanes of a wave.
__device__ __forceinline__ void trim_entry(int x, int mscore, int dscore, int &tab, int &sco)
{ int s = 0, mx = 0;
#pragma unroll
  for (int bit = AL_TRIM_LEN - 1; bit >= 0; bit--)
    { mx = (s > mx) ? s : mx;
      s += ((x >> bit) & 1) ? mscore : -dscore;
    }
  tab = (int) (int16_t) (s - mx);
  sco = (int) (int16_t) s;
}

__device__ __forceinline__ bool trim_ok(uint64_t b, int mscore, int dscore)
{ int t0, s0, t1, s1;
  trim_entry((int) (b & AL_TRIM_MASK), mscore, dscore, t0, s0);
  if (t0 < 0) return false;
  trim_entry((int) ((b >> AL_TRIM_LEN) & AL_TRIM_MASK), mscore, dscore, t1, s1);
  return (t1 + s0 >= 0);
}

__device__ __forceinline__ int wave_excl_scan_i(int v, int lane, int &total)
{ int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1)
    { int y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
  total = __shfl(x, 63);
  return x - v;
}

// State of one wave pass that survives into the trace back
struct WaveEnd
{ int trima, trimy, trimd, trimha, trimhb;
  int morea, morey, mored, moreha, morehb, morem;
  int ncell;
};

// ---------------------------------------------------------------------------------------
//  One d-wave pass.  DIR = +1 forward (align.c:353-893), -1 reverse (align.c:1015-1550).
//  Returns AL_OK or an error; on return *plow is the band's low diagonal (unused by callers).
// ---------------------------------------------------------------------------------------
template <int W, int DIR>
__device__ int wave_pass(const WaveWork<W> &wk, const AlignSpecDev &sp, const SeqView &S,
                         int k0, int mida, int aoff, int boff, WaveEnd &E)
{ const int  lane = threadIdx.x & 63;
  const int  TS   = sp.TS;
  Ring<W>    R{wk.band};
  const int  OUTV = (DIR > 0) ? -1 : AL_BIG;        // value of V outside the band
  // the reverse wave compares the base BEFORE position (aseq-1 / bseq-1, align.c:1017-1018)
  const int  soff = (DIR > 0) ? 0 : -1;

  int low = k0, hgh = k0, dif = 0, cur = 0;
  int more = 1;
  int aclip = (DIR > 0) ? AL_BIG : -AL_BIG;
  int bclip = (DIR > 0) ? -AL_BIG : AL_BIG;
  int besta, besty, trima, trimy, trimd, trimha, trimhb;
  int morea, morey, mored, moreha, morehb, morem, lasta;
  int ncell = 0;

  besta = trima = morea = lasta = mida;
  besty = trimy = morey = (mida - hgh) >> 1;
  trimd = mored = 0;
  trimha = moreha = 0;
  trimhb = morehb = 1;
  morem = -1;

  // ---- 0-wave on the seed diagonal (align.c:428-547 / 1090-1215); uniform on all lanes
  { const int k = k0;
    int y = (mida - k) >> 1;
    int na, nb, ha, hb, ham, hbm;
    if (wk.cellcap < 8) return AL_ERR_PEBBLES;
    if (DIR > 0)
      { na = (((y + k) + (TS - aoff)) / TS - 1) * TS + aoff;
        nb = ((y + (TS - boff)) / TS - 1) * TS + boff;
        if (lane == 0)
          { cell_put(wk, 0, make_int4(-1, k, 0, na));
            cell_put(wk, 1, make_int4(-1, k, 0, nb));
          }
        ham = na; hbm = nb;
        na += TS; nb += TS;
      }
    else
      { na = (((y + k) + (TS - aoff) - 1) / TS - 1) * TS + aoff;
        nb = ((y + (TS - boff) - 1) / TS - 1) * TS + boff;
        if (lane == 0)
          { cell_put(wk, 0, make_int4(-1, k, 0, y + k));
            cell_put(wk, 1, make_int4(-1, k, 0, y));
          }
        ham = y + k; hbm = y;
      }
    ha = 0; hb = 1;
    ncell = 2;
    { bool end;
      int  n = (DIR > 0) ? match_fwd(S, k + y, y, end) : match_rev(S, k + y, y, end);
      y += DIR * n;
      if (end)
        { // the next comparison meets a terminator: B's is looked at first (align.c:750-762)
          bool bend = (DIR > 0) ? (y == S.blen) : (y == 0);
          more = 0;
          if (bend) bclip = k; else aclip = k;
        }
    }
    int c = (y << 1) + k;
    if (DIR > 0)
      { while (y + k >= na)
          { if (ncell >= wk.cellcap) return AL_ERR_PEBBLES;
            if (lane == 0) cell_put(wk, ncell, make_int4(ha, k, 0, na));
            ha = ncell++; ham = na; na += TS;
          }
        while (y >= nb)
          { if (ncell >= wk.cellcap) return AL_ERR_PEBBLES;
            if (lane == 0) cell_put(wk, ncell, make_int4(hb, k, 0, nb));
            hb = ncell++; hbm = nb; nb += TS;
          }
        if (c > besta)
          { besta = trima = lasta = c; besty = trimy = y; trimha = ha; trimhb = hb; }
      }
    else
      { while (y + k <= na)
          { if (ncell >= wk.cellcap) return AL_ERR_PEBBLES;
            if (lane == 0) cell_put(wk, ncell, make_int4(ha, k, 0, na));
            ha = ncell++; ham = na; na -= TS;
          }
        while (y <= nb)
          { if (ncell >= wk.cellcap) return AL_ERR_PEBBLES;
            if (lane == 0) cell_put(wk, ncell, make_int4(hb, k, 0, nb));
            hb = ncell++; hbm = nb; nb -= TS;
          }
        if (c < besta)
          { besta = trima = lasta = c; besty = trimy = y; trimha = ha; trimhb = hb; }
      }
    if (lane == 0)
      { R.f(cur, F_V, k)   = c;
        R.f(cur, F_M, k)   = AL_PATH_LEN;
        R.f(cur, F_TL, k)  = (int) (uint32_t) (AL_PATH_INT & 0xffffffffu);
        R.f(cur, F_TH, k)  = (int) (uint32_t) (AL_PATH_INT >> 32);
        R.f(cur, F_HA, k)  = ha;
        R.f(cur, F_HB, k)  = hb;
        R.f(cur, F_HAM, k) = ham;
        R.f(cur, F_HBM, k) = hbm;
        R.na(k) = na;
        R.nb(k) = nb;
      }
    wave_lds_sync();
  }

  // boundary handling shared by the 0-wave and every later wave (align.c:549-577, 848-875)
  auto boundary = [&](bool setd)
  { if (more != 0) return;
    { int bi = besty + soff, ai = besta - besty + soff;
      if (bi >= 0 && bi < S.blen && ai >= 0 && ai < S.alen)
        more = 1;
    }
    if (DIR > 0)
      { if (hgh >= aclip)
          { hgh = aclip - 1;
            int m = R.f(cur, F_M, aclip);
            if (morem <= m)
              { morem = m; morea = R.f(cur, F_V, aclip); morey = (morea - aclip) / 2;
                if (setd) mored = dif;
                moreha = R.f(cur, F_HA, aclip); morehb = R.f(cur, F_HB, aclip);
              }
          }
        if (low <= bclip)
          { low = bclip + 1;
            int m = R.f(cur, F_M, bclip);
            if (morem <= m)
              { morem = m; morea = R.f(cur, F_V, bclip); morey = (morea - bclip) / 2;
                if (setd) mored = dif;
                moreha = R.f(cur, F_HA, bclip); morehb = R.f(cur, F_HB, bclip);
              }
          }
        aclip = AL_BIG;
        bclip = -AL_BIG;
      }
    else
      { if (low <= aclip)
          { low = aclip + 1;
            int m = R.f(cur, F_M, aclip);
            if (morem <= m)
              { morem = m; morea = R.f(cur, F_V, aclip); morey = (morea - aclip) / 2;
                if (setd) mored = dif;
                moreha = R.f(cur, F_HA, aclip); morehb = R.f(cur, F_HB, aclip);
              }
          }
        if (hgh >= bclip)
          { hgh = bclip - 1;
            int m = R.f(cur, F_M, bclip);
            if (morem <= m)
              { morem = m; morea = R.f(cur, F_V, bclip); morey = (morea - bclip) / 2;
                if (setd) mored = dif;
                moreha = R.f(cur, F_HA, bclip); morehb = R.f(cur, F_HB, bclip);
              }
          }
        aclip = -AL_BIG;
        bclip = AL_BIG;
      }
  };

  boundary(false);

  // ---- successive waves
  while (more && ((DIR > 0) ? (lasta >= besta - AL_TRIM_MLAG) : (lasta <= besta + AL_TRIM_MLAG)))
    { low -= 1;
      hgh += 1;
      if (hgh - low + 3 > W) return AL_ERR_BAND;
      if (dif > S.alen + S.blen + 64) return AL_ERR_LOOP;
      if (lane == 0)
        { R.na(low) = R.na(low + 1);
          R.nb(low) = R.nb(low + 1);
          R.na(hgh) = R.na(hgh - 1);
          R.nb(hgh) = R.nb(hgh - 1);
          R.f(cur, F_V, low) = OUTV;
          R.f(cur, F_V, hgh) = OUTV;
          R.f(cur, F_V, low - 1) = OUTV;
          R.f(cur, F_V, hgh + 1) = OUTV;
        }
      dif += 1;
      wave_lds_sync();
      const int nxt = cur ^ 1;

      // ---- phase 1: every diagonal from the frozen previous wave
      int hitA = (DIR > 0) ? AL_BIG : -AL_BIG;     // aclip candidate of this wave
      int hitB = (DIR > 0) ? -AL_BIG : AL_BIG;
      for (int base = low; base <= hgh; base += 64)
        { const int  k      = base + lane;
          const bool active = (k <= hgh);
          int c = 0, m = 0, ha = 0, hb = 0, ham = 0, hbm = 0, y = 0;
          uint64_t b = 0;
          int hita = 0, hitb = 0;
          int na = 0, nb = 0, cntA = 0, cntB = 0;
          if (active)
            { int am = R.f(cur, F_V, k - 1), ac = R.f(cur, F_V, k), ap = R.f(cur, F_V, k + 1);
              int s;
              if (DIR > 0)
                { if (ac < am) s = (am < ap) ? k + 1 : k - 1;
                  else         s = (ac < ap) ? k + 1 : k;
                  c = (s == k) ? ac + 2 : ((s > k) ? ap + 1 : am + 1);
                }
              else
                { if (ac > ap) s = (ap > am) ? k - 1 : k + 1;
                  else         s = (ac > am) ? k - 1 : k;
                  c = (s == k) ? ac - 2 : ((s > k) ? ap - 1 : am - 1);
                }
              m   = R.f(cur, F_M, s);
              b   = ((uint64_t) (uint32_t) R.f(cur, F_TH, s) << 32) | (uint32_t) R.f(cur, F_TL, s);
              ha  = R.f(cur, F_HA, s);
              hb  = R.f(cur, F_HB, s);
              ham = R.f(cur, F_HAM, s);
              hbm = R.f(cur, F_HBM, s);

              if ((b & AL_PATH_TOP) != 0) m -= 1;
              b <<= 1;

              y = (c - k) >> 1;
              { bool end;
                int  n = (DIR > 0) ? match_fwd(S, k + y, y, end) : match_rev(S, k + y, y, end);
                y += DIR * n;
                push_matches(b, m, n);
                if (end)
                  { bool bend = (DIR > 0) ? (y == S.blen) : (y == 0);
                    if (bend) hitb = 1; else hita = 1;
                  }
              }
              c = (y << 1) + k;

              // trace lines crossed on this diagonal (align.c:771-817): count the pebbles to drop
              na = R.na(k);
              nb = R.nb(k);
              if (DIR > 0)
                { int t = na; int h = ham;
                  while (y + k >= t) { if (h < t) { cntA += 1; h = t; } t += TS; }
                  t = nb; h = hbm;
                  while (y >= t) { if (h < t) { cntB += 1; h = t; } t += TS; }
                }
              else
                { int t = na; int h = ham;
                  while (y + k <= t) { if (h > t) { cntA += 1; h = t; } t -= TS; }
                  t = nb; h = hbm;
                  while (y <= t) { if (h > t) { cntB += 1; h = t; } t -= TS; }
                }
            }
          int tot = 0, off = 0;
          { const int      t1   = cntA + cntB;
            const uint64_t any  = __ballot(t1 != 0);
            if (any != 0)
              { if (__ballot(t1 > 1) == 0)           // common case: at most one pebble per lane
                  { off = __popcll(any & ((1ull << lane) - 1ull));
                    tot = __popcll(any);
                  }
                else
                  off = wave_excl_scan_i(t1, lane, tot);
              }
          }
          if (ncell + tot > wk.cellcap) return AL_ERR_PEBBLES;
          if (active)
            { int idx = ncell + off;
              if (DIR > 0)
                { while (y + k >= na)
                    { if (ham < na) { cell_put(wk, idx, make_int4(ha, k, dif, na)); ha = idx++; ham = na; }
                      na += TS;
                    }
                  while (y >= nb)
                    { if (hbm < nb) { cell_put(wk, idx, make_int4(hb, k, dif, nb)); hb = idx++; hbm = nb; }
                      nb += TS;
                    }
                }
              else
                { while (y + k <= na)
                    { if (ham > na) { cell_put(wk, idx, make_int4(ha, k, dif, na)); ha = idx++; ham = na; }
                      na -= TS;
                    }
                  while (y <= nb)
                    { if (hbm > nb) { cell_put(wk, idx, make_int4(hb, k, dif, nb)); hb = idx++; hbm = nb; }
                      nb -= TS;
                    }
                }
              R.na(k) = na;
              R.nb(k) = nb;
              R.f(nxt, F_V, k)   = c;
              R.f(nxt, F_M, k)   = m;
              R.f(nxt, F_TL, k)  = (int) (uint32_t) (b & 0xffffffffu);
              R.f(nxt, F_TH, k)  = (int) (uint32_t) (b >> 32);
              R.f(nxt, F_HA, k)  = ha;
              R.f(nxt, F_HB, k)  = hb;
              R.f(nxt, F_HAM, k) = ham;
              R.f(nxt, F_HBM, k) = hbm;
            }
          ncell += tot;

          // boundary clips: forward aclip = smallest k that hit A's end, bclip = largest k that
          // hit B's end (align.c:750-762); mirrored for the reverse wave (align.c:1405-1417)
          uint64_t ma = __ballot(active && hita), mb = __ballot(active && hitb);
          if (ma)
            { int lo = base + (__ffsll((long long) ma) - 1), hi = base + (63 - __clzll((long long) ma));
              if (DIR > 0) { if (lo < hitA) hitA = lo; }
              else         { if (hi > hitA) hitA = hi; }
            }
          if (mb)
            { int lo = base + (__ffsll((long long) mb) - 1), hi = base + (63 - __clzll((long long) mb));
              if (DIR > 0) { if (hi > hitB) hitB = hi; }
              else         { if (lo < hitB) hitB = lo; }
            }
        }
      if (hitA != ((DIR > 0) ? AL_BIG : -AL_BIG)) { more = 0; aclip = hitA; }
      if (hitB != ((DIR > 0) ? -AL_BIG : AL_BIG)) { more = 0; bclip = hitB; }
      wave_lds_sync();
      cur = nxt;

      // ---- phase 2: record setters of the running best, in the reference's loop order
      //      (descending k forward, align.c:819-833; ascending k reverse, align.c:1471-1485).
      //      Only lanes that beat the previous best can be record setters; they are visited one by
      //      one (usually one or two) with scalar lane reads.
      { const int nch = (hgh - low) / 64 + 1;
        for (int ci = 0; ci < nch; ci++)
          { const int  base   = (DIR > 0) ? low + 64 * (nch - 1 - ci) : low + 64 * ci;
            const int  k      = base + lane;
            const bool active = (k <= hgh);
            int c = active ? R.f(cur, F_V, k) : ((DIR > 0) ? -AL_BIG : AL_BIG);
            uint64_t better = __ballot(active && ((DIR > 0) ? (c > besta) : (c < besta)));
            if (better == 0) continue;
            int m = 0, tl = 0, th = 0, hav = 0, hbv = 0;
            if ((better >> lane) & 1)
              { m   = R.f(cur, F_M, k);
                tl  = R.f(cur, F_TL, k);
                th  = R.f(cur, F_TH, k);
                hav = R.f(cur, F_HA, k);
                hbv = R.f(cur, F_HB, k);
              }
            uint64_t rem = better;
            while (rem)
              { int l = (DIR > 0) ? (63 - __clzll((long long) rem)) : (__ffsll((long long) rem) - 1);
                rem &= ~(1ull << l);
                int cl = __builtin_amdgcn_readlane(c, l);
                if ((DIR > 0) ? (cl > besta) : (cl < besta))
                  { besta = cl;
                    besty = (cl - (base + l)) >> 1;
                    if (__builtin_amdgcn_readlane(m, l) >= sp.PATH_AVE)
                      { lasta = cl;
                        uint64_t bl = ((uint64_t) (uint32_t) __builtin_amdgcn_readlane(th, l) << 32)
                                      | (uint32_t) __builtin_amdgcn_readlane(tl, l);
                        if (trim_ok(bl, sp.mscore, sp.dscore))
                          { trima  = cl;
                            trimy  = besty;
                            trimd  = dif;
                            trimha = __builtin_amdgcn_readlane(hav, l);
                            trimhb = __builtin_amdgcn_readlane(hbv, l);
                          }
                      }
                  }
              }
          }
      }

      boundary(true);

      // ---- keep the points within WAVE_LAG of the best (align.c:877-885): trim both ends of the
      //      band up to the outermost diagonals that are still within the lag
      if (hgh >= low && hgh - low < 64)
        { const int  k   = low + lane;
          const bool act = (k <= hgh);
          const int  n   = (DIR > 0) ? besta - AL_WAVE_LAG : besta + AL_WAVE_LAG;
          int v = act ? R.f(cur, F_V, k) : 0;
          uint64_t keep = __ballot(act && ((DIR > 0) ? (v >= n) : (v <= n)));
          if (keep)
            { int l0 = __ffsll((long long) keep) - 1, h0 = 63 - __clzll((long long) keep);
              hgh = low + h0;
              low = low + l0;
            }
          else
            hgh = low - 1;
        }
      else if (DIR > 0)
        { int n = besta - AL_WAVE_LAG;
          while (hgh >= low)
            if (R.f(cur, F_V, hgh) < n)
              hgh -= 1;
            else
              { while (R.f(cur, F_V, low) < n) low += 1;
                break;
              }
        }
      else
        { int n = besta + AL_WAVE_LAG;
          while (hgh >= low)
            if (R.f(cur, F_V, hgh) > n)
              hgh -= 1;
            else
              { while (R.f(cur, F_V, low) > n) low += 1;
                break;
              }
        }
    }

  E.trima = trima; E.trimy = trimy; E.trimd = trimd; E.trimha = trimha; E.trimhb = trimhb;
  E.morea = morea; E.morey = morey; E.mored = mored; E.moreha = moreha; E.morehb = morehb;
  E.morem = morem;
  E.ncell = ncell;
  return AL_OK;
}


// ---------------------------------------------------------------------------------------
//  Register-resident d-wave pass for bands of at most REG_BAND diagonals (the common case:
//  average width 6-12).  Diagonal k lives in lane k & 63 for the whole pass (a ring, so nothing
//  moves when the band slides); a lane reads its neighbours' previous-wave state with DPP
//  wavefront rotations, no LDS.  Same arithmetic and the same ordered reductions as wave_pass;
//  returns AL_ERR_BAND when the band outgrows the ring and the caller re-runs the pass with the
//  LDS ring.
// ---------------------------------------------------------------------------------------
#ifdef DMH_ALIGN_PROFILE
// [0] cycles in register passes, [1] in fallback passes, [2] in trace backs, [3] #register passes,
// [4] #fallback passes, [5] sum of wave counts (trimd+mored proxy: ncell), [6] cycles per job, [7] #jobs, [8..11] passes with max band <=14, <=30, <=46, <=62, [12] sum of band widths, [13] wave steps
__device__ unsigned long long g_aprof[16];
#define APROF_T0() long long _t0 = clock64()
#define APROF_ADD(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_aprof[i], (unsigned long long) (v)); } while (0)
#define APROF_LAP(i) do { long long _t1 = clock64(); APROF_ADD(i, _t1 - _t0); _t0 = _t1; } while (0)
#else
#define APROF_T0() do {} while (0)
#define APROF_ADD(i, v) do {} while (0)
#define APROF_LAP(i) do {} while (0)
#endif

constexpr int REG_BAND = 62;

__device__ __forceinline__ int dpp_from_left(int v)      // lane i <- lane i-1 (wraps)
{ return __builtin_amdgcn_update_dpp(0, v, 0x13C, 0xf, 0xf, false); }
__device__ __forceinline__ int dpp_from_right(int v)     // lane i <- lane i+1 (wraps)
{ return __builtin_amdgcn_update_dpp(0, v, 0x134, 0xf, 0xf, false); }

__device__ __forceinline__ uint64_t rot_right64(uint64_t m, int r)
{ r &= 63;
  return r ? ((m >> r) | (m << (64 - r))) : m;
}

// TABLE/SCORE entries from three 5-column chunk lookups {sum, max proper-prefix sum} in LDS
__device__ __forceinline__ void trim_entry_tab(const int2 *tab, int x, int &t, int &sc)
{ int2 c0 = tab[(x >> 10) & 31], c1 = tab[(x >> 5) & 31], c2 = tab[x & 31];
  int sum = c0.x + c1.x + c2.x;
  int mx = c0.y;
  int v = c0.x + c1.y;
  mx = v > mx ? v : mx;
  v = c0.x + c1.x + c2.y;
  mx = v > mx ? v : mx;
  t  = (int) (int16_t) (sum - mx);
  sc = (int) (int16_t) sum;
}

__device__ __forceinline__ bool trim_ok_tab(const int2 *tab, uint64_t b)
{ int t0, s0, t1, s1;
  trim_entry_tab(tab, (int) (b & AL_TRIM_MASK), t0, s0);
  if (t0 < 0) return false;
  trim_entry_tab(tab, (int) ((b >> AL_TRIM_LEN) & AL_TRIM_MASK), t1, s1);
  return (t1 + s0 >= 0);
}

// fills the 32-entry chunk table (call with one wavefront, then synchronise)
__device__ __forceinline__ void build_trim_tab(int2 *tab, int mscore, int dscore)
{ int lane = threadIdx.x & 63;
  if (lane < 32)
    { int s = 0, mx = 0;
      for (int bit = 4; bit >= 0; bit--)
        { mx = (s > mx) ? s : mx;
          s += ((lane >> bit) & 1) ? mscore : -dscore;
        }
      tab[lane] = make_int2(s, mx);
    }
}


// ---------------------------------------------------------------------------------------
//  Lane groups.  An alignment occupies G lanes of a wavefront: G = 64 (one per wavefront) or G = 32
//  (two independent alignments per wavefront; bands of 10 kb reads at 15 % error peak below 30
//  diagonals, so half the lanes of a 64-wide band would idle).  Everything the pass needs from its
//  neighbours goes through these five primitives.
// ---------------------------------------------------------------------------------------
template <int G> struct Grp;

template <> struct Grp<64>
{ static constexpr int BAND = 62;
  static __device__ __forceinline__ int  gl()   { return threadIdx.x & 63; }
  static __device__ __forceinline__ int  from_left(int v)  { return __builtin_amdgcn_update_dpp(0, v, 0x13C, 0xf, 0xf, false); }
  static __device__ __forceinline__ int  from_right(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x134, 0xf, 0xf, false); }
  static __device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
  static __device__ __forceinline__ uint64_t rot_right(uint64_t m, int r)
  { r &= 63;
    return r ? ((m >> r) | (m << (64 - r))) : m;
  }
  static __device__ __forceinline__ int  read(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
  static __device__ __forceinline__ int  excl_scan(int v, int &total)
  { const int lane = threadIdx.x & 63;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
      { int y = __shfl_up(x, o);
        if (lane >= o) x += y;
      }
    total = __shfl(x, 63);
    return x - v;
  }
};

template <> struct Grp<32>
{ static constexpr int BAND = 30;
  static __device__ __forceinline__ int  gl()   { return threadIdx.x & 31; }
  static __device__ __forceinline__ int  base() { return threadIdx.x & 32; }
  // cross-lane reads inside a 32-lane half go through the LDS crossbar (ds_bpermute): one LDS-pipe
  // instruction each, off the VALU that bounds this kernel
  static __device__ __forceinline__ int  from_left(int v)
  { return __builtin_amdgcn_ds_bpermute((base() | ((threadIdx.x - 1) & 31)) << 2, v); }
  static __device__ __forceinline__ int  from_right(int v)
  { return __builtin_amdgcn_ds_bpermute((base() | ((threadIdx.x + 1) & 31)) << 2, v); }
  static __device__ __forceinline__ uint64_t ballot(bool p)
  { // the half of the wavefront's mask that belongs to this group: one select, not a 64-bit shift
    const uint64_t b = __ballot(p);
    return (uint64_t) ((threadIdx.x & 32) ? (uint32_t) (b >> 32) : (uint32_t) b);
  }
  static __device__ __forceinline__ uint64_t rot_right(uint64_t m, int r)
  { uint32_t x = (uint32_t) m;
    r &= 31;
    return r ? ((x >> r) | (x << (32 - r))) : x;
  }
  static __device__ __forceinline__ int  read(int x, int l)
  { return __builtin_amdgcn_ds_bpermute((base() | (l & 31)) << 2, x); }
  static __device__ __forceinline__ int  excl_scan(int v, int &total)
  { const int gl_ = threadIdx.x & 31;
    int x = v;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1)
      { int y = __shfl_up(x, o, 32);
        if (gl_ >= o) x += y;
      }
    total = __shfl(x, 31, 32);
    return x - v;
  }
};

template <int DIR, bool WANT_B, typename WK, int G = 64>
__device__ int wave_pass_reg(const WK &wk, const AlignSpecDev &sp, const SeqView &S,
                             int k0, int mida, int aoff, int boff, WaveEnd &E)
{ const int  lane = Grp<G>::gl();        // lane within the alignment's group
  const int  TS   = sp.TS;
  const int  OUTV = (DIR > 0) ? -1 : AL_BIG;
  const int  soff = (DIR > 0) ? 0 : -1;

  int low = k0, hgh = k0, dif = 0;
  int more = 1;
  int aclip = (DIR > 0) ? AL_BIG : -AL_BIG;
  int bclip = (DIR > 0) ? -AL_BIG : AL_BIG;
  int besta, besty, trima, trimy, trimd, trimha, trimhb;
  int morea, morey, mored, moreha, morehb, morem, lasta;
  int ncell = 0;
#ifdef DMH_ALIGN_PROFILE
  int prof_maxw = 1, prof_sumw = 0, prof_wide = 0, prof_wmode = 0, prof_wsteps = 0;
  long long prof_tprev = 0;
#endif

  // per-lane state of "my" diagonal
  int      V = OUTV, M = 0, HA = 0, HB = 0, HAM = 0, HBM = 0, NA = 0, NB = 0;
  uint32_t TL = 0, TH = 0;

  besta = trima = morea = lasta = mida;
  besty = trimy = morey = (mida - hgh) >> 1;
  trimd = mored = 0;
  trimha = moreha = 0;
  trimhb = morehb = 1;
  morem = -1;

  // ---- 0-wave on the seed diagonal (uniform on all lanes)
  { const int k = k0;
    int y = (mida - k) >> 1;
    int na, nb, ha, hb, ham, hbm;
    if (wk.cellcap < 8) return AL_ERR_PEBBLES;
    if (DIR > 0)
      { na = (((y + k) + (TS - aoff)) / TS - 1) * TS + aoff;
        nb = ((y + (TS - boff)) / TS - 1) * TS + boff;
        if (lane == 0)
          { cell_put(wk, 0, make_int4(-1, k, 0, na));
            if (WANT_B) cell_put(wk, 1, make_int4(-1, k, 0, nb));
          }
        ham = na; hbm = nb;
        na += TS; nb += TS;
      }
    else
      { na = (((y + k) + (TS - aoff) - 1) / TS - 1) * TS + aoff;
        nb = ((y + (TS - boff) - 1) / TS - 1) * TS + boff;
        if (lane == 0)
          { cell_put(wk, 0, make_int4(-1, k, 0, y + k));
            if (WANT_B) cell_put(wk, 1, make_int4(-1, k, 0, y));
          }
        ham = y + k; hbm = y;
      }
    ha = 0; hb = 1;
    ncell = 2;
    { bool end;
      int  n = (DIR > 0) ? match_fwd(S, k + y, y, end) : match_rev(S, k + y, y, end);
      y += DIR * n;
      if (end)
        { bool bend = (DIR > 0) ? (y == S.blen) : (y == 0);
          more = 0;
          if (bend) bclip = k; else aclip = k;
        }
    }
    int c = (y << 1) + k;
    if (DIR > 0)
      { while (y + k >= na)
          { if (ncell >= wk.cellcap) return AL_ERR_PEBBLES;
            if (lane == 0) cell_put(wk, ncell, make_int4(ha, k, 0, na));
            ha = ncell++; ham = na; na += TS;
          }
        while (WANT_B && y >= nb)
          { if (ncell >= wk.cellcap) return AL_ERR_PEBBLES;
            if (lane == 0) cell_put(wk, ncell, make_int4(hb, k, 0, nb));
            hb = ncell++; hbm = nb; nb += TS;
          }
        if (c > besta)
          { besta = trima = lasta = c; besty = trimy = y; trimha = ha; trimhb = hb; }
      }
    else
      { while (y + k <= na)
          { if (ncell >= wk.cellcap) return AL_ERR_PEBBLES;
            if (lane == 0) cell_put(wk, ncell, make_int4(ha, k, 0, na));
            ha = ncell++; ham = na; na -= TS;
          }
        while (WANT_B && y <= nb)
          { if (ncell >= wk.cellcap) return AL_ERR_PEBBLES;
            if (lane == 0) cell_put(wk, ncell, make_int4(hb, k, 0, nb));
            hb = ncell++; hbm = nb; nb -= TS;
          }
        if (c < besta)
          { besta = trima = lasta = c; besty = trimy = y; trimha = ha; trimhb = hb; }
      }
    if (lane == (k & (G - 1)))
      { V = c; M = AL_PATH_LEN;
        TL = (uint32_t) (AL_PATH_INT & 0xffffffffu);
        TH = (uint32_t) (AL_PATH_INT >> 32);
        HA = ha; HB = hb; HAM = ham; HBM = hbm; NA = na; NB = nb;
      }
  }

  auto boundary = [&](bool setd)
  { if (more != 0) return;
    { int bi = besty + soff, ai = besta - besty + soff;
      if (bi >= 0 && bi < S.blen && ai >= 0 && ai < S.alen)
        more = 1;
    }
    const int  ka = (DIR > 0) ? aclip : bclip;     // clip that lowers hgh forward / raises low reverse...
    (void) ka;
    if (DIR > 0)
      { if (hgh >= aclip)
          { hgh = aclip - 1;
            int l = aclip & (G - 1);
            int m = Grp<G>::read(M, l);
            if (morem <= m)
              { morem = m; morea = Grp<G>::read(V, l); morey = (morea - aclip) / 2;
                if (setd) mored = dif;
                moreha = Grp<G>::read(HA, l); morehb = Grp<G>::read(HB, l);
              }
          }
        if (low <= bclip)
          { low = bclip + 1;
            int l = bclip & (G - 1);
            int m = Grp<G>::read(M, l);
            if (morem <= m)
              { morem = m; morea = Grp<G>::read(V, l); morey = (morea - bclip) / 2;
                if (setd) mored = dif;
                moreha = Grp<G>::read(HA, l); morehb = Grp<G>::read(HB, l);
              }
          }
        aclip = AL_BIG;
        bclip = -AL_BIG;
      }
    else
      { if (low <= aclip)
          { low = aclip + 1;
            int l = aclip & (G - 1);
            int m = Grp<G>::read(M, l);
            if (morem <= m)
              { morem = m; morea = Grp<G>::read(V, l); morey = (morea - aclip) / 2;
                if (setd) mored = dif;
                moreha = Grp<G>::read(HA, l); morehb = Grp<G>::read(HB, l);
              }
          }
        if (hgh >= bclip)
          { hgh = bclip - 1;
            int l = bclip & (G - 1);
            int m = Grp<G>::read(M, l);
            if (morem <= m)
              { morem = m; morea = Grp<G>::read(V, l); morey = (morea - bclip) / 2;
                if (setd) mored = dif;
                moreha = Grp<G>::read(HA, l); morehb = Grp<G>::read(HB, l);
              }
          }
        aclip = -AL_BIG;
        bclip = AL_BIG;
      }
  };

  boundary(false);

  while (more && ((DIR > 0) ? (lasta >= besta - AL_TRIM_MLAG) : (lasta <= besta + AL_TRIM_MLAG)))
    { // an empty band cannot be represented here (its stale edge state matters to the reference)
      if (hgh < low) return AL_ERR_BAND;
      low -= 1;
      hgh += 1;
      if (hgh - low + 1 > Grp<G>::BAND) return AL_ERR_BAND;
      if (dif > S.alen + S.blen + 64) return AL_ERR_LOOP;
      dif += 1;
#ifdef DMH_ALIGN_PROFILE
      if (G == 32 && (threadIdx.x & 63) == __ffsll((long long) __ballot(true)) - 1)
        { const long long tnow = clock64();
          atomicAdd(&g_aprof[10], 1ull);
          if (prof_tprev != 0) atomicAdd(&g_aprof[15], (unsigned long long) (tnow - prof_tprev));
        }
      prof_tprev = clock64();
      if (hgh - low + 1 > prof_maxw) prof_maxw = hgh - low + 1;
      prof_sumw += hgh - low + 1;
      if (hgh - low + 1 > 14) prof_wide += 1;
      if (hgh - low + 1 > 14) prof_wmode = 1; else if (hgh - low + 1 <= 10) prof_wmode = 0;
      prof_wsteps += prof_wmode;
#endif

      const int  k      = low + ((lane - low) & (G - 1));     // the diagonal of this lane in the new band
      const bool active = (k <= hgh);
      const bool inold  = active && (k > low) && (k < hgh);
      const int  vold   = inold ? V : OUTV;

      // neighbours' previous-wave state (all lanes take part in the rotations)
      const int am = Grp<G>::from_left(vold), ap = Grp<G>::from_right(vold), ac = vold;
      const int nal = Grp<G>::from_left(NA),  nar = Grp<G>::from_right(NA);
      const int nbl = Grp<G>::from_left(NB),  nbr = Grp<G>::from_right(NB);
      const int Ml = Grp<G>::from_left(M),    Mr = Grp<G>::from_right(M);
      const int TLl = Grp<G>::from_left((int) TL), TLr = Grp<G>::from_right((int) TL);
      const int THl = Grp<G>::from_left((int) TH), THr = Grp<G>::from_right((int) TH);
      const int HAl = Grp<G>::from_left(HA),  HAr = Grp<G>::from_right(HA);
      const int HBl = Grp<G>::from_left(HB),  HBr = Grp<G>::from_right(HB);
      const int HAMl = Grp<G>::from_left(HAM), HAMr = Grp<G>::from_right(HAM);
      const int HBMl = Grp<G>::from_left(HBM), HBMr = Grp<G>::from_right(HBM);

      int na = NA, nb = NB;
      if (active && k == low && k != hgh) { na = nar; nb = nbr; }    // align.c:678-681
      else if (active && k == hgh && k != low) { na = nal; nb = nbl; }

      int c = 0, m = 0, ha = 0, hb = 0, ham = 0, hbm = 0, y = 0;
      uint64_t b = 0;
      int hita = 0, hitb = 0, cntA = 0, cntB = 0;
      if (active)
        { int s;      // -1: from k-1, 0: stay, +1: from k+1
          if (DIR > 0)
            { if (ac < am) s = (am < ap) ? 1 : -1;
              else         s = (ac < ap) ? 1 : 0;
              c = (s == 0) ? ac + 2 : ((s > 0) ? ap + 1 : am + 1);
            }
          else
            { if (ac > ap) s = (ap > am) ? -1 : 1;
              else         s = (ac > am) ? -1 : 0;
              c = (s == 0) ? ac - 2 : ((s > 0) ? ap - 1 : am - 1);
            }
          uint32_t tl, th;
          if (s == 0)     { m = M;  tl = TL;             th = TH;             ha = HA;  hb = HB;  ham = HAM;  hbm = HBM; }
          else if (s < 0) { m = Ml; tl = (uint32_t) TLl; th = (uint32_t) THl; ha = HAl; hb = HBl; ham = HAMl; hbm = HBMl; }
          else            { m = Mr; tl = (uint32_t) TLr; th = (uint32_t) THr; ha = HAr; hb = HBr; ham = HAMr; hbm = HBMr; }
          b = ((uint64_t) th << 32) | tl;

          if ((b & AL_PATH_TOP) != 0) m -= 1;
          b <<= 1;

          y = (c - k) >> 1;
          { bool end;
            int  n = (DIR > 0) ? match_fwd(S, k + y, y, end) : match_rev(S, k + y, y, end);
            y += DIR * n;
            push_matches(b, m, n);
            if (end)
              { bool bend = (DIR > 0) ? (y == S.blen) : (y == 0);
                if (bend) hitb = 1; else hita = 1;
              }
          }
          c = (y << 1) + k;

          if (DIR > 0)
            { int t = na; int h = ham;
              while (y + k >= t) { if (h < t) { cntA += 1; h = t; } t += TS; }
              t = nb; h = hbm;
              while (WANT_B && y >= t) { if (h < t) { cntB += 1; h = t; } t += TS; }
            }
          else
            { int t = na; int h = ham;
              while (y + k <= t) { if (h > t) { cntA += 1; h = t; } t -= TS; }
              t = nb; h = hbm;
              while (WANT_B && y <= t) { if (h > t) { cntB += 1; h = t; } t -= TS; }
            }
        }

      int tot = 0, off = 0;
      { const int      t1  = cntA + cntB;
        const uint64_t any = Grp<G>::ballot(t1 != 0);
        if (any != 0)
          { if (Grp<G>::ballot(t1 > 1) == 0)
              { off = __popcll(any & ((1ull << lane) - 1ull));
                tot = __popcll(any);
              }
            else
              off = Grp<G>::excl_scan(t1, tot);
          }
      }
      if (ncell + tot > wk.cellcap) return AL_ERR_PEBBLES;
      if (active)
        { int idx = ncell + off;
          if (DIR > 0)
            { while (y + k >= na)
                { if (ham < na) { cell_put(wk, idx, make_int4(ha, k, dif, na)); ha = idx++; ham = na; }
                  na += TS;
                }
              while (WANT_B && y >= nb)
                { if (hbm < nb) { cell_put(wk, idx, make_int4(hb, k, dif, nb)); hb = idx++; hbm = nb; }
                  nb += TS;
                }
            }
          else
            { while (y + k <= na)
                { if (ham > na) { cell_put(wk, idx, make_int4(ha, k, dif, na)); ha = idx++; ham = na; }
                  na -= TS;
                }
              while (WANT_B && y <= nb)
                { if (hbm > nb) { cell_put(wk, idx, make_int4(hb, k, dif, nb)); hb = idx++; hbm = nb; }
                  nb -= TS;
                }
            }
          V = c; M = m; TL = (uint32_t) (b & 0xffffffffu); TH = (uint32_t) (b >> 32);
          HA = ha; HB = hb; HAM = ham; HBM = hbm; NA = na; NB = nb;
        }
      ncell += tot;

      // boundary clips (bit j of a rotated mask <-> diagonal low + j)
      const int rl = low & (G - 1);
      { uint64_t ma = Grp<G>::rot_right(Grp<G>::ballot(active && hita), rl), mb = Grp<G>::rot_right(Grp<G>::ballot(active && hitb), rl);
        if (ma)
          { more = 0;
            aclip = (DIR > 0) ? low + (__ffsll((long long) ma) - 1) : low + (63 - __clzll((long long) ma));
          }
        if (mb)
          { more = 0;
            bclip = (DIR > 0) ? low + (63 - __clzll((long long) mb)) : low + (__ffsll((long long) mb) - 1);
          }
      }

      // record setters of the running best, visited in the reference's loop order
      { const bool beats = active && ((DIR > 0) ? (c > besta) : (c < besta));
        uint64_t better = Grp<G>::ballot(beats);
        if (better)
          { bool lflag = false, tflag = false;
            if (beats)
              { lflag = (m >= sp.PATH_AVE);
                if (lflag) tflag = trim_ok_tab(wk.trimtab, b);
              }
            uint64_t rb = Grp<G>::rot_right(better, rl);
            uint64_t rL = Grp<G>::rot_right(Grp<G>::ballot(lflag), rl);
            uint64_t rT = Grp<G>::rot_right(Grp<G>::ballot(tflag), rl);
            while (rb)
              { int j = (DIR > 0) ? (63 - __clzll((long long) rb)) : (__ffsll((long long) rb) - 1);
                rb &= ~(1ull << j);
                int kk = low + j, l = kk & (G - 1);
                int cl = Grp<G>::read(c, l);
                if ((DIR > 0) ? (cl > besta) : (cl < besta))
                  { besta = cl;
                    besty = (cl - kk) >> 1;
                    if ((rL >> j) & 1)
                      { lasta = cl;
                        if ((rT >> j) & 1)
                          { trima  = cl;
                            trimy  = besty;
                            trimd  = dif;
                            trimha = Grp<G>::read(ha, l);
                            trimhb = Grp<G>::read(hb, l);
                          }
                      }
                  }
              }
          }
      }

      boundary(true);

      // keep the points within WAVE_LAG of the best (align.c:877-885)
      if (hgh >= low)
        { const int  kk  = low + ((lane - low) & (G - 1));
          const bool act = (kk <= hgh);
          const int  n   = (DIR > 0) ? besta - AL_WAVE_LAG : besta + AL_WAVE_LAG;
          uint64_t keep = Grp<G>::rot_right(Grp<G>::ballot(act && ((DIR > 0) ? (V >= n) : (V <= n))), low & (G - 1));
          if (keep)
            { int l0 = __ffsll((long long) keep) - 1, h0 = 63 - __clzll((long long) keep);
              hgh = low + h0;
              low = low + l0;
            }
          else
            hgh = low - 1;
        }
    }

  E.trima = trima; E.trimy = trimy; E.trimd = trimd; E.trimha = trimha; E.trimhb = trimhb;
  E.morea = morea; E.morey = morey; E.mored = mored; E.moreha = moreha; E.morehb = morehb;
  E.morem = morem;
  E.ncell = ncell;
#ifdef DMH_ALIGN_PROFILE
  APROF_ADD(prof_maxw <= 14 ? 8 : (prof_maxw <= 30 ? 9 : (prof_maxw <= 46 ? 10 : 11)), 1);
  APROF_ADD(12, prof_sumw);
  APROF_ADD(13, dif);
  APROF_ADD(14, prof_wide);
  (void) prof_wsteps;
#endif
  return AL_OK;
}

// ---------------------------------------------------------------------------------------
//  FOUR alignments per wavefront: 16 lanes each.  The band of a 10 kb / 15 % read is 7.8 diagonals wide on
//  average and wider than 14 in 3 % of the wave steps only -- but two passes in three get there at some point,
//  so a 16-lane ring alone would send most work to the wider kernels.  Here every lane has TWO slots.  While the
//  band is NARROW (<= 14 diagonals) the ring has 16 positions, diagonal k lives in slot 0 of lane k & 15 and
//  slot 1 is not even executed (unless another group of the wavefront is wide).  When the band grows past 14 the
//  ring becomes 32 positions -- position p = k & 31 is slot p >> 4 of lane p & 15 -- and a diagonal moves at most
//  from slot 0 to slot 1 of its own lane; back to the 16-position ring at <= 10 diagonals.  Neighbour reads are
//  ds_bpermute on both slots with a select at the seam lanes (0 and 15); masks are 16-bit slices of the wavefront
//  ballots, joined and rotated into diagonal order with one funnel shift.  Arithmetic and ordered reductions are
//  those of wave_pass_reg (A view only: the -C runs use the one-per-wavefront kernel).
//  Measured (configs[1], 26 k candidates): 26 % fewer vector and 21 % fewer scalar instructions than the 32-lane
//  kernel, but 161 registers = three wavefronts per SIMD instead of four and 200 pebbles of LDS per group instead of
//  288 (more candidates re-run on the one-per-wavefront kernel): 16.0 ms against 14.1.  Opt-in ("align_g" = 16).
// ---------------------------------------------------------------------------------------
template <> struct Grp<16>
{ static constexpr int BAND   = 30;
  static constexpr int NARROW = 14;     // widest band of the 16-position ring
  static constexpr int BACK   = 10;     // back to it at this width
  static __device__ __forceinline__ int  gl()   { return threadIdx.x & 15; }
  static __device__ __forceinline__ int  base() { return threadIdx.x & 48; }
  static __device__ __forceinline__ int  perm(int addr, int v) { return __builtin_amdgcn_ds_bpermute(addr, v); }
  static __device__ __forceinline__ int  addr_of(int l) { return (base() | (l & 15)) << 2; }
  static __device__ __forceinline__ uint32_t ballot(bool p)
  { return (uint32_t) (__ballot(p) >> base()) & 0xffffu; }
  static __device__ __forceinline__ int  excl_scan(int v, int &total)
  { const int gl_ = threadIdx.x & 15;
    int x = v;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1)
      { int y = __shfl_up(x, o, 16);
        if (gl_ >= o) x += y;
      }
    total = __shfl(x, 15, 16);
    return x - v;
  }
};

template <int S> struct SlotC { static constexpr int value = S; };

template <int DIR, typename WK>
__device__ int wave_pass_reg16(const WK &wk, const AlignSpecDev &sp, const SeqView &S,
                               int k0, int mida, int aoff, WaveEnd &E)
{ using GR = Grp<16>;
  const int  lane = GR::gl();
  const int  TS   = sp.TS;
  const int  OUTV = (DIR > 0) ? -1 : AL_BIG;
  const int  soff = (DIR > 0) ? 0 : -1;
  const int  adl = GR::addr_of(lane - 1), adr = GR::addr_of(lane + 1);     // neighbour lanes on the 16-lane ring

  int low = k0, hgh = k0, dif = 0;
  int more = 1;
  int aclip = (DIR > 0) ? AL_BIG : -AL_BIG;
  int bclip = (DIR > 0) ? -AL_BIG : AL_BIG;
  int besta, besty, trima, trimy, trimd, trimha;
  int morea, morey, mored, moreha, morem, lasta;
  int ncell = 0;
  bool wide = false;                     // the group's ring has 32 positions
#ifdef DMH_ALIGN_PROFILE
  long long prof_tprev = 0;
#endif

  // per-lane state of the lane's (up to) two diagonals
  int      V[2] = {OUTV, OUTV}, M[2] = {0, 0}, HA[2] = {0, 0}, HAM[2] = {0, 0}, NA[2] = {0, 0};
  uint32_t TL[2] = {0, 0}, TH[2] = {0, 0};

  besta = trima = morea = lasta = mida;
  besty = trimy = morey = (mida - hgh) >> 1;
  trimd = mored = 0;
  trimha = moreha = 0;
  morem = -1;

  // ---- 0-wave on the seed diagonal (uniform on the group's lanes)
  { const int k = k0;
    int y = (mida - k) >> 1;
    int na, ha, ham;
    if (wk.cellcap < 8) return AL_ERR_PEBBLES;
    if (DIR > 0)
      { na = (((y + k) + (TS - aoff)) / TS - 1) * TS + aoff;
        if (lane == 0) cell_put(wk, 0, make_int4(-1, k, 0, na));
        ham = na;
        na += TS;
      }
    else
      { na = (((y + k) + (TS - aoff) - 1) / TS - 1) * TS + aoff;
        if (lane == 0) cell_put(wk, 0, make_int4(-1, k, 0, y + k));
        ham = y + k;
      }
    ha = 0;
    ncell = 2;
    { bool end;
      int  n = (DIR > 0) ? match_fwd(S, k + y, y, end) : match_rev(S, k + y, y, end);
      y += DIR * n;
      if (end)
        { bool bend = (DIR > 0) ? (y == S.blen) : (y == 0);
          more = 0;
          if (bend) bclip = k; else aclip = k;
        }
    }
    int c = (y << 1) + k;
    if (DIR > 0)
      { while (y + k >= na)
          { if (ncell >= wk.cellcap) return AL_ERR_PEBBLES;
            if (lane == 0) cell_put(wk, ncell, make_int4(ha, k, 0, na));
            ha = ncell++; ham = na; na += TS;
          }
        if (c > besta)
          { besta = trima = lasta = c; besty = trimy = y; trimha = ha; }
      }
    else
      { while (y + k <= na)
          { if (ncell >= wk.cellcap) return AL_ERR_PEBBLES;
            if (lane == 0) cell_put(wk, ncell, make_int4(ha, k, 0, na));
            ha = ncell++; ham = na; na -= TS;
          }
        if (c < besta)
          { besta = trima = lasta = c; besty = trimy = y; trimha = ha; }
      }
    if (lane == (k & 15))
      { V[0] = c; M[0] = AL_PATH_LEN;
        TL[0] = (uint32_t) (AL_PATH_INT & 0xffffffffu);
        TH[0] = (uint32_t) (AL_PATH_INT >> 32);
        HA[0] = ha; HAM[0] = ham; NA[0] = na;
      }
  }

  // two 16-bit ballot slices (slot 0, slot 1) -> one mask in diagonal order: bit j <-> diagonal low + j
  auto dmask = [&](uint32_t b0, uint32_t b1) -> uint32_t
  { const uint32_t x2 = b0 | ((wide ? b1 : b0) << 16);
    const uint32_t m  = __builtin_amdgcn_alignbit(x2, x2, (uint32_t) (low & (wide ? 31 : 15)));
    return wide ? m : (m & 0xffffu);
  };
  // the value diagonal kk holds (x0 / x1 = the lanes' slot 0 / slot 1 registers)
  auto rd = [&](int x0, int x1, int kk, bool anyw) -> int
  { const int p = kk & (wide ? 31 : 15);
    const int a = GR::addr_of(p);
    int v = GR::perm(a, x0);
    if (anyw)
      { const int v1 = GR::perm(a, x1);
        v = (p & 16) ? v1 : v;
      }
    return v;
  };

  auto boundary = [&](bool setd)
  { if (more != 0) return;
    { int bi = besty + soff, ai = besta - besty + soff;
      if (bi >= 0 && bi < S.blen && ai >= 0 && ai < S.alen)
        more = 1;
    }
    const bool anyw = __ballot(wide) != 0;
    auto clip_at = [&](int kc)
    { int m = rd(M[0], M[1], kc, anyw);
      if (morem <= m)
        { morem = m; morea = rd(V[0], V[1], kc, anyw); morey = (morea - kc) / 2;
          if (setd) mored = dif;
          moreha = rd(HA[0], HA[1], kc, anyw);
        }
    };
    if (DIR > 0)
      { if (hgh >= aclip) { hgh = aclip - 1; clip_at(aclip); }
        if (low <= bclip) { low = bclip + 1; clip_at(bclip); }
        aclip = AL_BIG;
        bclip = -AL_BIG;
      }
    else
      { if (low <= aclip) { low = aclip + 1; clip_at(aclip); }
        if (hgh >= bclip) { hgh = bclip - 1; clip_at(bclip); }
        aclip = -AL_BIG;
        bclip = AL_BIG;
      }
  };

  boundary(false);

  while (more && ((DIR > 0) ? (lasta >= besta - AL_TRIM_MLAG) : (lasta <= besta + AL_TRIM_MLAG)))
    { if (hgh < low) return AL_ERR_BAND;
      low -= 1;
      hgh += 1;
      if (hgh - low + 1 > GR::BAND) return AL_ERR_BAND;
      if (dif > S.alen + S.blen + 64) return AL_ERR_LOOP;
      dif += 1;

      if (!wide && hgh - low + 1 > GR::NARROW)
        { // to the 32-position ring: the diagonals with bit 4 set move to slot 1 of their lanes
          const int  lo = low + 1;
          const int  kn = lo + ((lane - lo) & 15);
          const bool up = (kn & 16) != 0;
          if (up)
            { V[1] = V[0]; M[1] = M[0]; HA[1] = HA[0]; HAM[1] = HAM[0]; NA[1] = NA[0]; TL[1] = TL[0]; TH[1] = TH[0];
              V[0] = OUTV;
            }
          else
            V[1] = OUTV;
          wide = true;
        }
      const bool anyw = __ballot(wide) != 0;          // (some group of the wavefront runs its second slot)
#ifdef DMH_ALIGN_PROFILE
      if ((threadIdx.x & 63) == __ffsll((long long) __ballot(true)) - 1)
        { const long long tnow = clock64();
          atomicAdd(&g_aprof[13], 1ull);
          atomicAdd(&g_aprof[12], (unsigned long long) __popcll(__ballot(true)) / 16);
          if (anyw) atomicAdd(&g_aprof[14], 1ull);
          if (prof_tprev != 0) atomicAdd(&g_aprof[15], (unsigned long long) (tnow - prof_tprev));
        }
      prof_tprev = clock64();
#endif

      // the lane's diagonals in the new band
      int  kk[2];
      bool act[2], inold[2];
      int  vold[2];
      kk[0]  = wide ? low + ((lane - low) & 31) : low + ((lane - low) & 15);
      kk[1]  = low + ((lane + 16 - low) & 31);
      act[0] = (kk[0] <= hgh);
      act[1] = wide && (kk[1] <= hgh);
#pragma unroll
      for (int s = 0; s < 2; s++)
        { inold[s] = act[s] && (kk[s] > low) && (kk[s] < hgh);
          vold[s]  = inold[s] ? V[s] : OUTV;
        }

      // neighbours' previous-wave state: position p - 1 / p + 1 of the ring
      int am[2], ap[2], nal[2], nar[2], Ml[2], Mr[2], TLl[2], TLr[2], THl[2], THr[2], HAl[2], HAr[2], HAMl[2], HAMr[2];
      struct X4 { int l0, l1, r0, r1; };
      auto xch = [&](int x0, int x1) -> X4
      { X4 q;
        const int l0 = GR::perm(adl, x0), r0 = GR::perm(adr, x0);
        if (anyw)
          { const int l1 = GR::perm(adl, x1), r1 = GR::perm(adr, x1);
            q.l0 = (wide && lane == 0) ? l1 : l0;   q.l1 = (lane == 0) ? l0 : l1;
            q.r0 = (wide && lane == 15) ? r1 : r0;  q.r1 = (lane == 15) ? r0 : r1;
          }
        else
          { q.l0 = l0; q.r0 = r0; q.l1 = l0; q.r1 = r0; }
        return q;
      };
#define DMH_XCH(X0, X1, L, R) { const X4 q_ = xch((int) (X0), (int) (X1)); L[0] = q_.l0; L[1] = q_.l1; R[0] = q_.r0; R[1] = q_.r1; }
      DMH_XCH(vold[0], vold[1], am, ap)
      DMH_XCH(NA[0], NA[1], nal, nar)
      DMH_XCH(M[0], M[1], Ml, Mr)
      DMH_XCH(TL[0], TL[1], TLl, TLr)
      DMH_XCH(TH[0], TH[1], THl, THr)
      DMH_XCH(HA[0], HA[1], HAl, HAr)
      DMH_XCH(HAM[0], HAM[1], HAMl, HAMr)
#undef DMH_XCH

      int      c[2] = {0, 0}, m[2] = {0, 0}, ha[2] = {0, 0}, ham[2] = {0, 0}, na[2] = {NA[0], NA[1]}, y[2] = {0, 0};
      uint64_t b[2] = {0, 0};
      int      hita[2] = {0, 0}, hitb[2] = {0, 0}, cntA[2] = {0, 0};
      auto slot_step = [&](auto sc)
      { constexpr int s = decltype(sc)::value;
        const int k = kk[s];
        if (act[s] && k == low && k != hgh) na[s] = nar[s];            // align.c:678-681
        else if (act[s] && k == hgh && k != low) na[s] = nal[s];
        if (act[s])
          { const int ac = vold[s];
            int sel;      // -1: from k-1, 0: stay, +1: from k+1
            if (DIR > 0)
              { if (ac < am[s]) sel = (am[s] < ap[s]) ? 1 : -1;
                else            sel = (ac < ap[s]) ? 1 : 0;
                c[s] = (sel == 0) ? ac + 2 : ((sel > 0) ? ap[s] + 1 : am[s] + 1);
              }
            else
              { if (ac > ap[s]) sel = (ap[s] > am[s]) ? -1 : 1;
                else            sel = (ac > am[s]) ? -1 : 0;
                c[s] = (sel == 0) ? ac - 2 : ((sel > 0) ? ap[s] - 1 : am[s] - 1);
              }
            const bool stay = (sel == 0), fl = (sel < 0);
            const int  ml_ = Ml[s], mr_ = Mr[s], tll_ = TLl[s], tlr_ = TLr[s], thl_ = THl[s], thr_ = THr[s];
            const int  hal_ = HAl[s], har_ = HAr[s], haml_ = HAMl[s], hamr_ = HAMr[s];
            m[s]   = stay ? M[s] : (fl ? ml_ : mr_);
            ha[s]  = stay ? HA[s] : (fl ? hal_ : har_);
            ham[s] = stay ? HAM[s] : (fl ? haml_ : hamr_);
            const uint32_t tl = stay ? TL[s] : (uint32_t) (fl ? tll_ : tlr_);
            const uint32_t th = stay ? TH[s] : (uint32_t) (fl ? thl_ : thr_);
            uint64_t bb = ((uint64_t) th << 32) | tl;
            if ((bb & AL_PATH_TOP) != 0) m[s] -= 1;
            bb <<= 1;

            int yy = (c[s] - k) >> 1;
            { bool end;
              int  n = (DIR > 0) ? match_fwd(S, k + yy, yy, end) : match_rev(S, k + yy, yy, end);
              yy += DIR * n;
              push_matches(bb, m[s], n);
              if (end)
                { bool bend = (DIR > 0) ? (yy == S.blen) : (yy == 0);
                  if (bend) hitb[s] = 1; else hita[s] = 1;
                }
            }
            c[s] = (yy << 1) + k;
            y[s] = yy;
            b[s] = bb;
            if (DIR > 0)
              { int t = na[s]; int h = ham[s];
                while (yy + k >= t) { if (h < t) { cntA[s] += 1; h = t; } t += TS; }
              }
            else
              { int t = na[s]; int h = ham[s];
                while (yy + k <= t) { if (h > t) { cntA[s] += 1; h = t; } t -= TS; }
              }
          }
      };
      slot_step(SlotC<0>());
      if (anyw) slot_step(SlotC<1>());

      // pebbles: bump allocation in lane order, a lane's slot 0 before its slot 1
      int tot = 0, off = 0;
      { const int      t1  = cntA[0] + cntA[1];
        const uint32_t any = GR::ballot(t1 != 0);
        if (any != 0)
          { if (GR::ballot(t1 > 1) == 0)
              { off = __popc(any & ((1u << lane) - 1u));
                tot = __popc(any);
              }
            else
              off = GR::excl_scan(t1, tot);
          }
      }
      if (ncell + tot > wk.cellcap) return AL_ERR_PEBBLES;
      auto slot_commit = [&](auto sc)
      { constexpr int s = decltype(sc)::value;
        if (act[s])
          { const int k = kk[s];
            int idx = ncell + off + (s ? cntA[0] : 0);
            if (DIR > 0)
              { while (y[s] + k >= na[s])
                  { if (ham[s] < na[s]) { cell_put(wk, idx, make_int4(ha[s], k, dif, na[s])); ha[s] = idx++; ham[s] = na[s]; }
                    na[s] += TS;
                  }
              }
            else
              { while (y[s] + k <= na[s])
                  { if (ham[s] > na[s]) { cell_put(wk, idx, make_int4(ha[s], k, dif, na[s])); ha[s] = idx++; ham[s] = na[s]; }
                    na[s] -= TS;
                  }
              }
            V[s] = c[s]; M[s] = m[s]; TL[s] = (uint32_t) (b[s] & 0xffffffffu); TH[s] = (uint32_t) (b[s] >> 32);
            HA[s] = ha[s]; HAM[s] = ham[s]; NA[s] = na[s];
          }
      };
      slot_commit(SlotC<0>());
      if (anyw) slot_commit(SlotC<1>());
      ncell += tot;

      // boundary clips
      { const uint32_t ma = dmask(GR::ballot(act[0] && hita[0]), GR::ballot(act[1] && hita[1]));
        const uint32_t mb = dmask(GR::ballot(act[0] && hitb[0]), GR::ballot(act[1] && hitb[1]));
        if (ma)
          { more = 0;
            aclip = (DIR > 0) ? low + (__ffs((int) ma) - 1) : low + (31 - __clz((int) ma));
          }
        if (mb)
          { more = 0;
            bclip = (DIR > 0) ? low + (31 - __clz((int) mb)) : low + (__ffs((int) mb) - 1);
          }
      }

      // record setters of the running best, visited in the reference's loop order
      { bool beats[2];
#pragma unroll
        for (int s = 0; s < 2; s++) beats[s] = act[s] && ((DIR > 0) ? (c[s] > besta) : (c[s] < besta));
        uint32_t rb = dmask(GR::ballot(beats[0]), GR::ballot(beats[1]));
        if (rb)
          { bool lflag[2] = {false, false}, tflag[2] = {false, false};
#pragma unroll
            for (int s = 0; s < 2; s++)
              if (beats[s])
                { lflag[s] = (m[s] >= sp.PATH_AVE);
                  if (lflag[s]) tflag[s] = trim_ok_tab(wk.trimtab, b[s]);
                }
            const uint32_t rL = dmask(GR::ballot(lflag[0]), GR::ballot(lflag[1]));
            const uint32_t rT = dmask(GR::ballot(tflag[0]), GR::ballot(tflag[1]));
            while (rb)
              { const int j = (DIR > 0) ? (31 - __clz((int) rb)) : (__ffs((int) rb) - 1);
                rb &= ~(1u << j);
                const int kd = low + j;
                const int cl = rd(c[0], c[1], kd, anyw);
                if ((DIR > 0) ? (cl > besta) : (cl < besta))
                  { besta = cl;
                    besty = (cl - kd) >> 1;
                    if ((rL >> j) & 1u)
                      { lasta = cl;
                        if ((rT >> j) & 1u)
                          { trima  = cl;
                            trimy  = besty;
                            trimd  = dif;
                            trimha = rd(ha[0], ha[1], kd, anyw);
                          }
                      }
                  }
              }
          }
      }

      boundary(true);

      // keep the points within WAVE_LAG of the best (align.c:877-885)
      if (hgh >= low)
        { const int  k0n = wide ? low + ((lane - low) & 31) : low + ((lane - low) & 15);
          const int  k1n = low + ((lane + 16 - low) & 31);
          const int  n   = (DIR > 0) ? besta - AL_WAVE_LAG : besta + AL_WAVE_LAG;
          const bool p0  = (k0n <= hgh) && ((DIR > 0) ? (V[0] >= n) : (V[0] <= n));
          const bool p1  = wide && (k1n <= hgh) && ((DIR > 0) ? (V[1] >= n) : (V[1] <= n));
          const uint32_t keep = dmask(GR::ballot(p0), GR::ballot(p1));
          if (keep)
            { const int l0 = __ffs((int) keep) - 1, h0 = 31 - __clz((int) keep);
              hgh = low + h0;
              low = low + l0;
            }
          else
            hgh = low - 1;
        }

      if (wide && hgh >= low && hgh - low + 1 <= GR::BACK)
        { // back to the 16-position ring: a diagonal kept in slot 1 returns to slot 0 of its lane
          const int  kn = low + ((lane - low) & 15);
          if (kn <= hgh && (kn & 16) != 0)
            { V[0] = V[1]; M[0] = M[1]; HA[0] = HA[1]; HAM[0] = HAM[1]; NA[0] = NA[1]; TL[0] = TL[1]; TH[0] = TH[1]; }
          wide = false;
        }
    }

  E.trima = trima; E.trimy = trimy; E.trimd = trimd; E.trimha = trimha; E.trimhb = 1;
  E.morea = morea; E.morey = morey; E.mored = mored; E.moreha = moreha; E.morehb = 1;
  E.morem = morem;
  E.ncell = ncell;
  return AL_OK;
}

// Reverse a pebble chain in place and return its first pebble (align.c:900-906)
__device__ __forceinline__ int reverse_chain(const CellsView &cells, int head)
{ int a = -1, h = head;
  while (h >= 0)
    { int nx = cells.at(h).x;
      cells.setx(h, a);
      a = h;
      h = nx;
    }
  return a;
}

// Trace back of the forward pass (align.c:895-1006).  Runs on lane 0 only.
template <int W, bool WANT_B>
__device__ int forward_trace(const WaveWork<W> &wk, const AlignSpecDev &sp, const WaveEnd &E, int mida,
                             PathRes &P, int &klow)
{ const CellsView cells = cells_of(wk);
  uint16_t *atrace = wk.atrace + P.aorg, *btrace = wk.btrace + P.borg;
  int trimx, trimy = E.trimy, trimd = E.trimd, trimha = E.trimha, trimhb = E.trimhb;
  int atlen = 0, btlen = 0;
  int a, b, k, h, d, e;
  const int room_a = wk.tcap - P.aorg - 4, room_b = wk.tcap - P.borg - 4;

  if (E.morem >= 0 && sp.REACH)
    { trimx = E.morea - E.morey; trimy = E.morey; trimd = E.mored; trimha = E.moreha; trimhb = E.morehb; }
  else
    trimx = E.trima - trimy;

  h = reverse_chain(cells, trimha);
  k = cells.at(h).y;
  b = (mida - k) / 2;
  e = 0;
  for (h = cells.at(h).x; h >= 0; h = cells.at(h).x)
    { int4 p = cells.at(h);
      k = p.y;
      a = p.w - k;
      d = p.z;
      if (atlen >= room_a) return AL_ERR_TRACE;
      atrace[atlen++] = (uint16_t) (d - e);
      atrace[atlen++] = (uint16_t) (a - b);
      b = a; e = d;
    }
  if (b + k != trimx)
    { if (atlen >= room_a) return AL_ERR_TRACE;
      atrace[atlen++] = (uint16_t) (trimd - e);
      atrace[atlen++] = (uint16_t) (trimy - b);
    }
  else if (b != trimy && atlen > 0)
    { atrace[atlen - 1] = (uint16_t) (atrace[atlen - 1] + (trimy - b));
      atrace[atlen - 2] = (uint16_t) (atrace[atlen - 2] + (trimd - e));
    }

  if (!WANT_B)
    { P.aepos = trimx; P.bepos = trimy; P.diffs = trimd; P.atlen = atlen; P.btlen = 0;
      return AL_OK;
    }
  h = reverse_chain(cells, trimhb);
  k = cells.at(h).y;
  b = (mida + k) / 2;
  e = 0;
  klow = k;
  for (h = cells.at(h).x; h >= 0; h = cells.at(h).x)
    { int4 p = cells.at(h);
      k = p.y;
      a = p.w + k;
      d = p.z;
      if (btlen >= room_b) return AL_ERR_TRACE;
      btrace[btlen++] = (uint16_t) (d - e);
      btrace[btlen++] = (uint16_t) (a - b);
      b = a; e = d;
    }
  if (b - k != trimy)
    { if (btlen >= room_b) return AL_ERR_TRACE;
      btrace[btlen++] = (uint16_t) (trimd - e);
      btrace[btlen++] = (uint16_t) (trimx - b);
    }
  else if (b != trimx && btlen > 0)
    { btrace[btlen - 1] = (uint16_t) (btrace[btlen - 1] + (trimx - b));
      btrace[btlen - 2] = (uint16_t) (btrace[btlen - 2] + (trimd - e));
    }

  P.aepos = trimx;
  P.bepos = trimy;
  P.diffs = trimd;
  P.atlen = atlen;
  P.btlen = btlen;
  return AL_OK;
}

// Trace back of the reverse pass (align.c:1554-1714): prepends to the forward trace.  Lane 0 only.
template <int W, bool WANT_B>
__device__ int reverse_trace(const WaveWork<W> &wk, const AlignSpecDev &sp, const WaveEnd &E,
                             int aoff, int boff, PathRes &P)
{ const CellsView cells = cells_of(wk);
  const int TS = sp.TS;
  uint16_t *atrace = wk.atrace + P.aorg, *btrace = wk.btrace + P.borg;
  int trimx, trimy = E.trimy, trimd = E.trimd, trimha = E.trimha, trimhb = E.trimhb;
  int atlen = 0, btlen = 0;
  int a, b, k, h, d, e;
  const int room_a = wk.tcap + P.aorg - 4, room_b = wk.tcap + P.borg - 4;

  if (E.morem >= 0 && sp.REACH)
    { trimx = E.morea - E.morey; trimy = E.morey; trimd = E.mored; trimha = E.moreha; trimhb = E.morehb; }
  else
    trimx = E.trima - trimy;

  // A view
  h = reverse_chain(cells, trimha);
  k = cells.at(h).y;
  b = cells.at(h).w - k;
  e = 0;
  if ((b + k) % TS != aoff)
    { h = cells.at(h).x;
      if (h < 0)
        { a = trimy; d = trimd; }
      else
        { int4 p = cells.at(h);
          k = p.y; a = p.w - k; d = p.z;
        }
      if (P.atlen == 0)
        { atrace[--atlen] = (uint16_t) (b - a);
          atrace[--atlen] = (uint16_t) (d - e);
        }
      else
        { atrace[1] = (uint16_t) (atrace[1] + (b - a));
          atrace[0] = (uint16_t) (atrace[0] + (d - e));
        }
      b = a; e = d;
    }
  if (h >= 0)
    { for (h = cells.at(h).x; h >= 0; h = cells.at(h).x)
        { int4 p = cells.at(h);
          k = p.y;
          a = p.w - k;
          d = p.z;
          if (-atlen >= room_a) return AL_ERR_TRACE;
          atrace[--atlen] = (uint16_t) (b - a);
          atrace[--atlen] = (uint16_t) (d - e);
          b = a; e = d;
        }
      if (b + k != trimx)
        { if (-atlen >= room_a) return AL_ERR_TRACE;
          atrace[--atlen] = (uint16_t) (b - trimy);
          atrace[--atlen] = (uint16_t) (trimd - e);
        }
      else if (b != trimy && (atlen < 0 || P.atlen > 0))
        { atrace[atlen + 1] = (uint16_t) (atrace[atlen + 1] + (b - trimy));
          atrace[atlen]     = (uint16_t) (atrace[atlen] + (trimd - e));
        }
    }

  // B view
  if (!WANT_B)
    { P.abpos = trimx; P.bbpos = trimy; P.diffs = P.diffs + trimd;
      P.atlen = P.atlen - atlen; P.aorg = P.aorg + atlen;
      return AL_OK;
    }
  h = reverse_chain(cells, trimhb);
  k = cells.at(h).y;
  b = cells.at(h).w + k;
  e = 0;
  if ((b - k) % TS != boff)
    { h = cells.at(h).x;
      if (h < 0)
        { a = trimx; d = trimd; }
      else
        { int4 p = cells.at(h);
          k = p.y; a = p.w + k; d = p.z;
        }
      if (P.btlen == 0)
        { btrace[--btlen] = (uint16_t) (b - a);     // sic (align.c:1670-1671): b-a in both slots
          btrace[--btlen] = (uint16_t) (b - a);
        }
      else
        { btrace[1] = (uint16_t) (btrace[1] + (b - a));
          btrace[0] = (uint16_t) (btrace[0] + (d - e));
        }
      b = a; e = d;
    }
  if (h >= 0)
    { for (h = cells.at(h).x; h >= 0; h = cells.at(h).x)
        { int4 p = cells.at(h);
          k = p.y;
          a = p.w + k;
          d = p.z;
          if (-btlen >= room_b) return AL_ERR_TRACE;
          btrace[--btlen] = (uint16_t) (b - a);
          btrace[--btlen] = (uint16_t) (d - e);
          b = a; e = d;
        }
      if (b - k != trimy)
        { if (-btlen >= room_b) return AL_ERR_TRACE;
          btrace[--btlen] = (uint16_t) (b - trimx);
          btrace[--btlen] = (uint16_t) (trimd - e);
        }
      else if (b != trimx && (btlen < 0 || P.btlen > 0))
        { btrace[btlen + 1] = (uint16_t) (btrace[btlen + 1] + (b - trimx));
          btrace[btlen]     = (uint16_t) (btrace[btlen] + (trimd - e));
        }
    }

  P.abpos = trimx;
  P.bbpos = trimy;
  P.diffs = P.diffs + trimd;
  P.atlen = P.atlen - atlen;
  P.aorg  = P.aorg + atlen;
  P.btlen = P.btlen - btlen;
  P.borg  = P.borg + btlen;
  return AL_OK;
}

// Lane 0 ran the (sequential) trace back; publish its results to the whole group through LDS.
template <int G>
__device__ __forceinline__ void bcast_path(int *mbox, PathRes &P, int &st, int &klow)
{ const int lane = Grp<G>::gl();
  if (lane == 0)
    { mbox[0] = P.abpos; mbox[1] = P.bbpos; mbox[2] = P.aepos; mbox[3] = P.bepos; mbox[4] = P.diffs;
      mbox[5] = P.atlen; mbox[6] = P.btlen; mbox[7] = P.aorg; mbox[8] = P.borg; mbox[9] = st; mbox[10] = klow;
    }
  wave_sync();
  P.abpos = mbox[0]; P.bbpos = mbox[1]; P.aepos = mbox[2]; P.bepos = mbox[3]; P.diffs = mbox[4];
  P.atlen = mbox[5]; P.btlen = mbox[6]; P.aorg = mbox[7]; P.borg = mbox[8]; st = mbox[9]; klow = mbox[10];
  wave_sync();
}

// Local_Alignment (align.c:1727-1946) for one seed (dg, anti); the whole wavefront takes part.
// On return P holds the A view in the frame the reference reports (flipped back for comp) with
// the A trace at wk.atrace+P.aorg; B = {babpos,bbbpos,baepos,bbepos} is the B view, its trace
// at wk.btrace+P.borg (complemented-frame order, as the reference leaves it for ACOMP).
// One pass (wave + trace back).  Fast path: register-resident band; when the B view is not
// wanted its pebbles live in the (otherwise idle) LDS ring, so the trace back chases pointers in
// LDS.  Falls back to the LDS-ring pass with pebbles in global memory when the band or the pebble
// count outgrows the fast path.

// With G = 32 (two alignments per wavefront) there is no in-kernel fallback: the error goes back to
// the host, which re-runs the candidate on the one-per-wavefront kernel.
template <int W, bool WANT_B, int DIR, int G = 64>
__device__ int run_pass(const WaveWork<W> &wk, const AlignSpecDev &sp, const SeqView &S,
                        int k0, int mida, int aoff, int boff, PathRes &P, int &klow)
{ const int lane = Grp<G>::gl();
  WaveEnd E;
  int st;
  WaveWork<W> wf = wk;
  if (!WANT_B && G == 64)
    { wf.cells   = (int4 *) wk.band;
      wf.cellcap = ((F_COUNT * 2 + 2) * W) / 4;
      wf.cells_hi = nullptr; wf.cellsplit = 0x7fffffff;
    }
  APROF_T0();
  if constexpr (G == 16)
    { static_assert(G != 16 || !WANT_B, "the four-per-wavefront pass keeps the A view only");
      st = wave_pass_reg16<DIR, WaveWork<W>>(wf, sp, S, k0, mida, aoff, E);
    }
  else
    st = wave_pass_reg<DIR, WANT_B, WaveWork<W>, G>(wf, sp, S, k0, mida, aoff, boff, E);
  APROF_LAP(0);
  APROF_ADD(3, 1);
  if (G == 64 && (st == AL_ERR_BAND || st == AL_ERR_PEBBLES))
    { wf = wk;
      st = wave_pass<W, DIR>(wf, sp, S, k0, mida, aoff, boff, E);
      APROF_LAP(1);
      APROF_ADD(4, 1);
      if (st != AL_OK) return st;
      wave_sync();
      if (lane == 0)
        st = (DIR > 0) ? forward_trace<W, WANT_B>(wf, sp, E, mida, P, klow)
                       : reverse_trace<W, WANT_B>(wf, sp, E, aoff, boff, P);
    }
  else
    { if (st != AL_OK) return st;
      wave_sync();
      if (lane == 0)
        st = (DIR > 0) ? forward_trace<W, WANT_B>(wf, sp, E, mida, P, klow)
                       : reverse_trace<W, WANT_B>(wf, sp, E, aoff, boff, P);
    }
  APROF_ADD(5, E.ncell);
  bcast_path<G>(wk.mbox, P, st, klow);
  APROF_LAP(2);
  if (st != AL_OK) return st;
  wave_sync();
  return AL_OK;
}

// Local_Alignment (align.c:1727-1946) for one seed (dg, anti); the whole wavefront takes part.
// On return P holds the A view in the frame the reference reports (flipped back for comp) with
// the A trace at wk.atrace+P.aorg; B = {babpos,bbbpos,baepos,bbepos} is the B view, its trace
// (WANT_B only) at wk.btrace+P.borg (complemented-frame order, as the reference leaves it).
template <int W, bool WANT_B, int G = 64>
__device__ int local_alignment(const WaveWork<W> &wk, const AlignSpecDev &sp, const SeqView &S,
                               int dg, int anti, PathRes &P, int *B)
{ const int lane = Grp<G>::gl();
  int aoff = S.comp ? (S.alen % sp.TS) : 0, boff = 0;
  int hgh = dg, low = dg;
  while (((anti - hgh) >> 1) < 0) hgh -= 1;     // align.c:1774 (never taken for damapper's seeds)
  low = hgh;

  int st, klow = low;
  P.abpos = P.bbpos = P.aepos = P.bepos = P.diffs = 0;
  P.atlen = P.btlen = 0;
  P.aorg = P.borg = 0;

  st = run_pass<W, WANT_B, +1, G>(wk, sp, S, low, anti, aoff, boff, P, klow);
  if (st != AL_OK) return st;
  int fshort = ((P.aepos + P.bepos) - anti < AL_DUB_TRIM);

  st = run_pass<W, WANT_B, -1, G>(wk, sp, S, low, anti, aoff, boff, P, klow);
  if (st != AL_OK) return st;
  int rshort = (anti - (P.abpos + P.bbpos) < AL_DUB_TRIM);

  if (fshort)
    { if (rshort)
        { P.aepos = P.abpos = (P.abpos + P.aepos) / 2;
          P.bepos = P.bbpos = (P.bbpos + P.bepos) / 2;
          P.atlen = P.btlen = 0;
        }
      else
        { low  = P.abpos - P.bbpos;
          anti = P.abpos + P.bbpos;
          P.atlen = P.btlen = 0;
          st = run_pass<W, WANT_B, +1, G>(wk, sp, S, low, anti, aoff, boff, P, klow);
          if (st != AL_OK) return st;
        }
    }
  else if (rshort)
    { low  = P.aepos - P.bepos;
      anti = P.aepos + P.bepos;
      P.atlen = P.btlen = 0;
      P.diffs = 0;
      st = run_pass<W, WANT_B, -1, G>(wk, sp, S, low, anti, aoff, boff, P, klow);
      if (st != AL_OK) return st;
    }

  // B view and, for a complemented A, the flip back to the forward frame (align.c:1857-1912)
  B[0] = P.bbpos; B[1] = P.abpos; B[2] = P.bepos; B[3] = P.aepos;
  if (S.comp)
    { int abp = S.alen - B[3], bbp = S.blen - B[2], aep = S.alen - B[1], bep = S.blen - B[0];
      P.abpos = abp; P.bbpos = bbp; P.aepos = aep; P.bepos = bep;
      if (lane == 0)
        { uint16_t *t = wk.atrace + P.aorg;
          int i = P.atlen - 2, j = 0;
          while (j < i)
            { uint16_t p = t[i]; t[i] = t[j]; t[j] = p;
              p = t[i + 1]; t[i + 1] = t[j + 1]; t[j + 1] = p;
              i -= 2;
              j += 2;
            }
        }
      wave_sync();
    }
  return AL_OK;
}

}  // namespace dmh
