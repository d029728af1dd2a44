// dmh_report.hip -- Reporter (map.c:3227-3319): candidate chains -> local alignments (d-wave
// kernel, one wavefront per candidate) -> redundancy removal, chain linking and selection on the
// host (sequential per read, a few records per read) -> per-"thread" .las files and -p track.

#include "dmh_state.h"
#include "dmh_spec.h"
#include "dmh_align.h"
#include <chrono>
#include <memory>
#include "dmh_chain.h"
#include "dmh_merge.h"
#include <algorithm>
#include <math.h>
#include <atomic>
#include <thread>
#include <future>
#include <mutex>
#include <condition_variable>
#include <deque>

namespace dmh {

constexpr int LDS_W  = 128;    // ring width of the LDS band
constexpr int GLB_W  = 8192;   // ring width of the global-memory fallback band
constexpr int CELLCAP = 16384; // pebbles per wavefront arena

int16_t *spec_device_tables(Spec *sp)
{ if (sp->d_tables == nullptr)
    { sp->d_tables = (int16_t *) dev_alloc(sizeof(int16_t) * 65536, "trim tables");
      copy_h2d(sp->d_tables, sp->score.data(), sizeof(int16_t) * 32768);
      copy_h2d(sp->d_tables + 32768, sp->table.data(), sizeof(int16_t) * 32768);
      HIP_CHECK(hipStreamSynchronize(stream()));
    }
  return sp->d_tables;
}

void spec_release_device(Spec *sp)
{ if (sp->d_tables) dev_free(sp->d_tables);
  sp->d_tables = nullptr;
}

// ---------------------------------------------------------------------------------------
//  Kernels
// ---------------------------------------------------------------------------------------

struct SeedJob          // one Local_Alignment call (stage-level API)
{ int64_t aoff, boff;
  int     alen, blen, comp, dg, ad;
  int     pad;
};

struct CandJob          // one candidate chain = a sequence of dependent Local_Alignment calls
{ int64_t  aoff, boff;  // first base of the read / contig inside the uploaded blocks
  int      alen, blen, comp;
  int      alast, blast;     // chain end (map.c:2487-2488)
  int      npts;
  uint32_t joff;             // first (adisp,bdisp) pair of the chain
  int      pad;
};

struct Scratch
{ int      *band;       // global band (fallback instantiation only)
  int4     *cells;
  uint16_t *traces;     // per wavefront: A buffer then B buffer, each 2*tcap values
  int       tcap;
};

template <int W, bool LDSBAND>
__device__ __forceinline__ WaveWork<W> make_work(const Scratch &sc, int *lds, int *mbox, int wave_global, int wave_local)
{ WaveWork<W> wk;
  wk.mbox = mbox + 16 * wave_local;
  if (LDSBAND)
    wk.band = lds + (size_t) wave_local * ((F_COUNT * 2 + 2) * W);
  else
    wk.band = sc.band + (size_t) wave_global * ((F_COUNT * 2 + 2) * W);
  wk.cells   = sc.cells + (size_t) wave_global * CELLCAP;
  wk.cellcap = CELLCAP;
  wk.tcap    = sc.tcap;
  wk.atrace  = sc.traces + (size_t) wave_global * (4 * (size_t) sc.tcap) + sc.tcap;
  wk.btrace  = wk.atrace + 2 * (size_t) sc.tcap;
  return wk;
}

template <int W, bool LDSBAND>
__global__ __launch_bounds__(256) void k_align_seeds(const uint64_t *apk, const uint64_t *arc, const uint64_t *bpk,
                                                      const SeedJob *jobs, int njobs, AlignSpecDev sp,
                                                      Scratch sc, unsigned int *ticket, int *out,
                                                      uint16_t *atr_out, uint16_t *btr_out, int stride,
                                                      const int *only)
{ __shared__ __attribute__((aligned(16))) int lds[LDSBAND ? 4 * (F_COUNT * 2 + 2) * W : 4];
  const int lane = threadIdx.x & 63, wl = threadIdx.x >> 6;
  const int wg = blockIdx.x * 4 + wl;
  __shared__ int  mbox[4 * 16];
  __shared__ int2 trimtab[32];
  if (wl == 0) build_trim_tab(trimtab, sp.mscore, sp.dscore);
  __syncthreads();
  WaveWork<W> wk = make_work<W, LDSBAND>(sc, lds, mbox, wg, wl);
  wk.trimtab = trimtab;
  while (true)
    { unsigned int t = 0;
      if (lane == 0) t = atomicAdd(ticket, 1u);
      t = __shfl(t, 0);
      if (t >= (unsigned int) njobs) break;
      int j = only ? only[t] : (int) t;
      SeedJob J = jobs[j];
      SeqView S{J.comp ? arc : apk, bpk, J.aoff, J.boff, J.alen, J.blen, J.comp};
      PathRes P;
      int B[4];
      int st = local_alignment<W, true>(wk, sp, S, J.dg, J.ad, P, B);
      if (lane == 0)
        { int *o = out + 16 * (size_t) j;
          o[0] = P.abpos; o[1] = P.bbpos; o[2] = P.aepos; o[3] = P.bepos; o[4] = P.diffs;
          o[5] = (st == AL_OK) ? P.atlen : 0;
          o[6] = (st == AL_OK) ? P.btlen : 0;
          o[7] = st;
          o[8] = B[0]; o[9] = B[1]; o[10] = B[2]; o[11] = B[3];
          o[12] = P.aorg; o[13] = P.borg; o[14] = wk.atrace[P.aorg]; o[15] = wk.atrace[0];
        }
      if (st == AL_OK)
        { int na = P.atlen < stride ? P.atlen : stride, nb = P.btlen < stride ? P.btlen : stride;
          for (int i = lane; i < na; i += 64) atr_out[(size_t) j * stride + i] = wk.atrace[P.aorg + i];
          if (btr_out != nullptr)
            for (int i = lane; i < nb; i += 64) btr_out[(size_t) j * stride + i] = wk.btrace[P.borg + i];
        }
      wave_sync();
    }
}

// Stage-level API on the lane-group passes (test option "align_g" = 16 / 32): G lanes per Local_Alignment, A view only
// (the B end points follow from the A view; B trace lengths are reported as 0).  Jobs whose band or pebbles outgrow
// the group come back with their error status and are re-run by k_align_seeds.
template <int G, int PEB>
__global__ __launch_bounds__(256, (G == 16 ? 3 : 4)) void k_align_seeds_g(const uint64_t *apk, const uint64_t *arc,
                                                                         const uint64_t *bpk, const SeedJob *jobs, int njobs,
                                                                         AlignSpecDev sp, uint16_t *traces, int tcap,
                                                                         unsigned int *ticket, int *out, uint16_t *atr_out,
                                                                         int stride)
{ constexpr int NG = 256 / G;
  __shared__ int4 peb[NG * PEB];
  __shared__ int  mbox[NG * 16];
  __shared__ int2 trimtab[32];
  const int lane = threadIdx.x & (G - 1), gid = threadIdx.x / G;
  const int gg = blockIdx.x * NG + gid;
  if (threadIdx.x < 64) build_trim_tab(trimtab, sp.mscore, sp.dscore);
  __syncthreads();
  WaveWork<LDS_W> wk;
  wk.band    = nullptr;
  wk.cells   = peb + gid * PEB;
  wk.cellcap = PEB;
  wk.tcap    = tcap;
  wk.atrace  = traces + (size_t) gg * (4 * (size_t) tcap) + tcap;
  wk.btrace  = wk.atrace + 2 * (size_t) tcap;
  wk.mbox    = mbox + 16 * gid;
  wk.trimtab = trimtab;
  while (true)
    { unsigned int t = 0;
      if (lane == 0) t = atomicAdd(ticket, 1u);
      t = __shfl(t, 0, G);
      if (t >= (unsigned int) njobs) break;
      const int j = (int) t;
      SeedJob J = jobs[j];
      SeqView S{J.comp ? arc : apk, bpk, J.aoff, J.boff, J.alen, J.blen, J.comp};
      PathRes P;
      int B[4];
      int st = local_alignment<LDS_W, false, G>(wk, sp, S, J.dg, J.ad, P, B);
      if (lane == 0)
        { int *o = out + 16 * (size_t) j;
          o[0] = P.abpos; o[1] = P.bbpos; o[2] = P.aepos; o[3] = P.bepos; o[4] = P.diffs;
          o[5] = (st == AL_OK) ? P.atlen : 0;
          o[6] = 0;
          o[7] = st;
          o[8] = B[0]; o[9] = B[1]; o[10] = B[2]; o[11] = B[3];
          o[12] = P.aorg; o[13] = 0; o[14] = 0; o[15] = 0;
        }
      if (st == AL_OK)
        { int na = P.atlen < stride ? P.atlen : stride;
          for (int i = lane; i < na; i += G) atr_out[(size_t) j * stride + i] = wk.atrace[P.aorg + i];
        }
      wave_sync();
    }
}

// Output pool record: 16-int header {cand, seq, abpos, bbpos, aepos, bepos, diffs, atlen,
// B view: abpos, bbpos, aepos, bepos, btlen, 0, 0, 0} followed by atlen uint16 A-trace values (padded
// to an even count) and, for -C, btlen B-trace values (padded likewise).
constexpr int REC_HDR = 16;

template <int W, bool LDSBAND, bool WANT_B>
__global__ __launch_bounds__(256, 4) void k_align_cands(const uint64_t *apk, const uint64_t *arc, const uint64_t *bpk,
                                                      const CandJob *jobs, int njobs,
                                                      const uint16_t *jumps, int hithr, AlignSpecDev sp,
                                                      Scratch sc, unsigned int *ticket, int *status,
                                                      int *pool, unsigned int *pool_used,
                                                      unsigned int pool_cap, const int *only)
{ __shared__ __attribute__((aligned(16))) int lds[LDSBAND ? 4 * (F_COUNT * 2 + 2) * W : 4];
  const int lane = threadIdx.x & 63, wl = threadIdx.x >> 6;
  const int wg = blockIdx.x * 4 + wl;
  __shared__ int  mbox[4 * 16];
  __shared__ int2 trimtab[32];
  if (wl == 0) build_trim_tab(trimtab, sp.mscore, sp.dscore);
  __syncthreads();
  WaveWork<W> wk = make_work<W, LDSBAND>(sc, lds, mbox, wg, wl);
  wk.trimtab = trimtab;
  while (true)
    { unsigned int t = 0;
      if (lane == 0) t = atomicAdd(ticket, 1u);
      t = __shfl(t, 0);
      if (t >= (unsigned int) njobs) break;
      const int j = only ? only[t] : (int) t;
      const CandJob J = jobs[j];
      SeqView S{J.comp ? arc : apk, bpk, J.aoff, J.boff, J.alen, J.blen, J.comp};
#ifdef DMH_ALIGN_PROFILE
      long long tjob = clock64();
#endif

      // map.c:2487-2579: walk the chain from its end; a point is a seed only if it lies left of
      // the last accepted alignment's start
      int apos = J.alast, bpos = J.blast, alast = J.alen + 1;
      int seq = 0, st = AL_OK;
      for (int n = 0; n < J.npts; n++)
        { apos -= (int) jumps[2 * ((size_t) J.joff + n)];
          bpos -= (int) jumps[2 * ((size_t) J.joff + n) + 1];
          if (apos >= alast) continue;
          int dg, ad;
          if (J.comp)
            { int ac = J.alen - apos, bc = J.blen - bpos;
              dg = ac - bc;
              ad = ac + bc;
            }
          else
            { dg = apos - bpos;
              ad = apos + bpos;
            }
          PathRes P;
          int B[4];
          st = local_alignment<W, WANT_B>(wk, sp, S, dg, ad, P, B);
          if (st != AL_OK) break;
          if (P.aepos - P.abpos >= hithr)
            { alast = P.abpos;
              const int    btl  = WANT_B ? P.btlen : 0;
              unsigned int need = REC_HDR + ((unsigned int) P.atlen + 1) / 2 + ((unsigned int) btl + 1) / 2;
              unsigned int at = 0;
              if (lane == 0) at = atomicAdd(pool_used, need);
              at = __shfl(at, 0);
              if (at + need > pool_cap) { st = 100; break; }
              int *r = pool + at;
              if (lane == 0)
                { r[0] = j; r[1] = seq; r[2] = P.abpos; r[3] = P.bbpos; r[4] = P.aepos; r[5] = P.bepos;
                  r[6] = P.diffs; r[7] = P.atlen;
                  r[8] = B[0]; r[9] = B[1]; r[10] = B[2]; r[11] = B[3]; r[12] = btl;
                  r[13] = 0; r[14] = 0; r[15] = 0;
                }
              uint16_t *tr = (uint16_t *) (r + REC_HDR);
              for (int i = lane; i < P.atlen; i += 64) tr[i] = wk.atrace[P.aorg + i];
              if (WANT_B)
                { uint16_t *tb2 = (uint16_t *) (r + REC_HDR + (P.atlen + 1) / 2);
                  for (int i = lane; i < btl; i += 64) tb2[i] = wk.btrace[P.borg + i];
                }
              seq += 1;
            }
          wave_sync();
        }
      if (lane == 0) status[j] = st;
#ifdef DMH_ALIGN_PROFILE
      APROF_ADD(6, clock64() - tjob);
      APROF_ADD(7, 1);
#endif
      wave_sync();
    }
}


// Two candidates per wavefront: each 32-lane half runs the chain walk of k_align_cands on its own
// job with the register band on a 32-lane ring (bands up to 30 diagonals) and its pebbles in LDS.
// The halves share one instruction stream, so a wavefront's two jobs should be alike (the host
// orders the tickets by orientation and read length).  A band or pebble overflow ends the job with
// AL_ERR_BAND / AL_ERR_PEBBLES; the host re-runs it on k_align_cands.
constexpr int PEB32 = 288;     // pebbles per half-wavefront
// Four per wavefront (16-lane groups, dmh_align.h wave_pass_reg16): 3 workgroups of 16 groups per CU, so a group's
// pebbles get 3.2 KB of LDS.
constexpr int PEB16 = 200;

template <int G, int PEB>
__device__ __forceinline__ void align_cands_groups(const uint64_t *apk, const uint64_t *arc, const uint64_t *bpk,
                                                   const CandJob *jobs, int njobs,
                                                   const uint16_t *jumps, int hithr, const AlignSpecDev &sp,
                                                   uint16_t *traces, int tcap, unsigned int *ticket, int *status,
                                                   int *pool, unsigned int *pool_used,
                                                   unsigned int pool_cap, const int *only, int4 *spill, int spillcap)
{ constexpr int NG = 256 / G;
  __shared__ int4 peb[NG * PEB];
  __shared__ int  mbox[NG * 16];
  __shared__ int2 trimtab[32];
  const int lane = threadIdx.x & (G - 1), gid = threadIdx.x / G;
  const int gg = blockIdx.x * NG + gid;
  if (threadIdx.x < 64) build_trim_tab(trimtab, sp.mscore, sp.dscore);
  __syncthreads();
  WaveWork<LDS_W> wk;
  wk.band    = nullptr;
  wk.cells   = peb + gid * PEB;
  wk.cellcap = PEB;
  if (spill != nullptr)
    { // a pass of more than PEB pebbles (a read beyond ~100 PEB bases) puts the rest into the group's arena in global
      // memory instead of ending the job for the one-per-wavefront kernel (align.c:771-817 has one arena)
      wk.cells_hi  = spill + (size_t) gg * (size_t) spillcap;
      wk.cellsplit = PEB;
      wk.cellcap   = PEB + spillcap;
    }
  wk.tcap    = tcap;
  wk.atrace  = traces + (size_t) gg * (4 * (size_t) tcap) + tcap;
  wk.btrace  = wk.atrace + 2 * (size_t) tcap;
  wk.mbox    = mbox + 16 * gid;
  wk.trimtab = trimtab;
  while (true)
    { unsigned int t = 0;
      if (lane == 0) t = atomicAdd(ticket, 1u);
      t = __shfl(t, 0, G);
      if (t >= (unsigned int) njobs) break;
      const int j = only ? only[t] : (int) t;
      const CandJob J = jobs[j];
      SeqView S{J.comp ? arc : apk, bpk, J.aoff, J.boff, J.alen, J.blen, J.comp};

      int apos = J.alast, bpos = J.blast, alast = J.alen + 1;     // map.c:2487-2579, as k_align_cands
      int seq = 0, st = AL_OK;
      for (int n = 0; n < J.npts; n++)
        { apos -= (int) jumps[2 * ((size_t) J.joff + n)];
          bpos -= (int) jumps[2 * ((size_t) J.joff + n) + 1];
          if (apos >= alast) continue;
          int dg, ad;
          if (J.comp)
            { int ac = J.alen - apos, bc = J.blen - bpos;
              dg = ac - bc;
              ad = ac + bc;
            }
          else
            { dg = apos - bpos;
              ad = apos + bpos;
            }
          PathRes P;
          int B[4];
          st = local_alignment<LDS_W, false, G>(wk, sp, S, dg, ad, P, B);
          if (st != AL_OK) break;
          if (P.aepos - P.abpos >= hithr)
            { alast = P.abpos;
              unsigned int need = REC_HDR + ((unsigned int) P.atlen + 1) / 2;
              unsigned int at = 0;
              if (lane == 0) at = atomicAdd(pool_used, need);
              at = __shfl(at, 0, G);
              if (at + need > pool_cap) { st = 100; break; }
              int *r = pool + at;
              if (lane == 0)
                { r[0] = j; r[1] = seq; r[2] = P.abpos; r[3] = P.bbpos; r[4] = P.aepos; r[5] = P.bepos;
                  r[6] = P.diffs; r[7] = P.atlen;
                  r[8] = B[0]; r[9] = B[1]; r[10] = B[2]; r[11] = B[3]; r[12] = 0;
                  r[13] = 0; r[14] = 0; r[15] = 0;
                }
              uint16_t *tr = (uint16_t *) (r + REC_HDR);
              for (int i = lane; i < P.atlen; i += G) tr[i] = wk.atrace[P.aorg + i];
              seq += 1;
            }
          wave_sync();
        }
      if (lane == 0) status[j] = st;
      wave_sync();
    }
}

__global__ __launch_bounds__(256, 4) void k_align_cands32(const uint64_t *apk, const uint64_t *arc, const uint64_t *bpk,
                                                        const CandJob *jobs, int njobs,
                                                        const uint16_t *jumps, int hithr, AlignSpecDev sp,
                                                        uint16_t *traces, int tcap, unsigned int *ticket, int *status,
                                                        int *pool, unsigned int *pool_used,
                                                        unsigned int pool_cap, const int *only, int4 *spill, int spillcap)
{ align_cands_groups<32, PEB32>(apk, arc, bpk, jobs, njobs, jumps, hithr, sp, traces, tcap, ticket, status, pool, pool_used,
                                pool_cap, only, spill, spillcap);
}

__global__ __launch_bounds__(256, 3) void k_align_cands16(const uint64_t *apk, const uint64_t *arc, const uint64_t *bpk,
                                                        const CandJob *jobs, int njobs,
                                                        const uint16_t *jumps, int hithr, AlignSpecDev sp,
                                                        uint16_t *traces, int tcap, unsigned int *ticket, int *status,
                                                        int *pool, unsigned int *pool_used,
                                                        unsigned int pool_cap, const int *only, int4 *spill, int spillcap)
{ align_cands_groups<16, PEB16>(apk, arc, bpk, jobs, njobs, jumps, hithr, sp, traces, tcap, ticket, status, pool, pool_used,
                                pool_cap, only, spill, spillcap);
}

// ---------------------------------------------------------------------------------------
//  Host helpers for launching the alignment kernels
// ---------------------------------------------------------------------------------------

struct AlignScratch
{ DevBuf<int4>     cells;
  DevBuf<uint16_t> traces;
  DevBuf<int>      band;
  DevBuf<unsigned int> ctl;
  int nwaves = 0, tcap = 0;
};

static AlignScratch &ascr()
{ static AlignScratch s;
  return s;
}

// Pebbles a lane group keeps in global memory behind the ones in its LDS (k_align_cands32 / 16): 288 + 1,760 cover a
// pass over 200 kb of read
constexpr int SPILLCAP = 1760;

static Scratch prepare_scratch(int nblocks, int tcap, bool global_band, int groups_per_block = 4, bool arenas = true)
{ AlignScratch &a = ascr();
  int nw = nblocks * groups_per_block;
  a.cells.ensure((size_t) nw * (arenas ? CELLCAP : SPILLCAP), "pebble arenas");
  a.traces.ensure((size_t) nw * 4 * tcap, "trace buffers");
  if (global_band)
    a.band.ensure((size_t) nw * (F_COUNT * 2 + 2) * GLB_W, "global d-wave bands");
  a.ctl.ensure(8, "alignment counters");
  Scratch sc;
  sc.band = a.band.p;
  sc.cells = a.cells.p;
  sc.traces = a.traces.p;
  sc.tcap = tcap;
  return sc;
}

static int trace_cap(int maxalen, int tspace)
{ // A view: <= alen/s + 2 pairs per direction; B view can be ~1.3x longer.  Two reverse passes
  // may stack on the low side (retry path), hence the factor 2 and the slack.
  int pairs = (int) ((int64_t) maxalen * 3 / 2 / tspace) + 16;
  return 2 * (2 * pairs) + 64;
}

static AlignSpecDev spec_dev(Spec *sp)
{ AlignSpecDev d;
  int16_t *t = spec_device_tables(sp);
  d.TS = sp->trace_space;
  d.PATH_AVE = sp->ave_path;
  d.REACH = sp->reach;
  d.mscore = sp->mscore;
  d.dscore = sp->dscore;
  d.SCORE = t;
  d.TABLE = t + 32768;
  return d;
}

// ---------------------------------------------------------------------------------------
//  Host: report/select stage (map.c:1953-2871), sequential per read
// ---------------------------------------------------------------------------------------

constexpr uint32_t COMP_FLAG  = 0x1;    // align.h:127
constexpr uint32_t START_FLAG = 0x4;    // align.h:133
constexpr uint32_t NEXT_FLAG  = 0x8;    // align.h:134
constexpr uint32_t BEST_FLAG  = 0x10;   // align.h:135
constexpr int      TRACE_XOVR = 125;    // align.h:21

constexpr double CHAIN_OFF  = 500.;     // map.c:42-49
constexpr double CHAIN_OVL  = 400.;
constexpr double CHAIN_PLAY = 1.4;
constexpr double DIFF_SCORE = 2.3;
constexpr int    TIE_SCORE  = 50;
constexpr int    TIE_GAP    = 500;

struct Ovl
{ int      tlen, diffs, abpos, bbpos, aepos, bepos;
  uint32_t flags;
  int      aread, bread;
  size_t   trace;       // offset into the read's trace buffer
};

struct Link { int score, link, mark; };
struct Zone { int beg, end, top; };

// map.c:1953-2058: do the two paths share a trace point?  Returns the minimum |b-offset| at the
// A trace lines both cross (0 => they meet, *where = that A coordinate), -1 if they share none.
static int entwine(const Ovl &J, const Ovl &K, const std::vector<uint16_t> &tb, int S, int *where)
{ const uint16_t *kt = tb.data() + K.trace, *jt = tb.data() + J.trace;
  int min = 10000, den = 0;
  int y2 = J.bbpos, j = J.abpos / S;
  int b2 = K.bbpos, k = K.abpos / S;
  int ac, i;

  if (J.abpos == K.abpos)
    { min = abs(y2 - b2);
      if (min == 0) *where = K.abpos;
    }
  if (j < k)
    { ac = k * S;
      j = 1 + 2 * (k - j);
      k = 1;
      for (i = 1; i < j; i += 2) y2 += jt[i];
    }
  else
    { ac = j * S;
      k = 1 + 2 * (j - k);
      j = 1;
      for (i = 1; i < k; i += 2) b2 += kt[i];
    }
  int ae = J.aepos;
  if (ae > K.aepos) ae = K.aepos;
  while (true)
    { ac += S;
      if (ac >= ae) break;
      y2 += jt[j];
      b2 += kt[k];
      j += 2;
      k += 2;
      i = abs(y2 - b2);
      if (i <= min)
        { min = i;
          if (i == 0) *where = ac;
        }
      den += 1;
    }
  if (J.aepos == K.aepos)
    { i = abs(J.bepos - K.bepos);
      if (i <= min)
        { min = i;
          if (i == 0) *where = K.aepos;
        }
    }
  return (den == 0) ? -1 : min;
}

// map.c:2065-2109: P1 := prefix of P1 up to trace line ap, then P2 from there on
static void fusion(Ovl &P1, int ap, const Ovl &P2, std::vector<uint16_t> &tb, int S)
{ int k1 = 2 * ((ap / S) - (P1.abpos / S));
  int k2 = 2 * ((ap / S) - (P2.abpos / S));
  size_t at = tb.size();
  int diff = 0;
  for (int k = 0; k < k1; k += 2)
    { uint16_t d = tb[P1.trace + k], v = tb[P1.trace + k + 1];
      tb.push_back(d);
      tb.push_back(v);
      diff += d;
    }
  for (int k = k2; k < P2.tlen; k += 2)
    { uint16_t d = tb[P2.trace + k], v = tb[P2.trace + k + 1];
      tb.push_back(d);
      tb.push_back(v);
      diff += d;
    }
  P1.aepos = P2.aepos;
  P1.bepos = P2.bepos;
  P1.diffs = diff;
  P1.trace = at;
  P1.tlen  = (int) (tb.size() - at);
}

static void copy_path(Ovl &dst, const Ovl &src)
{ dst.tlen = src.tlen; dst.diffs = src.diffs; dst.trace = src.trace;
  dst.abpos = src.abpos; dst.bbpos = src.bbpos; dst.aepos = src.aepos; dst.bepos = src.bepos;
}

// map.c:2116-2268.  bm (may be null) is the parallel array of B views (-C): a pair is fused only if
// the B views meet at a trace point as well (Entwine's argument order depends on the orientation).
static int handle_redundancies(Ovl *am, int novls, Ovl *bm, std::vector<uint16_t> &tb, int S, int cm)
{ int awhen = 0, bwhen = 0;
  const bool hasB = (bm != nullptr);
  for (int j = 1; j < novls; j++)
    { Ovl &J = am[j];
      for (int k = j - 1; k >= 0; k--)
        { Ovl &K = am[k];
          if (K.abpos < 0) continue;
          if (J.abpos < K.abpos)
            { if (K.abpos <= J.aepos && K.bbpos <= J.bepos)
                { int dist = entwine(J, K, tb, S, &awhen);
                  if (dist == 0)
                    { if (K.aepos > J.aepos)
                        { if (hasB)
                            { Ovl &JB = bm[j], &KB = bm[k];
                              if (cm)
                                { dist = entwine(KB, JB, tb, S, &bwhen);
                                  if (dist != 0) continue;
                                  fusion(J, awhen, K, tb, S);
                                  fusion(KB, bwhen, JB, tb, S);
                                  copy_path(JB, KB);
                                }
                              else
                                { dist = entwine(JB, KB, tb, S, &bwhen);
                                  if (dist != 0) continue;
                                  fusion(J, awhen, K, tb, S);
                                  fusion(JB, bwhen, KB, tb, S);
                                }
                            }
                          else
                            fusion(J, awhen, K, tb, S);
                        }
                      K.abpos = -1;
                      break;
                    }
                }
            }
          else
            { if (J.abpos <= K.aepos && J.bbpos <= K.bepos)
                { int dist = entwine(K, J, tb, S, &awhen);
                  if (dist == 0)
                    { if (K.abpos == J.abpos)
                        { if (K.aepos > J.aepos)
                            { copy_path(J, K);
                              if (hasB) copy_path(bm[j], bm[k]);
                            }
                        }
                      else if (J.aepos > K.aepos)
                        { if (hasB)
                            { Ovl &JB = bm[j], &KB = bm[k];
                              if (cm)
                                { dist = entwine(JB, KB, tb, S, &bwhen);
                                  if (dist != 0) continue;
                                  fusion(K, awhen, J, tb, S);
                                  copy_path(J, K);
                                  fusion(JB, bwhen, KB, tb, S);
                                }
                              else
                                { dist = entwine(KB, JB, tb, S, &bwhen);
                                  if (dist != 0) continue;
                                  fusion(K, awhen, J, tb, S);
                                  copy_path(J, K);
                                  fusion(KB, bwhen, JB, tb, S);
                                  copy_path(JB, KB);
                                }
                            }
                          else
                            { fusion(K, awhen, J, tb, S);
                              copy_path(J, K);
                            }
                        }
                      else
                        { copy_path(J, K);
                          if (hasB) copy_path(bm[j], bm[k]);
                        }
                      K.abpos = -1;
                      break;
                    }
                }
            }
        }
    }
  int no = 0;
  for (int j = 0; j < novls; j++)
    if (am[j].abpos >= 0)
      { if (hasB) bm[no] = bm[j];
        am[no++] = am[j];
      }
  return no;
}

// map.c:2270-2302
static int special_log(int cover)
{ static double Spow[51];
  static bool   first = true;
  if (first)
    { first = false;
      for (int m = 0; m <= 40; m++) Spow[m] = pow(10., m / 10.);
    }
  if (cover <= 1) return cover;
  if (cover >= 10000) return 40;
  double x = cover;
  int l = 0, r = 41;
  while (l < r)
    { int m = (l + r) >> 1;
      if (Spow[m] <= x) l = m + 1; else r = m;
    }
  return l - 1;
}

static void write_overlap(FILE *f, const Ovl &o, const uint16_t *trace, bool small, std::vector<uint8_t> &buf)
{ buf.resize(40 + 2 * (size_t) o.tlen);
  int32_t *rec = (int32_t *) buf.data();
  rec[0] = o.tlen; rec[1] = o.diffs; rec[2] = o.abpos; rec[3] = o.bbpos; rec[4] = o.aepos; rec[5] = o.bepos;
  rec[6] = (int32_t) o.flags; rec[7] = o.aread; rec[8] = o.bread;
  rec[9] = 0;                                  // struct padding (uninitialised in the reference)
  if (small)
    { uint8_t *t8 = buf.data() + 40;
      for (int j = 0; j < o.tlen; j++)
        { if (trace[j] > 255)               // align.c:3129-3137
            { fprintf(stderr, "damapper: Compression of trace to bytes fails, value too big\n");
              exit(1);
            }
          t8[j] = (uint8_t) trace[j];
        }
      fwrite(buf.data(), 1, 40 + (size_t) o.tlen, f);
    }
  else
    { memcpy(buf.data() + 40, trace, 2 * (size_t) o.tlen);
      fwrite(buf.data(), 1, 40 + 2 * (size_t) o.tlen, f);
    }
}

struct AlnRec
{ int seq, abpos, bbpos, aepos, bepos, diffs, tlen;
  int babpos, bbbpos, baepos, bbepos, btlen;        // B view (-C)
  const uint16_t *trace, *btrace;
};


// Pinned host buffers the alignment pool is downloaded into.  When the whole batch succeeded in one
// launch the report context takes the buffer as it is (no copy on the critical path) and gives it back
// when it is disposed of; acquired and recycled on the main thread only.
struct PinBuf
{ int   *p = nullptr;
  size_t cap = 0;
  std::atomic<bool> busy{false};
};
static std::vector<PinBuf *> g_pins;

static PinBuf *pin_acquire(size_t n)
{ PinBuf *pick = nullptr;
  for (PinBuf *b : g_pins)
    if (!b->busy.load() && (pick == nullptr || b->cap > pick->cap)) pick = b;
  if (pick == nullptr)
    { pick = new PinBuf();
      g_pins.push_back(pick);
    }
  if (pick->cap < n)
    { if (pick->p) (void) hipHostFree(pick->p);
      pick->cap = n + n / 4 + 4096;
      HIP_CHECK(hipHostMalloc((void **) &pick->p, sizeof(int) * pick->cap, hipHostMallocDefault));
    }
  pick->busy.store(true);
  return pick;
}

// Everything the per-read select/write stage needs, detached from the library state so that it
// can run on the report worker thread while the next read block is already being filtered.
struct ReportCtx
{ int S_, NT, nshift, nreads, astart, doA, doB, spacing, profile, verbose;
  double best_tie;
  std::string aname, bname, sort_path;
  std::vector<int> rlen;
  std::vector<ReadState> reads;   // built from raw by the report stage itself (select_and_write)
  CandRaw raw;
  PinBuf *raw_pin = nullptr;      // page-locked memory raw points into
  std::vector<int> job_of_first;
  std::vector<std::vector<AlnRec>> recs;
  std::vector<int> pool;
  PinBuf *pin = nullptr;         // the pool lives here instead when one launch aligned everything
  size_t deferred_got = 0;       // > 0: pool[0..deferred_got) still has to be parsed into recs
  struct Seg { PinBuf *pin; size_t got; std::vector<char> failed; };
  std::vector<Seg> segs;         // several launches: the page-locked copy of each one's records and the jobs whose
                                 // records do not count
  int *pool_base() { return pin ? pin->p : pool.data(); }
  ~ReportCtx()
  { if (pin) pin->busy.store(false);
    if (raw_pin) raw_pin->busy.store(false);
    for (Seg &g : segs) if (g.pin) g.pin->busy.store(false);
  }
};

// Records of the alignment pool: {cand, seq, abpos, bbpos, aepos, bepos, diffs, tlen} + trace
static void parse_pool(int *pool, size_t base, size_t got, int njobs, const std::vector<char> &failed,
                       std::vector<std::vector<AlnRec>> &recs, bool absolute = false)
{ size_t p = base;
  while (p + REC_HDR <= base + got)
    { int *r = pool + p;
      int cand = r[0], tlen = r[7];
      int btl = r[12];
      if (cand < 0 || cand >= njobs || tlen < 0 || btl < 0) break;   // unwritten tail (pool overflow)
      size_t need = REC_HDR + ((size_t) tlen + 1) / 2 + ((size_t) btl + 1) / 2;
      if (p + need > base + got) break;
      if (failed.empty() || !failed[cand])
        { AlnRec a;
          a.seq = r[1]; a.abpos = r[2]; a.bbpos = r[3]; a.aepos = r[4]; a.bepos = r[5];
          a.diffs = r[6]; a.tlen = tlen;
          a.babpos = r[8]; a.bbbpos = r[9]; a.baepos = r[10]; a.bbepos = r[11]; a.btlen = btl;
          if (absolute)
            { a.trace  = (const uint16_t *) (pool + p + REC_HDR);
              a.btrace = (const uint16_t *) (pool + p + REC_HDR + ((size_t) tlen + 1) / 2);
            }
          else
            { a.trace  = (const uint16_t *) (size_t) (p + REC_HDR);      // offsets, fixed up later
              a.btrace = (const uint16_t *) (size_t) (p + REC_HDR + ((size_t) tlen + 1) / 2);
            }
          recs[cand].push_back(a);
        }
      p += need;
    }
}

// turn the trace offsets into pointers once the host pool no longer moves; order by sequence no.
static void fixup_recs(int *pool, std::vector<std::vector<AlnRec>> &recs)
{ for (auto &v : recs)
    { for (auto &a : v)
        { a.trace  = (const uint16_t *) (pool + (size_t) a.trace);
          a.btrace = (const uint16_t *) (pool + (size_t) a.btrace);
        }
      if (v.size() > 1)
        std::sort(v.begin(), v.end(), [](const AlnRec &x, const AlnRec &y) { return x.seq < y.seq; });
    }
}

static void select_and_write(ReportCtx &C)
{ const int S_ = C.S_, NT = C.NT, nreads = C.nreads, astart = C.astart, doA = C.doA, doB = C.doB;
  if (C.deferred_got > 0)
    { parse_pool(C.pool_base(), 0, C.deferred_got, (int) C.recs.size(), std::vector<char>(), C.recs);
      fixup_recs(C.pool_base(), C.recs);
    }
  else if (!C.segs.empty())
    { for (const ReportCtx::Seg &g : C.segs)
        parse_pool(g.pin->p, 0, g.got, (int) C.recs.size(), g.failed, C.recs, true);
      for (auto &v : C.recs)
        if (v.size() > 1)
          std::sort(v.begin(), v.end(), [](const AlnRec &x, const AlnRec &y) { return x.seq < y.seq; });
    }
  HostTimer ht2("report_select_write");
  if (C.raw.valid)
    { cand_build(C.raw, C.reads, nullptr);
      C.raw = CandRaw();
      if (C.raw_pin) { C.raw_pin->busy.store(false); C.raw_pin = nullptr; }
    }
  // ---- 2. per read: redundancy removal, chain linking, selection, output (map.c:2449-2846)
  const bool small = (S_ <= TRACE_XOVR);
  std::vector<std::vector<uint8_t>> prof(C.profile ? nreads : 0);
  (void) special_log(2);                   // fill its table before the threads start

  // one output file per "thread" of the reference: the files cover disjoint read ranges and are independent,
  // so they are written by NT host threads
  auto do_file = [&](int ti) -> int64_t
  { std::vector<Ovl>      am, bm;
    std::vector<uint16_t> tb;
    std::vector<Link>     linker;
    std::vector<Link *>   perm;
    std::vector<Zone>     part;
    std::vector<uint8_t>  wbuf;
    { int abeg = (int) (((int64_t) ti * nreads) >> C.nshift);
      int aend = (ti == NT - 1) ? nreads : (int) (((int64_t) (ti + 1) * nreads) >> C.nshift);
      FILE *afile = nullptr;
      FILE *bfile = nullptr;
      // every file is written under <name>.tmp and renamed once it is complete (record count patched in, closed):
      // a name without .tmp is a whole file, whatever happened to the process (the dispatcher resumes by file)
      std::string aname_, bname_;
      int64_t ahits = 0, bhits = 0;
      if (doB)                                  // map.c:3256-3261
        { bname_ = C.sort_path + "/" + C.bname + "." + C.aname + ".R" + std::to_string(ti + 1) + ".las";
          const std::string fn = bname_ + ".tmp";
          bfile = fopen(fn.c_str(), "w");
          if (bfile == nullptr)
            { fprintf(stderr, "damapper: Cannot open %s for 'w'\n", fn.c_str());
              Clean_Exit(1);
            }
          setvbuf(bfile, nullptr, _IOFBF, 1 << 20);
          fwrite(&bhits, sizeof(int64_t), 1, bfile);
          fwrite(&C.spacing, sizeof(int), 1, bfile);
        }
      if (doA)
        { aname_ = C.sort_path + "/" + C.aname + "." + C.bname + ".M" + std::to_string(ti + 1) + ".las";
          const std::string fn = aname_ + ".tmp";
          afile = fopen(fn.c_str(), "w");
          if (afile == nullptr)
            { fprintf(stderr, "damapper: Cannot open %s for 'w'\n", fn.c_str());
              Clean_Exit(1);
            }
          setvbuf(afile, nullptr, _IOFBF, 1 << 20);
          fwrite(&ahits, sizeof(int64_t), 1, afile);
          fwrite(&C.spacing, sizeof(int), 1, afile);
        }

      for (int ar = abeg; ar < aend; ar++)
        { const int alen = C.rlen[ar];
          const int atck = (alen - 1) / S_ + 1;
          am.clear();
          bm.clear();
          tb.clear();
          int lovl = 0;
          auto &cl = C.reads[ar].cands;
          int job = C.job_of_first[ar];
          for (size_t ci = 0; ci < cl.size(); ci++, job++)
            { Candidate &c = *cl[ci];
              for (const AlnRec &a : C.recs[job])
                { Ovl o;
                  o.tlen = a.tlen; o.diffs = a.diffs;
                  o.abpos = a.abpos; o.bbpos = a.bbpos; o.aepos = a.aepos; o.bepos = a.bepos;
                  o.flags = c.comp ? COMP_FLAG : 0;
                  o.aread = ar + astart;
                  o.bread = c.bread;
                  o.trace = tb.size();
                  tb.insert(tb.end(), a.trace, a.trace + a.tlen);
                  am.push_back(o);
                  if (doB)                       // map.c:2556-2572
                    { Ovl b;
                      b.tlen = a.btlen; b.diffs = a.diffs;
                      b.abpos = a.babpos; b.bbpos = a.bbbpos; b.aepos = a.baepos; b.bepos = a.bbepos;
                      b.flags = o.flags;
                      b.aread = c.bread;
                      b.bread = ar + astart;
                      b.trace = tb.size();
                      tb.insert(tb.end(), a.btrace, a.btrace + a.btlen);
                      bm.push_back(b);
                    }
                }
              bool last = (ci + 1 == cl.size()) || cl[ci + 1]->bread != c.bread || cl[ci + 1]->comp != c.comp;
              if (last)
                { int novl = (int) am.size();
                  if (novl - lovl > 1)
                    { novl = lovl + handle_redundancies(am.data() + lovl, novl - lovl,
                                                        doB ? bm.data() + lovl : nullptr, tb, S_, c.comp);
                      am.resize(novl);
                      if (doB) bm.resize(novl);
                    }
                  if (novl - lovl > 1)
                    { // AMATCH_SORT (map.c:2304-2315): abpos descending, ties by descending original
                      // position (glibc sorts 48-byte records through pointers to the originals)
                      std::vector<std::pair<Ovl, int>> tmp;
                      for (int i = lovl; i < novl; i++) tmp.push_back({am[i], i});
                      std::sort(tmp.begin(), tmp.end(), [](const std::pair<Ovl, int> &x, const std::pair<Ovl, int> &y)
                                { if (x.first.abpos != y.first.abpos) return x.first.abpos > y.first.abpos;
                                  return x.second > y.second;
                                });
                      for (int i = lovl; i < novl; i++) am[i] = tmp[i - lovl].first;
                      if (doB)
                        { // BC_MATCH_SORT / BN_MATCH_SORT (map.c:2317-2341): the B views are sorted on
                          // their own keys -- bepos ascending (comp) or bbpos descending (normal)
                          tmp.clear();
                          for (int i = lovl; i < novl; i++) tmp.push_back({bm[i], i});
                          if (c.comp)
                            std::sort(tmp.begin(), tmp.end(), [](const std::pair<Ovl, int> &x, const std::pair<Ovl, int> &y)
                                      { if (x.first.bepos != y.first.bepos) return x.first.bepos < y.first.bepos;
                                        return x.second > y.second;
                                      });
                          else
                            std::sort(tmp.begin(), tmp.end(), [](const std::pair<Ovl, int> &x, const std::pair<Ovl, int> &y)
                                      { if (x.first.bbpos != y.first.bbpos) return x.first.bbpos > y.first.bbpos;
                                        return x.second > y.second;
                                      });
                          for (int i = lovl; i < novl; i++) bm[i] = tmp[i - lovl].first;
                        }
                    }
                  lovl = novl;
                }
            }

          const int novl = (int) am.size();
          linker.assign(std::max(novl, 1), Link{0, -1, 1});
          perm.assign(std::max(novl, 1), nullptr);
          part.assign(std::max(novl, 1), Zone{0, 0, 0});

          // LA-chain DP (map.c:2630-2710)
          if (novl > 0)
            { lovl = 0;
              linker[0].link = -1;
              linker[0].score = (int) ((am[0].aepos - am[0].abpos) - DIFF_SCORE * am[0].diffs);
              linker[0].mark = 1;
              perm[0] = &linker[0];
              int br = am[0].bread;
              for (int c = 1; c < novl; c++)
                { const Ovl &cp = am[c];
                  linker[c].link = -1;
                  linker[c].score = (int) ((cp.aepos - cp.abpos) - DIFF_SCORE * cp.diffs);
                  linker[c].mark = 1;
                  perm[c] = &linker[c];
                  if (am[c].bread != br)
                    { br = am[c].bread;
                      lovl = c;
                      continue;
                    }
                  int cor = (am[c].flags & COMP_FLAG);
                  for (int d = c - 1; d >= lovl; d--)
                    { int dor = (am[d].flags & COMP_FLAG);
                      if (dor != cor) continue;
                      const Ovl &dp = am[d];
                      if (dor)
                        { if (dp.bepos < cp.bepos) continue; }
                      else
                        { if (dp.bbpos < cp.bbpos) continue; }
                      if (dp.abpos <= cp.aepos - CHAIN_OVL || dp.bbpos <= cp.bepos - CHAIN_OVL)
                        continue;
                      double rat = ((dp.abpos - cp.aepos + CHAIN_OFF) / (dp.bbpos - cp.bepos + CHAIN_OFF));
                      if (1. > rat * CHAIN_PLAY || rat > CHAIN_PLAY)
                        continue;
                      int scr  = (int) (linker[d].score + (cp.aepos - cp.abpos) - DIFF_SCORE * cp.diffs);
                      int scr2 = linker[c].score;
                      if (scr < scr2 - TIE_SCORE)
                        continue;
                      if (scr <= scr2 + TIE_SCORE)
                        { int gap = dp.abpos - cp.aepos, gap2;
                          if (linker[d].link >= 0)
                            { // map.c:2689-2690 indexes amatch[] with linker[c].link, which may be -1
                              // (the reference then reads the record before the array)
                              int lc = linker[c].link;
                              if (lc < 0)
                                { // the reference's value here is whatever precedes its array on the heap, so
                                  // its choice between the two links cannot be reproduced; 0 is used.  Said once.
                                  static std::atomic<bool> told{false};
                                  if (!told.exchange(true))
                                    fprintf(stderr, "damapper_hip: Warning: read %d ties two chain links where map.c:2689-2690 "
                                                    "reads amatch[-1] (undefined in the reference); using 0 -- the reference "
                                                    "may link this read's alignments differently\n", ar + astart);
                                }
                              gap2 = (lc >= 0 ? am[lc].aepos : 0) - dp.abpos;
                            }
                          else
                            gap2 = 0;
                          if (gap > gap2 + TIE_GAP)
                            continue;
                          if (gap >= gap2 - TIE_GAP)
                            { if (scr < scr2) continue;
                              if (scr == scr2)
                                { if (gap >= gap2) continue; }
                            }
                        }
                      linker[c].link = d;
                      linker[c].score = scr;
                      linker[d].mark = 0;
                    }
                }
            }

          // LINK_SORT (map.c:2355-2360): score descending, stable (glibc merge sort on 8-byte items)
          std::stable_sort(perm.begin(), perm.begin() + novl, [](const Link *x, const Link *y) { return x->score > y->score; });

          int nparts = 0;
          for (int c = 0; c < novl && perm[c]->score >= 0; c++)
            if (perm[c]->mark == 1)
              { int b, e, p, n, best;
                b = e = (int) (perm[c] - linker.data());
                for (p = linker[b].link; p >= 0 && linker[p].mark >= 0; p = linker[p].link)
                  e = p;
                for (p = 0; p < nparts; p++)
                  if (am[b].abpos < part[p].end - 100 && am[e].aepos > part[p].beg + 100)
                    break;
                if (p >= nparts)
                  { part[p].beg = am[b].abpos;
                    part[p].end = am[e].aepos;
                    part[p].top = linker[b].score;
                    best = 1;
                    nparts += 1;
                  }
                else
                  { if (linker[b].score < C.best_tie * part[p].top)
                      continue;
                    best = (linker[b].score == part[p].top);
                  }
                int q = -1;
                for (p = b; true; p = n)
                  { linker[p].mark = -1;
                    if (doA)
                      { if (p == b)
                          { am[p].flags |= START_FLAG;
                            if (best) am[p].flags |= BEST_FLAG;
                          }
                        else
                          am[p].flags |= NEXT_FLAG;
                        write_overlap(afile, am[p], tb.data() + am[p].trace, small, wbuf);
                        ahits += 1;
                      }
                    n = linker[p].link;
                    if (doB)                     // map.c:2758-2777
                      { if (bm[p].flags & COMP_FLAG)
                          { linker[p].link = q;  // complemented chains run the other way in B: stack them
                            q = p;
                          }
                        else
                          { if (p == b)
                              { bm[p].flags |= START_FLAG;
                                if (best) bm[p].flags |= BEST_FLAG;
                              }
                            else
                              bm[p].flags |= NEXT_FLAG;
                            write_overlap(bfile, bm[p], tb.data() + bm[p].trace, small, wbuf);
                            bhits += 1;
                          }
                      }
                    if (p == e) break;
                  }
                if (doB && (bm[b].flags & COMP_FLAG))      // map.c:2796-2815
                  { e = b;
                    b = q;
                    for (p = b; true; p = linker[p].link)
                      { if (p == b)
                          { bm[p].flags |= START_FLAG;
                            if (best) bm[p].flags |= BEST_FLAG;
                          }
                        else
                          bm[p].flags |= NEXT_FLAG;
                        write_overlap(bfile, bm[p], tb.data() + bm[p].trace, small, wbuf);
                        bhits += 1;
                        if (p == e) break;
                      }
                  }
              }

          if (C.profile)                           // map.c:2835-2845
            { auto &cv = C.reads[ar].cover;
              prof[ar].resize(atck + 1);
              int c = 0;
              for (int i = 0; i <= atck; i++)
                { c += cv[i];
                  prof[ar][i] = (uint8_t) special_log(c);
                }
            }
        }

      if (doA)
        { rewind(afile);
          fwrite(&ahits, sizeof(int64_t), 1, afile);
          if (fclose(afile) != 0 || rename((aname_ + ".tmp").c_str(), aname_.c_str()) != 0)
            { fprintf(stderr, "damapper: Cannot complete %s\n", aname_.c_str());
              Clean_Exit(1);
            }
        }
      if (doB)
        { rewind(bfile);
          fwrite(&bhits, sizeof(int64_t), 1, bfile);
          if (fclose(bfile) != 0 || rename((bname_ + ".tmp").c_str(), bname_.c_str()) != 0)
            { fprintf(stderr, "damapper: Cannot complete %s\n", bname_.c_str());
              Clean_Exit(1);
            }
        }
      return doA ? ahits : bhits;
    }
  };
  std::vector<int64_t> hits(NT, 0);
  if (NT > 1)
    { std::vector<std::thread> th;
      for (int ti = 0; ti < NT; ti++) th.emplace_back([&, ti] { hits[ti] = do_file(ti); });
      for (auto &t : th) t.join();
    }
  else
    hits[0] = do_file(0);
  int64_t ncheck = 0;
  for (int ti = 0; ti < NT; ti++) ncheck += hits[ti];

  if (C.verbose)
    { printf("      %lld mapped segments\n", (long long) ncheck);
      fflush(stdout);
    }

  if (C.profile)                                   // map.c:3295-3318
    { std::string an = std::string(".") + C.aname + ".prof.anno", dn = std::string(".") + C.aname + ".prof.data";
      FILE *af = fopen(an.c_str(), "w"), *df = fopen(dn.c_str(), "w");
      if (af == nullptr || df == nullptr)
        { fprintf(stderr, "damapper: Cannot open profile track files for %s\n", C.aname.c_str());
          Clean_Exit(1);
        }
      int size = sizeof(int64_t);
      fwrite(&C.nreads, sizeof(int), 1, af);
      fwrite(&size, sizeof(int), 1, af);
      int64_t cnt = 0;
      for (int a = 0; a < nreads; a++)
        { fwrite(&cnt, sizeof(int64_t), 1, af);
          cnt += (C.rlen[a] - 1) / S_ + 2;
          fwrite(prof[a].data(), 1, prof[a].size(), df);
        }
      fwrite(&cnt, sizeof(int64_t), 1, af);
      fclose(af);
      fclose(df);
    }

}


// ---- report worker: one thread, FIFO; synchronous unless dmh_async_report(1) ---------------
static std::thread              *g_worker = nullptr;
static std::mutex                g_rmu;
static std::condition_variable   g_rcv;
static std::deque<ReportCtx *>   g_rqueue;
static int                       g_rbusy = 0;
static int                       g_async = 0;      // 0 synchronous, 1 report stage on a worker, 2 also the alignment kernel deferred
static bool                      g_stop  = false;
static std::vector<ReportCtx *>  g_rdone;
static int                       g_reports_done = 0;   // Reporter calls whose files are complete (dmh_reports_done)

static void dispose_done()
{ std::vector<ReportCtx *> d;
  { std::unique_lock<std::mutex> lk(g_rmu);
    d.swap(g_rdone);
  }
  for (auto *c : d) delete c;
}

static void worker_main()
{ while (true)
    { ReportCtx *C;
      { std::unique_lock<std::mutex> lk(g_rmu);
        g_rcv.wait(lk, [] { return !g_rqueue.empty() || g_stop; });
        if (g_rqueue.empty()) return;
        C = g_rqueue.front();
        g_rqueue.pop_front();
      }
      select_and_write(*C);
      // what this thread allocated (the per-read candidate objects, the parsed records) it frees itself
      { std::vector<ReadState>().swap(C->reads);
        for (auto &v : C->recs) std::vector<AlnRec>().swap(v);
      }
      { // the rest of the context was allocated piece by piece on the main thread: hand it back for disposal
        // there (freeing it here would fight the main thread for its malloc arena)
        std::unique_lock<std::mutex> lk(g_rmu);
        g_rdone.push_back(C);
        g_rbusy -= 1;
        g_reports_done += 1;
      }
      g_rcv.notify_all();
    }
}

void report_wait()
{ { std::unique_lock<std::mutex> lk(g_rmu);
    g_rcv.wait(lk, [] { return g_rbusy == 0; });
  }
  dispose_done();
}

static void report_submit(ReportCtx *C)
{ if (!g_async)
    { select_and_write(*C);
      delete C;
      { std::unique_lock<std::mutex> lk(g_rmu);
        g_reports_done += 1;
      }
      return;
    }
  { std::unique_lock<std::mutex> lk(g_rmu);
    if (g_worker == nullptr)
      { g_worker = new std::thread(worker_main);
        atexit([] { { std::unique_lock<std::mutex> lk2(g_rmu); g_stop = true; }
                    g_rcv.notify_all();
                    if (g_worker && g_worker->joinable()) g_worker->join();
                  });
      }
    g_rqueue.push_back(C);
    g_rbusy += 1;
  }
  g_rcv.notify_all();
}


// ---------------------------------------------------------------------------------------
//  One Reporter call = prepare (host + main stream) -> [launch a rung of the kernel ladder -> collect]* ->
//  hand the records to the report stage.  With dmh_async_report(2) the first launch goes to the
//  alignment stream and Reporter returns; the collection happens at the next Reporter / report_wait, so
//  the (VALU-bound) alignment kernel of block i runs next to the (latency-bound) merge and chain kernels
//  of block i+1.  The HBM-bound index sort of block i+1 is flushed BEFORE the alignment kernel may start
//  and held back while it runs, so the sort kernels never share the GPU.
// ---------------------------------------------------------------------------------------
struct PendingReport
{ ReportCtx *C = nullptr;
  std::vector<CandJob>  jobs;
  DevBuf<uint16_t> jumps;        // the block's jump pool, taken over from the candidate store (freed when collected)
  std::vector<std::vector<AlnRec>> recs;
  std::vector<int> pool;
  PinBuf *pool_pin = nullptr;
  size_t deferred_got = 0;
  std::vector<int> status, todo;
  int    njobs = 0, doB = 0, hithr = 0, tcap = 0, level = 0, round = 0;
  std::vector<ReportCtx::Seg> segs;
  std::vector<PinBuf *> uploads;  // page-locked copies of job / retry lists on their way to the device (freed with P)
  bool   polled = false, collected = false, jobs_up = false;      // report_poll has looked at the first round / everything is in
  size_t pool_cap = 0;
  AlignSpecDev spd;
  const uint64_t *apk = nullptr, *arc = nullptr, *bpk = nullptr;
  const DAZZ_DB *ablk = nullptr, *bblk = nullptr;
  BlockDev hold[2];            // device copies the kernel reads, taken out of the block registry meanwhile
  int      nhold = 0;
  bool     on_aux = false;     // the kernel runs on the alignment stream
  HostTimer *ht = nullptr;
  // deferred mode: the jobs are put together on a thread of their own while the main thread goes on with the next
  // block's filter; whoever looks next (report_poll, finish_report) launches the first rung once they are in
  std::future<void> builder;
  bool     building = false;
  size_t   pool_per_job = 0;   // ints of the record pool per job to start from
  int      pool_div = 0;       // ("pool_div" test option)
};

static PendingReport *g_pending = nullptr;
static hipStream_t    g_astream = nullptr;
static hipEvent_t     g_ev_ab = nullptr, g_ev_done = nullptr;
static DevBuf<CandJob>  d_jobs;
static DevBuf<int>      d_status, d_pool, d_only;

bool report_deferred_mode() { return g_async == 2; }

bool report_in_flight()
{ if (g_pending == nullptr || !g_pending->on_aux) return false;
  hipError_t e = hipEventQuery(g_ev_done);
  if (e == hipErrorNotReady) { (void) hipGetLastError(); return true; }
  return false;
}

// host -> device on the stream the kernel will run on, from a page-locked buffer that lives as long as P: no wait
// (copy_h2d would wait for everything queued on the MAIN stream -- the next block's index sort, or the chain kernels
// the filter has just queued when report_poll gets here)
static void upload_for(PendingReport &P, hipStream_t ks, void *dev, const void *host, size_t bytes)
{ PinBuf *b = pin_acquire(bytes / sizeof(int) + 16);
  memcpy(b->p, host, bytes);
  HIP_CHECK(hipMemcpyAsync(dev, b->p, bytes, hipMemcpyHostToDevice, ks));
  P.uploads.push_back(b);
}

static void launch_round(PendingReport &P)
{ hipStream_t ks = P.on_aux ? g_astream : stream();
  const int njobs = P.njobs;
  if (P.pool_cap > 0xfffffff0ull / 4) fatal("Reporter: alignment pool too large");
  if (!P.jobs_up)
    { d_jobs.ensure(njobs, "candidate jobs");
      if (P.jumps.p == nullptr) P.jumps.ensure(16, "candidate jumps");
      d_status.ensure(njobs, "candidate status");
    }
  d_pool.ensure(P.pool_cap, "alignment pool");
  int nrun = P.todo.empty() ? njobs : (int) P.todo.size();
  const int *only = nullptr;
  if (P.on_aux)
    { // everything queued on the main stream so far (packing, the flushed index sort -- and whatever main-stream kernel
      // may still be using a pool block that d_jobs / d_only have just been given) comes first: the uploads below and
      // the kernel run on the alignment stream
      HIP_CHECK(hipEventRecord(g_ev_ab, stream()));
      HIP_CHECK(hipStreamWaitEvent(ks, g_ev_ab, 0));
    }
  if (!P.jobs_up)
    { upload_for(P, ks, d_jobs.p, P.jobs.data(), sizeof(CandJob) * (size_t) njobs);
      P.jobs_up = true;
    }
  if (!P.todo.empty())
    { d_only.ensure(P.todo.size(), "retry list");
      upload_for(P, ks, d_only.p, P.todo.data(), sizeof(int) * P.todo.size());
      only = d_only.p;
    }
  const int level = P.level, doB = P.doB;
  // level 0: two candidates per wavefront (k_align_cands32); four (k_align_cands16) with the "align_g" test option = 16
  // or DMH_ALIGN_G=16 -- measured slower, see DESIGN.md 3
  static const int env_g = getenv("DMH_ALIGN_G") ? atoi(getenv("DMH_ALIGN_G")) : 32;
  const bool g16 = (level == 0) && (S().opt_align_g > 0 ? S().opt_align_g == 16 : env_g == 16);
  const int per_block = (level == 0) ? (g16 ? 16 : 8) : 4;
  // Four workgroups per CU take every SIMD's registers; a DEFERRED kernel shares the machine with the next block's
  // filter, and how many workgroups it should get depends on what that filter is made of: 448 against a random
  // GRCh38-sized reference (merge-bound: 59.9 ms per block, 65.3 at 704), 576 on configs[1] (34.1 against 35.7 at 448
  // and 34.9 at 704), all 1024 against the repeat-rich reference (262.8 against 270.3 at 448).  So the grid is tuned
  // while the process runs: each candidate gets three read blocks (the first one, a mix of two settings, is not
  // counted), the one with the shortest time between deferred launches is kept.  DMH_DEFER_GRID fixes it.
  static const int fixed_grid = getenv("DMH_DEFER_GRID") ? std::max(64, atoi(getenv("DMH_DEFER_GRID"))) : 0;
  int aux_grid = fixed_grid;
  if (P.on_aux && fixed_grid == 0 && level == 0)
    { static const int cand[4] = {448, 704, 1024, 576};
      static int    phase = 0;            // blocks launched while exploring; >= 12: decided
      static int    best = 448;
      static double best_t = 1e30, acc = 0.0;
      static double last = 0.0;
      const double now = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
      if (phase < 12)
        { const int c = phase / 3, k = phase % 3;
          if (k > 0) acc += now - last;                 // (two periods per candidate: the kernel launched at k = 0 and
          if (k == 0 && c > 0)                          //  k = 1 ran next to the filters that end at k = 1 and k = 2)
            { if (acc < best_t) { best_t = acc; best = cand[c - 1]; }
              acc = 0.0;
            }
          aux_grid = cand[c];
          phase += 1;
        }
      else
        { if (phase == 12)
            { if (acc < best_t) { best_t = acc; best = cand[3]; }
              phase = 13;
              if (getenv("DMH_STATS")) fprintf(stderr, "dmh: deferred alignment grid: %d workgroups\n", best);
            }
          aux_grid = best;
        }
      last = now;
    }
  else if (aux_grid == 0)
    aux_grid = 448;
  int nblocks = std::min((nrun + per_block - 1) / per_block, P.on_aux ? aux_grid : 256 * 4);
  Scratch sc = prepare_scratch(nblocks, P.tcap, level == 2, per_block, level != 0);
  if (P.on_aux)
    { // (prepare_scratch may have grown its buffers under the main stream's scope since the event above)
      HIP_CHECK(hipEventRecord(g_ev_ab, stream()));
      HIP_CHECK(hipStreamWaitEvent(ks, g_ev_ab, 0));
    }
  HIP_CHECK(hipMemsetAsync(ascr().ctl.p, 0, 8 * sizeof(unsigned int), ks));
  HIP_CHECK(hipMemsetAsync(d_pool.p, 0xff, sizeof(int) * P.pool_cap, ks));   // cand = -1 ends the parse
  // "pebble_spill" test option / DMH_PEBBLE_SPILL = 0: a lane group's pass ends at the pebbles its LDS holds (round 3)
  static const int env_spill = getenv("DMH_PEBBLE_SPILL") ? atoi(getenv("DMH_PEBBLE_SPILL")) : 1;
  int4 *spill = (S().opt_pebble_spill >= 0 ? S().opt_pebble_spill : env_spill) != 0 ? sc.cells : nullptr;
  { StageTimer tm(level == 2 ? "align_global_band" : (level == 1 ? "align_wave64" : "align"), 0.0, ks);
    if (level == 0 && g16)
      hipLaunchKernelGGL(k_align_cands16, dim3(nblocks), dim3(256), 0, ks,
                         P.apk, P.arc, P.bpk, (const CandJob *) d_jobs.p, nrun,
                         (const uint16_t *) P.jumps.p, P.hithr, P.spd, sc.traces, sc.tcap, ascr().ctl.p, d_status.p,
                         d_pool.p, ascr().ctl.p + 1, (unsigned int) P.pool_cap, only, spill, SPILLCAP);
    else if (level == 0)
      hipLaunchKernelGGL(k_align_cands32, dim3(nblocks), dim3(256), 0, ks,
                         P.apk, P.arc, P.bpk, (const CandJob *) d_jobs.p, nrun,
                         (const uint16_t *) P.jumps.p, P.hithr, P.spd, sc.traces, sc.tcap, ascr().ctl.p, d_status.p,
                         d_pool.p, ascr().ctl.p + 1, (unsigned int) P.pool_cap, only, spill, SPILLCAP);
    else if (level == 2)
      { if (doB)
          hipLaunchKernelGGL((k_align_cands<GLB_W, false, true>), dim3(nblocks), dim3(256), 0, ks,
                             P.apk, P.arc, P.bpk, (const CandJob *) d_jobs.p, nrun,
                             (const uint16_t *) P.jumps.p, P.hithr, P.spd, sc, ascr().ctl.p, d_status.p, d_pool.p,
                             ascr().ctl.p + 1, (unsigned int) P.pool_cap, only);
        else
          hipLaunchKernelGGL((k_align_cands<GLB_W, false, false>), dim3(nblocks), dim3(256), 0, ks,
                             P.apk, P.arc, P.bpk, (const CandJob *) d_jobs.p, nrun,
                             (const uint16_t *) P.jumps.p, P.hithr, P.spd, sc, ascr().ctl.p, d_status.p, d_pool.p,
                             ascr().ctl.p + 1, (unsigned int) P.pool_cap, only);
      }
    else
      { if (doB)
          hipLaunchKernelGGL((k_align_cands<LDS_W, true, true>), dim3(nblocks), dim3(256), 0, ks,
                             P.apk, P.arc, P.bpk, (const CandJob *) d_jobs.p, nrun,
                             (const uint16_t *) P.jumps.p, P.hithr, P.spd, sc, ascr().ctl.p, d_status.p, d_pool.p,
                             ascr().ctl.p + 1, (unsigned int) P.pool_cap, only);
        else
          hipLaunchKernelGGL((k_align_cands<LDS_W, true, false>), dim3(nblocks), dim3(256), 0, ks,
                             P.apk, P.arc, P.bpk, (const CandJob *) d_jobs.p, nrun,
                             (const uint16_t *) P.jumps.p, P.hithr, P.spd, sc, ascr().ctl.p, d_status.p, d_pool.p,
                             ascr().ctl.p + 1, (unsigned int) P.pool_cap, only);
      }
    HIP_CHECK(hipGetLastError());
  }
  if (P.on_aux) HIP_CHECK(hipEventRecord(g_ev_done, ks));
}

// Waits for the launched rung, fetches its records and decides what has to run again.  True when done.
static bool collect_round(PendingReport &P)
{ hipStream_t ks = P.on_aux ? g_astream : stream();
  const int njobs = P.njobs;
  dispose_done();        // free finished report contexts while the kernel runs
  unsigned int used = 0;
  { HostTimer hw("report_kernel_wait");
    if (P.on_aux)
      { // nothing here may wait for the MAIN stream: the next block's filter is queued on it
        PinBuf *ps = pin_acquire((size_t) njobs + 16);
        HIP_CHECK(hipMemcpyAsync(ps->p, ascr().ctl.p + 1, sizeof(unsigned int), hipMemcpyDeviceToHost, ks));
        HIP_CHECK(hipMemcpyAsync(ps->p + 1, d_status.p, sizeof(int) * njobs, hipMemcpyDeviceToHost, ks));
        HIP_CHECK(hipStreamSynchronize(ks));
        used = (unsigned int) ps->p[0];
        memcpy(P.status.data(), ps->p + 1, sizeof(int) * njobs);
        ps->busy.store(false);
      }
    else
      { copy_d2h(&used, ascr().ctl.p + 1, sizeof(unsigned int));
        copy_d2h(P.status.data(), d_status.p, sizeof(int) * njobs);
        HIP_CHECK(hipStreamSynchronize(stream()));
      }
  }
  HostTimer hp("report_pool_parse");
  size_t got = std::min<size_t>(used, P.pool_cap);
  PinBuf *pb = pin_acquire(got + 16);
  if (got > 0)
    { HIP_CHECK(hipMemcpyAsync(pb->p, d_pool.p, sizeof(int) * got, hipMemcpyDeviceToHost, ks));
      if (P.on_aux)
        HIP_CHECK(hipStreamSynchronize(ks));
      else
        { static hipEvent_t ev_pool = nullptr;
          if (ev_pool == nullptr) HIP_CHECK(hipEventCreateWithFlags(&ev_pool, hipEventDisableTiming));
          HIP_CHECK(hipEventRecord(ev_pool, ks));
          bg_pump(4);        // runs while the records are handed to the report stage
          HIP_CHECK(hipEventSynchronize(ev_pool));
        }
    }

  // which jobs of this run failed, and why
  std::vector<int> ran;
  if (P.todo.empty()) { ran.resize(njobs); for (int i = 0; i < njobs; i++) ran[i] = i; }
  else ran = P.todo;
  std::vector<int> again;
  bool need_pool = false, need_up = false;
  { // why candidates have to run again, per rung of the ladder (dmh_timing_get: "count:align...")
    int64_t nband = 0, npeb = 0, npool = 0;
    for (int j : ran)
      if (P.status[j] == AL_ERR_BAND) nband += 1;
      else if (P.status[j] == AL_ERR_PEBBLES) npeb += 1;
      else if (P.status[j] == 100) npool += 1;
    static const char *cname[3][4] = {
      {"count:align32_candidates", "count:align32_band_over_30", "count:align32_pebbles_over_288", "count:align32_pool_full"},
      {"count:align64_candidates", "count:align64_band_over_ring", "count:align64_pebbles", "count:align64_pool_full"},
      {"count:align_global_candidates", "count:align_global_band", "count:align_global_pebbles", "count:align_global_pool_full"}};
    const int lv = std::min(2, std::max(0, P.level));
    timing_count(cname[lv][0], (int64_t) ran.size());
    if (nband) timing_count(cname[lv][1], nband);
    if (npeb)  timing_count(cname[lv][2], npeb);
    if (npool) timing_count(cname[lv][3], npool);
  }
  for (int j : ran)
    if (P.status[j] != AL_OK)
      { again.push_back(j);
        if (P.status[j] == 100) need_pool = true;
        else if (P.level == 0 && (P.status[j] == AL_ERR_BAND || P.status[j] == AL_ERR_PEBBLES)) need_up = true;
        else if (P.level == 1 && P.status[j] == AL_ERR_BAND) need_up = true;
        else fatal("Reporter: alignment of candidate %d failed (status %d)", j, P.status[j]);
      }
  // parse the records of this run (records of failed jobs are dropped below); when the whole
  // batch succeeded at once the parse is left to the report stage (worker thread)
  if (again.empty() && P.round == 0)
    { P.deferred_got = got;
      P.pool_pin = pb;
      return true;
    }
  std::vector<char> failed(njobs, 0);
  for (int j : again) failed[j] = 1;
  P.segs.push_back(ReportCtx::Seg{pb, got, std::move(failed)});        // parsed by the report stage, on its thread
  if (again.empty()) return true;
  std::sort(again.begin(), again.end());
  P.todo = again;
  if (need_pool) P.pool_cap *= 4;
  else if (need_up) P.level += 1;
  if (P.round == 7) fatal("Reporter: alignment retries exhausted");
  P.round += 1;
  return false;
}

static void finish_report(PendingReport *P)
{ if (P->building)
    { P->builder.get();
      P->building = false;
      if (P->njobs > 0) launch_round(*P);
    }
  if (P->njobs > 0)
    { if (!P->collected)
        while (!collect_round(*P)) launch_round(*P);
    }
  delete P->ht;
  HostTimer ht_tail("report_tail");
  ReportCtx *C = P->C;
  C->recs = std::move(P->recs);
  C->pool = std::move(P->pool);
  C->pin = P->pool_pin;
  C->deferred_got = P->deferred_got;
  C->segs = std::move(P->segs);
  report_submit(C);
  for (int i = 0; i < P->nhold; i++) free_blockdev(P->hold[i]);
  P->jumps.release();
  for (PinBuf *b : P->uploads) b->busy.store(false);      // (every launch that read them has been collected)
  delete P;
}

bool report_adopt_block(const DAZZ_DB *block)
{ PendingReport *P = g_pending;
  if (P == nullptr || !P->on_aux || (block != P->ablk && block != P->bblk) || P->nhold >= 2) return false;
  State &st = S();
  auto it = st.blocks.find(block);
  if (it == st.blocks.end()) return false;
  P->hold[P->nhold++] = it->second;
  st.blocks.erase(it);
  return true;
}

void report_poll()
{ PendingReport *P = g_pending;
  static const bool off = getenv("DMH_NO_REPORT_POLL") != nullptr;
  if (off || P == nullptr || !P->on_aux || P->polled) return;
  if (P->building)
    { if (P->builder.wait_for(std::chrono::seconds(0)) != std::future_status::ready) return;
      P->builder.get();
      P->building = false;
      if (P->njobs > 0) launch_round(*P);
      return;
    }
  if (P->njobs == 0) return;
  const hipError_t e = hipEventQuery(g_ev_done);
  if (e != hipSuccess) { (void) hipGetLastError(); return; }
  P->polled = true;
  HostTimer ht("report_poll");
  if (collect_round(*P)) P->collected = true;
  else launch_round(*P);                        // (collected by the next Reporter / report_wait)
}

void report_finish_pending()
{ if (g_pending == nullptr) return;
  PendingReport *P = g_pending;
  finish_report(P);          // (g_pending stays set while collecting: background stages remain held back)
  g_pending = nullptr;
}

}  // namespace dmh

using namespace dmh;

extern "C" {

#ifdef DMH_ALIGN_PROFILE
void dmh_align_profile(unsigned long long *out, int reset)
{ HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_aprof), sizeof(unsigned long long) * 16));
  if (reset)
    { unsigned long long z[16] = {0};
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_aprof), z, sizeof(z)));
    }
}
#endif

void Reporter(char *aname, DAZZ_DB *ablock, char *bname, DAZZ_DB *bblock, Align_Spec *asettings, int mflag)
{ State &st = S();
  const Params &pm = st.params;
  Spec *sp = (Spec *) asettings;
  if (sp == nullptr || sp->magic != SPEC_MAGIC)
    fatal("Reporter: the Align_Spec was not created by this library's New_Align_Spec");
  const int cslot = st.cand_slot_of(ablock);
  if (cslot < 0 || st.cand_slot[cslot].nreads != ablock->nreads)
    fatal("Reporter: called for a READS block that has not been through Match_Filter");
  CandStore &cstore = st.cand_slot[cslot];
  // mode 2: the (VALU-bound) alignment kernel of this block is left running, at a partial grid, next to the filter
  // of the NEXT block, whose kernels mostly wait for memory.  (Round 1 deferred only for one-block references: the
  // per-pass host round trips of a many-block filter made the overlap a loss; with the passes batched it wins
  // there too -- 69.3 -> 65.7 ms per read block against a GRCh38-sized random reference.  DMH_DEFER_MAX_PASSES=2
  // restores the old rule.)
  // what is left of this block's filter (the batched passes) comes first: it is the work that runs next to the
  // previous block's alignment kernel, which is only collected further down
  static const int defer_max = getenv("DMH_DEFER_MAX_PASSES") ? atoi(getenv("DMH_DEFER_MAX_PASSES")) : 1 << 30;
  const bool defer = (g_async == 2) && st.mf_calls <= defer_max;
  st.mf_calls_block = st.mf_calls;
  st.mf_calls = 0;
  // the index sort of the NEXT block: all of it now (deferred mode) -- ahead of this block's alignment kernel, and on
  // the main stream while the finish stream drains this block's last batches (a dozen small kernels and two round
  // trips each, which leave the machine nearly empty) -- or a few stages for the GPU to do while the jobs are put together
  // (not ahead of a batch of sparse passes that is still open: its sort and chain kernels are this block's work)
  const bool index_first = defer && !filter_batch().open;
  { HostTimer hb("report_bg_flush");
    if (index_first && !bg_on_tail()) bg_flush();      // (a build on the tail stream goes on stage by stage with the passes)
  }
  { HostTimer hf("report_flush");
    st.tfilt += filter_flush_block(cslot);     // (this block's batches: the next block's may be in flight behind them)
  }
  if (defer && !index_first && !bg_on_tail()) bg_flush();
  // the candidate lists: copied on the finish stream (the main stream is busy with the index sort).  Deferred mode:
  // nobody waits for the copy here -- the jobs of the alignment kernel are put together on a thread of their own
  // (2.4 ms per block of 25,000 reads, plus the copy), which starts by waiting for it; the main thread goes back to
  // the filter of the next block, and the first report_poll that finds the jobs ready launches the kernel
  // (DMH_SYNC_REPORT_JOBS: everything here, as before)
  static const bool sync_jobs = getenv("DMH_SYNC_REPORT_JOBS") != nullptr;
  // (not for a block of a few passes: there the alignment kernel is the longest thing on the GPU and has to start now,
  //  not at the next poll)
  const bool async_build = defer && !sync_jobs && st.mf_calls_block > 6;
  PendingReport *P = new PendingReport();
  hipEvent_t ev_raw = nullptr;
  if (async_build) HIP_CHECK(hipEventCreateWithFlags(&ev_raw, hipEventDisableTiming));
  CandRaw raw;
  PinBuf *raw_pin = nullptr;
  { HostTimer hm("report_mirror");
    StreamScope on_finish(index_first ? finish_stream() : main_stream());
    const size_t nb = cand_fetch_bytes(cslot);
    raw_pin = pin_acquire(nb / sizeof(int) + 64);
    cand_fetch(raw, raw_pin->p, sizeof(int) * raw_pin->cap, cslot, ev_raw);      // (their Jump lists stay in HBM)
  }
  { HostTimer hq("report_previous");
    report_finish_pending();                   // the previous block's alignments (deferred mode)
  }
  if (!defer) bg_pump(8);
  const int doA = (mflag & FLAG_DOA) != 0;
  const int doB = (mflag & FLAG_DOB) != 0;
  const int S_  = SPACING;
  const int nreads = ablock->nreads;

  ablock->maxlen += SPACING;                  // map.c:3246

  P->ht = new HostTimer("report_align");
  P->doB   = doB;
  P->hithr = 3 * pm.Kmer;
  // kernel ladder: 0 = two candidates per wavefront (register band <= 30 diagonals, LDS pebbles),
  // 1 = one per wavefront (register band <= 62, LDS ring 128, global pebbles), 2 = global ring 8192
  static const bool no_pack = getenv("DMH_NO_PACK32") != nullptr;
  P->level = (doB || no_pack) ? 1 : 0;
  if (st.opt_align_level >= 0) P->level = std::max(P->level, std::min(2, st.opt_align_level));

  ReportCtx *C = new ReportCtx();
  C->S_ = S_; C->NT = pm.NTHREADS; C->nshift = pm.NSHIFT; C->nreads = nreads; C->astart = ablock->tfirst;
  C->doA = doA; C->doB = doB;
  C->spacing = SPACING; C->profile = PROFILE; C->verbose = VERBOSE; C->best_tie = BEST_TIE;
  C->aname = aname; C->bname = bname; C->sort_path = SORT_PATH;
  C->rlen.resize(nreads);
  int maxalen = 0;
  for (int i = 0; i < nreads; i++)
    { C->rlen[i] = ablock->reads[i].rlen;
      maxalen = std::max(maxalen, ablock->reads[i].rlen);
    }
  const size_t npool = raw.npool;              // candidates, tombstones included: at least as many as there will be jobs
  C->raw = std::move(raw);
  C->raw_pin = raw_pin;
  if (st.cur_reads == ablock)
    { st.reads.clear();
      st.cur_reads = nullptr;
    }
  cstore.block = nullptr;                  // (the store is free for the block after next)
  cstore.nreads = 0;
  cstore.mirror_valid = false;
  P->jumps = cstore.jumps;                 // the kernel reads the Jump lists where the filter stored them
  cstore.jumps = DevBuf<uint16_t>();
  cstore.jtop = 0;
  P->C = C;
  P->pool_per_job = 2 * (size_t) (REC_HDR + (doB ? 3 : 1) * (maxalen / S_ + 8));
  P->pool_div = st.opt_pool_div;

  // ---- 1. all candidates of the block become one batch of chain jobs for the d-wave kernel: per read newest first,
  //         tombstones left out -- the order in which cand_build numbers the Candidate objects of the report stage.
  //         (What it reads of the two blocks is copied first: the thread may outlive this call.)
  struct Where { int64_t boff; int rlen; };
  std::vector<Where> aw(nreads), bw(bblock->nreads);
  for (int i = 0; i < nreads; i++) aw[i] = Where{ablock->reads[i].boff, ablock->reads[i].rlen};
  for (int i = 0; i < bblock->nreads; i++) bw[i] = Where{bblock->reads[i].boff, bblock->reads[i].rlen};
  auto build = [P, C, ev_raw, nreads](const std::vector<Where> &aw, const std::vector<Where> &bw)
    { if (ev_raw != nullptr)
        { HIP_CHECK(hipEventSynchronize(ev_raw));
          HIP_CHECK(hipEventDestroy(ev_raw));
        }
      HostTimer hj("report_jobs");
      const CandRaw &raw = C->raw;
      std::vector<int> job_of_first(nreads + 1, 0);   // first job index of each read
      for (int ar = 0; ar < nreads; ar++)
        { job_of_first[ar] = (int) P->jobs.size();
          if (raw.npool == 0) continue;
          for (uint32_t i = raw.len[ar]; i-- > 0;)
            { const DCand &c = raw.pool[raw.off[ar] + i];
              if (c.flags & DC_TOMB) continue;
              CandJob J;
              J.aoff  = aw[ar].boff;
              J.alen  = aw[ar].rlen;
              J.boff  = bw[c.bread].boff;
              J.blen  = bw[c.bread].rlen;
              J.comp  = (c.flags & DC_COMP) ? 1 : 0;
              J.alast = c.alast;
              J.blast = c.blast;
              J.npts  = (int) c.jlen;
              J.joff  = c.joff;
              J.pad   = 0;
              P->jobs.push_back(J);
            }
        }
      job_of_first[nreads] = (int) P->jobs.size();
      const int njobs = (int) P->jobs.size();
      P->njobs = njobs;
      P->recs.resize(njobs);
      C->job_of_first = std::move(job_of_first);
      if (njobs == 0) return;
      // pool: one record per accepted alignment; start from 2 per candidate and grow on demand
      P->pool_cap = (size_t) njobs * P->pool_per_job + 4096;
      if (P->pool_div > 1) P->pool_cap = P->pool_cap / P->pool_div + 64;
      P->status.resize(njobs);
      if (P->level == 0)
        { // the halves of a wavefront run in lock step: hand out like jobs together (same orientation,
          // similar length, equal chain end = equal first seed), long ones first:
          // (comp asc, alen desc, alast desc, index asc) packed into one integer key per job
          const std::vector<CandJob> &jobs = P->jobs;
          std::vector<int> &todo = P->todo;
          std::vector<uint64_t> key(njobs);
          for (int i = 0; i < njobs; i++)
            { uint64_t len = 0xfffffu - (uint64_t) std::min(jobs[i].alen, 0xfffff);
              uint64_t end = 0xfffffu - (uint64_t) std::min(std::max(jobs[i].alast, 0), 0xfffff);
              key[i] = ((uint64_t) (jobs[i].comp != 0) << 63) | (len << 43) | (end << 23);
            }
          if (njobs < (1 << 23))
            { for (int i = 0; i < njobs; i++) key[i] |= (uint64_t) i;
              std::sort(key.begin(), key.end());
              todo.resize(njobs);
              for (int i = 0; i < njobs; i++) todo[i] = (int) (key[i] & 0x7fffffu);
            }
          else
            { todo.resize(njobs);
              for (int i = 0; i < njobs; i++) todo[i] = i;
              std::stable_sort(todo.begin(), todo.end(), [&](int x, int y) { return key[x] < key[y]; });
            }
        }
    };

  if (npool > 0)
    { // what the kernel reads of the two blocks (main stream; launch_round orders the alignment stream behind it)
      BlockDev &ad = upload_block(ablock, false);
      BlockDev &bd = upload_block(bblock, false);
      ensure_packed(ad, true);
      ensure_packed(bd, false);
      P->apk = ad.d_pk(); P->arc = ad.d_rc(); P->bpk = bd.d_pk();
      P->spd = spec_dev(sp);
      P->tcap = trace_cap(maxalen, S_);
      if (defer)
        { if (g_astream == nullptr)
            { // (DMH_ASTREAM_PRIO=1: the deferred alignment kernel on a low-priority stream -- experiments)
              static const int prio = getenv("DMH_ASTREAM_PRIO") ? atoi(getenv("DMH_ASTREAM_PRIO")) : 0;
              if (prio != 0)
                { int least = 0, greatest = 0;
                  HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
                  HIP_CHECK(hipStreamCreateWithPriority(&g_astream, hipStreamNonBlocking, prio > 0 ? least : greatest));
                }
              else
                HIP_CHECK(hipStreamCreateWithFlags(&g_astream, hipStreamNonBlocking));
              HIP_CHECK(hipEventCreateWithFlags(&g_ev_ab, hipEventDisableTiming));
              HIP_CHECK(hipEventCreateWithFlags(&g_ev_done, hipEventDisableTiming));
            }
          P->on_aux = true;
          P->ablk = ablock;
          P->bblk = bblock;
          // the kernel outlives this call: the device copies it reads leave the registry, so that neither
          // the clean-up below nor a re-used host block can free them under it
          for (const DAZZ_DB *blk : {(const DAZZ_DB *) ablock, (const DAZZ_DB *) bblock})
            { auto it = st.blocks.find(blk);
              if (it != st.blocks.end() && !it->second.pinned)
                { P->hold[P->nhold++] = it->second;
                  st.blocks.erase(it);
                }
            }
        }
    }
  if (async_build && npool > 0)
    { P->building = true;
      P->builder = std::async(std::launch::async, build, std::move(aw), std::move(bw));
    }
  else
    { build(aw, bw);
      if (P->njobs > 0) launch_round(*P);
    }
  const int njobs = P->building ? (int) npool : P->njobs;

  // the READS index is not needed any more (the alignment kernel reads the packed bases only); the
  // driver free()s the stub
  for (auto it = st.indices.begin(); it != st.indices.end();)
    { if (it->second.block == ablock && !st.retain_ref)
        { dev_free(it->second.keys); dev_free(it->second.vals);
          dev_free(it->second.d_boff); dev_free(it->second.d_chunk);
          it = st.indices.erase(it);
        }
      else
        ++it;
    }

  if (defer && njobs > 0)
    { g_pending = P;           // collected by the next Reporter / dmh_report_wait
      return;
    }
  finish_report(P);
  if (st.blocks.count(ablock) && !st.blocks[ablock].pinned) drop_block(ablock);
  if (st.blocks.count(bblock) && !st.blocks[bblock].pinned) drop_block(bblock);
}

void dmh_async_report(int on)
{ report_finish_pending();
  report_wait();
  g_async = (on < 0) ? 0 : (on > 2 ? 2 : on);
}

void dmh_report_wait(void)
{ report_finish_pending();
  report_wait();
}

int dmh_reports_done(void)
{ std::unique_lock<std::mutex> lk(g_rmu);
  return g_reports_done;
}

int dmh_local_alignment(Align_Spec *spec, const uint8_t *abases, int64_t abytes, const uint8_t *bbases,
                        int64_t bbytes, int njobs, const int64_t *aoff, const int *alen, const int64_t *boff,
                        const int *blen, const int *comp, const int *dg, const int *ad, int *out,
                        uint16_t *trace_out, uint16_t *btrace_out, int trace_stride)
{ Spec *sp = (Spec *) spec;
  if (sp == nullptr || sp->magic != SPEC_MAGIC) fatal("dmh_local_alignment: bad Align_Spec");
  if (njobs <= 0) return 0;
  AlignSpecDev spd = spec_dev(sp);
  uint8_t *da = (uint8_t *) dev_alloc(abytes + 256, "A bases");
  uint8_t *db = (uint8_t *) dev_alloc(bbytes + 256, "B bases");
  copy_h2d(da, abases, abytes);
  copy_h2d(db, bbases, bbytes);
  // 2-bit packed copies; read boundaries of A (for the reverse complements) are its separators
  std::vector<int64_t> hboff;
  for (int64_t q = 0; q < abytes; q++)
    if (abases[q] > 3) hboff.push_back(q + 1);
  if (hboff.size() < 2) { hboff.clear(); hboff.push_back(0); hboff.push_back(abytes + 1); }
  int      nar = (int) hboff.size() - 1;
  int64_t *d_hboff = (int64_t *) dev_alloc(sizeof(int64_t) * hboff.size(), "A boundaries");
  copy_h2d(d_hboff, hboff.data(), sizeof(int64_t) * hboff.size());
  uint64_t *apkraw = (uint64_t *) dev_alloc(sizeof(uint64_t) * ((abytes >> 5) + 8), "packed A");
  uint64_t *arcraw = (uint64_t *) dev_alloc(sizeof(uint64_t) * ((abytes >> 5) + 8), "packed A rc");
  uint64_t *bpkraw = (uint64_t *) dev_alloc(sizeof(uint64_t) * ((bbytes >> 5) + 8), "packed B");
  pack_bases(da, abytes, apkraw);
  pack_revcomp(da, abytes, d_hboff, nar, arcraw);
  pack_bases(db, bbytes, bpkraw);
  std::vector<SeedJob> jobs(njobs);
  int maxalen = 1;
  for (int i = 0; i < njobs; i++)
    { jobs[i] = SeedJob{aoff[i], boff[i], alen[i], blen[i], comp[i], dg[i], ad[i], 0};
      maxalen = std::max(maxalen, alen[i]);
    }
  SeedJob *dj = (SeedJob *) dev_alloc(sizeof(SeedJob) * njobs, "seed jobs");
  copy_h2d(dj, jobs.data(), sizeof(SeedJob) * njobs);
  int      *dout = (int *) dev_alloc(sizeof(int) * 16 * njobs, "seed results");
  uint16_t *dat = (uint16_t *) dev_alloc(sizeof(uint16_t) * (size_t) njobs * trace_stride, "seed traces");
  uint16_t *dbt = (uint16_t *) dev_alloc(sizeof(uint16_t) * (size_t) njobs * trace_stride, "seed traces");
  int tcap = trace_cap(maxalen, sp->trace_space);

  std::vector<int> res(16 * (size_t) njobs);
  std::vector<int> todo;
  int *d_only = nullptr;
  bool listed = false;          // the first wavefront-per-job pass runs a list (what the lane-group pass left over)
  const int optg = S().opt_align_g;
  if (optg == 16 || optg == 32)
    { // A view on 16- or 32-lane groups first; band / pebble overflows go on to the passes below
      const int per_block = 256 / optg;
      int nblocks = std::min((njobs + per_block - 1) / per_block, 256 * (optg == 16 ? 3 : 4));
      Scratch sc = prepare_scratch(nblocks, tcap, false, per_block, false);
      HIP_CHECK(hipMemsetAsync(ascr().ctl.p, 0, 8 * sizeof(unsigned int), stream()));
      HIP_CHECK(hipMemsetAsync(dat, 0, sizeof(uint16_t) * (size_t) njobs * trace_stride, stream()));
      HIP_CHECK(hipMemsetAsync(dbt, 0, sizeof(uint16_t) * (size_t) njobs * trace_stride, stream()));
      { StageTimer tm("align_seeds_groups", 0.0);
        if (optg == 16)
          hipLaunchKernelGGL((k_align_seeds_g<16, PEB16>), dim3(nblocks), dim3(256), 0, stream(),
                             (const uint64_t *) (apkraw + 2), (const uint64_t *) (arcraw + 2), (const uint64_t *) (bpkraw + 2),
                             (const SeedJob *) dj, njobs, spd, sc.traces, sc.tcap, ascr().ctl.p, dout, dat, trace_stride);
        else
          hipLaunchKernelGGL((k_align_seeds_g<32, PEB32>), dim3(nblocks), dim3(256), 0, stream(),
                             (const uint64_t *) (apkraw + 2), (const uint64_t *) (arcraw + 2), (const uint64_t *) (bpkraw + 2),
                             (const SeedJob *) dj, njobs, spd, sc.traces, sc.tcap, ascr().ctl.p, dout, dat, trace_stride);
        HIP_CHECK(hipGetLastError());
      }
      copy_d2h(res.data(), dout, sizeof(int) * 16 * njobs);
      HIP_CHECK(hipStreamSynchronize(stream()));
      for (int i = 0; i < njobs; i++)
        if (res[16 * (size_t) i + 7] == AL_ERR_BAND || res[16 * (size_t) i + 7] == AL_ERR_PEBBLES) todo.push_back(i);
      listed = true;
    }
  for (int pass = 0; pass < 2; pass++)
    { bool global_band = (pass == 1);
      int  nrun = (global_band || listed) ? (int) todo.size() : njobs;
      if (nrun == 0) break;
      int nblocks = std::min((nrun + 3) / 4, 256 * 4);
      Scratch sc = prepare_scratch(nblocks, tcap, global_band);
      HIP_CHECK(hipMemsetAsync(ascr().ctl.p, 0, 8 * sizeof(unsigned int), stream()));
      if (global_band || listed)
        { if (d_only) dev_free(d_only);
          d_only = (int *) dev_alloc(sizeof(int) * nrun, "retry list");
          copy_h2d(d_only, todo.data(), sizeof(int) * nrun);
        }
      { StageTimer tm(global_band ? "align_seeds_global_band" : "align_seeds", 0.0);
        if (global_band)
          hipLaunchKernelGGL((k_align_seeds<GLB_W, false>), dim3(nblocks), dim3(256), 0, stream(),
                             (const uint64_t *) (apkraw + 2), (const uint64_t *) (arcraw + 2), (const uint64_t *) (bpkraw + 2),
                             (const SeedJob *) dj, nrun, spd, sc, ascr().ctl.p, dout, dat, dbt, trace_stride,
                             (const int *) d_only);
        else
          hipLaunchKernelGGL((k_align_seeds<LDS_W, true>), dim3(nblocks), dim3(256), 0, stream(),
                             (const uint64_t *) (apkraw + 2), (const uint64_t *) (arcraw + 2), (const uint64_t *) (bpkraw + 2),
                             (const SeedJob *) dj, nrun, spd, sc, ascr().ctl.p, dout, dat, dbt, trace_stride,
                             listed ? (const int *) d_only : (const int *) nullptr);
        HIP_CHECK(hipGetLastError());
      }
      copy_d2h(res.data(), dout, sizeof(int) * 16 * njobs);
      HIP_CHECK(hipStreamSynchronize(stream()));
      todo.clear();
      listed = false;
      for (int i = 0; i < njobs; i++)
        if (res[16 * (size_t) i + 7] == AL_ERR_BAND) todo.push_back(i);
    }
  for (int i = 0; i < njobs; i++)
    for (int f = 0; f < 16; f++) out[16 * (size_t) i + f] = res[16 * (size_t) i + f];
  copy_d2h(trace_out, dat, sizeof(uint16_t) * (size_t) njobs * trace_stride);
  if (btrace_out != nullptr)
    copy_d2h(btrace_out, dbt, sizeof(uint16_t) * (size_t) njobs * trace_stride);
  HIP_CHECK(hipStreamSynchronize(stream()));
  dev_free(da); dev_free(db); dev_free(dj); dev_free(dout); dev_free(dat); dev_free(dbt);
  dev_free(d_hboff); dev_free(apkraw); dev_free(arcraw); dev_free(bpkraw);
  if (d_only) dev_free(d_only);
  return 0;
}

}  // extern "C"
