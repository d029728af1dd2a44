// dmh_chain.hip -- sparse chain scoring of the sorted seed hits (map.c:1463-1922).
//
// The reference keeps the live k-mer hits of one (read, contig) group in a splay tree keyed by
// (diagonal, apos) and walks the hits sequentially.  Only three things about that tree are
// observable: which hits are live, the nearest qualifying neighbour on either
// side in key order, and the order in which chain ends are reported.  Here one wavefront owns a
// group: the live hits are a window of the sorted list (hits expire in insertion order, the
// window is [qhead, j)), the two neighbour queries are wavefront min/max reductions over that
// window, and the scalar chain bookkeeping (cost / from / origin / best end) runs uniformly on
// all lanes.  Groups are handed out through an atomic ticket so that skewed group sizes do not
// leave wavefronts idle.
//
// The per-read candidate list with its domination rule (map.c:1668-1766) is host code below: it
// is sequential per read and touches a handful of records per read.

#include "dmh_state.h"
#include "dmh_device.h"
#include "dmh_chain.h"
#include <algorithm>
#include <sched.h>

namespace dmh {

struct ChainArgs
{ const uint64_t *ab;          // the sorted hits, one word each: apos in the low half, bpos in the high half (0-based)
  const uint32_t *gstart;
  uint32_t        ngroups;
  uint64_t        pmask;
  int             K, hithr;
  int32_t        *cost;        // per hit: chain score ending here
  uint32_t       *from;        //          predecessor (NIL for an origin)
  uint32_t       *rootof;      //          origin of the hit's chain tree (itself for an origin)
  uint32_t       *best;        // per origin: current best end of its tree (map.c: orig->orig)
  int32_t        *bestcost;    //             and that end's score
  uint32_t       *state;       // ST_*
  uint32_t       *tmp;         // reporting scratch
  uint4          *live;        // live-set entries {hit, apos, bpos, cost}
  unsigned int   *ticket;
  uint32_t        size_lo, size_hi;   // this launch takes the groups with size_lo <= hits < size_hi
  const uint32_t *only;               // if set: the launch walks only[0 .. *nonly) instead of all groups
  const uint32_t *nonly;
  uint32_t        nhits;              // hits in keys / vals
  uint32_t       *retry;              // G < 64: groups whose live window outgrew LDS are appended here
  unsigned int   *nretry;             //         (the 64-lane launch that follows walks this list)
  ChainSummary   *out;
  uint32_t       *gcount;             // per group: number of summaries it reported and the slot of the first one in `out`
  uint32_t       *gbase;              //   (k_order_summaries puts them into the reference's reporting order)
  const uint32_t *glast;              // cluster mode: per group the largest apos + 1 of its (aread, bread) pair; else nullptr
  uint32_t       *gdone;              // per group: 1 = chained by k_chain_simple (k_chain_lane leaves it alone); may be nullptr
  uint32_t       *wide;               // k_chain_lane: groups of walk_max hits or more that k_chain_simple has left are not
  unsigned int   *nwide;              //   walked by ONE lane but appended here (the wavefront-per-group launch's list)
  uint32_t        walk_max;
  uint32_t        pile_cap;           // k_chain_pile: entries of LDS it may use (<= PILE_CAP; smaller in tests)
};

// Reductions over the G lanes of a group (G = 16, 32 or 64; groups are aligned).  The 16 lanes of a DPP row
// are reduced with four data-parallel-primitive moves (lane ^ 1, lane ^ 2, mirror within 8, mirror within 16)
// that cost one VALU operation each -- a __shfl_xor goes through the LDS crossbar (ds_bpermute) and a
// 64-bit butterfly of them was a third of the time of a hit.  64-bit keys are reduced in two 32-bit
// rounds: the high words first, then the low words of the lanes that hold the winning high word.
template <bool MAX>
__device__ __forceinline__ uint32_t row16_reduce_u32(uint32_t v)
{ auto op = [](uint32_t a, uint32_t b) -> uint32_t { return MAX ? (a > b ? a : b) : (a < b ? a : b); };
  v = op(v, (uint32_t) __builtin_amdgcn_update_dpp((int) v, (int) v, 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]
  v = op(v, (uint32_t) __builtin_amdgcn_update_dpp((int) v, (int) v, 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]
  v = op(v, (uint32_t) __builtin_amdgcn_update_dpp((int) v, (int) v, 0x141, 0xf, 0xf, false));    // row_half_mirror
  v = op(v, (uint32_t) __builtin_amdgcn_update_dpp((int) v, (int) v, 0x140, 0xf, 0xf, false));    // row_mirror
  return v;
}

template <int G, bool MAX>
__device__ __forceinline__ uint32_t group_reduce_u32(uint32_t v)
{ auto op = [](uint32_t a, uint32_t b) -> uint32_t { return MAX ? (a > b ? a : b) : (a < b ? a : b); };
  v = row16_reduce_u32<MAX>(v);
  if (G == 32)
    v = op(v, (uint32_t) __shfl_xor((int) v, 16));
  else if (G == 64)
    { uint32_t a = (uint32_t) __builtin_amdgcn_readlane((int) v, 0),  b = (uint32_t) __builtin_amdgcn_readlane((int) v, 16);
      uint32_t c = (uint32_t) __builtin_amdgcn_readlane((int) v, 32), d = (uint32_t) __builtin_amdgcn_readlane((int) v, 48);
      v = op(op(a, b), op(c, d));
    }
  return v;
}

template <int G>
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v)
{ const uint32_t hi = (uint32_t) (v >> 32);
  const uint32_t mh = group_reduce_u32<G, false>(hi);
  const uint32_t ml = group_reduce_u32<G, false>(hi == mh ? (uint32_t) v : 0xffffffffu);
  return ((uint64_t) mh << 32) | ml;
}

template <int G>
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v)
{ const uint32_t hi = (uint32_t) (v >> 32);
  const uint32_t mh = group_reduce_u32<G, true>(hi);
  const uint32_t ml = group_reduce_u32<G, true>(hi == mh ? (uint32_t) v : 0u);
  return ((uint64_t) mh << 32) | ml;
}

// G lanes per group: the live window of a 10 kb read holds a few dozen hits, so a whole wavefront
// per group leaves most lanes idle and -- the walk over the hits being one long dependent chain of
// memory round trips -- the kernel is bound by how many groups are in flight, not by issue slots.
// 64/G groups share a wavefront's instruction stream.  Groups with thousands of hits (repeats: the
// live window itself holds thousands) want all 64 lanes on their scan instead, so the host launches
// the kernel twice: G = 16 over the small groups, G = 64 over the big ones.
template <int G, int LCAP = (G < 64 ? 96 : 0), int BLOCK = 256>
__global__ __launch_bounds__(BLOCK) void k_chain(ChainArgs A)
{ const int      lane  = threadIdx.x & (G - 1);                 // lane within the group
  const int      sbase = (threadIdx.x & 63) & ~(G - 1);         // first lane of the group in its wavefront
  const uint64_t gmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
  const uint64_t lt    = (1ull << lane) - 1ull;
  const int      K     = A.K;
  auto gballot = [&](bool p) -> uint64_t { return (__ballot(p) >> sbase) & gmask; };

  // G < 64: the live window sits in LDS together with each entry's chain origin, so a hit costs one
  // LDS scan and ONE global round trip (the origin's best score) instead of three (window, origin,
  // best score).  A window that outgrows LCAP entries hands its group to the 64-lane launch.
  constexpr bool LDSLIVE = (LCAP > 0);
  __shared__ uint4    s_live[LDSLIVE ? (BLOCK / G) * LCAP : 1];
  __shared__ uint32_t s_root[LDSLIVE ? (BLOCK / G) * LCAP : 1];
  uint4    *const lds_live = s_live + (LDSLIVE ? (threadIdx.x / G) * LCAP : 0);
  uint32_t *const lds_root = s_root + (LDSLIVE ? (threadIdx.x / G) * LCAP : 0);

  const uint32_t nwork = A.only ? *A.nonly : A.ngroups;
  while (true)
    { // one ticket per wavefront covers its 64/G groups: a device-scope atomic on one address costs
      // several ns of memory-side service time, which adds up over 10^5 groups
      uint32_t g = 0;
      if ((threadIdx.x & 63) == 0) g = atomicAdd(A.ticket, (uint32_t) (64 / G));
      g = __shfl(g, 0) + (uint32_t) (sbase / G);
      if (g >= nwork) break;
      if (A.only) g = A.only[g];

      const uint32_t g0 = A.gstart[g], g1 = A.gstart[g + 1];
      if (!A.only && (g1 - g0 < A.size_lo || g1 - g0 >= A.size_hi)) continue;      // the other launch's group
      // The live set (hits in the reference's tree) is an unordered list of self-contained entries
      // {hit, apos, bpos, cost} in this group's slice of A.live.  One coalesced scan per hit drops
      // expired / removed entries (compacting in place) and answers both neighbour queries, so a
      // hit costs O(live/64) even when thousands of dead hits share its 1000-bp window.
      uint4   *live  = A.live + g0;
      uint32_t nlive = 0;
      bool     spill = false;

      for (uint32_t j = g0; j < g1; j++)
        { const int apos = (int) (uint32_t) A.ab[j] + 1;              // map.c:1784-1785
          const int bpos = (int) (A.ab[j] >> 32) + 1;
          const int diag = apos - bpos;

          // ---- expiry (map.c:1787-1796) + neighbour queries (predOf/leftmost/succOf, map.c:1262-1318)
          // Every lane keeps the best candidate of ITS entries over all chunks of the window; the two
          // group-wide reductions happen once, after the scan (a repeat pile's window has thousands of
          // entries = dozens of chunks: reducing per chunk was most of the big groups' time).  The keys
          // are unique per entry, so "best of the lanes' bests" picks the same entry.
          uint64_t myL = ~0ull, myR = 0;
          uint4    myLe = make_uint4(NIL, 0, 0, 0), myRe = make_uint4(NIL, 0, 0, 0);
          uint32_t myLpos = 0, myRpos = 0, myLroot = 0, myRroot = 0;
          const int thresh = bpos - MAX_GAP;
          uint32_t  wr = 0;
          for (uint32_t base = 0; base < nlive; base += G)
            { uint32_t i  = base + lane;
              uint4    en = make_uint4(NIL, 0, 0, 0);
              uint32_t rt = 0;
              bool     keep = false;
              uint64_t lk = ~0ull, rk = 0;
              if (i < nlive)
                { if (LDSLIVE) { en = lds_live[i]; rt = lds_root[i]; }
                  else         en = live[i];
                  if (en.x != NIL)                       // NIL = removed by the slope rule
                    { int ax = (int) en.y;
                      if (ax < apos - MAX_GAP)
                        A.state[en.x] = ST_DEAD;      // (only 'live or not' is ever read back)
                      else
                        { int bx = (int) en.z;
                          int dx = ax - bx;
                          uint64_t db = (uint64_t) ((uint32_t) dx + 0x80000000u);
                          keep = true;
                          if (dx > diag)
                            { if (bx >= thresh) lk = (db << 32) | (uint32_t) (0x7fffffff - ax); }
                          else
                            { if (bx <= bpos) rk = (db << 32) | (uint32_t) ax; }
                        }
                    }
                }
              uint64_t km  = gballot(keep);
              uint32_t dst = wr + (uint32_t) __popcll(km & lt);
              if (lk < myL) { myL = lk; myLe = en; myLpos = dst; myLroot = rt; }
              if (rk > myR) { myR = rk; myRe = en; myRpos = dst; myRroot = rt; }
              // in-place compaction: the write index never passes this chunk's first read index, and
              // the stored value is the loaded one, so the stores cannot overtake the loads
              if (keep && dst != i)               // (entries that stay where they are need no store)
                { if (LDSLIVE) { lds_live[dst] = en; lds_root[dst] = rt; }
                  else         live[dst] = en;
                }
              wr += (uint32_t) __popcll(km);
            }
          uint4    Le = make_uint4(NIL, 0, 0, 0), Re = make_uint4(NIL, 0, 0, 0);
          uint32_t Lpos = 0, Rpos = 0, Lroot = 0, Rroot = 0;
          { uint64_t lm = wave_min_u64<G>(myL);
            if (lm != ~0ull)
              { int sl = __ffsll((long long) gballot(myL == lm)) - 1;
                Le.x = __shfl(myLe.x, sl, G); Le.y = __shfl(myLe.y, sl, G); Le.z = __shfl(myLe.z, sl, G); Le.w = __shfl(myLe.w, sl, G);
                Lpos = __shfl(myLpos, sl, G);
                if (LDSLIVE) Lroot = __shfl(myLroot, sl, G);
              }
            uint64_t rm = wave_max_u64<G>(myR);
            if (rm != 0)
              { int sl = __ffsll((long long) gballot(myR == rm)) - 1;
                Re.x = __shfl(myRe.x, sl, G); Re.y = __shfl(myRe.y, sl, G); Re.z = __shfl(myRe.z, sl, G); Re.w = __shfl(myRe.w, sl, G);
                Rpos = __shfl(myRpos, sl, G);
                if (LDSLIVE) Rroot = __shfl(myRroot, sl, G);
              }
          }
          nlive = wr;

          // ---- extend the better neighbour's chain (map.c:1811-1862); uniform on all lanes
          int lcost = 0, rcost = 0;
          if (Le.x != NIL)
            lcost = (apos >= (int) Le.y + K) ? (int) Le.w + K : (int) Le.w + (apos - (int) Le.y);
          if (Re.x != NIL)
            rcost = (bpos >= (int) Re.z + K) ? (int) Re.w + K : (int) Re.w + (bpos - (int) Re.z);
          if (lcost > rcost) rcost = 0; else lcost = 0;

          uint4    pe = make_uint4(NIL, 0, 0, 0);
          uint32_t ppos = 0, proot = 0;
          int      c = K;
          if (lcost > 0)      { pe = Le; ppos = Lpos; proot = Lroot; c = lcost; }
          else if (rcost > 0) { pe = Re; ppos = Rpos; proot = Rroot; c = rcost; }

          uint32_t jroot = j;                    // origin of the chain the new hit ends
          if (pe.x != NIL)
            { uint32_t root;
              int      bc;
              if (LDSLIVE)
                { // bestcost[] is written by lane 0 only: let lane 0 read it back (a thread sees its own
                  // stores) and hand the value round
                  root = proot;
                  bc   = 0;
                  if (lane == 0) bc = A.bestcost[root];
                  bc = __shfl(bc, 0, G);
                }
              else
                { root = A.rootof[pe.x];
                  bc   = A.bestcost[root];
                }
              jroot = root;
              bool     kill = false;
              if (c >= bc)
                { int pa = (int) pe.y;
                  int pd = pa - (int) pe.z;
                  int dd = pd - diag;
                  if (dd < 0) dd = -dd;
                  kill = ((double) dd <= .2 * (double) (apos - pa));      // map.c:1837,1852
                }
              if (lane == 0)
                { A.from[j]   = pe.x;
                  A.cost[j]   = c;
                  A.rootof[j] = root;
                  A.state[j]  = ST_LIVE;
                  if (c >= bc)
                    { A.best[root]     = j;
                      A.bestcost[root] = c;
                      if (kill)
                        { A.state[pe.x] = ST_DEAD;
                          if (LDSLIVE) lds_live[ppos].x = NIL;
                          else         live[ppos].x = NIL;
                        }
                    }
                }
            }
          else if (lane == 0)
            { A.from[j]     = NIL;
              A.cost[j]     = K;
              A.rootof[j]   = j;
              A.best[j]     = j;
              A.bestcost[j] = K;
              A.state[j]    = ST_LIVE;
            }
          if (LDSLIVE)
            { if (nlive >= (uint32_t) LCAP) { spill = true; break; }
              if (lane == 0)
                { lds_live[nlive] = make_uint4(j, (uint32_t) apos, (uint32_t) bpos, (uint32_t) c);
                  lds_root[nlive] = jroot;
                }
              nlive += 1;
              // the lanes of a group meet again only through LDS, which serves a wavefront's accesses in order
              __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
              __builtin_amdgcn_wave_barrier();
            }
          else
            { if (lane == 0) live[nlive] = make_uint4(j, (uint32_t) apos, (uint32_t) bpos, (uint32_t) c);
              nlive += 1;
              __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            }
        }
      if (spill)
        { if (lane == 0) A.retry[atomicAdd(A.nretry, 1u)] = g;
          continue;
        }
      if (LDSLIVE) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");    // the report reads what lane 0 stored

      // ---- report the best end of every origin with cost >= hithr (map.c:1634-1649).  Order:
      //      live ends by (diag,apos) descending, then expired ends most recent first.
      uint32_t E = 0;
      for (uint32_t base = g0; base < g1; base += G)
        { uint32_t x = base + lane;
          bool     f = false;
          if (x < g1 && A.cost[x] >= A.hithr && A.from[x] != NIL)
            f = (A.best[A.rootof[x]] == x);
          uint64_t m = gballot(f);
          if (f) A.tmp[g0 + E + __popcll(m & lt)] = x;
          E += __popcll(m);
        }
      if (E == 0) continue;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");

      // A reported end closes a chain of at least HITMIN hits and belongs to a tree of its own, so a group of n
      // hits reports at most n / HITMIN ends: its summaries go to the slots from ceil(g0 / HITMIN) on -- no counter
      // shared by 10^6 groups, and the ranges of consecutive groups cannot overlap.
      const uint32_t obase = (g0 + HITMIN - 1) / HITMIN;
      if (lane == 0)
        { A.gcount[g] = E;
          A.gbase[g]  = obase;
        }

      for (uint32_t e = lane; e < E; e += G)
        { uint32_t x    = A.tmp[g0 + e];
          int      ax   = (int) (uint32_t) A.ab[x] + 1;
          int      bx   = (int) (A.ab[x] >> 32) + 1;
          int      dx   = ax - bx;
          // still in the tree when the PAIR's last hit has been processed (a cluster's own hits may end earlier)
          bool     live_x = (A.state[x] == ST_LIVE) && (A.glast == nullptr || ax >= (int) A.glast[g] - MAX_GAP);
          uint32_t     root = A.rootof[x];
          ChainSummary s;
          s.group = g;
          s.live  = live_x ? 1u : 0u;
          // expired ends are reported most recent first = by descending position in the pair's (apos, bpos) order
          s.okey  = live_x ? (((uint64_t) ((uint32_t) dx + 0x80000000u) << 32) | (uint32_t) ax)
                           : (((uint64_t) (uint32_t) ax << 32) | (uint32_t) bx);
          s.cost  = A.cost[x];
          s.ab    = (int) (uint32_t) A.ab[root] + 1 - K;
          s.bb    = (int) (A.ab[root] >> 32) + 1 - K;
          s.ae    = ax;
          s.be    = bx;
          s.end   = x;
          A.out[obase + e] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------
//  Small groups (<= CAP hits; in cluster mode nearly all of them): everything in LDS.  G lanes load the group's
//  hits, walk them with the per-hit state (cost / from / origin / the origin's best end) in LDS arrays indexed by
//  the hit's position in the group, and write `from[]` and the summaries at the end -- no global round trip
//  inside the walk, no ticket (groups are dealt to the lane groups round-robin: they are small and there are
//  millions of them), no live-window list: hit j scans the earlier hits of the group from the first that has
//  not expired.
// ---------------------------------------------------------------------------------------
template <int G, int CAP>
__global__ __launch_bounds__(256) void k_chain_small(ChainArgs A, uint32_t size_max)
{ constexpr int      GPB = 256 / G;
  constexpr uint16_t NIL16 = 0xffff;
  __shared__ uint32_t s_ap[GPB][CAP], s_bp[GPB][CAP];
  __shared__ int32_t  s_cost[GPB][CAP], s_bcost[GPB][CAP];
  __shared__ uint16_t s_from[GPB][CAP], s_root[GPB][CAP], s_best[GPB][CAP];
  __shared__ uint8_t  s_alive[GPB][CAP];
  const int      lane  = threadIdx.x & (G - 1);
  const int      sbase = (threadIdx.x & 63) & ~(G - 1);
  const uint64_t gmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
  const uint64_t lt    = (1ull << lane) - 1ull;
  const int      K     = A.K;
  const int      gi    = threadIdx.x / G;
  auto gballot = [&](bool p) -> uint64_t { return (__ballot(p) >> sbase) & gmask; };
  uint32_t *ap = s_ap[gi], *bp = s_bp[gi];
  int32_t  *cs = s_cost[gi], *bc = s_bcost[gi];
  uint16_t *fr = s_from[gi], *rt = s_root[gi], *be = s_best[gi];
  uint8_t  *al = s_alive[gi];

  const uint32_t stride = gridDim.x * GPB;
  for (uint32_t g = blockIdx.x * GPB + gi; g < A.ngroups; g += stride)
    { const uint32_t g0 = A.gstart[g], g1 = A.gstart[g + 1];
      const uint32_t n = g1 - g0;
      if (n > size_max) continue;
      for (uint32_t i = lane; i < n; i += G)
        { const uint64_t hb = A.ab[g0 + i];
          ap[i] = (uint32_t) hb + 1u;        // map.c:1784-1785
          bp[i] = (uint32_t) (hb >> 32) + 1u;
        }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();

      uint32_t head = 0;
      for (uint32_t j = 0; j < n; j++)
        { const int apos = (int) ap[j], bpos = (int) bp[j];
          const int diag = apos - bpos, thresh = bpos - MAX_GAP, old = apos - MAX_GAP;
          // ---- neighbour queries (map.c:1262-1318) over the hits [head, j) that are still in the tree; hits
          //      expire in order (map.c:1787-1796), so whole chunks of expired hits are left behind for good
          uint64_t myL = ~0ull, myR = 0;
          uint32_t myLi = 0, myRi = 0;
          for (uint32_t base = head; base < j; base += G)
            { const uint32_t i = base + lane;
              const bool in = i < j;
              const int  ax = in ? (int) ap[i] : 0;
              const bool fresh = in && ax >= old;
              if (gballot(fresh) == 0 && base == head && base + G <= j) { head += G; continue; }
              if (fresh && al[i])
                { const int bx = (int) bp[i];
                  const int dx = ax - bx;
                  const uint64_t db = (uint64_t) ((uint32_t) dx + 0x80000000u);
                  if (dx > diag)
                    { if (bx >= thresh)
                        { const uint64_t lk = (db << 32) | (uint32_t) (0x7fffffff - ax);
                          if (lk < myL) { myL = lk; myLi = i; }
                        }
                    }
                  else if (bx <= bpos)
                    { const uint64_t rk = (db << 32) | (uint32_t) ax;
                      if (rk > myR) { myR = rk; myRi = i; }
                    }
                }
            }
          uint32_t li = NIL, ri = NIL;
          { const uint64_t lm = wave_min_u64<G>(myL);
            if (lm != ~0ull) li = (uint32_t) __shfl((int) myLi, __ffsll((long long) gballot(myL == lm)) - 1, G);
            const uint64_t rm = wave_max_u64<G>(myR);
            if (rm != 0) ri = (uint32_t) __shfl((int) myRi, __ffsll((long long) gballot(myR == rm)) - 1, G);
          }
          // ---- extend the better neighbour's chain (map.c:1811-1862); uniform on the lanes of the group
          int lcost = 0, rcost = 0;
          if (li != NIL)
            { const int la = (int) ap[li], lc = cs[li];
              lcost = (apos >= la + K) ? lc + K : lc + (apos - la);
            }
          if (ri != NIL)
            { const int rb = (int) bp[ri], rc = cs[ri];
              rcost = (bpos >= rb + K) ? rc + K : rc + (bpos - rb);
            }
          if (lcost > rcost) rcost = 0; else lcost = 0;
          uint32_t pi = NIL;
          int      c = K;
          if (lcost > 0)      { pi = li; c = lcost; }
          else if (rcost > 0) { pi = ri; c = rcost; }
          if (pi != NIL)
            { const uint32_t root = rt[pi];
              const int      bcst = bc[root];
              bool kill = false;
              if (c >= bcst)
                { const int pa = (int) ap[pi];
                  const int pd = pa - (int) bp[pi];
                  int dd = pd - diag;
                  if (dd < 0) dd = -dd;
                  kill = ((double) dd <= .2 * (double) (apos - pa));      // map.c:1837,1852
                }
              if (lane == 0)
                { fr[j] = (uint16_t) pi;
                  cs[j] = c;
                  rt[j] = (uint16_t) root;
                  al[j] = 1;
                  if (c >= bcst)
                    { be[root] = (uint16_t) j;
                      bc[root] = c;
                      if (kill) al[pi] = 0;
                    }
                }
            }
          else if (lane == 0)
            { fr[j] = NIL16;
              cs[j] = K;
              rt[j] = (uint16_t) j;
              be[j] = (uint16_t) j;
              bc[j] = K;
              al[j] = 1;
            }
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }

      // ---- from[] for the Jump lists, and the best end of every origin with cost >= hithr (map.c:1634-1649)
      const int last = A.glast ? (int) A.glast[g] : (int) ap[n - 1];
      const uint32_t obase = (g0 + HITMIN - 1) / HITMIN;
      uint32_t E = 0;
      for (uint32_t base = 0; base < n; base += G)
        { const uint32_t x = base + lane;
          bool f = false;
          if (x < n)
            { const uint16_t fx = fr[x];
              A.from[g0 + x] = (fx == NIL16) ? NIL : g0 + fx;
              f = cs[x] >= A.hithr && fx != NIL16 && be[rt[x]] == (uint16_t) x;
            }
          const uint64_t m = gballot(f);
          if (f)
            { const int  ax = (int) ap[x], bx = (int) bp[x], dx = ax - bx;
              const bool live_x = al[x] && ax >= last - MAX_GAP;
              const uint32_t root = rt[x];
              ChainSummary s;
              s.group = g;
              s.live  = live_x ? 1u : 0u;
              s.okey  = live_x ? (((uint64_t) ((uint32_t) dx + 0x80000000u) << 32) | (uint32_t) ax)
                               : (((uint64_t) (uint32_t) ax << 32) | (uint32_t) bx);
              s.cost  = cs[x];
              s.ab    = (int) ap[root] - K;
              s.bb    = (int) bp[root] - K;
              s.ae    = ax;
              s.be    = bx;
              s.end   = g0 + x;
              A.out[obase + E + (uint32_t) __popcll(m & lt)] = s;
            }
          E += (uint32_t) __popcll(m);
        }
      if (lane == 0)
        { A.gcount[g] = E;
          A.gbase[g]  = obase;
        }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------
//  Simple groups without a walk.  Measured on the bench workload (DMH_CHAIN_PROFILE): 95.7 % of the hits of the
//  lane-per-group kernel find exactly ONE live hit in their window -- the hit before them -- take it as their
//  predecessor and remove it (the slope rule, map.c:1835-1841), so that the window holds one hit again; 4 % find none
//  (the first hit of a group, or one more than MAX_GAP behind its predecessor in A).  A group that consists of such
//  steps only is a sequence of plain chains: the score of a hit is its predecessor's plus min(K, advance) -- a
//  segmented prefix sum --, every chain's best end is its last hit, and nothing depends on the order of evaluation.
//  This kernel checks, for every group of up to size_max hits, that each hit P -> H (consecutive in the group's
//  (apos, bpos) order) is either
//     a break:  P.apos < H.apos - MAX_GAP (P has left the window when H scans it: H starts a chain), or
//     a step:   P is H's only candidate -- on a larger diagonal with P.bpos >= H.bpos - MAX_GAP (map.c:1806-1807, the
//               "left" neighbour) or on H's or a smaller diagonal with P.bpos <= H.bpos (the "right" one) -- and
//               5 |diagonal difference| <= H.apos - P.apos, so that H removes P;
//  anything else (P stays live beside H, or is no candidate) leaves the group to k_chain_lane.  For the groups that
//  pass it writes what the walk would: the `from` links, one summary per chain that scores, gcount / gbase, and marks
//  the group done.  One 16-lane group per chain group, sixteen hits per round; streaming: 8 B read and 4 B written
//  per hit.  Same results by construction -- and by the parity tests, which run every route.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chain_simple(ChainArgs A, uint32_t size_max)
{ constexpr int G = 16, GPB = 256 / G;
  const int lane = threadIdx.x & (G - 1), sub = (threadIdx.x & 63) / G;
  const int K = A.K;
  const uint64_t stride = (uint64_t) gridDim.x * GPB;
  uint64_t c0 = (uint64_t) blockIdx.x * GPB + threadIdx.x / G;
  // the bounds of the lane group's next chain group are asked for a turn ahead (a group is some 200 bytes: without that
  // every turn starts with a memory round trip for two numbers)
  // ... and its first sixteen hits, with the hit behind each, and its pair's last apos: two turns ahead the bounds, one
  // turn ahead what they point at
  const uint32_t hlast = A.nhits - 1;
  uint32_t ng0 = 0, ng1 = 0, fg0 = 0, fg1 = 0, nlastv = 0;
  uint64_t fw = 0, fnw = 0;
  if (c0 < A.ngroups)
    { ng0 = A.gstart[c0]; ng1 = A.gstart[c0 + 1];
      fw = A.ab[min(ng0 + (uint32_t) lane, hlast)]; fnw = A.ab[min(ng0 + (uint32_t) lane + 1u, hlast)];
      nlastv = A.glast ? A.glast[c0] : (uint32_t) A.ab[min(ng1 - 1u, hlast)];
    }
  if (c0 + stride < A.ngroups) { fg0 = A.gstart[c0 + stride]; fg1 = A.gstart[c0 + stride + 1]; }
  for (; __any(c0 < A.ngroups); c0 += stride)
    { const bool mine = c0 < A.ngroups;
      const uint32_t g = mine ? (uint32_t) c0 : 0;
      const uint32_t g0 = ng0;
      uint32_t n = mine ? ng1 - ng0 : 0;
      const uint64_t w0 = fw, nw0 = fnw;              // the first round's hits, asked for a turn ago
      const uint32_t lastr = nlastv;
      ng0 = fg0; ng1 = fg1;
      if (c0 + stride < A.ngroups)
        { fw = A.ab[min(ng0 + (uint32_t) lane, hlast)]; fnw = A.ab[min(ng0 + (uint32_t) lane + 1u, hlast)];
          nlastv = A.glast ? A.glast[c0 + stride] : (uint32_t) A.ab[min(ng1 - 1u, hlast)];
        }
      if (c0 + 2 * stride < A.ngroups) { fg0 = A.gstart[c0 + 2 * stride]; fg1 = A.gstart[c0 + 2 * stride + 1]; }
      if (n > size_max) n = 0;                        // (not ours: the listed-group kernels)
      // One pass: links and chain ends are written as if every hit were a break or a step; whether that was so is
      // known at the end, and only then is the group marked done (k_chain_lane walks the others and writes over this).
      const uint32_t obase = (g0 + HITMIN - 1) / HITMIN;
      uint32_t E = 0;
      bool     bad = false;
      int      ccost = 0, coa = 0, cob = 0;           // running chain at the end of the previous round: score, origin
      uint64_t plast = 0;                             // the last hit of the previous round
      const int lastv = (int) lastr + (A.glast ? 0 : 1);
      for (uint32_t r0 = 0; __any(r0 < n); r0 += G)
        { const uint32_t i = r0 + lane;
          const bool in = i < n;
          uint64_t w, nw;
          if (r0 == 0) { w = in ? w0 : 0; nw = (in && i + 1 < n) ? nw0 : 0; }
          else
            { w  = in ? A.ab[g0 + i] : 0;
              nw = (in && i + 1 < n) ? A.ab[g0 + i + 1] : 0;                   // (the same cache line, but for the round's last lane)
            }
          uint64_t pw = ((uint64_t) (uint32_t) __shfl_up((int) (uint32_t) (w >> 32), 1, G) << 32) | (uint64_t) (uint32_t) __shfl_up((int) (uint32_t) w, 1, G);
          if (lane == 0) pw = plast;
          plast = ((uint64_t) (uint32_t) __shfl((int) (uint32_t) (w >> 32), G - 1, G) << 32) | (uint64_t) (uint32_t) __shfl((int) (uint32_t) w, G - 1, G);
          const int a = (int) (uint32_t) w + 1, b = (int) (w >> 32) + 1, pa = (int) (uint32_t) pw + 1, pb = (int) (pw >> 32) + 1;
          const int d = a - b, pd = pa - pb;
          const bool brk = in && (i == 0 || pa < a - MAX_GAP);
          { const bool cand = (pd > d) ? (pb >= b - MAX_GAP) : (pb <= b);
            int dd = pd - d;
            if (dd < 0) dd = -dd;
            if (in && !brk && !(cand && 5 * dd <= a - pa)) bad = true;
          }
          const int  adv = (pd > d) ? a - pa : b - pb;                      // left neighbour: advance in A; right: in B
          int inc = brk ? K : (adv >= K ? K : adv);
          if (!in) inc = 0;
          // segmented inclusive sum over the 16 lanes; a lane before the round's first break continues the carried chain
          int  x = inc, oa = brk ? a : 0, ob = brk ? b : 0;
          bool f = brk;                                                       // a break at or below me in this round
#pragma unroll
          for (int o = 1; o < G; o <<= 1)
            { const int  y  = __shfl_up(x, o, G), ya = __shfl_up(oa, o, G), yb = __shfl_up(ob, o, G);
              const bool yf = __shfl_up((int) f, o, G) != 0;
              if (lane >= o && !f) { x += y; oa = ya; ob = yb; f = yf; }
            }
          if (!f) { x += ccost; oa = coa; ob = cob; }
          // the end of a chain: the group's last hit, or the hit before a break
          const int  na = (int) (uint32_t) nw + 1;
          const bool last = in && (i + 1 == n);
          const bool end  = in && (last || a < na - MAX_GAP);
          const bool emit = end && x >= A.hithr;
          const uint32_t em = (uint32_t) ((__ballot(emit) >> (sub * G)) & 0xffffull);
          if (emit)
            { const bool live = last && a >= lastv - MAX_GAP;                 // (its end is in the window when the group ends)
              ChainSummary sm;
              sm.group = g;
              sm.live  = live ? 1u : 0u;
              sm.okey  = live ? (((uint64_t) ((uint32_t) d + 0x80000000u) << 32) | (uint32_t) a) : (((uint64_t) (uint32_t) a << 32) | (uint32_t) b);
              sm.cost  = x;
              sm.ab    = oa - K;
              sm.bb    = ob - K;
              sm.ae    = a;
              sm.be    = b;
              sm.end   = g0 + i;
              A.out[obase + E + (uint32_t) __popc(em & ((1u << lane) - 1u))] = sm;
            }
          E += (uint32_t) __popc(em);
          if (in) A.from[g0 + i] = brk ? NIL : g0 + i - 1;
          // carry: the chain running at the round's last lane
          ccost = __shfl(x, G - 1, G); coa = __shfl(oa, G - 1, G); cob = __shfl(ob, G - 1, G);
        }
      const bool anybad = ((__ballot(bad) >> (sub * G)) & 0xffffull) != 0;
      if (n > 0 && !anybad && lane == 0)
        { A.gcount[g] = E;
          A.gbase[g]  = obase;
          A.gdone[g]  = 1u;
        }
    }
}

// ---------------------------------------------------------------------------------------
//  Groups that fall apart BY DIAGONAL (round 4).  What k_chain_simple (or the per-read kernel, dmh_readsort.h) leaves
//  is, on a repeat-rich reference, almost always a cluster that holds TWO OR MORE loci side by side -- a stretch of the
//  read opposite two copies of a repeat a few kb apart in the contig: 99.7 % of the hits left in groups of up to 1,023
//  (DMH_RS_PROFILE counters).  Their hits alternate between the copies in (apos, bpos) order, so no hit's predecessor is
//  its only candidate, yet the copies never meet: a hit N and a live hit X (X.apos within MAX_GAP behind N.apos) whose
//  DIAGONALS differ by more than MAX_GAP are no candidates for each other -- on a larger diagonal X would need
//  X.bpos >= N.bpos - MAX_GAP, i.e. a diagonal at most MAX_GAP + (X.apos - N.apos) <= MAX_GAP above N's; on a smaller one
//  X.bpos <= N.bpos, i.e. at most N.apos - X.apos <= MAX_GAP below (map.c:1262-1318).  So the diagonal CLASSES of a group --
//  runs of occupied 1024-wide diagonal bins with a free bin between -- chain independently, exactly as the B-clusters
//  of a pair do, and each class is usually a sequence of plain chains.  One wavefront per group left (up to 1,024 hits in
//  its registers, up to eight classes): per class the plain-chain test and the segmented prefix sum of k_chain_simple over
//  the class's hits alone (the others are transparent), links to the previous hit OF THE CLASS, chain ends reported when
//  the next hit of the class breaks (or the class ends).  A group all of whose classes pass is done; the others
//  (one in 350) stay for k_chain_lane.  No hit moves: the classes stay interleaved, only the links know them.
// ---------------------------------------------------------------------------------------
constexpr int CLS_MAX    = 8;         // classes of a group

// CLS_ROUNDS x 64 hits of a group at most: 2 for the bulk (two copies of a repeat: 80 hits; few registers, many groups in
// flight -- what bounds the kernel is the chain of memory round trips per group), 16 for the rest; a launch takes the groups
// of size_min .. size_max hits
template <int CLS_ROUNDS>
__global__ __launch_bounds__(256) void k_chain_classes(ChainArgs A, uint32_t size_min, uint32_t size_max)
{ __shared__ unsigned long long s_occ[4][4];                  // per wavefront: the occupied diagonal bins (256)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint32_t nwaves = gridDim.x * 4u, wave = blockIdx.x * 4u + (uint32_t) wv;
  const int K = A.K;
  for (uint64_t gb = (uint64_t) wave * 64u; gb < A.ngroups; gb += (uint64_t) nwaves * 64u)
    { // 64 groups at a time: which of them are left, and of a size for this kernel?
      const uint64_t gl = gb + (uint32_t) lane;
      uint32_t q0 = 0, q1 = 0;
      bool todo = false;
      if (gl < A.ngroups)
        { q0 = A.gstart[gl]; q1 = A.gstart[gl + 1];
          todo = A.gdone[gl] == 0u && q1 - q0 <= size_max && q1 - q0 >= size_min;
        }
      for (uint64_t tm = __ballot(todo); tm != 0; tm &= tm - 1)
        { const int      src = __ffsll((long long) tm) - 1;
          const uint32_t g   = (uint32_t) gb + (uint32_t) src;
          const uint32_t g0  = (uint32_t) __builtin_amdgcn_readlane((int) q0, src);
          const uint32_t n   = (uint32_t) __builtin_amdgcn_readlane((int) q1, src) - g0;
          const uint32_t rounds = (n + 63u) / 64u;            // <= CLS_ROUNDS
          // the group's hits, and the range of their diagonals
          uint64_t wq[CLS_ROUNDS];
          int dmin = 0x7fffffff, dmax = (int) 0x80000000;
#pragma unroll
          for (int j = 0; j < CLS_ROUNDS; j++)
            { wq[j] = 0;
              if ((uint32_t) j < rounds)
                { const uint32_t i = 64u * j + lane;
                  wq[j] = A.ab[g0 + (i < n ? i : n - 1)];
                  const int dg = (int) (uint32_t) wq[j] - (int) (wq[j] >> 32);
                  dmin = min(dmin, dg); dmax = max(dmax, dg);
                }
            }
          dmin = (int) (group_reduce_u32<64, false>((uint32_t) dmin + 0x80000000u) - 0x80000000u);
          dmax = (int) (group_reduce_u32<64, true>((uint32_t) dmax + 0x80000000u) - 0x80000000u);
          if (((dmax - dmin) >> 10) >= 256) continue;         // (uniform)
          // occupied bins -> class starts (a bin whose lower neighbour is free) -> the class of every hit
          if (lane < 4) s_occ[wv][lane] = 0ull;
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int j = 0; j < CLS_ROUNDS; j++)
            if ((uint32_t) j < rounds)
              { const uint32_t i = 64u * j + lane;
                const int b = (((int) (uint32_t) wq[j] - (int) (wq[j] >> 32)) - dmin) >> 10;
                if (i < n) atomicOr(&s_occ[wv][b >> 6], 1ull << (b & 63));
              }
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
          __builtin_amdgcn_wave_barrier();
          unsigned long long cst[4];
          uint32_t ncls = 0;
#pragma unroll
          for (int q = 0; q < 4; q++)
            { const unsigned long long oc = s_occ[wv][q], lowc = (q > 0) ? (s_occ[wv][q - 1] >> 63) : 0ull;
              cst[q] = oc & ~((oc << 1) | lowc);
              ncls += (uint32_t) __popcll(cst[q]);
            }
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
          __builtin_amdgcn_wave_barrier();                    // (s_occ is the next group's from here on)
          if (ncls < 2u || ncls > (uint32_t) CLS_MAX) continue;           // one class: a walk; more than eight: rare
          uint32_t clsv[CLS_ROUNDS];
#pragma unroll
          for (int j = 0; j < CLS_ROUNDS; j++)
            { const int b = (((int) (uint32_t) wq[j] - (int) (wq[j] >> 32)) - dmin) >> 10;
              uint32_t c = 0;
#pragma unroll
              for (int q = 0; q < 4; q++)
                { if (q < (b >> 6)) c += (uint32_t) __popcll(cst[q]);
                  else if (q == (b >> 6)) c += (uint32_t) __popcll(cst[q] & (~0ull >> (63 - (b & 63))));
                }
              clsv[j] = c - 1u;
            }
          const uint32_t obase = (g0 + HITMIN - 1) / HITMIN;
          const int lastv = (int) (A.glast ? A.glast[g] : (uint32_t) A.ab[g0 + n - 1] + 1u);
          uint32_t E = 0;
          bool bad = false;
          auto report = [&](bool who, int cost, int oa, int ob, int ea, int eb, uint32_t ex)
          { const uint64_t em = __ballot(who);
            if (who)
              { const bool live = ea >= lastv - MAX_GAP;      // (a chain's end is its last hit: never removed by the slope rule)
                const int  d = ea - eb;
                ChainSummary sm;
                sm.group = g;
                sm.live  = live ? 1u : 0u;
                sm.okey  = live ? (((uint64_t) ((uint32_t) d + 0x80000000u) << 32) | (uint32_t) ea) : (((uint64_t) (uint32_t) ea << 32) | (uint32_t) eb);
                sm.cost  = cost;
                sm.ab    = oa - K;
                sm.bb    = ob - K;
                sm.ae    = ea;
                sm.be    = eb;
                sm.end   = g0 + ex;
                A.out[obase + E + (uint32_t) __popcll(em & lt)] = sm;
              }
            E += (uint32_t) __popcll(em);
          };
          for (uint32_t k = 0; k < ncls; k++)
            { bool     have_prev = false;                         // the class's last hit so far, and the chain running at it:
              uint64_t prevw = 0, corig = 0;                      //   its score up to there, its first hit
              uint32_t prev_i = 0;
              int      ccost = 0;
#pragma unroll
              for (int j = 0; j < CLS_ROUNDS; j++)
                { if ((uint32_t) j >= rounds) continue;
                  const uint32_t i = 64u * j + lane;
                  const bool member = i < n && clsv[j] == k;
                  const uint64_t M = __ballot(member);
                  if (M == 0) continue;                           // (uniform)
                  // the hit of the class before this one: a lane below, or the last of an earlier round
                  const uint64_t w = wq[j];
                  const uint64_t below = M & lt;
                  const int  sl = below ? 63 - __clzll((long long) below) : 0;
                  uint64_t pw = ((uint64_t) (uint32_t) __shfl((int) (uint32_t) (w >> 32), sl) << 32) | (uint64_t) (uint32_t) __shfl((int) (uint32_t) w, sl);
                  uint32_t pi = 64u * j + (uint32_t) sl;
                  bool havep = true;
                  if (!below) { pw = prevw; pi = prev_i; havep = have_prev; }
                  const int a = (int) (uint32_t) w + 1, b = (int) (w >> 32) + 1, pa = (int) (uint32_t) pw + 1, pb = (int) (pw >> 32) + 1;
                  const int d = a - b, pd = pa - pb;
                  const bool brk = member && (!havep || pa < a - MAX_GAP);
                  { const bool cand = (pd > d) ? (pb >= b - MAX_GAP) : (pb <= b);
                    int dd = pd - d;
                    if (dd < 0) dd = -dd;
                    if (member && !brk && !(cand && 5 * dd <= a - pa)) bad = true;
                  }
                  const int adv = (pd > d) ? a - pa : b - pb;
                  const int inc = !member ? 0 : (brk ? K : (adv >= K ? K : adv));
                  // The running score of a chain = a prefix sum over the lanes (the hits of other classes add nothing) minus the
                  // sum below the chain's first hit -- the last break at or below the lane, from the ballot -- or, with no
                  // break in this round so far, plus the score carried in.  No scan through LDS: a DPP sum and lane reads.
                  const uint64_t BR = __ballot(brk);
                  const int P = (int) wave_incl_scan_dpp((uint32_t) inc);
                  // a hit that breaks ends the chain of the hit before it (lane sl of this round, or the carried one)
                  { const uint64_t brp = below ? (BR & ((1ull << sl) | ((1ull << sl) - 1ull))) : 0ull;      // breaks at or below the predecessor
                    const int stp = brp ? 63 - __clzll((long long) brp) : 0;
                    const int Pp  = __shfl(P, sl), Pb = __shfl(P, stp > 0 ? stp - 1 : 0);
                    const uint64_t ow = ((uint64_t) (uint32_t) __shfl((int) (uint32_t) (w >> 32), stp) << 32) | (uint64_t) (uint32_t) __shfl((int) (uint32_t) w, stp);
                    int px; uint64_t po;
                    if (!below)   { px = ccost; po = corig; }
                    else if (brp) { px = Pp - (stp > 0 ? Pb : 0); po = ow; }
                    else          { px = Pp + ccost; po = corig; }
                    report(member && brk && havep && px >= A.hithr, px, (int) (uint32_t) po + 1, (int) (po >> 32) + 1, pa, pb, pi);
                  }
                  if (member) A.from[g0 + i] = brk ? NIL : g0 + pi;
                  // carry: the class's last hit of this round and the chain running at it
                  { const int ll = 63 - __clzll((long long) M);
                    const uint64_t brl = BR & ((ll == 63) ? ~0ull : ((1ull << (ll + 1)) - 1ull));
                    const int Pl = __builtin_amdgcn_readlane(P, ll);
                    if (brl)
                      { const int stl = 63 - __clzll((long long) brl);
                        ccost = Pl - (stl > 0 ? __builtin_amdgcn_readlane(P, stl > 0 ? stl - 1 : 0) : 0);
                        corig = wave_lane_u64(w, stl);
                      }
                    else
                      ccost += Pl;
                    prevw = wave_lane_u64(w, ll);
                    prev_i = 64u * j + (uint32_t) ll;
                    have_prev = true;
                  }
                }
              // the class's last chain
              report(lane == 0 && have_prev && ccost >= A.hithr, ccost, (int) (uint32_t) corig + 1, (int) (corig >> 32) + 1,
                     (int) (uint32_t) prevw + 1, (int) (prevw >> 32) + 1, prev_i);
            }
          if (__ballot(bad) == 0 && lane == 0)
            { A.gcount[g] = E;
              A.gbase[g]  = obase;
              A.gdone[g]  = 1u;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
//  One LANE per group.  A lane group per chain group keeps 15 of its 16 lanes idle through the sequential part of
//  every hit, and the 10^6-10^7 groups of a pass on a repeat-rich genome (B-clusters of some twenty hits) offer all
//  the parallelism a GPU wants without any inside a group.  Here every lane walks groups of its own: the live
//  window sits in registers (RW slots, scanned by an unrolled loop; a hit takes the first free slot), the chain
//  trees with a live member in a small LDS table private to the lane (best score, origin, best end: all a
//  summary needs, so nothing is read back from global memory), hits are fetched a chunk ahead, and a lane that
//  finishes a group starts its next one at the wavefront's following step while the other lanes carry on (the group
//  after that is looked up in the background).  A window of more than RW live hits, or more than RT trees with live
//  members, sends the group to the lane-group kernels (A.retry).  Same rules, same results.
// ---------------------------------------------------------------------------------------
// RW window slots of a lane (registers) and RT tree slots (LDS).  Measured on the bench workload (DMH_CHAIN_PROFILE,
// profiles/scripts/rs_profile.py): 95.7 % of the hits find ONE live hit in the window, 4.0 % none, 0.34 % two, 0.01 %
// three; one tree 92.5 %, two 3.8 %, three 0.06 %.  So the launch is <3, 3>: the unrolled scan of the window is what a
// hit costs, and at eight slots it was five times what the data needs.  A group that outgrows its slots (one in 400)
// goes to the lane-group kernel.
constexpr int LANE_RW = 3, LANE_RT = 3;
#ifndef DMH_LANE_WALKS
#define DMH_LANE_WALKS 4
#endif
#ifndef DMH_LANE_CH
#define DMH_LANE_CH 8
#endif
#ifdef DMH_CHAIN_PROFILE
__device__ unsigned long long g_lprof[24];     // [0..8] hits by live window entries before the hit, [12..17] by live trees
#endif
constexpr int LANE_WALKS = DMH_LANE_WALKS;    // hits a lane walks per step of the wavefront (divides the chunk)
constexpr int LANE_BLOCK = 64;    // one wavefront per workgroup: 7.5 KB of LDS each, so the registers bound the occupancy

template <int RW, int RT>
__global__ __launch_bounds__(LANE_BLOCK, 5) void k_chain_lane(ChainArgs A, uint32_t size_max)
{ constexpr int NW = LANE_BLOCK / 64;
  static_assert(LANE_WALKS == 4 && DMH_LANE_CH == 8, "a chunk is two steps of four hits");
  __shared__ int32_t  t_bc[NW][RT][64];                       // per tree slot: best score ...
  __shared__ uint32_t t_oa[NW][RT][64], t_ob[NW][RT][64];      //   origin (apos, bpos)
  __shared__ uint32_t t_ea[NW][RT][64], t_eb[NW][RT][64];      //   best end (apos, bpos)
  __shared__ uint32_t t_ex[NW][RT][64];                       //   best end's position in the group
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int K = A.K;
  const uint32_t nthreads = gridDim.x * LANE_BLOCK;

  // the window: slot s holds a live hit iff bit s of vmask
  uint32_t w_ap[RW], w_bp[RW], w_meta[RW];     // meta = position in the group | tree slot << 24
  int32_t  w_cost[RW];
#pragma unroll
  for (int s = 0; s < RW; s++) { w_ap[s] = 0; w_bp[s] = 0; w_meta[s] = 0; w_cost[s] = 0; }
  uint32_t vmask = 0, tmask = 0;

  // the group being walked
  bool     active = false;
  uint32_t g = 0, g0 = 0, n = 0, j = 0, E = 0, obase = 0;
  // Hits are read -- and the `from` links written -- eight at a time per lane: a lane walks its own stretch of the
  // sorted hits, and a cache line fetched for ONE 8- or 4-byte access of it is long gone when the lane comes back for
  // the next (10^6 lanes stream at once): 22 GB of HBM traffic per launch against 1.5 GB of data.  A chunk is
  // the lane's next 64 bytes (and the 32 bytes of links that go with them), requested together; the chunk after it
  // (or the first one of the lane's next group) is on its way while this one is walked.
  constexpr int CH = DMH_LANE_CH, HC = CH / 2;
  // (the chunk in two halves of four: hit j of the group is element u = j mod 4 of the half j & 4 picks -- a select
  //  between two registers; as ONE array of eight the compiler keeps it in scratch memory and loads by index)
  uint64_t kb0[HC], kb1[HC], nb0[HC], nb1[HC];
  uint32_t fb0[HC], fb1[HC];
#pragma unroll
  for (int t = 0; t < HC; t++) { kb0[t] = 0; kb1[t] = 0; nb0[t] = 0; nb1[t] = 0; fb0[t] = 0; fb1[t] = 0; }
  bool     copy_nb = false;   // the chunk was finished in the last step: its successor is waiting in nb0 / nb1
  bool     flush = false, ending = false;    // ... its links are to be written; ... and it was the group's last
  uint32_t last_r = 0;        // apos of the group's last hit (live test of its ends), asked for when the group is taken
  uint32_t lq0 = 0, lq1 = 0, ldone = 0;  // bounds of group pg on their way (pst 2), and whether k_chain_simple has done it
  int      nb_for = 0;        // the look-ahead chunk holds: 0 nothing, 1 the next chunk of this group, 2 the first chunk of group pg
  // the group after it: pst 1 = look its bounds up, 2 = bounds on their way, 4 = bounds there and fine, 3 = and its first
  // chunk requested, 9 = none left
  uint32_t pg = blockIdx.x * LANE_BLOCK + threadIdx.x, q0 = 0, q1 = 0;
  int      pst = 1;
  const uint32_t hlast = A.nhits - 1;
  auto load_chunk = [&](uint32_t at)
  {
#pragma unroll
    for (int t = 0; t < HC; t++)
      { const uint32_t x0 = (at + t < hlast) ? at + t : hlast, x1 = (at + HC + t < hlast) ? at + HC + t : hlast;
        nb0[t] = A.ab[x0];
        nb1[t] = A.ab[x1];
      }
  };

  auto emit = [&](uint32_t t, bool at_end, int lastv)
  { // the best end of tree slot t, if it scores (map.c:1642-1649)
    const int bcst = t_bc[wv][t][lane];
    if (bcst < A.hithr) return;
    const uint32_t oa = t_oa[wv][t][lane], ob = t_ob[wv][t][lane];
    const uint32_t ea = t_ea[wv][t][lane], eb = t_eb[wv][t][lane], ex = t_ex[wv][t][lane];
    bool live_x = false;
    if (at_end)
      {
#pragma unroll
        for (int s = 0; s < RW; s++)
          if (((vmask >> s) & 1u) && (w_meta[s] & 0xffffffu) == ex) live_x = true;
        live_x = live_x && (int) ea >= lastv - MAX_GAP;
      }
    ChainSummary sm;
    sm.group = g;
    sm.live  = live_x ? 1u : 0u;
    const int dx = (int) ea - (int) eb;
    sm.okey  = live_x ? (((uint64_t) ((uint32_t) dx + 0x80000000u) << 32) | ea) : (((uint64_t) ea << 32) | eb);
    sm.cost  = bcst;
    sm.ab    = (int) oa - K;
    sm.bb    = (int) ob - K;
    sm.ae    = (int) ea;
    sm.be    = (int) eb;
    sm.end   = g0 + ex;
    A.out[obase + E] = sm;
    E += 1;
  };

  // A wavefront waits for ALL its outstanding loads at once, whichever lane asked for them: a load whose result is used
  // a few instructions later costs every lane a memory round trip.  So a step of the walk first takes in whatever
  // earlier steps have asked for (the chunk after the one just finished, the bounds and the first chunk of the next
  // group), then issues this step's loads, and only then walks its hit: the loads have the whole walk to arrive.
  while (true)
    { // ---- take in (a lane that is about to leave its group, or has none, and whose next group's first chunk has
      //      been asked for takes that chunk now: nothing below reads kb before the walk)
      if (copy_nb || ((!active || ending) && pst == 3))
        {
#pragma unroll
          for (int t = 0; t < HC; t++) { kb0[t] = nb0[t]; kb1[t] = nb1[t]; }
          nb_for = 0;
          copy_nb = false;
        }
      if (pst == 2)
        { q0 = lq0; q1 = lq1;            // (the registers a load is on its way to are read nowhere else)
          const uint32_t sz = q1 - q0;
          if (sz <= size_max && ldone == 0 && sz < A.walk_max) pst = 4;      // (not a listed group's size, not chained already)
          else
            { if (sz <= size_max && ldone == 0) A.wide[atomicAdd(A.nwide, 1u)] = pg;      // (too long a walk for one lane)
              pg += nthreads; pst = 1;
            }
        }
      // ---- what the last step left to be written: its group's ends (they read last_r: before any store, or the
      //      wait for it would wait for the stores too), its chunk's links
      const bool was_flush = flush;
      const uint32_t jl = j - 1;
      uint32_t *fo = A.from + g0 + (jl & ~(uint32_t) (CH - 1));
      if (ending)
        { const int lastv = (int) last_r + (A.glast ? 0 : 1);
          for (uint32_t m = tmask; m != 0; m &= m - 1)
            emit((uint32_t) __ffs((int) m) - 1u, true, lastv);
          A.gcount[g] = E;
          A.gbase[g]  = obase;
          active = false;
          ending = false;
        }
      if (was_flush)
        {
#pragma unroll
          for (int t = 0; t < HC; t++)
            { if ((uint32_t) t <= (jl & (CH - 1))) fo[t] = fb0[t];
              if ((uint32_t) (t + HC) <= (jl & (CH - 1))) fo[t + HC] = fb1[t];
            }
          flush = false;
        }
      bool taken = false;
      if (!active && pst == 3)
        { active = true; taken = true;
          g = pg; g0 = q0; n = q1 - q0; j = 0; E = 0;
          obase = (g0 + HITMIN - 1) / HITMIN;
          vmask = 0; tmask = 0;
          pg += nthreads;
          pst = 1;
        }
      // ---- ask for: the group's last apos, the bounds of the lane's next group, the next chunk (of this group, or the
      //      first one of the next group once its bounds are known)
      if (taken)
        last_r = A.glast ? A.glast[g] : (uint32_t) A.ab[g0 + n - 1];      // (0-based there: see emit)
      if (pst == 1)
        { if (pg < A.ngroups)
            { lq0 = A.gstart[pg];
              lq1 = A.gstart[pg + 1];
              ldone = A.gdone ? A.gdone[pg] : 0u;
              pst = 2;
            }
          else
            pst = 9;
        }
      if (nb_for == 0)
        { uint32_t at = 0;
          int      what = 0;
          const uint32_t nxt = (j & ~(uint32_t) (CH - 1)) + CH;
          if (active && nxt < n) { at = g0 + nxt; what = 1; }
          else if (pst == 4)     { at = q0; what = 2; }
          if (what != 0)
            { load_chunk(at);
              nb_for = what;
              if (what == 2) pst = 3;
            }
        }
      asm volatile("" ::: "memory");        // (no store of the above is to be moved behind the walk)
      if (__ballot(active || pst != 9) == 0) break;
      // ---- LANE_WALKS hits per step: the memory operations above have that much longer to complete, and their
      //      bookkeeping is shared (a chunk is a whole number of steps; a lane whose group ends inside one idles)
      // (a lane starts a group at the head of a step and walks one hit per turn while it walks at all, so j = u (mod 4):
      //  hit j is register u or u + 4 of the chunk -- one select instead of a search through the eight)
#pragma unroll
      for (int u = 0; u < LANE_WALKS; u++)
        { if (!active || ending || copy_nb) continue;

          // ---- hit j of the group (map.c:1781-1872)
          const uint32_t t8 = j & (CH - 1);
          const bool hi4 = (j & 4u) != 0;
          const uint64_t hk = hi4 ? kb1[u] : kb0[u];
          const int apos = (int) (uint32_t) hk + 1, bpos = (int) (hk >> 32) + 1;
          const int diag = apos - bpos, thresh = bpos - MAX_GAP, old = apos - MAX_GAP;
          // The window scan, written without branches (selects only): the left neighbour is the live hit with the
          // smallest (diagonal, -apos) among those with a larger diagonal and bpos >= bpos - MAX_GAP, the right one the
          // largest (diagonal, apos) among those with diagonal <= diag and bpos <= bpos (map.c:1262-1318); keys are
          // unique inside a group.  Slots no lane of the wavefront uses are skipped as a whole.
          int      Ldx = 0x7fffffff, Lax = -1, Lc = 0, Rdx = (int) 0x80000000, Rax = -1, Rc = 0;
          uint32_t Lm = 0, Ls = 0, Rm = 0, Rs = 0, emask = 0;
#pragma unroll
          for (int s = 0; s < RW; s++)
            { const bool v = (vmask >> s) & 1u;
              if (__ballot(v) == 0) continue;
              const int  ax = (int) w_ap[s], bx = (int) w_bp[s], dx = ax - bx;
              const bool ex = v && ax < old;                            // expired (map.c:1787-1796)
              emask |= ex ? (1u << s) : 0u;
              const bool ok  = v && !ex;
              const bool isL = ok && dx > diag && bx >= thresh;
              const bool isR = ok && dx <= diag && bx <= bpos;
              const bool bl  = isL && (dx < Ldx || (dx == Ldx && ax > Lax));
              const bool br  = isR && (dx > Rdx || (dx == Rdx && ax > Rax));
              Ldx = bl ? dx : Ldx; Lax = bl ? ax : Lax; Lc = bl ? w_cost[s] : Lc; Lm = bl ? w_meta[s] : Lm; Ls = bl ? (uint32_t) s : Ls;
              Rdx = br ? dx : Rdx; Rax = br ? ax : Rax; Rc = br ? w_cost[s] : Rc; Rm = br ? w_meta[s] : Rm; Rs = br ? (uint32_t) s : Rs;
            }
          vmask &= ~emask;
#ifdef DMH_CHAIN_PROFILE
          atomicAdd(&g_lprof[__popc(vmask)], 1ull);
          atomicAdd(&g_lprof[12 + __popc(tmask)], 1ull);
#endif
          const int  Rbx    = Rax - Rdx;
          const int  lcost0 = (Lax >= 0) ? ((apos >= Lax + K) ? Lc + K : Lc + (apos - Lax)) : 0;
          const int  rcost0 = (Rax >= 0) ? ((bpos >= Rbx + K) ? Rc + K : Rc + (bpos - Rbx)) : 0;
          const bool useL   = lcost0 > rcost0;                        // map.c:1823-1826 (ties go right)
          const bool havep  = useL || rcost0 > 0;
          const int      c  = useL ? lcost0 : (havep ? rcost0 : K);
          const uint32_t pm = useL ? Lm : Rm, ps = useL ? Ls : Rs;
          const int      pa = useL ? Lax : Rax, pd = useL ? Ldx : Rdx;
          bool     overflow = false;
          uint32_t tj = pm >> 24;
          if (!havep)
            { if ((tmask & ((1u << RT) - 1u)) == ((1u << RT) - 1u))
                { // no tree slot left: the trees without a live member are complete -- report them and take their slots
                  uint32_t used = 0;
#pragma unroll
                  for (int s = 0; s < RW; s++)
                    if ((vmask >> s) & 1u) used |= 1u << (w_meta[s] >> 24);
                  const uint32_t dead = tmask & ~used;
                  for (uint32_t m = dead; m != 0; m &= m - 1)
                    emit((uint32_t) __ffs((int) m) - 1u, false, 0);
                  tmask &= ~dead;
                }
              if ((tmask & ((1u << RT) - 1u)) == ((1u << RT) - 1u))
                { overflow = true; tj = 0; }
              else
                { tj = (uint32_t) __ffs((int) ~tmask) - 1u;
                  tmask |= 1u << tj;
                  t_oa[wv][tj][lane] = (uint32_t) apos;
                  t_ob[wv][tj][lane] = (uint32_t) bpos;
                }
            }
          const int  bcst = havep ? t_bc[wv][tj][lane] : 0;
          const bool improved = !havep || c >= bcst;                   // (a new origin is its tree's best end)
          { const uint32_t fv = havep ? g0 + (pm & 0xffffffu) : NIL;
            fb0[u] = hi4 ? fb0[u] : fv;
            fb1[u] = hi4 ? fv : fb1[u];
          }
          if (improved && !overflow)
            { t_bc[wv][tj][lane] = c;
              t_ea[wv][tj][lane] = (uint32_t) apos;
              t_eb[wv][tj][lane] = (uint32_t) bpos;
              t_ex[wv][tj][lane] = j;
            }
          { int dd = pd - diag;
            if (dd < 0) dd = -dd;
            // map.c:1837,1852: (double) dd <= .2 * (double) (apos - pa).  In integers: the double nearest to .2 lies
            // above it by 5.6e-17 of its value, so the rounded product is x / 5 or the next double above, never below,
            // and for integers dd, x < 2^31 the test is 5 dd <= x exactly (x / 5 + eps reaches no further integer)
            const bool kill = havep && improved && (5 * dd <= apos - pa);
            vmask &= ~(kill ? (1u << ps) : 0u);
          }
          { const uint32_t fm = ~vmask & ((1u << RW) - 1u);
            overflow = overflow || fm == 0;
            const int f = __ffs((int) fm) - 1;             // (-1 when the window is full: no slot matches)
#pragma unroll
            for (int s = 0; s < RW; s++)
              { const bool h = (s == f);
                w_ap[s] = h ? (uint32_t) apos : w_ap[s]; w_bp[s] = h ? (uint32_t) bpos : w_bp[s];
                w_cost[s] = h ? c : w_cost[s]; w_meta[s] = h ? (j | (tj << 24)) : w_meta[s];
              }
            vmask |= (fm != 0) ? (1u << f) : 0u;
          }
          j += 1;
          if (overflow)
            { A.retry[atomicAdd(A.nretry, 1u)] = g;          // (the lane-group kernel starts the group afresh)
              active = false;
              if (nb_for == 1) nb_for = 0;                   // (nb0 / nb1 hold the abandoned group's next chunk)
            }
          else if (j == n || t8 == CH - 1)
            { // the chunk's links go out together -- at the head of the next step, like the group's ends when this was
              // its last hit (stores count as outstanding memory operations too)
              flush = true;
              if (j == n) ending = true;
              else        copy_nb = true;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
//  Repeat piles: one WORKGROUP per group.  A (read, contig) group of a repeat family has tens of thousands
//  of hits and a live window of thousands; one wavefront walking it (k_chain<64>) scans dozens of 64-entry
//  chunks of global memory per hit and sets the run time of the whole pass.  Here the window lives in LDS
//  (PILE_CAP entries + their chain origins), PILE_THREADS lanes scan it, entries are never moved while
//  hits are added (expired and removed ones become tombstones; the list is compacted when it has doubled),
//  so a hit costs one strided LDS scan, one reduction over the wavefronts and the sequential update.
//  Same rules and same results as k_chain (the window is an unordered set with unique keys).  A window
//  that outgrows LDS sends its group to the 64-lane launch (A.retry).
// ---------------------------------------------------------------------------------------
constexpr int PILE_THREADS = 512;
constexpr int PILE_WAVES   = PILE_THREADS / 64;
constexpr int PILE_CAP     = 6912;                   // 108 KB of entries + 27 KB of origins
constexpr int PILE_BCACHE  = 2048;                   // direct-mapped cache of bestcost[] (thread 0 is its only writer)

#ifdef DMH_CHAIN_PROFILE
__device__ unsigned long long g_cprof[16];
#define CPROF_T0() long long _t0 = clock64()
#define CPROF_LAP(i) do { long long _t1 = clock64(); if (threadIdx.x == 0) atomicAdd(&g_cprof[i], (unsigned long long) (_t1 - _t0)); _t0 = _t1; } while (0)
#define CPROF_ADD(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_cprof[i], (unsigned long long) (v)); } while (0)
#else
#define CPROF_T0() do {} while (0)
#define CPROF_LAP(i) do {} while (0)
#define CPROF_ADD(i, v) do {} while (0)
#endif

// The waves of a pile's workgroup meet twice per hit and exchange LDS data only: a barrier that waits for the
// wave's LDS operations but NOT for its global stores (thread 0's six per hit, each acknowledged by the L2
// long after it was issued; __syncthreads would wait for them every time).
__device__ __forceinline__ void lds_barrier()
{ asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(PILE_THREADS) void k_chain_pile(ChainArgs A)
{ __shared__ uint4    s_live[PILE_CAP];
  __shared__ uint32_t s_root[PILE_CAP];
  __shared__ uint64_t s_key[2][PILE_WAVES];
  __shared__ uint32_t s_idx[2][PILE_WAVES];
  __shared__ uint32_t s_cnt[PILE_WAVES];
  __shared__ uint32_t s_g;
  __shared__ uint2    s_bc[PILE_BCACHE];             // {origin, its best score}: saves the global round trip per hit
  const int      tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint64_t lt  = (1ull << lane) - 1ull;
  const int      K   = A.K;
  const uint32_t nwork = *A.nonly;

  while (true)
    { __syncthreads();
      if (tid == 0) s_g = atomicAdd(A.ticket, 1u);
      __syncthreads();
      uint32_t g = s_g;
      if (g >= nwork) break;
      g = A.only[g];
      const uint32_t g0 = A.gstart[g], g1 = A.gstart[g + 1];
      uint32_t n = 0;                      // entries in the list, tombstones included
      uint32_t hw = A.pile_cap < 1024u ? A.pile_cap : 1024u;      // compact when the list reaches this length
      bool     spill = false;
      for (int x = tid; x < PILE_BCACHE; x += PILE_THREADS) s_bc[x] = make_uint2(NIL, 0);
      // The next hit, fetched one round ahead -- through the scalar cache (the address is the workgroup's, and nothing
      // writes the hits here): a vector load would share the wavefront's one in-order counter with thread 0's
      // stores of the round before, and waiting for the hit would wait for their acknowledgement, every round.
      typedef const __attribute__((address_space(4))) uint64_t *const_hits;
      const const_hits cab = (const_hits) (uintptr_t) A.ab;
      uint64_t nkey = cab[g0];

      for (uint32_t j = g0; j < g1; j++)
        { const int apos = (int) (uint32_t) nkey + 1;             // map.c:1784-1785
          const int bpos = (int) (nkey >> 32) + 1;
          const int diag = apos - bpos;
          const int thresh = bpos - MAX_GAP;
          { const uint32_t jn = (j + 1 < g1) ? j + 1 : j;
            nkey = cab[jn];
          }

          // ---- expiry (map.c:1787-1796) + neighbour queries (map.c:1262-1318) over the thread's entries
          CPROF_T0();
          CPROF_ADD(6, 1);
          CPROF_ADD(7, n);
          uint64_t myL = ~0ull, myR = 0;
          uint32_t myLi = 0, myRi = 0;
#pragma unroll 2
          for (uint32_t i = tid; i < n; i += PILE_THREADS)
            { const uint4 en = s_live[i];
              const int  ax = (int) en.y, bx = (int) en.z;
              const int  dx = ax - bx;
              const bool there = (en.x != NIL);
              const bool old   = there && (ax < apos - MAX_GAP);
              if (old)
                { A.state[en.x] = ST_DEAD;
                  s_live[i].x = NIL;
                }
              const uint64_t db = (uint64_t) ((uint32_t) dx + 0x80000000u);
              const uint64_t lk = (db << 32) | (uint32_t) (0x7fffffff - ax);
              const uint64_t rk = (db << 32) | (uint32_t) ax;
              const bool cl = there && !old && dx > diag && bx >= thresh;
              const bool cr = there && !old && dx <= diag && bx <= bpos;
              if (cl && lk < myL) { myL = lk; myLi = i; }
              if (cr && rk > myR) { myR = rk; myRi = i; }
            }
          CPROF_LAP(0);
          { uint64_t lm = wave_min_u64<64>(myL);
            uint64_t rm = wave_max_u64<64>(myR);
            uint64_t bl = __ballot(myL == lm), br = __ballot(myR == rm);
            if (lane == __ffsll((long long) bl) - 1) { s_key[0][w] = lm; s_idx[0][w] = myLi; }
            if (lane == __ffsll((long long) br) - 1) { s_key[1][w] = rm; s_idx[1][w] = myRi; }
          }
          CPROF_LAP(1);
          lds_barrier();
          CPROF_LAP(2);

          // ---- the sequential part (map.c:1811-1862), on thread 0; the others wait at the barrier
          if (tid == 0)
            { uint64_t lm = ~0ull, rm = 0;
              uint32_t li = 0, ri = 0;
              for (int x = 0; x < PILE_WAVES; x++)
                { if (s_key[0][x] < lm) { lm = s_key[0][x]; li = s_idx[0][x]; }
                  if (s_key[1][x] > rm) { rm = s_key[1][x]; ri = s_idx[1][x]; }
                }
              uint4 Le = make_uint4(NIL, 0, 0, 0), Re = make_uint4(NIL, 0, 0, 0);
              if (lm != ~0ull) Le = s_live[li];
              if (rm != 0)     Re = s_live[ri];
              int lcost = 0, rcost = 0;
              if (Le.x != NIL)
                lcost = (apos >= (int) Le.y + K) ? (int) Le.w + K : (int) Le.w + (apos - (int) Le.y);
              if (Re.x != NIL)
                rcost = (bpos >= (int) Re.z + K) ? (int) Re.w + K : (int) Re.w + (bpos - (int) Re.z);
              if (lcost > rcost) rcost = 0; else lcost = 0;

              uint4    pe = make_uint4(NIL, 0, 0, 0);
              uint32_t ppos = 0;
              int      c = K;
              if (lcost > 0)      { pe = Le; ppos = li; c = lcost; }
              else if (rcost > 0) { pe = Re; ppos = ri; c = rcost; }

              uint32_t jroot = j;
              if (pe.x != NIL)
                { const uint32_t root = s_root[ppos];
                  // (one 8-byte read and a branch: as two 4-byte reads the compiler turns "cached score or the one in
                  // global memory" into a flat load from a selected address, waited for -- together with every store
                  // before it -- at every hit)
                  const uint64_t cbw  = *(const uint64_t *) &s_bc[root & (PILE_BCACHE - 1)];
                  const uint2    cb   = make_uint2((uint32_t) cbw, (uint32_t) (cbw >> 32));
                  int            bc   = (int) cb.y;
                  if (cb.x != root)
                    { bc = A.bestcost[root];
                      asm volatile("" : "+v"(bc));        // (the wait for it here, not behind the join at every hit)
                    }
                  jroot = root;
                  A.from[j]   = pe.x;
                  A.cost[j]   = c;
                  A.rootof[j] = root;
                  A.state[j]  = ST_LIVE;
                  if (c >= bc)
                    { int pa = (int) pe.y;
                      int pd = pa - (int) pe.z;
                      int dd = pd - diag;
                      if (dd < 0) dd = -dd;
                      const bool kill = ((double) dd <= .2 * (double) (apos - pa));      // map.c:1837,1852
                      A.best[root]     = j;
                      A.bestcost[root] = c;
                      s_bc[root & (PILE_BCACHE - 1)] = make_uint2(root, (uint32_t) c);
                      if (kill)
                        { A.state[pe.x] = ST_DEAD;
                          s_live[ppos].x = NIL;
                        }
                    }
                  else if (cb.x != root)
                    s_bc[root & (PILE_BCACHE - 1)] = make_uint2(root, (uint32_t) bc);
                }
              else
                { A.from[j]     = NIL;
                  A.cost[j]     = K;
                  A.rootof[j]   = j;
                  A.best[j]     = j;
                  A.bestcost[j] = K;
                  A.state[j]    = ST_LIVE;
                  s_bc[j & (PILE_BCACHE - 1)] = make_uint2(j, (uint32_t) K);
                }
              s_live[n] = make_uint4(j, (uint32_t) apos, (uint32_t) bpos, (uint32_t) c);      // (n < PILE_CAP, see below)
              s_root[n] = jroot;
            }
          n += 1;
          CPROF_LAP(3);
          lds_barrier();
          CPROF_LAP(4);

          // ---- the list has doubled (or is full): squeeze the tombstones out, through the group's slice
          //      of the global scratch arrays (any order will do)
          if (n >= hw)
            { uint32_t mine = 0;
              for (uint32_t base = (uint32_t) w * 64; base < n; base += PILE_THREADS)
                { uint32_t i = base + lane;
                  mine += (uint32_t) __popcll(__ballot(i < n && s_live[i].x != NIL));
                }
              if (lane == 0) s_cnt[w] = mine;
              __syncthreads();
              uint32_t pre = 0, alive = 0;
              for (int x = 0; x < PILE_WAVES; x++)
                { if (x < w) pre += s_cnt[x];
                  alive += s_cnt[x];
                }
              uint4    *gl = A.live + g0;
              uint32_t *gr = A.tmp + g0;
              for (uint32_t base = (uint32_t) w * 64; base < n; base += PILE_THREADS)
                { uint32_t i = base + lane;
                  bool     keep = (i < n && s_live[i].x != NIL);
                  uint64_t km = __ballot(keep);
                  if (keep)
                    { uint32_t dst = pre + (uint32_t) __popcll(km & lt);
                      gl[dst] = s_live[i];
                      gr[dst] = s_root[i];
                    }
                  pre += (uint32_t) __popcll(km);
                }
              __syncthreads();
              for (uint32_t i = tid; i < alive; i += PILE_THREADS)
                { s_live[i] = gl[i];
                  s_root[i] = gr[i];
                }
              n  = alive;
              hw = 2 * alive + 512;
              if (hw > A.pile_cap) hw = A.pile_cap;
              __syncthreads();
              CPROF_LAP(5);
              CPROF_ADD(8, 1);
              if (n + 64 >= A.pile_cap) { spill = true; break; }      // the window itself outgrew LDS
            }
        }
      if (spill)
        { if (tid == 0) A.retry[atomicAdd(A.nretry, 1u)] = g;
          CPROF_ADD(9, 1);
          continue;
        }
      __syncthreads();

      // ---- report (as k_chain's, by the first wavefront)
      if (w == 0)
        { uint32_t E = 0;
          for (uint32_t base = g0; base < g1; base += 64)
            { uint32_t x = base + lane;
              bool     f = false;
              if (x < g1 && A.cost[x] >= A.hithr && A.from[x] != NIL)
                f = (A.best[A.rootof[x]] == x);
              uint64_t m = __ballot(f);
              if (f) A.tmp[g0 + E + __popcll(m & lt)] = x;
              E += __popcll(m);
            }
          if (E != 0)
            { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
              const uint32_t obase = (g0 + HITMIN - 1) / HITMIN;
              if (lane == 0)
                { A.gcount[g] = E;
                  A.gbase[g]  = obase;
                }
              for (uint32_t e = lane; e < E; e += 64)
                { uint32_t x    = A.tmp[g0 + e];
                  int      ax   = (int) (uint32_t) A.ab[x] + 1;
                  int      bx   = (int) (A.ab[x] >> 32) + 1;
                  int      dx   = ax - bx;
                  bool     live_x = (A.state[x] == ST_LIVE) && (A.glast == nullptr || ax >= (int) A.glast[g] - MAX_GAP);
                  uint32_t root = A.rootof[x];
                  ChainSummary s;
                  s.group = g;
                  s.live  = live_x ? 1u : 0u;
                  s.okey  = live_x ? (((uint64_t) ((uint32_t) dx + 0x80000000u) << 32) | (uint32_t) ax)
                                   : (((uint64_t) (uint32_t) ax << 32) | (uint32_t) bx);
                  s.cost  = A.cost[x];
                  s.ab    = (int) (uint32_t) A.ab[root] + 1 - K;
                  s.bb    = (int) (A.ab[root] >> 32) + 1 - K;
                  s.ae    = ax;
                  s.be    = bx;
                  s.end   = x;
                  A.out[obase + e] = s;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
//  Device-side candidate lists: reporting order, domination rule, -p deltas, Jump lists
//  (map.c:1634-1766).  The reference does this per read on the thread that owns the read; here the
//  chain ends of a pass are put into the reference's order by rank (k_order_summaries), every read's
//  ends are contiguous (groups are sorted by read), and one 16-lane group per read applies the
//  domination rule to its list in that order (k_dominate).  Nothing but a few counters goes to the host.
// ---------------------------------------------------------------------------------------

struct OrdChain
{ int32_t  cost, ab, ae, bb, be;
  uint32_t end;
  uint64_t gkey;          // (aread | pass | bread) of the group = its hits' sort key without the position bits
};

// Reporting order inside a group (map.c:1634-1642): live ends by (diagonal, apos) descending, then expired ends
// most recent first -- i.e. descending (live, okey).  The keys of a group are unique, so the rank of an end is
// the number of its siblings that precede it.  The siblings are the ends of all groups of the same (aread, bread)
// pair -- the group itself, or in cluster mode the pair's clusters c0..c1-1 (their numbers are consecutive), of which
// only a few hold an end at all.  So the ends are first packed (goff = exclusive scan of the groups' end counts:
// the ends of group g go to slots goff[g]..., a pair's ends to goff[c0] .. goff[c1]) together with their pair's slot
// range, and then every end is ranked by its own thread among that short range.
// The order of two sibling ends in one word: live before expired, then okey.  A live okey is (diagonal + 2^31, apos)
// and an expired one (apos, bpos), with positions below 2^31, so the low half gives up its top bit to make room.
__device__ __forceinline__ uint64_t order_word(uint32_t live, uint64_t okey)
{ return ((uint64_t) (live & 1u) << 63) | ((okey >> 32) << 31) | (okey & 0x7fffffffull); }

__global__ __launch_bounds__(256) void k_pack_summaries(const ChainSummary *__restrict__ out, uint32_t ngroups,
                                                        const uint32_t *__restrict__ gbase,
                                                        const uint32_t *__restrict__ gcount,
                                                        const uint32_t *__restrict__ goff,
                                                        const uint32_t *__restrict__ cog,
                                                        const uint32_t *__restrict__ ocfirst,
                                                        ChainSummary *__restrict__ packed, uint2 *__restrict__ range,
                                                        uint64_t *__restrict__ oword,
                                                        const uint32_t *__restrict__ gstart, const uint32_t *__restrict__ glast)
{ for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += gridDim.x * blockDim.x)
    { const uint32_t En = gcount[g];
      if (En == 0) continue;
      uint32_t c0 = g, c1 = g + 1;
      if (cog != nullptr)
        { const uint32_t p = cog[g];
          c0 = ocfirst[p];
          c1 = ocfirst[p + 1];
        }
      const uint2 r = make_uint2(goff[c0], goff[c1]);
      const uint32_t dst = goff[g], mb = gbase[g];
      for (uint32_t e = 0; e < En; e++)
        { ChainSummary s = out[mb + e];
          if (s.live == LIVE_LATER)
            { // an end reported by the per-read kernel (dmh_readsort.h, step 5): the chain's last hit; live when it is the
              // group's last hit and still in the window at the pair's last hit (k_chain_simple's rule)
              const bool live = (s.end + 1u == gstart[g + 1]) && s.ae >= (int) glast[g] - MAX_GAP;
              const int  d = s.ae - s.be;
              s.group = g;
              s.live  = live ? 1u : 0u;
              s.okey  = live ? (((uint64_t) ((uint32_t) d + 0x80000000u) << 32) | (uint32_t) s.ae)
                             : (((uint64_t) (uint32_t) s.ae << 32) | (uint32_t) s.be);
            }
          packed[dst + e] = s;
          range[dst + e]  = r;
          oword[dst + e]  = order_word(s.live, s.okey);
        }
    }
}

// The sibling ranges are short and the ends of a pair sit next to each other, so neighbouring lanes walk the same
// order words (one 8-byte load per sibling, mostly the same address across the wavefront).
__global__ __launch_bounds__(256) void k_order_summaries(const ChainSummary *__restrict__ packed,
                                                         const uint2 *__restrict__ range,
                                                         const uint64_t *__restrict__ oword, uint32_t nsum,
                                                         const uint64_t *__restrict__ gkey, OrdChain *__restrict__ ord)
{ const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nsum) return;
  const uint2 r = range[i];
  const uint64_t mine = oword[i];
  uint32_t rank = 0, j = r.x;
  for (; j + 4 <= r.y; j += 4)
    { const uint64_t t0 = oword[j], t1 = oword[j + 1], t2 = oword[j + 2], t3 = oword[j + 3];
      rank += (t0 > mine ? 1u : 0u) + (t1 > mine ? 1u : 0u) + (t2 > mine ? 1u : 0u) + (t3 > mine ? 1u : 0u);
    }
  for (; j < r.y; j++) rank += (oword[j] > mine) ? 1u : 0u;
  const ChainSummary s = packed[i];
  OrdChain o;
  o.cost = s.cost; o.ab = s.ab; o.ae = s.ae; o.bb = s.bb; o.be = s.be; o.end = s.end;
  o.gkey = gkey[s.group];
  ord[r.x + rank] = o;
}

// rgfirst[r] = first group whose read index is >= r (r = 0..nreads): read r's groups are [rgfirst[r], rgfirst[r+1])
__global__ void k_read_groups(const uint64_t *__restrict__ gkey, uint32_t ngroups, int ashift, int nreads,
                              uint32_t *__restrict__ rgfirst)
{ uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g > ngroups) return;
  int64_t cur  = (g < ngroups) ? (int64_t) (gkey[g] >> ashift) : (int64_t) nreads;
  int64_t prev = (g > 0) ? (int64_t) (gkey[g - 1] >> ashift) : -1;
  if (cur > nreads) cur = nreads;
  for (int64_t r = prev + 1; r <= cur; r++) rgfirst[r] = g;
}

// records the new list of read r may need: what it has plus one per chain end of this pass
__global__ void k_dom_need(const uint32_t *__restrict__ olen, const uint32_t *__restrict__ goff,
                           const uint32_t *__restrict__ rgfirst, int nreads, uint32_t *__restrict__ need)
{ int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nreads) return;
  need[r] = olen[r] + (goff[rgfirst[r + 1]] - goff[rgfirst[r]]);
}

struct PassInfo { int32_t bbase, comp; };     // per pass of a batch: first contig of the REF block, orientation

struct DomArgs
{ const OrdChain *ord;
  const uint32_t *goff, *rgfirst;
  const DCand    *oldp;
  const uint32_t *ooff, *olen;
  DCand          *newp;
  const uint32_t *noff;
  uint32_t       *nlen;
  int             nreads;
  int             bbits, passbits;
  PassInfo        pass[64];
  int16_t        *cover;            // nullptr without -p
  const uint32_t *cov_off;
  int             spacing;
  unsigned int   *ctl;              // [0] += candidates added, [1] += removed
  uint32_t       *fresh;            // the records added by this pass that survived it (positions in newp) ...
  unsigned int   *nfresh;           // ... and their number
};

// map.c:1654-1730 for all reads at once: a 16-lane group per read walks the read's chain ends of this pass in the
// reference's order; for each, the list is scanned newest first, 16 records at a time -- the newest record that
// dominates the chain drops it, and the records it dominates are removed up to there (map.c:1668-1713: "break"
// leaves the deletions made so far in place); a chain that survives the scan is appended.  The list (oldest
// first, tombstones squeezed out) is first copied from the other pool.
constexpr int DOM_G = 16;
constexpr int DOM_CAP = 96;        // records of a read's list whose scan fields are kept in LDS

// Round 4: (1) a read's chain ends are fetched SIXTEEN at a time (score and A interval, a lane each) and handed round the
// group with lane reads -- one end after the other cost a memory round trip each, 116 per read and pass, which was the
// kernel's time (0.5 ms); (2) what the scan of the list looks at -- score, A interval, flags of its first DOM_CAP records --
// lives in LDS (records beyond are read where they are; the records themselves stay in the pool); (3) the records this
// pass added that survived it are listed here (k_fresh_list walked every read's list again for them).
__global__ __launch_bounds__(256) void k_dominate(DomArgs D)
{ const int      lane  = threadIdx.x & (DOM_G - 1);
  const int      sbase = (threadIdx.x & 63) & ~(DOM_G - 1);
  const uint64_t lt    = (1ull << lane) - 1ull;
  auto gballot = [&](bool p) -> uint32_t { return (uint32_t) ((__ballot(p) >> sbase) & 0xffffull); };
  const uint64_t bmask = (D.bbits >= 64) ? ~0ull : ((1ull << D.bbits) - 1ull);
  const uint64_t qmask = (1ull << D.passbits) - 1ull;
  __shared__ int32_t  l_sc[256 / DOM_G][DOM_CAP], l_af[256 / DOM_G][DOM_CAP], l_al[256 / DOM_G][DOM_CAP];
  __shared__ uint32_t l_fl[256 / DOM_G][DOM_CAP];
  const int gi = threadIdx.x / DOM_G;
  const int stride = gridDim.x * (256 / DOM_G);
  // (the four groups of a wavefront make the same number of turns: the fresh records of a turn take ONE atomic)
  const int wfirst = blockIdx.x * (256 / DOM_G) + (threadIdx.x >> 6) * (64 / DOM_G);
  for (int ar0 = wfirst; ar0 < D.nreads; ar0 += stride)
    { const int  ar = ar0 + (gi & (64 / DOM_G - 1));
      const bool mine = ar < D.nreads;
      const uint32_t ooff = mine ? D.ooff[ar] : 0u, olen = mine ? D.olen[ar] : 0u, nb = mine ? D.noff[ar] : 0u;
      const uint32_t s0 = mine ? D.goff[D.rgfirst[ar]] : 0u, s1 = mine ? D.goff[D.rgfirst[ar + 1]] : 0u;
      uint32_t n = 0;
      for (uint32_t base = 0; base < olen; base += DOM_G)
        { const uint32_t i = base + lane;
          DCand e;
          bool  keep = false;
          if (i < olen)
            { e = D.oldp[ooff + i];
              keep = !(e.flags & DC_TOMB);
            }
          const uint32_t m = gballot(keep);
          if (keep)
            { const uint32_t at = n + (uint32_t) __popc(m & (uint32_t) lt);
              D.newp[nb + at] = e;
              if (at < (uint32_t) DOM_CAP) { l_sc[gi][at] = e.score; l_af[gi][at] = e.afirst; l_al[gi][at] = e.alast; l_fl[gi][at] = e.flags; }
            }
          n += (uint32_t) __popc(m);
        }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      const uint32_t n0 = n;                                    // what this pass adds lies behind
      uint32_t added = 0, removed = 0;
      int16_t *cov = (D.cover && mine) ? D.cover + D.cov_off[ar] : nullptr;
      for (uint32_t c0 = s0; c0 < s1; c0 += DOM_G)
        { // sixteen ends: score and A interval, a lane each
          OrdChain Se;                                           // (the whole record: the lane that holds it appends it)
          Se.cost = 0; Se.ab = 0; Se.ae = 0; Se.bb = 0; Se.be = 0; Se.end = 0; Se.gkey = 0;
          if (c0 + lane < s1) Se = D.ord[c0 + lane];
          const int e_cost = Se.cost, e_ab = Se.ab, e_ae = Se.ae;
          const uint32_t nc = min((uint32_t) DOM_G, s1 - c0);
          for (uint32_t k = 0; k < nc; k++)
            { const int S_cost = __shfl(e_cost, (int) k, DOM_G), S_ab = __shfl(e_ab, (int) k, DOM_G), S_ae = __shfl(e_ae, (int) k, DOM_G);
              if (cov != nullptr && lane == 0)                           // map.c:1654-1666
                { const int tb = S_ab / D.spacing, te = (S_ae - 1) / D.spacing + 1;
                  const int cb = cov[tb], ce = cov[te];
                  if (cb < 0x7fff && ce > -0xffff)
                    { cov[tb] = (int16_t) (cb + 1);
                      cov[te] = (int16_t) (ce - 1);
                    }
                }
              bool dropped = false;
              for (int hi = (int) n; hi > 0; hi -= DOM_G)
                { const int i = hi - 1 - lane;             // lane 0 looks at the newest record of the chunk
                  int      sc = 0, af = 0, al = 0;
                  uint32_t fl = DC_TOMB;
                  if (i >= 0)
                    { if (i < DOM_CAP) { sc = l_sc[gi][i]; af = l_af[gi][i]; al = l_al[gi][i]; fl = l_fl[gi][i]; }
                      else
                        { const DCand *e = D.newp + nb + i;
                          sc = e->score; af = e->afirst; al = e->alast; fl = e->flags;
                        }
                    }
                  const bool live = !(fl & DC_TOMB);
                  const bool dhc  = live && af < S_ab + MIN_PIECE && al > S_ae - MIN_PIECE;
                  const bool chd  = live && S_ab < af + MIN_PIECE && S_ae > al - MIN_PIECE;
                  const bool dom  = dhc && (.9 * (double) sc >= (double) S_cost);
                  bool       del  = chd && !dom && ((double) sc <= .9 * (double) S_cost);
                  const uint32_t md = gballot(dom);
                  if (md != 0)
                    { del = del && lane < (__ffs((int) md) - 1);
                      dropped = true;
                    }
                  if (del)
                    { D.newp[nb + i].flags = fl | DC_TOMB;
                      if (i < DOM_CAP) l_fl[gi][i] = fl | DC_TOMB;
                    }
                  removed += (uint32_t) __popc(gballot(del));
                  if (dropped) break;
                }
              if (!dropped)
                { if ((uint32_t) lane == k)
                    { const OrdChain &S = Se;
                      const uint32_t q = (uint32_t) ((S.gkey >> D.bbits) & qmask);
                      DCand e;
                      e.score = S.cost; e.afirst = S.ab; e.alast = S.ae; e.bfirst = S.bb; e.blast = S.be;
                      e.bread = (int32_t) (S.gkey & bmask) + D.pass[q].bbase;
                      e.end   = S.end;
                      e.joff  = 0; e.jlen = 0;
                      e.flags = DC_FRESH | (D.pass[q].comp ? DC_COMP : 0u);
                      D.newp[nb + n] = e;
                      if (n < (uint32_t) DOM_CAP) { l_sc[gi][n] = e.score; l_af[gi][n] = e.afirst; l_al[gi][n] = e.alast; l_fl[gi][n] = e.flags; }
                    }
                  n += 1;
                  added += 1;
                }
              // the next end reads what the lanes stored: LDS is served in order within the wavefront; the records beyond
              // DOM_CAP are read from the pool, behind a fence
              __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
              __builtin_amdgcn_wave_barrier();
              if (n > (uint32_t) DOM_CAP) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            }
        }
      // the records this pass added that are still there: listed for the Jump lists (one atomic per wavefront and turn)
      { uint32_t mycnt = 0;
        for (uint32_t base = n0; base < n; base += DOM_G)
          { const uint32_t i = base + lane;
            uint32_t fl = DC_TOMB;
            if (i < n) fl = (i < (uint32_t) DOM_CAP) ? l_fl[gi][i] : D.newp[nb + i].flags;
            mycnt += (uint32_t) __popc(gballot((fl & DC_FRESH) && !(fl & DC_TOMB)));
          }
        // (mycnt is the group's count in every lane of the group)
        const uint32_t g0c = (uint32_t) __builtin_amdgcn_readlane((int) mycnt, 0), g1c = (uint32_t) __builtin_amdgcn_readlane((int) mycnt, 16),
                       g2c = (uint32_t) __builtin_amdgcn_readlane((int) mycnt, 32), g3c = (uint32_t) __builtin_amdgcn_readlane((int) mycnt, 48);
        uint32_t wbase = 0;
        if ((threadIdx.x & 63) == 0 && g0c + g1c + g2c + g3c > 0) wbase = atomicAdd(D.nfresh, g0c + g1c + g2c + g3c);
        wbase = (uint32_t) __builtin_amdgcn_readfirstlane((int) wbase);
        const int gw = (threadIdx.x & 63) / DOM_G;
        uint32_t at = wbase + (gw > 0 ? g0c : 0u) + (gw > 1 ? g1c : 0u) + (gw > 2 ? g2c : 0u);
        for (uint32_t base = n0; base < n; base += DOM_G)
          { const uint32_t i = base + lane;
            uint32_t fl = DC_TOMB;
            if (i < n) fl = (i < (uint32_t) DOM_CAP) ? l_fl[gi][i] : D.newp[nb + i].flags;
            const bool f = (fl & DC_FRESH) && !(fl & DC_TOMB);
            const uint32_t m = gballot(f);
            if (f) D.fresh[at + (uint32_t) __popc(m & (uint32_t) lt)] = nb + i;
            at += (uint32_t) __popc(m);
          }
      }
      if (lane == 0 && mine)
        { D.nlen[ar] = n;
          if (added)   atomicAdd(D.ctl, added);
          if (removed) atomicAdd(D.ctl + 1, removed);
        }
    }
}

// Compressed Jump displacements of a kept chain (chain_length, map.c:1243-1260, and the encoding
// loop map.c:1741-1759): links on one diagonal closer than 100 bp to the last kept point are
// skipped.  Pass 0 counts (jlen[c], 0 beyond the list), pass 1 writes the pairs at jbase + joff[c] of the jump pool
// and completes the candidate's record.
// One WAVEFRONT per surviving fresh candidate (round 4; a thread per candidate chased the links one memory round trip at
// a time: 0.5 ms per launch for the 400 links of a read's true locus).  A chain's links point a few hits back -- the hit
// before, in a plain chain --, so the wavefront keeps a WINDOW of 64 consecutive hits (words and links, two coalesced
// loads) in its lanes, follows the links inside it with lane reads, and fetches a new window only when a link leaves it.
constexpr int CJ_BLOCK = 256;
__global__ __launch_bounds__(CJ_BLOCK) void k_cand_jumps(const uint64_t *__restrict__ ab, const uint32_t *__restrict__ from,
                                                         DCand *__restrict__ pool, const uint32_t *__restrict__ fresh,
                                                         const unsigned int *__restrict__ nfresh, uint32_t cap,
                                                         uint32_t *__restrict__ jlen, const uint32_t *__restrict__ joff,
                                                         uint16_t *__restrict__ jout, uint32_t jbase, int write)
{ const int lane = threadIdx.x & 63;
  const uint32_t nwaves = (gridDim.x * CJ_BLOCK) >> 6, wave = (blockIdx.x * CJ_BLOCK + threadIdx.x) >> 6;
  const uint32_t nf = min(*nfresh, cap);
  if (!write)
    for (uint32_t c = nf + blockIdx.x * CJ_BLOCK + threadIdx.x; c < cap; c += gridDim.x * CJ_BLOCK) jlen[c] = 0;
  for (uint32_t c = wave; c < nf; c += nwaves)
    { const uint32_t slot = fresh[c];
      uint32_t x = pool[slot].end;
      uint32_t base = (x >= 63u) ? x - 63u : 0u;                 // the window: hits base .. base + 63 (x is always inside)
      uint64_t wv = ab[base + lane];
      uint32_t fv = from[base + lane];
      int ax = (int) (uint32_t) wave_lane_u64(wv, (int) (x - base)), bx = (int) (wave_lane_u64(wv, (int) (x - base)) >> 32);
      uint32_t y = (uint32_t) __builtin_amdgcn_readlane((int) fv, (int) (x - base));
      uint32_t cnt = 0;
      uint32_t *o = write ? (uint32_t *) (jout + 2 * ((size_t) jbase + joff[c])) : nullptr;      // (a pair is one 32-bit word)
      while (y != NIL)
        { if (y < base)                                          // (links point backwards)
            { base = (y >= 63u) ? y - 63u : 0u;
              wv = ab[base + lane];
              fv = from[base + lane];
            }
          const uint64_t ky = wave_lane_u64(wv, (int) (y - base));
          const uint32_t fy = (uint32_t) __builtin_amdgcn_readlane((int) fv, (int) (y - base));
          const int ay = (int) (uint32_t) ky, by = (int) (ky >> 32);
          const int da = ax - ay, db = bx - by;
          if (!(da == db && da < 100))
            { if (write && lane == 0) o[cnt] = (uint32_t) (uint16_t) da | ((uint32_t) (uint16_t) db << 16);
              cnt += 1;
              x = y; ax = ay; bx = by;
            }
          y = fy;
        }
      if (lane == 0)
        { if (!write) jlen[c] = cnt;
          else
            { pool[slot].joff  = jbase + joff[c];
              pool[slot].jlen  = cnt;
              pool[slot].flags &= ~DC_FRESH;
            }
        }
    }
}

// groups by size, in any order: more than `small` and fewer than `big` hits -> mlist (the 16-lane launch with the
// window in LDS), at least `big` -> list (one wavefront each), at least `pile` -> plist (one workgroup each).
// One atomic per wavefront and list.
__global__ void k_big_groups(const uint32_t *__restrict__ gstart, uint32_t ngroups, uint32_t small, uint32_t big,
                             uint32_t pile, uint32_t *__restrict__ mlist, unsigned int *__restrict__ nmlist,
                             uint32_t *__restrict__ list, unsigned int *__restrict__ nlist,
                             uint32_t *__restrict__ plist, unsigned int *__restrict__ nplist)
{ const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  const int      lane = threadIdx.x & 63;
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint32_t sz = (g < ngroups) ? gstart[g + 1] - gstart[g] : 0;
  const bool isp = sz >= pile, isb = !isp && sz >= big, ism = !isp && !isb && sz > small;
  const uint64_t mp = __ballot(isp), mb = __ballot(isb), mm = __ballot(ism);
  uint32_t bp = 0, bb = 0, bm = 0;
  if (lane == 0)
    { if (mp) bp = atomicAdd(nplist, (uint32_t) __popcll(mp));
      if (mb) bb = atomicAdd(nlist, (uint32_t) __popcll(mb));
      if (mm) bm = atomicAdd(nmlist, (uint32_t) __popcll(mm));
    }
  bp = __shfl(bp, 0); bb = __shfl(bb, 0); bm = __shfl(bm, 0);
  if (isp)
    { plist[bp + (uint32_t) __popcll(mp & lt)] = g;
      atomicAdd(nplist + 2, sz);                 // (counts[3]: hits in piles, for the stage timer's algorithmic bytes)
    }
  if (isb) list[bb + (uint32_t) __popcll(mb & lt)] = g;
  if (ism) mlist[bm + (uint32_t) __popcll(mm & lt)] = g;
}

// small groups four to a wavefront (live window in LDS), big ones and windows that outgrew LDS one per
// wavefront, repeat piles one per workgroup; dmh_test_option("chain_g") / DMH_CHAIN_G force one group width
// for everything below the pile threshold, "chain_pile" / DMH_CHAIN_PILE move that threshold (0 = no piles)
constexpr uint32_t BIG_GROUP = 1024;
constexpr uint32_t SIMPLE_MAX = 4096;

constexpr uint32_t SMALL_CAP = 64;       // k_chain_small<16, 64>

// the size classes of the chain kernels (before the groups exist: k_read_sort counts by them)
void chain_group_limits(MergeResult &R)
{ State &st = S();
  static const int env_g    = getenv("DMH_CHAIN_G") ? atoi(getenv("DMH_CHAIN_G")) : 0;
  static const int env_pile = getenv("DMH_CHAIN_PILE") ? atoi(getenv("DMH_CHAIN_PILE")) : 4096;
  static const int env_small = getenv("DMH_CHAIN_SMALL") ? atoi(getenv("DMH_CHAIN_SMALL")) : (int) SMALL_CAP;
  const int chain_g = st.opt_chain_g ? st.opt_chain_g : env_g;
  R.pile_min = (st.opt_chain_pile >= 0) ? (uint32_t) st.opt_chain_pile : (uint32_t) env_pile;
  // with k_chain_simple on, groups of up to SIMPLE_MAX hits get their chance there -- the true locus of a 30 kb read is
  // a plain chain of 1,200 hits -- and k_chain_lane hands the ones of BIG_GROUP hits and more that are left to the
  // wavefront-per-group launch; without it those groups are listed for that launch right away
  static const int env_lane0 = getenv("DMH_CHAIN_LANE") ? atoi(getenv("DMH_CHAIN_LANE")) : 1;
  static const int env_simple0 = getenv("DMH_CHAIN_SIMPLE") ? atoi(getenv("DMH_CHAIN_SIMPLE")) : 1;
  const bool simple_on = (st.opt_chain_lane >= 0 ? st.opt_chain_lane : env_lane0) != 0 &&
                         (st.opt_chain_simple >= 0 ? st.opt_chain_simple : env_simple0) != 0 &&
                         (st.opt_chain_small >= 0 ? st.opt_chain_small : env_small) != 0;
  R.small_hi = (chain_g == 0) ? (simple_on ? SIMPLE_MAX : BIG_GROUP) : 0xffffffffu;
  if (R.pile_min > 0) R.small_hi = std::min(R.small_hi, R.pile_min);
  // groups of up to small_cap hits go to the all-LDS kernel (not when a group width is forced for everything)
  int sc = (st.opt_chain_small >= 0) ? st.opt_chain_small : env_small;
  if (chain_g != 0) sc = 0;
  R.small_cap = (uint32_t) std::min<int>(std::max(sc, 0), (int) SMALL_CAP);
  if (R.small_cap >= R.small_hi) R.small_cap = R.small_hi - 1;
  // one lane per group (k_chain_lane) for everything below small_hi instead; "chain_lane" test option / DMH_CHAIN_LANE: 0 off
  static const int env_lane = getenv("DMH_CHAIN_LANE") ? atoi(getenv("DMH_CHAIN_LANE")) : 1;
  const int ln = (st.opt_chain_lane >= 0) ? st.opt_chain_lane : env_lane;
  R.lane_max = (ln != 0 && chain_g == 0 && sc != 0) ? std::min<uint32_t>(R.small_hi - 1, 0xffffffu) : 0;
}

// known (may be null): {groups of small_hi .. pile_min - 1 hits, piles, hits in piles} counted already (k_read_sort;
// only with the lane kernel on, when no group is "medium" at this point): the lists are made, nothing is fetched
void chain_classify_groups(MergeResult &R, const uint32_t *known)
{ chain_group_limits(R);
  const uint32_t list_from = R.lane_max > 0 ? R.lane_max : R.small_cap;      // smaller groups are not listed
  R.biglist.ensure((size_t) R.ngroups + 64, "big groups");
  R.pilelist.ensure((size_t) R.ngroups + 64, "repeat piles");
  R.medlist.ensure((size_t) R.ngroups + 64, "medium groups");
  R.gcounts.ensure(4, "group list lengths");
  HIP_CHECK(hipMemsetAsync(R.gcounts.p, 0, 4 * sizeof(uint32_t), stream()));
  R.npiles = 0;
  R.nbig   = 0;
  R.nmed   = 0;
  R.classified = true;
  if (R.ngroups == 0) return;
  hipLaunchKernelGGL(k_big_groups, dim3(ceil_div(R.ngroups, 256)), dim3(256), 0, stream(),
                     (const uint32_t *) R.gstart.p, (uint32_t) R.ngroups, list_from, R.small_hi,
                     R.pile_min > 0 ? R.pile_min : 0xffffffffu, R.medlist.p, R.gcounts.p + 2, R.biglist.p, R.gcounts.p,
                     R.pilelist.p, R.gcounts.p + 1);
  HIP_CHECK(hipGetLastError());
  if (known != nullptr && R.lane_max > 0)
    { R.nbig = known[0]; R.npiles = known[1]; R.nmed = 0; R.pile_hits = known[2];
      return;
    }
  uint32_t cnt[4] = {0, 0, 0, 0};
  copy_d2h(cnt, R.gcounts.p, 4 * sizeof(uint32_t));
  R.nbig   = cnt[0];
  R.npiles = cnt[1];
  R.nmed   = cnt[2];
  R.pile_hits = cnt[3];
}

// ---------------------------------------------------------------------------------------
//  Host: per-read candidate lists
// ---------------------------------------------------------------------------------------

#ifdef DMH_CHAIN_PROFILE
}  // namespace dmh
extern "C" void dmh_lane_profile(unsigned long long *out, int reset)
{ HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(dmh::g_lprof), sizeof(unsigned long long) * 24));
  if (reset)
    { unsigned long long z[24] = {0};
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(dmh::g_lprof), z, sizeof(z)));
    }
}
extern "C" void dmh_chain_profile(unsigned long long *out, int reset)
{ HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(dmh::g_cprof), sizeof(unsigned long long) * 16));
  if (reset)
    { unsigned long long z[16] = {0};
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(dmh::g_cprof), z, sizeof(z)));
    }
}
namespace dmh {
#endif

static size_t g_jump_hint = 0;       // Jump pairs the largest READS block so far ended with

void chain_reset_reads(const DAZZ_DB *ablock)
{ State &st = S();
  // The block's store: its own when it has one (a restart: what it had collected is dropped), else a free one, else
  // the one of the block before the current one (a block that never reached Reporter and never will: at most ONE
  // block may wait for its Reporter call while the next one is being filtered)
  int slot = st.cand_slot_of(ablock);
  if (slot < 0)
    for (int i = 0; i < 2 && slot < 0; i++)
      if (st.cand_slot[i].block == nullptr) slot = i;
  if (slot < 0) slot = st.cand_cur ^ 1;
  filter_discard_block(slot);               // hits of a block that never reached Reporter
  st.cand_cur = slot;
  CandStore &cs = st.cands();
  const int n = ablock->nreads;
  st.cur_reads = ablock;
  st.reads.clear();
  st.next_cand_id = 0;
  st.tfilt = 0;
  cs.block  = ablock;
  cs.nreads = n;
  cs.cur    = 0;
  cs.total_len = 0;
  { // the jump pool starts at the size the block before this one ended with: it grows by copying (finish_step_c)
    if (cs.jtop > g_jump_hint) g_jump_hint = cs.jtop;
    if (g_jump_hint > 0) cs.jumps.ensure(2 * (g_jump_hint + g_jump_hint / 8) + 1024, "jump pool");
  }
  cs.jtop   = 0;
  cs.mirror_valid = false;
  cs.mirror_jumps = false;
  cs.seg_off[0].ensure((size_t) n + 1, "candidate list offsets");
  cs.seg_len[0].ensure((size_t) n + 1, "candidate list lengths");
  cs.ctl.ensure(16, "candidate counters");
  HIP_CHECK(hipMemsetAsync(cs.seg_off[0].p, 0, sizeof(uint32_t) * ((size_t) n + 1), stream()));
  HIP_CHECK(hipMemsetAsync(cs.seg_len[0].p, 0, sizeof(uint32_t) * ((size_t) n + 1), stream()));
  if (PROFILE)
    { cs.h_cov_off.resize((size_t) n + 1);
      uint32_t tot = 0;
      for (int i = 0; i < n; i++)
        { cs.h_cov_off[i] = tot;
          tot += (uint32_t) ((ablock->reads[i].rlen - 1) / SPACING + 2);
        }
      cs.h_cov_off[n] = tot;
      cs.cov_off.ensure((size_t) n + 1, "cover offsets");
      cs.cover.ensure((size_t) tot + 1, "cover deltas");
      copy_h2d(cs.cov_off.p, cs.h_cov_off.data(), sizeof(uint32_t) * ((size_t) n + 1));
      HIP_CHECK(hipMemsetAsync(cs.cover.p, 0, sizeof(int16_t) * ((size_t) tot + 1), stream()));
    }
  else
    cs.h_cov_off.clear();
  for (int i = 0; i < n; i++)
    ablock->reads[i].coff = -1;
}

struct FinishState
{ int       stage = 0;             // steps done
  uint32_t *h_pin = nullptr;       // 32 page-locked words: [0] chain ends, [1] its tag; [8..15] the candidate counters, [16] their tag
  uint32_t  seq = 0;               // tag of the step under way: k_post_words writes it behind the words (the host polls it)
  hipEvent_t ev = nullptr;         // ... and the event behind that kernel (errors surface here)
  uint32_t  nsum = 0;
  int64_t   result = 0;
};


// Everything the chain stage of one batch keeps between chain_submit and chain_finish
struct ChainCtx
{ DevBuf<int32_t>  cost, bestcost;
  DevBuf<uint32_t> from, rootof, best, state, tmp, ctl, gcnt, gbase, pileretry;
  DevBuf<uint4>    livebuf;
  DevBuf<ChainSummary> out;
  ChainArgs  A;
  hipEvent_t piles_done = nullptr;
  hipEvent_t submitted = nullptr;    // the batch's chain kernels are queued on the main stream up to here
  hipEvent_t lane_done = nullptr, tail_done = nullptr;     // ... and its tail launches on the tail stream (has_tail)
  bool       has_tail = false;
  bool       has_piles = false, all64 = false;
  // chain_finish: buffers that live from one of its steps to the next, and where it stands
  DevBuf<uint32_t> goff, fresh, jl;
  DevBuf<uint32_t> gdone;            // groups chained by k_chain_simple
  bool       prefilled = false;      // fused route (dmh_readsort.h): from, out, gcnt, gbase and gdone hold the per-read kernel's
                                     //   results when chain_submit is called -- not to be cleared, dropped or made again
  FinishState *fin = nullptr;
};

// Device words -> page-locked host memory, a tag behind them.  The host polls the tag: hipEventSynchronize on the
// finish stream came back 1.6 ms after the copy had landed whenever the main stream was busy (r03 traces: the step after
// the wait was queued the moment the main stream ran dry).
__global__ void k_post_words(volatile uint32_t *host, const uint32_t *__restrict__ src, int n, int tag_at, uint32_t tag)
{ if ((int) threadIdx.x < n) host[threadIdx.x] = src[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) host[tag_at] = tag;
}

static void finish_wait_tag(FinishState &F, int tag_at)
{ static const bool use_event = getenv("DMH_FINISH_EVENT_WAIT") != nullptr;
  if (use_event) { HIP_CHECK(hipEventSynchronize(F.ev)); return; }
  uint64_t spins = 0;
  const bool yield = host_crowded();
  while (__atomic_load_n(&F.h_pin[tag_at], __ATOMIC_ACQUIRE) != F.seq)
    { __builtin_ia32_pause();
      if (yield && (spins & 0xffu) == 0xffu) sched_yield();
      if ((++spins & 0xfffffu) == 0)
        { const hipError_t e = hipEventQuery(F.ev);             // (a fault on the device must not leave the host spinning)
          if (e == hipErrorNotReady) (void) hipGetLastError();
          else if (e != hipSuccess) HIP_CHECK(e);
          else
            { // the kernel that posts the tag has completed: the tag must be visible (coherent page-locked memory).  If it
              // is not -- non-coherent host memory (HIP_HOST_COHERENT=0), or a stale sequence number -- say so instead of
              // spinning for ever
              __atomic_thread_fence(__ATOMIC_SEQ_CST);
              if (__atomic_load_n(&F.h_pin[tag_at], __ATOMIC_ACQUIRE) == F.seq) break;
              HIP_CHECK(hipEventSynchronize(F.ev));
              if (__atomic_load_n(&F.h_pin[tag_at], __ATOMIC_ACQUIRE) != F.seq)
                fatal("chain_finish: the device has finished but its counters never arrived (tag %u, expected %u): "
                      "page-locked memory is not coherent (HIP_HOST_COHERENT=0?)", F.h_pin[tag_at], F.seq);
              break;
            }
        }
    }
}

static FinishState &fin_of(ChainCtx &C)
{ if (C.fin == nullptr)
    { C.fin = new FinishState();
      HIP_CHECK(hipHostMalloc((void **) &C.fin->h_pin, 32 * sizeof(uint32_t), hipHostMallocCoherent));
      memset(C.fin->h_pin, 0, 32 * sizeof(uint32_t));
      HIP_CHECK(hipEventCreateWithFlags(&C.fin->ev, hipEventDisableTiming));
    }
  return *C.fin;
}

void chain_release(MergeResult &R)
{ if (R.cc == nullptr) return;
  ChainCtx &C = *R.cc;
  C.cost.release(); C.bestcost.release();
  C.from.release(); C.rootof.release(); C.best.release(); C.state.release(); C.tmp.release();
  C.gcnt.release(); C.gbase.release(); C.pileretry.release();
  C.livebuf.release();
  C.out.release();
  C.goff.release(); C.fresh.release(); C.jl.release(); C.gdone.release();
}

// A batch in flight is dropped (filter_discard): its pile kernel may still be writing the chain arrays on the pile
// stream, which join_side_streams does not cover -- whatever the main stream does next comes after it
void chain_abandon(MergeResult &R)
{ if (R.cc != nullptr && R.cc->fin != nullptr) { R.cc->fin->stage = 0; R.cc->fin->result = 0; }
  if (R.cc != nullptr && R.cc->has_tail)
    { HIP_CHECK(hipStreamWaitEvent(main_stream(), R.cc->tail_done, 0));
      if (finish_stream() != main_stream()) HIP_CHECK(hipStreamWaitEvent(finish_stream(), R.cc->tail_done, 0));
      R.cc->has_tail = false;
    }
  if (R.cc == nullptr || !R.cc->has_piles) return;
  HIP_CHECK(hipStreamWaitEvent(main_stream(), R.cc->piles_done, 0));
  if (finish_stream() != main_stream()) HIP_CHECK(hipStreamWaitEvent(finish_stream(), R.cc->piles_done, 0));
  R.cc->has_piles = false;
}

static ChainCtx *g_alloc_ctx = nullptr;      // chain_submit is allocating this batch's buffers (none of them used yet)

bool chain_drop_unused()
{ if (g_alloc_ctx == nullptr) return false;
  ChainCtx &C = *g_alloc_ctx;
  bool got = C.cost.p || C.rootof.p || C.best.p || C.bestcost.p || C.livebuf.p || C.state.p || C.tmp.p;
  C.cost.release(); C.bestcost.release();
  C.rootof.release(); C.best.release(); C.state.release(); C.tmp.release();
  C.livebuf.release();
  if (!C.prefilled)                    // (fused route: these hold results already)
    { got = got || C.from.p || C.out.p;
      C.from.release(); C.out.release(); C.gcnt.release(); C.gbase.release();
    }
  return got;
}

bool chain_simple_wanted(const MergeResult &R)
{ static const int env_simple = getenv("DMH_CHAIN_SIMPLE") ? atoi(getenv("DMH_CHAIN_SIMPLE")) : 1;
  const State &st = S();
  return R.lane_max > 0 && (st.opt_chain_simple >= 0 ? st.opt_chain_simple : env_simple) != 0;
}

// Fused route (dmh_readsort.h): the per-read kernel writes links and chain ends, k_cluster_compact the per-group
// counters -- both before chain_submit.  The buffers are the batch's chain buffers, made here.
void chain_prepare_hits(MergeResult &R, uint32_t **from, ChainSummary **out)
{ if (R.cc == nullptr) R.cc = new ChainCtx();
  ChainCtx &C = *R.cc;
  C.from.ensure(R.nhits, "chain from");
  C.out.ensure(R.nhits / HITMIN + 64, "chain summaries");
  C.prefilled = true;
  *from = C.from.p;
  *out  = C.out.p;
}

void chain_prepare_groups(MergeResult &R, uint32_t **gcount, uint32_t **gbase, uint32_t **gdone)
{ ChainCtx &C = *R.cc;
  C.gcnt.ensure((size_t) R.ngroups + 1, "chain ends per group");
  C.gbase.ensure((size_t) R.ngroups + 1, "first chain end per group");
  C.gdone.ensure((size_t) R.ngroups + 1, "groups chained without a walk");
  *gcount = C.gcnt.p; *gbase = C.gbase.p; *gdone = C.gdone.p;
}

void chain_submit(MergeResult &R, const DAZZ_DB *ablock)
{ State &st = S();
  const Params &pm = st.params;
  if (st.cand_slot[R.cslot].block != ablock || st.cand_slot[R.cslot].nreads != ablock->nreads)
    fatal("Match_Filter: first call for a READS block must have start != 0 (map.c:1574)");
  if (R.cc == nullptr) R.cc = new ChainCtx();
  ChainCtx &C = *R.cc;
  DevBuf<int32_t>  &cost = C.cost, &bestcost = C.bestcost;
  DevBuf<uint32_t> &from = C.from, &rootof = C.rootof, &best = C.best, &state = C.state, &tmp = C.tmp, &ctl = C.ctl;
  DevBuf<uint32_t> &gcnt = C.gcnt, &gbase = C.gbase, &pileretry = C.pileretry;
  DevBuf<uint4>    &livebuf = C.livebuf;
  DevBuf<ChainSummary> &out = C.out;

  const int64_t nhits = R.nhits;
  // (Nothing has used these buffers yet: when one of the allocations runs out of memory the reclaim may drop the ones
  //  already made -- chain_drop_unused: finishing the batches in flight takes memory itself -- and the round is made again.)
  g_alloc_ctx = &C;
  for (int round = 0; round < 4; round++)
    { cost.ensure(nhits, "chain cost");
      from.ensure(nhits, "chain from");
      rootof.ensure(nhits, "chain origin");
      best.ensure(nhits, "chain best end");
      bestcost.ensure(nhits, "chain best score");
      livebuf.ensure(nhits + 1, "chain live set");
      state.ensure(nhits, "chain state");
      tmp.ensure(nhits, "chain scratch");
      ctl.ensure(8, "chain counters");
      out.ensure(nhits / HITMIN + 64, "chain summaries");
      gcnt.ensure((size_t) R.ngroups + 1, "chain ends per group");
      gbase.ensure((size_t) R.ngroups + 1, "first chain end per group");
      if (cost.p && from.p && rootof.p && best.p && bestcost.p && livebuf.p && state.p && tmp.p && ctl.p && out.p && gcnt.p && gbase.p)
        break;
      if (round == 3)
        fatal("Match_Filter: out of device memory for the chain buffers of a pass of %lld hits (the allocations keep evicting each other)",
              (long long) nhits);
    }
  g_alloc_ctx = nullptr;
  const bool prefilled = C.prefilled;       // (this batch's; the flag is the ring slot's and is set again by the next fused pass)
  C.prefilled = false;
  HIP_CHECK(hipMemsetAsync(ctl.p, 0, 8 * sizeof(uint32_t), stream()));
  if (!prefilled) HIP_CHECK(hipMemsetAsync(gcnt.p, 0, sizeof(uint32_t) * ((size_t) R.ngroups + 1), stream()));

  ChainArgs &A = C.A;
  A.ab = R.sab;
  A.gstart = R.gstart.p;
  A.ngroups = (uint32_t) R.ngroups;
  A.nhits = (uint32_t) nhits;
  A.pmask = ((uint64_t) 1 << R.pbits) - 1;
  A.K = pm.Kmer;
  A.hithr = HITMIN * pm.Kmer;
  A.cost = cost.p;
  A.from = from.p;
  A.rootof = rootof.p;
  A.best = best.p;
  A.bestcost = bestcost.p;
  A.live = livebuf.p;
  A.state = state.p;
  A.tmp = tmp.p;
  A.only = nullptr;
  A.nonly = nullptr;
  A.ticket = ctl.p;
  A.out = out.p;
  A.gcount = gcnt.p;
  A.gbase  = gbase.p;
  A.glast  = R.clustered ? R.glast.p : nullptr;
  A.gdone  = nullptr;
  A.wide = nullptr; A.nwide = nullptr; A.walk_max = 0xffffffffu;
  A.pile_cap = PILE_CAP;

  static const int env_g = getenv("DMH_CHAIN_G") ? atoi(getenv("DMH_CHAIN_G")) : 0;
  const int chain_g = st.opt_chain_g ? st.opt_chain_g : env_g;
  if (!R.classified) fatal("chain_and_filter: groups not classified");
  A.retry  = R.biglist.p;        // list of the 64-lane launch: big groups + spilled windows
  A.nretry = R.gcounts.p;
  { StageTimer tm("chain", 16.0 * nhits);
    const bool all64 = (chain_g == 64);
    const int  gsmall = (chain_g == 32) ? 32 : 16;
    if (!all64)
      { const int gpb = 256 / gsmall;
        (void) gpb;
        hipEvent_t piles_done = nullptr;
        if (R.npiles > 0)
          { // the piles first: they are the longest jobs of the pass.  (Launched only when there are any:
            // a workgroup wants nearly all of a CU's LDS and would wait for whatever else holds some.)
            ChainArgs P = A;
            P.ticket = ctl.p + 4;
            P.only   = R.pilelist.p;
            P.nonly  = R.gcounts.p + 1;
            pileretry.ensure((size_t) R.npiles + 64, "spilled piles");
            P.retry  = pileretry.p;
            P.nretry = ctl.p + 6;
            P.pile_cap = (st.opt_pile_cap >= 128 && st.opt_pile_cap < PILE_CAP) ? (uint32_t) st.opt_pile_cap : (uint32_t) PILE_CAP;
            // on a stream of their own, next to the other two launches; piles whose window outgrows LDS are
            // listed for a last 64-lane launch behind everything
            // (round 4: FOUR such streams taken in turn -- a pass's piles are a handful of workgroups that run for 12 ms
            //  alone and 30 next to the other streams' kernels, four passes of a read block have them, and on one stream
            //  those launches stood in a queue that was as long as the read block itself: 126 ms per read block with one
            //  stream, 117 with two or four, 120 with three; DMH_PILE_STREAMS)
            constexpr int MAXPS = 4;
            static hipStream_t pstreams[MAXPS] = {nullptr, nullptr, nullptr, nullptr};
            static hipEvent_t  ev_go = nullptr;
            static int         turn = 0;
            // (on a runtime that runs only four streams side by side four pile streams would sit on every queue the other
            //  streams have: one then)
            static const int   nps = std::max(1, std::min(MAXPS, getenv("DMH_PILE_STREAMS") ? atoi(getenv("DMH_PILE_STREAMS"))
                                                                                             : (hw_queues_seen() >= 8 ? 4 : 1)));
            hipEvent_t &ev_done = C.piles_done;
            if (pstreams[0] == nullptr)
              { for (int q = 0; q < nps; q++) HIP_CHECK(hipStreamCreateWithFlags(&pstreams[q], hipStreamNonBlocking));
                HIP_CHECK(hipEventCreateWithFlags(&ev_go, hipEventDisableTiming));
              }
            hipStream_t pstream = pstreams[turn];
            turn = (turn + 1) % nps;
            if (ev_done == nullptr) HIP_CHECK(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
            HIP_CHECK(hipEventRecord(ev_go, stream()));
            HIP_CHECK(hipStreamWaitEvent(pstream, ev_go, 0));
            { StageTimer t0("chain_piles", 16.0 * R.pile_hits, pstream);
              hipLaunchKernelGGL(k_chain_pile, dim3(std::min<uint32_t>(R.npiles, 256)), dim3(PILE_THREADS), 0, pstream, P);
            }
            HIP_CHECK(hipEventRecord(ev_done, pstream));
            piles_done = ev_done;
          }
        bool tail_from_lane = false;
        if (R.lane_max > 0)
          { // groups that are plain chains need no walk (k_chain_simple: "chain_simple" test option / DMH_CHAIN_SIMPLE = 0: off)
            static const int env_simple = getenv("DMH_CHAIN_SIMPLE") ? atoi(getenv("DMH_CHAIN_SIMPLE")) : 1;
            static const int env_tail0 = getenv("DMH_CHAIN_TAIL") ? atoi(getenv("DMH_CHAIN_TAIL")) : 3;
            bool simple_on_tail = false;
            if (prefilled)
              A.gdone = C.gdone.p;            // (the plain chains are done: dmh_readsort.h step 5)
            else if ((st.opt_chain_simple >= 0 ? st.opt_chain_simple : env_simple) != 0)
              { C.gdone.ensure((size_t) R.ngroups + 1, "groups chained without a walk");
                HIP_CHECK(hipMemsetAsync(C.gdone.p, 0, sizeof(uint32_t) * ((size_t) R.ngroups + 1), stream()));
                A.gdone = C.gdone.p;
                // (the whole chain stage of a dense pass on the tail stream: 150.4 ms per read block; DMH_CHAIN_TAIL=1: from
                //  the lane kernel on, 152.5; = 2: only the launches behind the lane kernel, 158.5; = 0: none, 164.7)
                simple_on_tail = (env_tail0 == 3) && tail_stream() != main_stream() && stream() == main_stream();
                if (simple_on_tail)
                  { if (C.lane_done == nullptr)
                      { HIP_CHECK(hipEventCreateWithFlags(&C.lane_done, hipEventDisableTiming));
                        HIP_CHECK(hipEventCreateWithFlags(&C.tail_done, hipEventDisableTiming));
                      }
                    HIP_CHECK(hipEventRecord(C.lane_done, stream()));
                    HIP_CHECK(hipStreamWaitEvent(tail_stream(), C.lane_done, 0));
                  }
                StreamScope on_tail0(simple_on_tail ? tail_stream() : stream());
                StageTimer t0("chain_simple_groups", 12.0 * nhits);
                const int nbs = (int) std::min<int64_t>((R.ngroups + 15) / 16, 256 * 8);
                hipLaunchKernelGGL(k_chain_simple, dim3(nbs), dim3(256), 0, stream(), A, R.lane_max);
              }
            // groups that fall apart by diagonal (k_chain_classes above; "chain_classes" test option / DMH_CHAIN_CLASSES = 0: off)
            static const int env_classes = getenv("DMH_CHAIN_CLASSES") ? atoi(getenv("DMH_CHAIN_CLASSES")) : 1;
            const bool classes_on = A.gdone != nullptr && (st.opt_chain_classes >= 0 ? st.opt_chain_classes : env_classes) != 0;
            // the tail: from here on the launches walk what k_chain_simple has left (7 % of the hits of a dense pass,
            // the time of each launch the LENGTH of its longest walk) -- on the tail stream, next to the following
            // pass's match kernels
            static const int env_tail = getenv("DMH_CHAIN_TAIL") ? atoi(getenv("DMH_CHAIN_TAIL")) : 3;
            tail_from_lane = (env_tail == 1 || env_tail == 3) && tail_stream() != main_stream() && stream() == main_stream();
            if (tail_from_lane && !simple_on_tail)
              { if (C.lane_done == nullptr)
                  { HIP_CHECK(hipEventCreateWithFlags(&C.lane_done, hipEventDisableTiming));
                    HIP_CHECK(hipEventCreateWithFlags(&C.tail_done, hipEventDisableTiming));
                  }
                HIP_CHECK(hipEventRecord(C.lane_done, stream()));
                HIP_CHECK(hipStreamWaitEvent(tail_stream(), C.lane_done, 0));
              }
            StreamScope on_tail(tail_from_lane ? tail_stream() : stream());
            if (classes_on)
              { StageTimer t0("chain_class_groups", 12.0 * nhits);
                hipLaunchKernelGGL(k_chain_classes<2>, dim3(4096), dim3(256), 0, stream(), A, 2u, std::min<uint32_t>(R.lane_max, 128u));
                if (R.lane_max > 128u)
                  hipLaunchKernelGGL(k_chain_classes<16>, dim3(2048), dim3(256), 0, stream(), A, 129u, std::min<uint32_t>(R.lane_max, 1024u));
              }
            // one lane per group: every group below the wavefront-per-group size tries this first; the ones whose
            // window or tree table overflows are appended to the medium list
            ChainArgs Q = A;
            Q.retry  = R.medlist.p;
            Q.nretry = R.gcounts.p + 2;
            Q.wide   = R.biglist.p;
            Q.nwide  = R.gcounts.p;
            static const uint32_t env_walk = getenv("DMH_WALK_MAX") ? (uint32_t) atoi(getenv("DMH_WALK_MAX")) : BIG_GROUP;
            Q.walk_max = env_walk;
            StageTimer t0("chain_lane_groups", 16.0 * nhits);        // 16 B per hit, read once by the chain stage
            // persistent lanes: exactly as many workgroups as the GPU holds at once (a second round of them would run
            // on a half-empty GPU)
            static int resident = 0;
            if (resident == 0)
              { int per_cu = 0, dev = 0;
                hipDeviceProp_t prop;
                HIP_CHECK(hipGetDevice(&dev));
                HIP_CHECK(hipGetDeviceProperties(&prop, dev));
                HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_chain_lane<LANE_RW, LANE_RT>, LANE_BLOCK, 0));
                resident = std::max(1, per_cu) * std::max(1, prop.multiProcessorCount);
                if (getenv("DMH_STATS")) fprintf(stderr, "dmh: k_chain_lane: %d workgroups of %d per CU\n", per_cu, LANE_BLOCK);
              }
            const int nb = (int) std::min<int64_t>((R.ngroups + LANE_BLOCK - 1) / LANE_BLOCK, resident);
            hipLaunchKernelGGL((k_chain_lane<LANE_RW, LANE_RT>), dim3(nb), dim3(LANE_BLOCK), 0, stream(), Q, R.lane_max);
          }
        else if (R.small_cap > 0)
          { // the all-LDS kernel: groups dealt round-robin, as many lane groups as the GPU holds
            StageTimer t0("chain_lds_groups", 0.0);
            const int nb = (int) std::min<int64_t>((R.ngroups + 15) / 16, 256 * 8);
            hipLaunchKernelGGL((k_chain_small<16, (int) SMALL_CAP>), dim3(nb), dim3(256), 0, stream(), A, R.small_cap);
          }
        // the tail: what follows walks the lists the lane kernel has left -- on the tail stream, behind that kernel
        hipStream_t ts = (R.lane_max > 0) ? tail_stream() : main_stream();
        const bool tail = ts != main_stream() && stream() == main_stream();
        if (tail && !tail_from_lane)
          { if (C.lane_done == nullptr)
              { HIP_CHECK(hipEventCreateWithFlags(&C.lane_done, hipEventDisableTiming));
                HIP_CHECK(hipEventCreateWithFlags(&C.tail_done, hipEventDisableTiming));
              }
            HIP_CHECK(hipEventRecord(C.lane_done, stream()));
            HIP_CHECK(hipStreamWaitEvent(ts, C.lane_done, 0));
          }
        {
        StreamScope on_tail(tail ? ts : stream());
        if (R.nmed > 0 || R.lane_max > 0)
          { // medium groups: window in LDS, a list with its own ticket counter (ctl[3])
            ChainArgs M = A;
            M.size_lo = 0; M.size_hi = 0xffffffffu;
            M.ticket = ctl.p + 3;
            M.only   = R.medlist.p;
            M.nonly  = R.gcounts.p + 2;
            const int gpbm = 256 / gsmall;
            const int nbm = (R.lane_max > 0) ? 256 * 4
                                             : (int) std::min<int64_t>(((int64_t) R.nmed + gpbm - 1) / gpbm, 256 * 8);
            StageTimer t1("chain_small_groups", 0.0);
            if (gsmall == 32) hipLaunchKernelGGL(k_chain<32>, dim3(nbm), dim3(256), 0, stream(), M);
            else              hipLaunchKernelGGL(k_chain<16>, dim3(nbm), dim3(256), 0, stream(), M);
          }
        // the list (its own ticket counter ctl[2]); on read-vs-genome data it is empty
        A.ticket = ctl.p + 2;
        A.only   = R.biglist.p;
        A.nonly  = R.gcounts.p;
        { // one group per wavefront: a handful of big groups on read-vs-random-genome data (plus the windows
          // that outgrew LDS), most groups of the pass on a repeat-rich genome -- size the grid for the list
          const uint32_t lblocks = std::min<uint32_t>(std::max<uint32_t>((R.nbig + 3) / 4 + 64, 256), 256 * 8);
          StageTimer t2("chain_listed_groups", 0.0);
          hipLaunchKernelGGL(k_chain<64>, dim3(lblocks), dim3(256), 0, stream(), A);
        }
        if (tail)
          { HIP_CHECK(hipEventRecord(C.tail_done, stream()));
            C.has_tail = true;
          }
        }
        C.has_piles = (piles_done != nullptr);
      }
    else
      { int nblocks = (int) std::min<int64_t>((R.ngroups + 3) / 4, 256 * 8);
        A.size_lo = 0; A.size_hi = 0xffffffffu;
        hipLaunchKernelGGL(k_chain<64>, dim3(nblocks), dim3(256), 0, stream(), A);
      }
    HIP_CHECK(hipGetLastError());
  }
  if (C.submitted == nullptr) HIP_CHECK(hipEventCreateWithFlags(&C.submitted, hipEventDisableTiming));
  HIP_CHECK(hipEventRecord(C.submitted, stream()));
}

// Runs on the FINISH stream (dmh_common.h): a dozen small, latency-bound kernels and two counter round trips per
// batch that depend on nothing the main stream does meanwhile (the merge, sort and chain kernels of later batches) and
// on which only the next batch's finish and Reporter depend.  The caller (finish_oldest) makes the main stream wait
// for its end before the batch's ring slot is used again.
// In THREE steps, each ending where the host needs a number from the device (chain ends; candidates and Jump pairs):
// the number travels to page-locked memory behind an event, and the next step picks it up.  The filter calls the steps
// at points of a pass where it has just queued work on the main stream (chain_finish_step), so that the host waits
// for the finish stream's small kernels -- slow next to a machine-filling chain kernel -- as little as possible;
// chain_finish runs whatever steps are left in one go.

static void finish_step_a(MergeResult &R)
{ State &st = S();
  ChainCtx &C = *R.cc;
  FinishState &F = fin_of(C);
  StreamScope on_finish(finish_stream());
  if (stream() != main_stream()) HIP_CHECK(hipStreamWaitEvent(stream(), C.submitted, 0));
  if (C.has_tail)
    { HIP_CHECK(hipStreamWaitEvent(stream(), C.tail_done, 0));
      C.has_tail = false;
    }
  ChainArgs A = C.A;
  DevBuf<uint32_t> &ctl = C.ctl, &gcnt = C.gcnt, &pileretry = C.pileretry;
  if (C.has_piles)
    { // piles whose window outgrew the workgroup kernel's LDS: a last 64-lane launch (the piles have had the time of
      // the batches submitted since to finish)
      HIP_CHECK(hipStreamWaitEvent(stream(), C.piles_done, 0));
      A.ticket = ctl.p + 7;
      A.only   = pileretry.p;
      A.nonly  = ctl.p + 6;
      StageTimer t3("chain_spilled_piles", 0.0);
      hipLaunchKernelGGL(k_chain<64>, dim3(std::min<uint32_t>((R.npiles + 3) / 4, 256)), dim3(256), 0, stream(), A);
      HIP_CHECK(hipGetLastError());
      C.has_piles = false;
    }
  // chain ends per group -> first slot of every group's ends in the reference's order, and their number
  const uint32_t ng = (uint32_t) R.ngroups;
  C.goff.ensure((size_t) ng + 2, "chain ends per group");
  exclusive_scan_u32((const uint32_t *) gcnt.p, C.goff.p, ng, st.sort_fin.bsum, C.goff.p + ng);
  F.seq += 1; if (F.seq == 0) F.seq = 1;
  hipLaunchKernelGGL(k_post_words, dim3(1), dim3(64), 0, stream(), (volatile uint32_t *) F.h_pin,
                     (const uint32_t *) (C.goff.p + ng), 1, 1, F.seq);
  HIP_CHECK(hipEventRecord(F.ev, stream()));
  F.stage = 1;
}

static void finish_step_b(MergeResult &R)
{ State &st = S();
  ChainCtx &C = *R.cc;
  FinishState &F = fin_of(C);
  StreamScope on_finish(finish_stream());
  { HostTimer ht("chain_wait_a");
    finish_wait_tag(F, 1);
  }
  const uint32_t nsum = F.h_pin[0];
  F.nsum = nsum;
  DevBuf<uint32_t> &from = C.from, &gcnt = C.gcnt, &gbase = C.gbase;
  DevBuf<ChainSummary> &out = C.out;
  const int64_t nhits = R.nhits;
  const uint32_t ng = (uint32_t) R.ngroups;
  { static const bool stats = getenv("DMH_STATS") != nullptr;
    if (stats)
      { uint32_t c3[3] = {0, 0, 0};
        copy_d2h(c3, R.gcounts.p, 3 * sizeof(uint32_t));
        fprintf(stderr, "dmh: lists after the chain kernels: big %u piles %u medium %u\n", c3[0], c3[1], c3[2]);
      }
    if (stats)
      fprintf(stderr, "dmh: pass hits %lld groups %lld (clustered %d) lds<=%u medium %u big %u piles %u chain ends %u\n",
              (long long) nhits, (long long) R.ngroups, (int) R.clustered, R.small_cap, R.nmed, R.nbig, R.npiles, nsum);
  }
  F.result = 0;
  if (nsum == 0) { F.stage = 3; return; }        // no chain reached the threshold: the lists stay as they are

  // ---- reporting order, domination rule and -p deltas on the device (map.c:1634-1730)
  CandStore &cs = st.cand_slot[R.cslot];
  const int nreads = cs.nreads;
  static DevBuf<OrdChain> ord;
  static DevBuf<uint32_t> rgfirst;
  static DevBuf<ChainSummary> packed;
  static DevBuf<uint2> prange;
  static DevBuf<uint64_t> oword;
  DevBuf<uint32_t> &fresh = C.fresh, &jl = C.jl, &goff = C.goff;
  if (st.opt_oom_finish > 0 && --st.opt_oom_finish == 0)
    simulate_out_of_memory();            // (test: the batch being finished must survive the reclaim, dmh_merge.hip)
  ord.ensure(nsum, "ordered chain ends");
  packed.ensure(nsum, "packed chain ends");
  prange.ensure(nsum, "sibling ranges of the chain ends");
  oword.ensure(nsum, "order words of the chain ends");
  rgfirst.ensure((size_t) nreads + 2, "groups per read");
  fresh.ensure(nsum, "fresh candidates");
  jl.ensure(2 * (size_t) nsum + 4, "jump lengths");
  const int nx = cs.cur ^ 1;
  cs.pool[nx].ensure(cs.total_len + nsum, "candidate lists");
  cs.seg_off[nx].ensure((size_t) nreads + 1, "candidate list offsets");
  cs.seg_len[nx].ensure((size_t) nreads + 1, "candidate list lengths");
  if (cs.pool[cs.cur].p == nullptr) cs.pool[cs.cur].ensure(16, "candidate lists");
  { StageTimer tm("chain_dominate", 40.0 * nsum);
    HIP_CHECK(hipMemsetAsync(cs.ctl.p, 0, 16 * sizeof(unsigned int), stream()));
    hipLaunchKernelGGL(k_pack_summaries, dim3(std::min<uint32_t>(ceil_div(ng, 256), 8192)), dim3(256), 0, stream(),
                       (const ChainSummary *) out.p, ng, (const uint32_t *) gbase.p,
                       (const uint32_t *) gcnt.p, (const uint32_t *) goff.p,
                       R.clustered ? (const uint32_t *) R.cog.p : (const uint32_t *) nullptr,
                       R.clustered ? (const uint32_t *) R.ocfirst.p : (const uint32_t *) nullptr, packed.p, prange.p,
                       oword.p, (const uint32_t *) R.gstart.p, R.clustered ? (const uint32_t *) R.glast.p : (const uint32_t *) nullptr);
    hipLaunchKernelGGL(k_order_summaries, dim3(ceil_div(nsum, 256)), dim3(256), 0, stream(),
                       (const ChainSummary *) packed.p, (const uint2 *) prange.p, (const uint64_t *) oword.p, nsum,
                       (const uint64_t *) R.gkey.p, ord.p);
    hipLaunchKernelGGL(k_read_groups, dim3(ceil_div((int64_t) ng + 1, 256)), dim3(256), 0, stream(),
                       (const uint64_t *) R.gkey.p, ng, R.bbits + R.passbits, nreads, rgfirst.p);
    hipLaunchKernelGGL(k_dom_need, dim3(ceil_div(nreads, 256)), dim3(256), 0, stream(),
                       (const uint32_t *) cs.seg_len[cs.cur].p, (const uint32_t *) goff.p, (const uint32_t *) rgfirst.p,
                       nreads, cs.seg_off[nx].p);
    exclusive_scan_u32(cs.seg_off[nx].p, nreads, st.sort_fin.bsum, cs.seg_off[nx].p + nreads);
    DomArgs D;
    D.ord = ord.p; D.goff = goff.p; D.rgfirst = rgfirst.p;
    D.oldp = cs.pool[cs.cur].p; D.ooff = cs.seg_off[cs.cur].p; D.olen = cs.seg_len[cs.cur].p;
    D.newp = cs.pool[nx].p; D.noff = cs.seg_off[nx].p; D.nlen = cs.seg_len[nx].p;
    D.nreads = nreads;
    D.bbits = R.bbits; D.passbits = R.passbits;
    for (int q = 0; q < R.npass; q++)
      { D.pass[q].bbase = R.pass[q].bbase;
        D.pass[q].comp  = R.pass[q].comp;
      }
    D.cover = PROFILE ? cs.cover.p : nullptr;
    D.cov_off = cs.cov_off.p;
    D.spacing = SPACING;
    D.ctl = cs.ctl.p;
    D.fresh = fresh.p; D.nfresh = cs.ctl.p + 2;
    hipLaunchKernelGGL(k_dominate, dim3(std::min(ceil_div(nreads, 256 / DOM_G), 256 * 8)), dim3(256), 0, stream(), D);
    HIP_CHECK(hipGetLastError());
  }
  { StageTimer tm("chain_jumps", 0.0);
    hipLaunchKernelGGL(k_cand_jumps, dim3(std::min<uint32_t>(ceil_div((int64_t) nsum * 64, CJ_BLOCK), 2048u)), dim3(CJ_BLOCK), 0, stream(),
                       (const uint64_t *) R.sab, (const uint32_t *) from.p, cs.pool[nx].p,
                       (const uint32_t *) fresh.p, (const unsigned int *) (cs.ctl.p + 2), nsum, jl.p,
                       (const uint32_t *) nullptr, (uint16_t *) nullptr, 0u, 0);
    exclusive_scan_u32((const uint32_t *) jl.p, jl.p + nsum + 1, nsum, st.sort_fin.bsum, cs.ctl.p + 3);
    HIP_CHECK(hipMemcpyAsync(cs.ctl.p + 4, cs.seg_off[nx].p + nreads, sizeof(uint32_t), hipMemcpyDeviceToDevice, stream()));
    // added, removed, fresh survivors, jump pairs, records in the new lists
    F.seq += 1; if (F.seq == 0) F.seq = 1;
    hipLaunchKernelGGL(k_post_words, dim3(1), dim3(64), 0, stream(), (volatile uint32_t *) (F.h_pin + 8),
                       (const uint32_t *) cs.ctl.p, 8, 8, F.seq);
    HIP_CHECK(hipEventRecord(F.ev, stream()));
  }
  F.stage = 2;
}

static void finish_step_c(MergeResult &R)
{ State &st = S();
  ChainCtx &C = *R.cc;
  FinishState &F = fin_of(C);
  StreamScope on_finish(finish_stream());
  { HostTimer ht("chain_wait_b");
    finish_wait_tag(F, 16);
  }
  CandStore &cs = st.cand_slot[R.cslot];
  const uint32_t nsum = F.nsum;
  uint32_t h[8];
  for (int i = 0; i < 8; i++) h[i] = F.h_pin[8 + i];
  const int nx = cs.cur ^ 1;
  { StageTimer tm("chain_jumps", 0.0);
    const size_t need = cs.jtop + h[3];
    if (need > cs.jumps.cap / 2)
      { // grow the jump pool, keeping what earlier passes stored
        HostTimer hg("jump_pool_grow");
        DevBuf<uint16_t> nj;
        nj.ensure(2 * (need + need / 2) + 1024, "jump pool");
        if (cs.jtop > 0)
          HIP_CHECK(hipMemcpyAsync(nj.p, cs.jumps.p, sizeof(uint16_t) * 2 * cs.jtop, hipMemcpyDeviceToDevice, stream()));
        HIP_CHECK(hipStreamSynchronize(stream()));
        cs.jumps.release();
        cs.jumps = nj;
      }
    if (h[2] > 0)
      hipLaunchKernelGGL(k_cand_jumps, dim3(std::min<uint32_t>(ceil_div((int64_t) h[2] * 64, CJ_BLOCK), 2048u)), dim3(CJ_BLOCK), 0, stream(),
                         (const uint64_t *) R.sab, (const uint32_t *) C.from.p, cs.pool[nx].p,
                         (const uint32_t *) C.fresh.p, (const unsigned int *) (cs.ctl.p + 2), nsum, C.jl.p,
                         (const uint32_t *) (C.jl.p + nsum + 1), cs.jumps.p, (uint32_t) cs.jtop, 1);
    HIP_CHECK(hipGetLastError());
  }
  cs.jtop += h[3];
  if (cs.jtop > g_jump_hint) g_jump_hint = cs.jtop;
  if (cs.jtop >= 0x7fffffffull) fatal("Match_Filter: Jump lists of one READS block exceed 2^31 pairs");
  cs.cur = nx;
  cs.total_len = h[4];
  cs.mirror_valid = false;
  cs.mirror_jumps = false;
  F.result = (int64_t) h[0] - (int64_t) h[1];
  F.stage = 3;
}

// one step of the batch's finish; true when nothing is left to do (*nfilt += the change in candidates)
bool chain_finish_step(MergeResult &R, int64_t *nfilt)
{ FinishState &F = fin_of(*R.cc);
  if (F.stage == 0) finish_step_a(R);
  else if (F.stage == 1) finish_step_b(R);
  else if (F.stage == 2) finish_step_c(R);
  if (F.stage < 3) return false;
  *nfilt += F.result;
  F.stage = 0;
  F.result = 0;
  return true;
}

bool chain_finish_started(MergeResult &R)
{ return R.cc != nullptr && R.cc->fin != nullptr && R.cc->fin->stage > 0; }

int chain_finish_stage(MergeResult &R)
{ return (R.cc != nullptr && R.cc->fin != nullptr) ? R.cc->fin->stage : 0; }

// true when the batch's next step would not block the host nor park the finish stream behind a repeat pile: the
// number the step starts from has landed (steps b, c), the piles of the batch have been walked (step a)
bool chain_finish_ready(MergeResult &R)
{ if (R.cc == nullptr) return true;
  ChainCtx &C = *R.cc;
  const int stage = chain_finish_stage(R);
  if (stage >= 3 || (stage == 0 && !C.has_piles)) return true;
  if (stage > 0) return __atomic_load_n(&C.fin->h_pin[stage == 1 ? 1 : 16], __ATOMIC_ACQUIRE) == C.fin->seq;
  const hipError_t e = hipEventQuery(C.piles_done);
  if (e == hipErrorNotReady) { (void) hipGetLastError(); return false; }
  HIP_CHECK(e);
  return true;
}

int64_t chain_finish(MergeResult &R, const DAZZ_DB *ablock)
{ (void) ablock;
  int64_t nfilt = 0;
  while (!chain_finish_step(R, &nfilt)) ;
  return nfilt;
}

// ---------------------------------------------------------------------------------------
//  Host view of the device lists (Reporter, stage-level API)
// ---------------------------------------------------------------------------------------
// The candidate lists as they are on the device (no Jump lists: they stay in HBM for the alignment kernel).
size_t cand_fetch_bytes(int slot)
{ State &st = S();
  if (slot < 0) slot = st.cand_cur;
  CandStore &cs = st.cand_slot[slot];
  st.tfilt += filter_flush_block(slot);   // (the passes still waiting in the batch)
  const size_t n = (size_t) cs.nreads;
  size_t b = sizeof(DCand) * (size_t) cs.total_len + 64 + sizeof(uint32_t) * 2 * (n + 1) + 64;
  if (!cs.h_cov_off.empty()) b += sizeof(int16_t) * (size_t) cs.h_cov_off[n] + 64;
  return b;
}

void cand_fetch(CandRaw &raw, void *dst, size_t dst_bytes, int slot, hipEvent_t landed)
{ State &st = S();
  if (slot < 0) slot = st.cand_cur;
  CandStore &cs = st.cand_slot[slot];
  st.tfilt += filter_flush_block(slot);
  const int n = cs.nreads;
  raw.nreads = n;
  raw.npool = cs.total_len;
  raw.cov_off = cs.h_cov_off;
  const size_t ncov = cs.h_cov_off.empty() ? 0 : (size_t) cs.h_cov_off[n];
  if (dst != nullptr && dst_bytes >= cand_fetch_bytes(slot))
    { // straight into the caller's page-locked memory: one wait, no staging copy
      uint8_t *q = (uint8_t *) dst;
      auto take = [&](size_t bytes) { uint8_t *r = q; q += (bytes + 63) & ~(size_t) 63; return r; };
      DCand    *pp = (DCand *) take(sizeof(DCand) * raw.npool);
      uint32_t *po = (uint32_t *) take(sizeof(uint32_t) * ((size_t) n + 1));
      uint32_t *pl = (uint32_t *) take(sizeof(uint32_t) * ((size_t) n + 1));
      int16_t  *pc = (int16_t *) take(sizeof(int16_t) * ncov);
      if (raw.npool > 0)
        { HIP_CHECK(hipMemcpyAsync(pp, cs.pool[cs.cur].p, sizeof(DCand) * raw.npool, hipMemcpyDeviceToHost, stream()));
          HIP_CHECK(hipMemcpyAsync(po, cs.seg_off[cs.cur].p, sizeof(uint32_t) * ((size_t) n + 1), hipMemcpyDeviceToHost, stream()));
          HIP_CHECK(hipMemcpyAsync(pl, cs.seg_len[cs.cur].p, sizeof(uint32_t) * (size_t) n, hipMemcpyDeviceToHost, stream()));
        }
      if (ncov > 0)
        HIP_CHECK(hipMemcpyAsync(pc, cs.cover.p, sizeof(int16_t) * ncov, hipMemcpyDeviceToHost, stream()));
      if (landed != nullptr) HIP_CHECK(hipEventRecord(landed, stream()));      // (the caller waits, when it needs the lists)
      else HIP_CHECK(hipStreamSynchronize(stream()));
      raw.pool = pp; raw.off = po; raw.len = pl; raw.cov = pc;
    }
  else
    { raw.v_pool.resize(raw.npool);
      raw.v_off.assign((size_t) n + 1, 0);
      raw.v_len.assign((size_t) n + 1, 0);
      raw.v_cov.resize(ncov);
      if (raw.npool > 0)
        { copy_d2h(raw.v_pool.data(), cs.pool[cs.cur].p, sizeof(DCand) * raw.npool);
          copy_d2h(raw.v_off.data(), cs.seg_off[cs.cur].p, sizeof(uint32_t) * ((size_t) n + 1));
          copy_d2h(raw.v_len.data(), cs.seg_len[cs.cur].p, sizeof(uint32_t) * (size_t) n);
        }
      if (ncov > 0) copy_d2h(raw.v_cov.data(), cs.cover.p, sizeof(int16_t) * ncov);
      HIP_CHECK(hipStreamSynchronize(stream()));
      raw.pool = raw.v_pool.data(); raw.off = raw.v_off.data(); raw.len = raw.v_len.data(); raw.cov = raw.v_cov.data();
    }
  raw.valid = true;
}

// Per read: its candidates newest first (the reference's list order), tombstones left out; ids count them in that
// order over the block.  jd (may be null): the block's Jump pool.
void cand_build(const CandRaw &raw, std::vector<ReadState> &reads, const std::vector<uint16_t> *jd)
{ const int n = raw.nreads;
  reads.clear();
  reads.resize(n);
  int id = 0;
  for (int r = 0; r < n; r++)
    { ReadState &rs = reads[r];
      if (raw.npool > 0)
        for (uint32_t i = raw.len[r]; i-- > 0;)
          { const DCand &d = raw.pool[raw.off[r] + i];
            if (d.flags & DC_TOMB) continue;
            std::unique_ptr<Candidate> c(new Candidate());
            c->score = d.score; c->length = (int) d.jlen; c->bread = d.bread; c->comp = (d.flags & DC_COMP) ? 1 : 0;
            c->afirst = d.afirst; c->alast = d.alast; c->bfirst = d.bfirst; c->blast = d.blast;
            c->id = id++;
            c->joff = d.joff;
            if (jd != nullptr && d.jlen > 0)
              c->jumps.assign(jd->begin() + 2 * (size_t) d.joff, jd->begin() + 2 * ((size_t) d.joff + d.jlen));
            rs.cands.push_back(std::move(c));
          }
      if (!raw.cov_off.empty())
        rs.cover.assign(raw.cov + raw.cov_off[r], raw.cov + raw.cov_off[r + 1]);
    }
}

void cand_mirror(bool with_jumps)
{ State &st = S();
  CandStore &cs = st.cands();
  st.tfilt += filter_flush();             // (the passes still waiting in the batch)
  if (cs.mirror_valid && (cs.mirror_jumps || !with_jumps)) return;
  CandRaw raw;
  cand_fetch(raw);
  std::vector<uint16_t> jd;
  if (with_jumps && cs.jtop > 0)
    { jd.resize(2 * cs.jtop);
      copy_d2h(jd.data(), cs.jumps.p, sizeof(uint16_t) * 2 * cs.jtop);
      HIP_CHECK(hipStreamSynchronize(stream()));
    }
  cand_build(raw, st.reads, with_jumps ? &jd : nullptr);
  cs.mirror_valid = true;
  cs.mirror_jumps = with_jumps;
}

}  // namespace dmh

using namespace dmh;

extern "C" {

int dmh_read_candidates(DAZZ_DB *ablock, int ar, int *out, int maxc)
{ State &st = S();
  if (st.cur_reads == ablock) cand_mirror(true);
  if (st.cur_reads != ablock || ar < 0 || ar >= (int) st.reads.size()) return -1;
  auto &L = st.reads[ar].cands;
  int n = 0;
  for (auto &c : L)
    { if (n < maxc)
        { int *o = out + 9 * n;
          o[0] = c->score; o[1] = c->length; o[2] = c->bread; o[3] = c->comp;
          o[4] = c->afirst; o[5] = c->alast; o[6] = c->bfirst; o[7] = c->blast; o[8] = c->id;
        }
      n += 1;
    }
  return n;
}

int dmh_candidate_jumps(DAZZ_DB *ablock, int ar, int k, uint16_t *adisp, uint16_t *bdisp, int maxj)
{ State &st = S();
  if (st.cur_reads == ablock) cand_mirror(true);
  if (st.cur_reads != ablock || ar < 0 || ar >= (int) st.reads.size()) return -1;
  auto &L = st.reads[ar].cands;
  if (k < 0 || k >= (int) L.size()) return -1;
  Candidate &c = *L[k];
  int n = (int) (c.jumps.size() / 2);
  for (int i = 0; i < n && i < maxj; i++)
    { adisp[i] = c.jumps[2 * i];
      bdisp[i] = c.jumps[2 * i + 1];
    }
  return n;
}

int dmh_read_cover(DAZZ_DB *ablock, int ar, int16_t *out, int maxn)
{ State &st = S();
  if (st.cur_reads == ablock) cand_mirror(false);
  if (st.cur_reads != ablock || ar < 0 || ar >= (int) st.reads.size()) return -1;
  auto &cv = st.reads[ar].cover;
  int n = (int) cv.size();
  for (int i = 0; i < n && i < maxn; i++) out[i] = cv[i];
  return n;
}

}  // extern "C"
