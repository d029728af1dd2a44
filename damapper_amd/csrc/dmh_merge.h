// dmh_merge.h -- result of the seed-hit merge + seed sort of one Match_Filter call.
#pragma once
#include "dmh_state.h"

namespace dmh {

constexpr int MAXPASS = 32;        // Match_Filter passes one batch can hold (5 pass bits in the seed key)

// A row of the merge as k_row_list leaves it: partners (RW_MB bits; < MAXGRAM under the -M limit rule) | apos << RW_MB |
// first partner's index in the REF index << 32
constexpr int      RW_MB = 14, RW_AP = 18;
constexpr uint64_t RW_MBMASK = (1ull << RW_MB) - 1ull, RW_APMASK = (1ull << RW_AP) - 1ull;

// Seed hits on their way from the merge to the chain kernels.  With batching on, consecutive Match_Filter calls of
// a READS block only APPEND their hits (the pass number becomes part of the sort key, between aread and bread);
// the batch is sorted, chained and put through the domination rule in one go when it is closed (filter_flush:
// the hit budget is reached, Reporter starts, or somebody looks at the candidates).  Passes are walked in call
// order per read, so the result is what the calls would have produced one by one (map.c:2889-3209).
struct ChainCtx;                   // the chain stage's per-batch buffers (dmh_chain.hip)

struct MergeResult
{ bool    open    = false;       // a batch is being collected: the key layout below is fixed
  ChainCtx *cc    = nullptr;
  DevBuf<uint64_t> alt_keys;     // sort partner of keys / vals (the sorted hits may end up here)
  DevBuf<uint32_t> alt_vals;
  const DAZZ_DB *ablock = nullptr;
  int     cslot   = 0;           // the candidate store of the batch's READS block (State::cand_slot)
  int     npass   = 0, passbits = 0, qbits = 0;
  struct { int bbase, comp; } pass[MAXPASS];
  int64_t nhits   = 0;
  int64_t limit   = 0;
  int64_t ngroups = 0;
  int     pbits = 0, bbits = 0, abits = 0;     // key = aread | bread | apos (high -> low)
  DevBuf<uint64_t> keys;
  DevBuf<uint32_t> vals;
  uint64_t *sab = nullptr;                     // the sorted hits as the chain kernels read them: one word per hit,
                                               // apos in the low half, bpos (0-based last base of the REF k-mer) in the high
  DevBuf<uint32_t> gstart;                     // ngroups+1 first-hit indices of the groups
  // A group is what one lane group of the chain kernels walks: all hits of an (aread, bread) pair, or -- in
  // cluster mode -- one B-CLUSTER of such a pair: a maximal run of its hits whose bpos values lie at most
  // MAX_GAP apart when sorted.  Hits only interact within MAX_GAP in A and in B (map.c:1262-1318,1787-1796),
  // so the clusters of a pair chain independently; the hits are then sorted (cluster, apos, bpos) and the
  // sort key holds the cluster number instead of (aread, bread).
  bool     clustered = false;
  bool     word = false;                       // cluster mode with one word per hit through the seed sort (no vals)
  bool     read_sorted = false;                // ... and the hits written read by read, to be ordered per read in LDS
  bool     fused = false;                      // ... or not written at all: the per-read kernel expands the merge's rows itself,
                                               //   orders the hits and scores the plain chains (dmh_readsort.h, round 4)
  const uint64_t *f_rows = nullptr;            //   the pass's row table, the READS index's read offsets, the REF index's
  const int64_t  *f_rowbase = nullptr;         //   positions: valid from merge_pass to the close that follows it
  const uint32_t *f_bvals = nullptr;
  int             f_bposbits = 0;
  DevBuf<uint32_t> rs_lcnt;                    //   per cluster of a read: chain ends reported | done << 31
  DevBuf<uint64_t> gkey;                       // per group: (aread << bbits) | bread
  DevBuf<uint32_t> cog, ocfirst, glast;        // cluster mode: pair number of each cluster, first cluster of each pair
                                               // (+ end), and per cluster its pair's largest apos + 1
  // the groups by size, for the chain kernels (chain_classify_groups): groups of small_hi..pile_min-1 hits
  // (list of the one-wavefront-per-group launch; windows that outgrow LDS are appended later) and repeat
  // piles of >= pile_min hits (one workgroup each)
  DevBuf<uint32_t> biglist, pilelist, medlist, gcounts;  // gcounts[0] = length of biglist, [1] = of pilelist, [2] = of medlist (device)
  uint32_t small_hi = 0xffffffffu, pile_min = 0, npiles = 0, nbig = 0, nmed = 0, pile_hits = 0;
  uint32_t lane_max = 0;                       // > 0: groups of up to this many hits try the lane-per-group kernel first
  uint32_t small_cap = 0;                      // groups of up to this many hits go to the all-LDS kernel
  bool     classified = false;
  // the per-read seed order of a dense pass (dmh_readsort.h): first hit of each read, size-class lists and their
  // lengths, clusters per read, cluster records, cluster counters, clusters listed for the wide sorts; largest apos per pair
  DevBuf<uint32_t> rs_start, rs_stats, rs_lists, rs_ncl, rs_wlist[2], pmax;
  DevBuf<uint64_t> rs_lcl;
  DevBuf<unsigned int> rs_wcnt;
  uint32_t  *h_known = nullptr;                // page-locked: the counters of a close that stopped behind the per-read sort
  hipEvent_t ev_known = nullptr;               //   (close_batch phases 1 / 2)
};

// classifies the groups of R.gstart on the device and fetches the number of piles
void chain_group_limits(MergeResult &R);        // (the size classes alone)
void chain_classify_groups(MergeResult &R, const uint32_t *known = nullptr);

// One Match_Filter pass up to the expanded hits (map.c:2984-3099): returns the number of hits it added.  May close the
// batch first (when the pass does not fit it) and closes it afterwards unless batching is on; *nfilt accumulates the
// change in the number of candidates of whatever was closed.
int64_t merge_pass(const DAZZ_DB *ablock, const DAZZ_DB *bblock, IndexDev &A, IndexDev &B,
                   int Kshift, int comp, int64_t *nfilt);

// Sort, group, chain and filter what the batch holds and everything still in flight (no-op when there is nothing);
// returns the change in candidates.
int64_t filter_flush();
int64_t filter_flush_block(int cslot);     // ... of ONE READS block only (the other's batches stay open / in flight)
void    filter_discard();          // forget the batches of a READS block that never reached Reporter
void    filter_discard_block(int cslot);
MergeResult &filter_batch();
// the driver's word on the pass that follows the one about to be made (dmh_next_pass): its front -- match + count, which
// depends on the two indices only -- may be queued before this pass's last round trip
void    filter_hint_next_pass(IndexDev *A, IndexDev *B, int Kshift);
void    filter_forget_index(const IndexDev *ix);      // an index is being freed

}  // namespace dmh
