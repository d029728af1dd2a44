// dmh_chain.h -- chain scoring stage of Match_Filter.
#pragma once
#include "dmh_merge.h"

namespace dmh {

constexpr int MAX_GAP   = 1000;   // map.c:36
constexpr int MIN_PIECE = 300;    // map.c:37
constexpr int HITMIN    = 3;      // map.c:34

constexpr uint32_t NIL = 0xffffffffu;

constexpr uint32_t ST_LIVE    = 1;   // in the tree
constexpr uint32_t ST_DEAD    = 0;   // removed by the slope rule, or expired

// The best end of a chain tree that scores, as the chain kernels leave it (slots derived from the group's first hit).
struct ChainSummary
{ uint32_t group, live;     // live: the end was still in the window when its group finished; LIVE_LATER: k_pack_summaries
                            //   works group, live and okey out (the per-read kernel of dmh_readsort.h knows neither the
                            //   group's number nor its pair's last hit)
  // reporting order inside a group (map.c:1634-1642): live ends by (diagonal, apos) descending, then
  // expired ends most recent first -- the host sorts on (group, live desc, okey desc)
  uint64_t okey;
  int32_t  cost, ab, ae, bb, be;
  uint32_t end;
};
constexpr uint32_t LIVE_LATER = 2;

// map.c:1574-1588: reset the per-read candidate lists and -p counters of the READS block.
void chain_reset_reads(const DAZZ_DB *ablock);

// map.c:1463-1922: score the chains of every (read, contig) group of the sorted hits, apply the
// per-read domination rule and store the surviving candidates.  Returns the change in the number
// of candidates (the reference's nfilt).
//   chain_submit queues the chain kernels of a batch (repeat piles on a stream of their own) and returns;
//   chain_finish waits for them and does the rest -- reporting order, domination rule, -p deltas, Jump lists.
// Batches are finished in the order they were submitted (the candidate lists carry over from one to the next), but a
// few of them may be in flight: a pile is a sequential walk of tens of milliseconds on ONE workgroup, and the
// merge, sorts and chain kernels of the following passes do not depend on it.
void    chain_submit(MergeResult &R, const DAZZ_DB *ablock);
// fused route (dmh_readsort.h): the batch's link / chain-end buffers and per-group counters, for the kernels that fill
// them BEFORE chain_submit (which then leaves them as they are and skips k_chain_simple)
void    chain_prepare_hits(MergeResult &R, uint32_t **from, ChainSummary **out);
void    chain_prepare_groups(MergeResult &R, uint32_t **gcount, uint32_t **gbase, uint32_t **gdone);
bool    chain_simple_wanted(const MergeResult &R);     // k_chain_simple would run for this batch (lane kernel and option on)
int64_t chain_finish(MergeResult &R, const DAZZ_DB *ablock);
// ... step by step (three of them, each up to the next number the host needs from the device): true when the batch is
// finished, *nfilt then has the change in candidates added to it
bool    chain_finish_step(MergeResult &R, int64_t *nfilt);
bool    chain_finish_started(MergeResult &R);
int     chain_finish_stage(MergeResult &R);        // steps of the finish done so far (0..2)
bool    chain_finish_ready(MergeResult &R);        // the next step would neither block the host nor wait for a pile

// Drops the chain stage's per-batch buffers of a ring slot that is not in use (memory pressure).
void    chain_release(MergeResult &R);
bool    chain_drop_unused();          // memory ran out inside chain_submit's allocations: the ones made so far are given back
// A submitted batch is dropped unfinished: later work on the main and finish streams waits for its pile kernel.
void    chain_abandon(MergeResult &R);

// Brings State::reads (the host view of the candidate lists: list order, -p deltas) up to date with the
// device lists; with_jumps also fetches every candidate's Jump list.
void cand_mirror(bool with_jumps);
// The two halves of it: the lists as they are on the device, and the per-read objects made from them.
// dst / dst_bytes: page-locked memory to copy into directly (cand_fetch_bytes() tells how much); nullptr = own vectors.
// slot: the store (State::cand_slot) to read; -1 = the current block's
size_t cand_fetch_bytes(int slot = -1);
// landed (only with dst): the copies are queued and the event recorded behind them instead of waited for.
void cand_fetch(CandRaw &raw, void *dst = nullptr, size_t dst_bytes = 0, int slot = -1, hipEvent_t landed = nullptr);
void cand_build(const CandRaw &raw, std::vector<ReadState> &reads, const std::vector<uint16_t> *jd);

}  // namespace dmh
