// dmh_chain.h -- chain scoring stage of Match_Filter.
#pragma once
#include "dmh_merge.h"

namespace dmh {

// map.c:1574-1588: reset the per-read candidate lists and -p counters of the READS block.
void chain_reset_reads(const DAZZ_DB *ablock);

// map.c:1463-1922: score the chains of every (read, contig) group of the sorted hits, apply the
// per-read domination rule and store the surviving candidates.  Returns the change in the number
// of candidates (the reference's nfilt).
int64_t chain_and_filter(MergeResult &R, const DAZZ_DB *ablock);

// Brings State::reads (the host view of the candidate lists: list order, -p deltas) up to date with the
// device lists; with_jumps also fetches every candidate's Jump list.
void cand_mirror(bool with_jumps);

}  // namespace dmh
