// dmh_state.h -- process-wide state of libdamapper_hip (the reference's map.c keeps the
// equivalent in file statics, map.c:114-122, 2885-2887: one filter per process, not re-entrant).
#pragma once

#include "dmh_common.h"
#include <list>
#include <map>
#include <memory>

namespace dmh {

constexpr uint64_t INDEX_MAGIC = 0x444d48494e444558ull;   // "DMHINDEX"

// What Sort_Kmers hands back to the driver: a malloc()'d stub (the driver free()s it,
// damapper.c:879); the device storage lives in the registry under `id`.
struct IndexStub
{ uint64_t magic;
  int32_t  id;
  int32_t  len;
};

// Device copy of a DAZZ_DB block's bases + read offsets.
struct BlockDev
{ uint8_t *d_raw   = nullptr;   // L+1 bytes: leading separator, then the block as in DB.c:1402-1437
  int64_t  L       = 0;         // reads[nreads].boff
  int64_t *d_boff  = nullptr;   // nreads+1 offsets
  int      nreads  = 0;
  uint64_t *d_pkraw = nullptr;  // 2-bit packed bases (32 per word), two pad words in front
  uint64_t *d_rcraw = nullptr;  // same for the per-read reverse complement (READS blocks only)
  uint64_t stamp   = 0;         // cheap content fingerprint (detects in-place complement)
  bool     pinned  = false;     // uploaded through dmh_preload_block
  bool     decoded = false;     // made by dmh_preload_block_2bit: the host block has no bases
  void    *d_tmp[2] = {nullptr, nullptr};   // upload scratch of the device-side decode, until block_ready()
  hipEvent_t uploaded = nullptr; // bases still on their way (DMA on the upload stream): block_ready() waits
  const uint8_t *d_bases() const { return d_raw + 1; }
  const uint64_t *d_pk() const { return d_pkraw + 2; }
  const uint64_t *d_rc() const { return d_rcraw + 2; }
};

struct IndexDev
{ int64_t   n      = 0;         // k-mers kept
  uint64_t *keys   = nullptr;   // sorted codes
  uint32_t *vals   = nullptr;   // base offset q of the k-mer's last base inside the block -- or, once the index is
                                //   complete and posbits > 0, (read << posbits) | rpos: what the merge turns into hits
  int       posbits = 0;
  int       maxlen  = 0;        // longest read of the block (for the packing above)
  int64_t  *d_boff = nullptr;   // nreads+1 read offsets (device)
  uint32_t *d_chunk = nullptr;  // read index of position (c << CHUNK_SHIFT), c = 0..nchunk-1
  int64_t   nchunk = 0;
  int       nreads = 0;
  int64_t   L      = 0;
  std::vector<int64_t> h_boff;  // host copy
  DAZZ_DB  *block  = nullptr;
  uint32_t *d_bstart = nullptr; // bucket table over the top bits of the codes (merge look-ups), built at first use
  int       bshift  = 0;        //   bucket of a code = code >> bshift
  int       probe_keys = 2;     //   keys of a bucket a look-up fetches before it would bisect (2 or 4)
  size_t    low_off = 0;        //   != 0: 2-byte array of the keys' low `bshift` bits at d_bstart + low_off (uint32 units)
  size_t    bstart_bytes = 0;
  // REF index cache (dmh_ref_cache): identity of the block the index was built from, and whether this
  // record only borrows the device buffers of a cache entry
  uint64_t  ckey[4] = {0, 0, 0, 0};
  bool      cacheable = false;
  bool      borrowed  = false;
  int       borrows   = 0;      // cache entry: records currently borrowing its buffers (never evicted meanwhile)
  bool      pending = false;    // a deferred build (dmh_sort_kmers_async) has stages left to queue
  hipEvent_t built = nullptr;   // the build ran on the tail stream: recorded behind its last stage (the main stream waits at first use)
};

constexpr int CHUNK_SHIFT = 9;

struct Params
{ int      Kmer = 0, Suppress = 0;
  int      Kshift = 0;
  uint64_t Kmask = 0;
  int      TooFrequent = INT32_MAX;
  int      NTHREADS = 1, NSHIFT = 0;
  bool     set = false;
};

// A candidate chain of one read (map.c:1386-1397) with its Jump displacements (map.c:1380-1384)
struct Candidate
{ int score, length, bread, comp;
  int afirst, alast, bfirst, blast;
  int id;
  uint32_t joff = 0;                     // first (adisp, bdisp) pair of the chain in the device jump pool
  std::vector<uint16_t> jumps;           // (adisp, bdisp) pairs, end -> start (map.c:1380-1384); filled on request
};

// The candidate lists live on the DEVICE between Match_Filter calls (map.c:1386-1397 keeps them behind
// DAZZ_READ.coff): one segment of DCand records per read, oldest first (the reference's list order --
// newest first -- is the segment read backwards), rewritten into the other pool by every domination pass.
struct DCand
{ int32_t  score, afirst, alast, bfirst, blast;
  int32_t  bread;              // contig index in the whole reference
  uint32_t end;                // hit index of the chain's end (valid until its Jump list has been stored)
  uint32_t joff, jlen;         // Jump list: jlen (adisp, bdisp) pairs from pair joff of the jump pool
  uint32_t flags;              // DC_*
};
constexpr uint32_t DC_COMP  = 1u;            // found against the complemented REF block
constexpr uint32_t DC_FRESH = 2u;            // added by the current pass: Jump list not stored yet
constexpr uint32_t DC_TOMB  = 0x80000000u;   // removed by the domination rule

struct CandStore
{ const DAZZ_DB *block = nullptr;
  int      nreads = 0;
  int      cur = 0;                           // pool[cur] / seg_*[cur] hold the lists
  DevBuf<DCand>    pool[2];
  DevBuf<uint32_t> seg_off[2], seg_len[2];    // per read: first record and number of records (tombstones included)
  size_t   total_len = 0;                     // sum of seg_len[cur] (host copy)
  DevBuf<uint16_t> jumps;                     // jump pool, 2 values per pair
  size_t   jtop = 0;                          // pairs in use
  DevBuf<int16_t>  cover;                     // -p deltas, read r at cov_off[r] (map.c:1654-1666)
  DevBuf<uint32_t> cov_off;
  std::vector<uint32_t> h_cov_off;
  DevBuf<unsigned int> ctl;                   // [0] candidates added, [1] removed (current pass), [2] fresh survivors, ...
  bool     mirror_valid = false;              // State::reads reflects the device lists
  bool     mirror_jumps = false;              //   ... including the Jump lists
};

struct ReadState
{ std::vector<std::unique_ptr<Candidate>> cands;   // index 0 = list head (newest)
  std::vector<int16_t> cover;                       // -p deltas
};

// The candidate lists of a READS block as they come off the device (cand_fetch): what Reporter needs to make its
// alignment jobs; the per-read Candidate objects of the report stage are built from it later (cand_build, on the
// report stage's own thread).
struct CandRaw
{ int nreads = 0;
  // views (into the vectors below, or into page-locked memory the caller lent to cand_fetch)
  const DCand    *pool = nullptr;
  size_t          npool = 0;
  const uint32_t *off = nullptr, *len = nullptr;
  const int16_t  *cov = nullptr;
  std::vector<uint32_t> cov_off;
  std::vector<DCand>    v_pool;
  std::vector<uint32_t> v_off, v_len;
  std::vector<int16_t>  v_cov;
  bool valid = false;
};

struct State
{ Params params;
  std::map<int, IndexDev>          indices;
  std::map<const DAZZ_DB *, BlockDev> blocks;
  int  next_index_id = 1;
  bool retain_ref    = false;
  bool capture_seeds = false;
  bool capture_append = false;    // ... of a later range of a pass made in ranges of reads: appended to the earlier ranges' 
  int  last_limit    = 0;
  std::vector<int32_t> last_seeds;     // SeedPair records (4 ints each)

  // cross-call filter state (map.c:2885-2887 + DAZZ_READ.coff lists)
  // Two READS blocks may be open at a time: the one the filter works on (cur_reads, cand_cur) and the one before it,
  // whose last batches are still being finished and whose Reporter call is still to come -- a driver that starts the
  // next block's passes first does not leave the GPU idle at the block boundary (damapper_amd/driver.py).  A store
  // belongs to a block from its first Match_Filter call (start != 0) to its Reporter call.
  const DAZZ_DB *cur_reads = nullptr;
  CandStore cand_slot[2];           // the candidate lists (device)
  int       cand_cur = 0;
  CandStore &cands() { return cand_slot[cand_cur]; }
  int cand_slot_of(const DAZZ_DB *b) const
  { for (int i = 0; i < 2; i++) if (b != nullptr && cand_slot[i].block == b) return i;
    return -1;
  }
  std::vector<ReadState> reads;     // host mirror of them, built on demand (cand_mirror)
  int      next_cand_id = 0;
  int64_t  tfilt = 0;

  // test / experiment switches (dmh_test_option): every setting must give byte-identical results
  int  opt_align_level = -1;    // first rung of the alignment kernel ladder (0 two per wavefront, 1, 2); -1 default
  int  opt_chain_g     = 0;     // lanes per chain group for ALL groups (16/32/64); 0 = 16 + 64 by group size
  int  mf_calls = 0;            // Match_Filter calls since the last Reporter (passes per READS block)
  int     opt_defer_close = -1; // 0: a pass is closed before Match_Filter returns (no second half left to the next pass)
  int     opt_front_early = -1; // 0: a pass's front is never queued by the pass before it (dmh_next_pass ignored)
  int     opt_tails = -1;       // 0: nothing moves to the tail stream (every kernel of a pass has the GPU to itself: bench.py's "alone" figures)
  int     opt_hit_limit = 0;    // > 0: a pass of this many hits is refused (test of the 2^32 refusal)
  int     mf_calls_block = 0;   // ... of the block Reporter is working on
  int  opt_merge_dense = -1;    // REF bucket table: 1 = one bucket per entry + four-key probes, 0 = four buckets per entry + two-key probes, -1 = by size
  int  opt_pile_cap    = 0;     // k_chain_pile's LDS window capacity in entries (128..); 0 = all of it
  int  opt_chain_pile  = -1;    // hits from which a group gets a whole workgroup (k_chain_pile); 0 never, -1 default
  int  opt_chain_small = -1;    // groups of up to this many hits take the all-LDS chain kernel (0 none, -1 default 64)
  int  opt_chain_lane  = -1;    // one lane per group for the groups below the wavefront-per-group size (0 off, -1 default on)
  int  opt_cluster_sort = -1;   // 0: clusters ordered by the global radix sort instead of where they lie
  int  opt_in_flight   = -1;    // 0: every batch is finished before the next pass starts
  int  opt_align_g     = -1;    // 16: the first rung of the alignment ladder runs four candidates per wavefront (default two)
  bool simulated_oom   = false;  // the reclaim hook runs for the "alloc_fail" test option (nothing is learnt from it)
  int  opt_fin_stream  = -1;    // 0: batches are finished on the main stream (no second stream)
  int  opt_seed_word   = -1;    // 0: cluster mode sorts (key, value) pairs even when a hit fits one word
  int  opt_expand      = -1;    // 0: hits expanded entry by entry (one thread / wavefront per entry) instead of by output tiles
  int  opt_batch       = -1;    // collect sparse Match_Filter passes into batches (0 off, -1 default on)
  int  opt_cluster     = -1;    // B-clusters as chain groups: 1 always, 0 never, -1 by hit density
  int  opt_chain_simple = -1;   // 0: no group is chained without a walk (k_chain_simple off)
  int  opt_oom_finish  = 0;     // n > 0: the n-th chain_finish from now meets a (simulated) out-of-memory in its middle
  int  opt_read_sort   = -1;    // cluster mode: 0 = hits ordered by the global radix sort instead of per read in LDS; 2 = the 1024-thread workgroups from 129 hits; 3 = reads of more than 512 hits go through global memory (k_read_sort_big); 5 = ranks from ballots; 6 = no 1024-thread workgroups (as while an alignment kernel is deferred); 7 = 3 and 6
  int  opt_match_compact = -1;  // 0: k_merge_match finishes its look-ups item by item (no candidate list in LDS)
  int  opt_sort_ballot = -1;    // 1: k_radix_scatter ranks by ballots whatever the device does (no ordered LDS atomics)
  int  opt_front_ahead = -1;    // 0: the match kernels of a REF block are queued when its own pass comes (no front stream)
  int  opt_pebble_spill = -1;   // 0: k_align_cands32's pebbles end with its LDS (longer passes are re-run one rung up)
  int  opt_chain_classes = -1;  // 0: groups that fall apart by diagonal are walked like any other (k_chain_classes off)
  int  opt_fuse        = -1;    // 0: dense passes take round 3's route (hits expanded to HBM, k_read_sort, k_chain_simple) instead of the fused per-read kernel
  int  opt_pool_div    = 1;     // initial alignment pool capacity is divided by this (forces the grow path)

  // REF indices kept after Match_Filter consumed them (opt-in, dmh_ref_cache): a later Sort_Kmers of a
  // block with the same content finds its index here -- the unmodified C driver reloads and re-sorts
  // every REF block for every READS block (damapper.c:834-862)
  bool   ref_cache = false;
  size_t ref_cache_bytes = 0, ref_cache_cap = (size_t) 160 << 30;
  std::list<IndexDev> ref_cached;      // most recently used first

  SortScratch sort;
  SortScratch sort_aux;         // scratch of the deferred index build (lives across Match_Filter calls)
  SortScratch sort_fin;         // scratch of the scans of chain_finish (its own stream)
  SortScratch sort_front;       // ... of the scans of a front made ahead (the front stream)
  SortScratch sort_tail;        // ... and of the scans behind the per-read seed order (the tail stream)
};

State &S();

BlockDev &upload_block(const DAZZ_DB *block, bool pin);
void      block_ready(BlockDev &bd);        // orders the current stream behind the block's upload
void      ensure_packed(BlockDev &bd, bool want_rc);
void      pack_bases(const uint8_t *d_bases, int64_t L, uint64_t *d_pkraw, hipStream_t on = nullptr);      // nullptr: the main stream
void      pack_revcomp(const uint8_t *d_bases, int64_t L, const int64_t *d_boff, int nreads, uint64_t *d_rcraw, hipStream_t on = nullptr);
void      drop_block(const DAZZ_DB *block);
void      free_blockdev(BlockDev &bd);

void kmer_emit(const uint8_t *d_bases, int64_t L, const int64_t *d_boff, int nreads, int K,
               uint64_t *d_keys, uint32_t *d_vals, int64_t nk, int part = 0, int nparts = 1, int posbits = 0);

IndexDev *index_of(void *table, const char *who);
void      release_ref_index(IndexDev &ix);    // Match_Filter is done with a REF index: cache it or free it
void      index_wait_ready(IndexDev &ix);     // queue what is left of a deferred build
void      index_pack_positions(IndexDev &ix); // vals -> (read, rpos) words, when they fit (once per index)

}  // namespace dmh
