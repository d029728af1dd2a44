/* damapper_hip.h -- C-ABI of libdamapper_hip.so (MI355X / gfx950).
 *
 * The library is a drop-in for the reference's filter API (map.h) and for the small part of
 * align.h that damapper.c's main() uses; every entry point below cites the reference
 * declaration it replaces.  A maintainer links the reference driver against it instead of
 * map.c + align.c:
 *
 *     gcc -O3 damapper.c DB.c QV.c -L. -ldamapper_hip -lpthread -lm -o damapper
 *
 * (INTEGRATION.md has the details).  All signatures are plain C: pointers, sizes, scalars.
 *
 * Section 1 is the drop-in boundary proper.  Section 2 is the stage-level test/bench surface
 * (`dmh_*`): the same hot path exposed stage by stage so that the parity tests can compare
 * each kernel with the oracle, and so that bench.py can read kernel timings.
 */
#ifndef DAMAPPER_HIP_H
#define DAMAPPER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- DAZZ_DB in-memory layout (binary compatible with DB.h:285-295 and DB.h:390-420).  When
 *      compiling together with the reference's own DB.h define DMH_HAVE_DAZZ_DB first.      */
#ifndef DMH_HAVE_DAZZ_DB
typedef struct
  { int     origin;
    int     rlen;     /* length of the sequence                                             */
    int     fpulse;
    int64_t boff;     /* offset of the read's bases inside DAZZ_DB.bases (in-memory block)  */
    int64_t coff;     /* scratch: the filter keeps the head of the read's candidate list    */
                      /*   here between Match_Filter calls (map.c:1578,1875)                */
    int     flags;
  } DAZZ_READ;

typedef struct
  { int        ureads, treads, cutoff, allarr;
    float      freq[4];
    int        maxlen;
    int64_t    totlen;
    int        nreads, trimmed, part, ufirst, tfirst;
    char      *path;
    int        loaded;
    void      *bases;   /* [4] r0 [4] r1 ... [4], one byte per base 0..3 (DB.c:1402-1437)   */
    DAZZ_READ *reads;   /* valid for [-1..nreads]                                           */
    void      *tracks;  /* DAZZ_TRACK list; the first one is the k-mer mask (map.c:701)         */
  } DAZZ_DB;

/* DB.h:306-316.  The filter reads only the first track of a block, as an interval mask: for read i
 * data[anno[i] .. anno[i+1]) holds sorted [begin,end) pairs, anno counted in ints (damapper.c:386-388);
 * k-mers touching a masked base are left out of the index (map.c:481-543).                        */
typedef struct _dmh_track
  { struct _dmh_track *next;
    char    *name;
    int      size;     /* bytes per anno record as recorded by the driver (see sizeof_DB, DB.c:1063) */
    int      nreads;
    void    *anno;     /* int64[nreads+1]                                                          */
    int     *alen;
    void    *data;     /* int[]                                                                    */
    int      loaded;
    int64_t  dmax;
  } DAZZ_TRACK;
#endif

typedef void Align_Spec;   /* opaque, align.h:186 */

/* ============================ 1. drop-in boundary ======================================= */

/* Globals the reference driver owns (damapper.c:58-65, declared extern in map.h:17-24).  The
 * library carries default definitions; when the driver defines them too, ELF symbol
 * interposition makes both sides share the driver's copy.                                  */
extern int      VERBOSE;
extern int      PROFILE;
extern int      SPACING;
extern double   BEST_TIE;
extern char    *SORT_PATH;
extern uint64_t MEM_LIMIT;
extern uint64_t MEM_PHYSICAL;

/* map.h:39 / damapper.c:543 -- fatal-error callback.  Weak default: exit(val).              */
void Clean_Exit(int val);

/* map.h:26 (map.c:124-150).  Returns 1 if kmer <= 1, else 0.  nthreads only decides how many
 * per-"thread" output files Reporter writes (NTHREADS = 2^floor(log2 nthreads)).            */
int Set_Filter_Params(int kmer, int suppress, int nthreads);

/* map.h:28 (map.c:655-822).  Builds the sorted k-mer index of `block` in HBM.  The result is
 * an opaque handle allocated with malloc() -- the driver releases the READS handle with plain
 * free() (damapper.c:879) -- whose device storage the library tracks.  *len = number of k-mers
 * kept; NULL and *len = 0 if the block has no k-mers.                                       */
void *Sort_Kmers(DAZZ_DB *block, int *len);

/* map.h:30 (map.c:2889-3209).  ablock = READS block, bblock = REF block (naming as in map.c).
 * Consumes btable (as the reference does, map.c:3073-3076,3181-3182).  comp != 0 when bblock
 * was reverse-complemented in place; start != 0 resets the per-read candidate lists.        */
void Match_Filter(DAZZ_DB *ablock, DAZZ_DB *bblock,
                  void *atable, int alen, void *btable, int blen, int comp, int start);

#define FLAG_DOA 0x1   /* map.h:33 */
#define FLAG_DOB 0x2   /* map.h:34 */

/* map.h:36 (map.c:3227-3319).  Turns the candidate chains into alignments (d-wave kernel),
 * selects/links them and writes SORT_PATH/<aname>.<bname>.M<i>.las, i = 1..NTHREADS
 * (and the R files for FLAG_DOB, the .prof track for PROFILE).                              */
void Reporter(char *aname, DAZZ_DB *ablock, char *bname, DAZZ_DB *bblock,
              Align_Spec *asettings, int mflag);

/* align.h:186-199 (align.c:222-287): the subset damapper.c uses.                            */
Align_Spec *New_Align_Spec(double ave_corr, int trace_space, float *freq, int reach);
void        Free_Align_Spec(Align_Spec *spec);
int         Trace_Spacing(Align_Spec *spec);
double      Average_Correlation(Align_Spec *spec);
float      *Base_Frequencies(Align_Spec *spec);
int         Overlap_If_Possible(Align_Spec *spec);

/* ============================ 2. stage-level surface ==================================== */

/* Device / library state */
int         dmh_device_count(void);
const char *dmh_last_error(void);
void        dmh_reset(void);                 /* drop every cached device buffer and all state */

/* Upload `block`'s bases ahead of time so Sort_Kmers / Reporter find them resident in HBM.   */
/* REF index cache for a driver that stays with the reference's loop (damapper.c:834-862 reloads,
 * complements and re-indexes every REF block for every READS block): with the cache on, the index
 * Match_Filter has consumed stays in HBM (up to max_gb, least recently used first out; <= 0 keeps the
 * default of 160 GB) and a later Sort_Kmers of a block with the same content returns it without upload
 * or sort.  Content identity = two 64-bit hashes over EVERY base and read boundary of the block (hashed on the
 * host, ~10 ms for a 375 Mbp block) plus k and -t.  Opt-in because it holds HBM; also switched on by the
 * environment variable DMH_REF_CACHE=1 (unmodified driver binary).                                       */
void dmh_ref_cache(int on, double max_gb);

/* Sort_Kmers whose radix passes are queued later, in the gaps between the GPU round trips of the
 * work on the current READS block; Match_Filter queues whatever is left before it uses the index.
 * Same index as Sort_Kmers.  For software-pipelining the block loop (index of READS block i+1 built
 * while block i is filtered and aligned).  Blocks with a mask track or under -t are built at once.  */
void *dmh_sort_kmers_async(DAZZ_DB *block, int *len);

/* Switches between code paths that must give byte-identical results (used by the parity tests to
 * drive every kernel variant; -1 / 0 restores the default where not said otherwise):
 *   "align_level" 0/1/2  first rung of the alignment kernel ladder (two candidates per wavefront / one per
 *                        wavefront with LDS fallback / global-memory band)
 *   "align_g" 16/32      16: the first rung runs four candidates per wavefront on 16-lane groups with two slots per lane
 *                        instead of two on 32-lane groups; 16 or 32 also make dmh_local_alignment try that lane-group
 *                        pass first (A view only)
 *   "chain_g" 16/32/64   lanes per chain group for ALL groups below the pile size (0 default: by group size)
 *   "chain_lane" 0       no lane-per-group kernel (groups go to the lane-group kernels)
 *   "chain_small" n      groups of up to n hits take the all-LDS lane-group kernel when chain_lane is off (0 none)
 *   "chain_pile" n       groups of >= n hits get a whole workgroup (0 never, -1 default 4096); "pile_cap" n = that
 *                        kernel may hold n window entries in LDS (forces its spill path)
 *   "cluster" 1/0        chain groups are the B-clusters of an (aread, bread) pair / whole pairs (-1: by hit density)
 *   "cluster_sort" 0     clusters are ordered by a second global radix sort instead of where they lie
 *   "alloc_fail" n       the n-th device allocation from now behaves as if memory had run out: the batches in flight
 *                        are finished and the idle buffers of the filter released (what a pass of 10^9 hits triggers)
 *   "seed_word" 0        cluster mode sorts (key, value) pairs even when aread | bread | bpos | apos fit one word
 *   "batch" 0            every Match_Filter pass is chained by itself (no batches of sparse passes)
 *   "in_flight" 0        every batch is finished before the next pass starts
 *   "fin_stream" 0       batches are finished on the main stream instead of a stream of their own
 *   "expand" 0           hits are expanded entry by entry instead of by output tiles
 *   "merge_dense" 1/0/2  REF bucket table with one bucket per entry and four-key probes / four buckets per entry
 *                        and two-key probes / 16 undecided bits per bucket, four-key probes on the 2-byte low-bits
 *                        array (-1: by index size; takes effect for indices built afterwards)
 *   "pool_div" n         start with 1/n of the alignment pool (forces the grow-and-retry path)
 *   "read_sort" n        cluster mode, dense passes: 0 = the global radix sort instead of one workgroup per read; 2 = the
 *                        1024-thread workgroups from 129 hits; 3 = reads of more than 512 hits through global memory
 *                        (k_read_sort_big); 5 = ranks from ballots instead of ordered LDS atomics; 6 = no 1024-thread
 *                        workgroups (as while an alignment kernel is deferred); 7 = 3 and 6
 *   "chain_simple" 0     plain-chain groups are walked like the others (no k_chain_simple)
 *   "oom_finish" n       memory "runs out" inside the n-th finish of a batch from now
 *   "tails" 0            nothing runs on the tail stream: every kernel of a pass has the GPU to itself
 *   "defer_close" 1      the close of a dense pass is taken up by the next Match_Filter call (experiment)
 *   "hit_limit" n        a pass of n hits or more is refused (the refusal at 2^32 hits, at a testable size)
 *   "match_compact" 0    k_merge_match finishes its look-ups item by item (no list of the look-ups that become entries)
 *   "sort_ballot" 1      k_radix_scatter ranks by ballots (not by the values ordered LDS atomic adds return)
 *   "front_ahead" 0      the match kernels of a REF block are queued when its own pass comes, on the main stream (default:
 *                        those of the REF block that follows run on a stream of their own next to the block at hand's
 *                        per-read kernels, when the driver names the passes with dmh_next_pass)
 *   "pebble_spill" 0     a lane group's d-wave pass ends at the 288 pebbles its LDS holds (the candidate is re-run one
 *                        rung up) instead of spilling the rest into global memory
 * Returns 0, or -1 for an unknown key.                                                                       */
int dmh_test_option(const char *key, int value);

/* Pipelining hook (optional; results unchanged): called BEFORE a Match_Filter call, it names the two index handles of
 * the Match_Filter call that will FOLLOW that one (same READS index, the next REF index).  The library then queues the
 * following pass's match + count kernels -- which depend on the two indices only (map.c:2984-2988) -- in front of
 * the coming call's last wait for the device, so that the GPU goes straight on while the host queues the rest of
 * that call's work.  A missing or wrong hint costs time, never results; (NULL, NULL) withdraws it.
 * (Measured on the GRCh38-sized bench workload: no gain -- 148.5 against 144.9 ms per read block, DESIGN.md 3b --, so
 * damapper_amd/driver.py only uses it with DMH_HINT_PASSES=1.)                                                   */
void dmh_next_pass(void *atable, void *btable);

/* Page-locked host memory for a driver that loads blocks itself: a block whose `bases` live in such
 * memory is uploaded by one DMA without staging (a block in ordinary malloc memory, as the reference
 * driver's, goes through two pinned staging buffers).  NULL when no device is available.              */
void *dmh_host_alloc(size_t bytes);
void  dmh_host_free(void *p);

int  dmh_preload_block(DAZZ_DB *block);
void dmh_release_block(DAZZ_DB *block);
/* Also derive what Reporter's kernel reads from a resident block now instead of at its first Reporter: the
 * 2-bit packed bases, and for the READS side (reads_side != 0) those of the reverse complements.  -1 if the
 * block is not resident.                                                                              */
int  dmh_prepare_block(DAZZ_DB *block, int reads_side);
/* Free and total memory of the device this process uses (hipMemGetInfo); 0, or -1 without a device. */
int  dmh_device_memory(unsigned long long *free_bytes, unsigned long long *total_bytes);

/* Load_All_Reads on the device (replaces DB.c:1389-1437 + Uncompress_Read DB.c:342-361 for a block the
 * driver has only opened): `bps` = the block's slice of the .bps file as it lies on disk (4 bases per byte,
 * first base in the top bits), `coff[i]` = byte offset of read i's record inside the slice (ascending),
 * block->reads[0..nreads] = rlen and boff exactly as Load_All_Reads sets them (boff[i+1] = boff[i] + rlen + 1).
 * The compressed bytes (a quarter of the block) are uploaded and unpacked by a kernel into the same device
 * layout dmh_preload_block produces; block->bases is never read and may be NULL.  The copy stays resident
 * until dmh_release_block.  `bps` may be freed on return -- unless it lies in page-locked memory
 * (dmh_host_alloc): then upload and decode run asynchronously on the upload stream and the slice has to stay
 * unchanged until a library call that uses the block has returned.  dmh_download_block copies the device image
 * (separator + bases, reads[nreads].boff + 1 bytes) back, for tests.  Return 0.                           */
int  dmh_preload_block_2bit(DAZZ_DB *block, const unsigned char *bps, int64_t nbytes, const int64_t *coff);
int  dmh_download_block(DAZZ_DB *block, unsigned char *out);

/* Keep REF indices after Match_Filter (index replicated per GPU): when set,
 * Match_Filter does not consume btable; release it with dmh_free_index.                     */
void dmh_retain_ref_index(int on);
void dmh_free_index(void *table);

/* Copy an index to the host in the reference's KmerPos layout {uint64 code; int32 rpos;
 * int32 read} (map.c:78-82), `len` records.                                                 */
int dmh_index_download(void *table, void *kmerpos_out);

/* Seed hits of the most recent Match_Filter in the reference's SeedPair layout
 * {int32 diag, apos, bread, aread} (map.c:84-89), sorted (aread,bread,apos,bpos).
 * Capture must be switched on first (it costs a device->host copy).                         */
void    dmh_capture_seeds(int on);
int64_t dmh_last_seed_count(void);
int     dmh_last_seeds(void *seedpair_out);
int     dmh_last_limit(void);

/* Candidate lists kept between Match_Filter calls (map.c:1386-1397): per candidate 9 ints
 * {score,length,bread,comp,afirst,alast,bfirst,blast,id}, list order (newest first).         */
int dmh_read_candidates(DAZZ_DB *ablock, int aread, int *out, int maxc);
int dmh_candidate_jumps(DAZZ_DB *ablock, int aread, int k, uint16_t *adisp, uint16_t *bdisp, int maxj);
/* -p coverage deltas of one read (map.c:1654-1666): (rlen-1)/SPACING+2 int16 values.         */
int dmh_read_cover(DAZZ_DB *ablock, int aread, int16_t *out, int maxn);

/* Radix sort of (uint64 key, uint32 value) pairs on key bits [0,nbits), stable (the kernel
 * under Sort_Kmers and the seed sort).  Host buffers in, sorted host buffers out.            */
int dmh_sort_pairs(uint64_t *keys, uint32_t *vals, int64_t n, int nbits);
/* the device-wide exclusive scan of the sorts and the filter (one look-back kernel; three kernels above 1 M values):
   data[0..n) in place, *total = their sum, *ms (may be null) = mean device time of `reps` scans */
int dmh_exclusive_scan(uint32_t *data, int64_t n, int reps, uint32_t *total, double *ms);

/* Batched Local_Alignment (align.c:1727-1946) on the d-wave kernel.  Sequences are one byte per
 * base with a 4 on both sides of every sequence (the in-memory DAZZ_DB layout).  For job i:
 * A = abases[aoff[i] .. +alen[i]), B = bbases[boff[i] .. +blen[i]); comp[i] != 0 aligns the
 * reverse complement of A with ACOMP_FLAG semantics (map.c:2474-2513); seed diagonal dg[i],
 * anti-diagonal ad[i] in the (possibly complemented) A frame.  out: per job 16 ints
 * {abpos,bbpos,aepos,bepos,diffs,tlen, b-view tlen, status, b-view abpos,bbpos,aepos,bepos, 4 debug words};
 * A-view traces (uint16) are written
 * at trace_out + i*trace_stride, B-view traces at btrace_out + i*trace_stride (may be NULL). */
int dmh_local_alignment(Align_Spec *spec,
                        const uint8_t *abases, int64_t abytes, const uint8_t *bbases, int64_t bbytes,
                        int njobs, const int64_t *aoff, const int *alen,
                        const int64_t *boff, const int *blen, const int *comp,
                        const int *dg, const int *ad,
                        int *out, uint16_t *trace_out, uint16_t *btrace_out, int trace_stride);

/* mode 1: Reporter's last stage (per-read selection + .las writing, host only) runs on a worker
 * thread so that the driver may start filtering the next read block at once.
 * mode 2: in addition Reporter may return while the alignment kernel of its block is still
 * running (on a second stream, at half grid, next to the next block's Match_Filter); its records
 * are collected by the next Reporter or by dmh_report_wait().  The library does this only for a
 * READS block that went through at most two Match_Filter passes (one-block reference); the device
 * copies the kernel reads stay alive until it is collected, whatever the driver does with its
 * blocks.  Output files are byte-identical in every mode.
 * mode 0 (default): Reporter is synchronous, as the reference driver expects (it reads the
 * files right after Reporter returns).  dmh_report_wait() blocks until every queued report has
 * been written; a driver that uses mode 1 or 2 must call it before it reads the files or exits. */
void dmh_async_report(int mode);
void dmh_report_wait(void);
/* Reporter calls, since the library was loaded, whose files are completely written: with mode 1 or 2 a driver runs its
 * post-processing of READS block i (damapper.c:894-910: LAsort, LAcat) once this exceeds i, without waiting for the
 * blocks behind it.  (Calls complete in the order they were made.)                                                   */
int  dmh_reports_done(void);

/* Kernel timing (HIP events on the library's stream).  Stages are named; dmh_timing_get
 * returns accumulated milliseconds and launch count since the last dmh_timing_reset.
 * dmh_timing_enable: 0 off, 1 every stage (about 1 ms of event handling per read block),
 * 2 only the roofline kernel (k_radix_scatter).                                               */
void   dmh_timing_enable(int on);
void   dmh_timing_reset(void);
int    dmh_timing_count(void);
int    dmh_timing_get(int idx, char *name, int namelen, double *ms, int64_t *launches, double *bytes);
/* Blocks until all queued device work is complete.                                          */
void   dmh_sync(void);

#ifdef __cplusplus
}
#endif
#endif
