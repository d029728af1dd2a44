/* oracle.h -- TEST INFRASTRUCTURE ONLY: plain-C restatement of the damapper hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it. */
#ifndef DAMAPPER_ORACLE_H
#define DAMAPPER_ORACLE_H
#include <stdint.h>

/* align.c:1727-1946 as called from map.c:2513 (low = hgh = dg, lbord = hbord = -1).
   aseq/bseq: one byte per base, a 4 before and after; for acomp != 0 aseq is already the reverse
   complement and ACOMP_FLAG semantics apply.  apath_out/bpath_out: {abpos,bbpos,aepos,bepos,diffs,tlen}. */
int oracle_local_alignment(const uint8_t *aseq, int alen, const uint8_t *bseq, int blen, int acomp,
                           int tspace, int ave_path, int reach, const int16_t *score, const int16_t *table,
                           int dg, int anti, int *apath_out, uint16_t *atrace_out,
                           int *bpath_out, uint16_t *btrace_out);

/* align.c:222-269: trim tables and ave_path from (-e, base frequencies) */
void oracle_align_spec(double ave_corr, const float *freq, int *ave_path, int16_t *score, int16_t *table);

/* map.c:545-576 + 655-822: sorted k-mer index of a block; returns the number of records kept.
   bases: [4] r0 [4] r1 ... [4] with boff[i] the offset of read i (boff[nreads] = total). */
int64_t oracle_sort_kmers(const uint8_t *bases, const int64_t *boff, int nreads, int kmer, int suppress,
                          uint64_t *code, int32_t *read, int32_t *rpos);

/* map.c:881-1002 + 2992-3055 + 3110-3126: merge two indices, apply the limit rule, return the seed
   hits sorted by (aread,bread,apos,bpos).  mem_avail < 0 means MEM_LIMIT = 0 (no cap). */
int64_t oracle_seed_hits(const uint64_t *acode, const int32_t *aread, const int32_t *arpos, int64_t alen,
                         const uint64_t *bcode, const int32_t *bread, const int32_t *brpos, int64_t blen,
                         int64_t mem_avail, int *limit_out, int32_t **hits_out);

/* map.c:1781-1872 + 1634-1649: chains of one (read,contig) group of sorted hits.
   hits: n x {apos,bpos} (0-based k-mer end positions).  out: per reported chain
   {cost,ab,ae,bb,be,end} in the reference's reporting order; returns the count. */
int oracle_chain_group(const int32_t *apos, const int32_t *bpos, int n, int kmer,
                       int32_t *out, int maxout, int32_t *from_out);

void oracle_free(void *p);
#endif
