/* simulate.c -- host-side synthetic PacBio-like read generator for tests and bench (no GPU code).
 * Reads are sampled uniformly (position, strand) from one contig, fixed template length, i.i.d.
 * per-base errors at rate `err` split ins:del:sub = 60:25:15, inserted/substituted base uniform
 * (SURVEY.md 8d).  Each read has its own RNG stream derived from (seed, read index), so the result
 * does not depend on the thread count.  Output is the DAZZ_DB in-memory block layout
 * [4] r0 [4] r1 ... [4] (DB.c:1402-1437).
 */
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <omp.h>

static inline uint64_t splitmix(uint64_t *s)
{ uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

/* A seed for one independent stream out of (seed, index): both go through the splitmix finaliser, so that streams of
 * neighbouring indices (or seeds) are not shifted copies of each other (s + i*G with G the generator's own
 * increment would be exactly that). */
static inline uint64_t stream_seed(uint64_t seed, uint64_t index)
{ uint64_t a = seed, b;
  a = splitmix(&a);
  b = a ^ (index * 0xd6e8feb86659fd93ull + 0x2545f4914f6cdd1dull);
  return splitmix(&b);
}

/* Generates read i; if out != NULL writes its bases there.  Returns the read length. */
static int gen_read(const uint8_t *g, int64_t glen, int rl, double err, uint64_t seed, int i, uint8_t *out)
{ uint64_t s = stream_seed(seed, (uint64_t) i);
  int64_t  pos = (int64_t) (splitmix(&s) % (uint64_t) (glen - rl + 1));
  int      rev = (int) (splitmix(&s) & 1);
  uint32_t tins = (uint32_t) (err * 0.60 * 4294967296.0);
  uint32_t tdel = (uint32_t) (err * 0.85 * 4294967296.0);
  uint32_t tsub = (uint32_t) (err * 4294967296.0);
  int n = 0, j;
  for (j = 0; j < rl; j++)
    { uint64_t r = splitmix(&s);
      uint32_t u = (uint32_t) r;
      int b = rev ? 3 - g[pos + rl - 1 - j] : g[pos + j];
      if (u < tins)
        { if (out) out[n] = (uint8_t) ((r >> 32) & 3);
          n += 1;
          if (out) out[n] = (uint8_t) b;
          n += 1;
        }
      else if (u < tdel)
        continue;
      else if (u < tsub)
        { if (out) out[n] = (uint8_t) ((b + 1 + ((r >> 32) % 3)) & 3);
          n += 1;
        }
      else
        { if (out) out[n] = (uint8_t) b;
          n += 1;
        }
    }
  return n;
}

/* Pass 1: read lengths.  Pass 2 (bases != NULL): fill `bases` (which starts with the leading 4)
 * given boff[] = offsets of the reads relative to bases+1. */
void dmh_sim_lengths(const uint8_t *g, int64_t glen, int nreads, int rl, double err, uint64_t seed, int *rlen)
{ int i;
#pragma omp parallel for schedule(static)
  for (i = 0; i < nreads; i++)
    rlen[i] = gen_read(g, glen, rl, err, seed, i, NULL);
}

void dmh_sim_fill(const uint8_t *g, int64_t glen, int nreads, int rl, double err, uint64_t seed,
                  const int64_t *boff, uint8_t *bases)
{ int i;
  bases[0] = 4;
#pragma omp parallel for schedule(static)
  for (i = 0; i < nreads; i++)
    { int n = gen_read(g, glen, rl, err, seed, i, bases + 1 + boff[i]);
      bases[1 + boff[i] + n] = 4;
    }
}

/* In-place reverse complement of every read of an in-memory block (damapper.c:417-469,
 * complement_DB with inplace != 0): bases points at the block's first base, boff[i] are the read
 * offsets, boff[nreads] the total. */
#include <string.h>

static void comp_one(uint8_t *s, int64_t len)
{ uint8_t *restrict t = (uint8_t *) malloc(len > 0 ? len : 1);
  const uint8_t *restrict r = s;
  int64_t i;
  for (i = 0; i < len; i++)
    t[i] = (uint8_t) (3 - r[len-1-i]);
  memcpy(s, t, len);
  free(t);
}

void dmh_host_complement(uint8_t *bases, const int64_t *boff, int nreads)
{ int i;
  for (i = 0; i < nreads; i++)
    comp_one(bases + boff[i], boff[i+1] - boff[i] - 1);
}

/* ---- synthetic reference genomes ------------------------------------------------------
 * dmh_sim_genome: one contig of `n` uniform random bases, seeded with copies of a repeat family `fam`
 * (length flen) on both strands, each copy mutated i.i.d. at rate `div` (substitutions), the copies
 * `gap_lo`..`gap_hi`-1 bases apart, plus one tandem array of `tcopies` x `tunit` random bases.  `fam` holds
 * `nfam` consensus sequences one after the other; every copy draws its family uniformly.  With nfam = 1 the same
 * model as dazzdb.repeat_rich_genome (one 2 kb family at 3 % divergence every 1-6 kb + a 60 x 150 bp
 * array), generated natively so that a GRCh38-sized reference takes seconds.  Every 1 MB chunk of the
 * background and every copy has its own RNG stream derived from (seed, index): the contig depends on
 * `seed` only, not on the thread count.  flen = 0 gives a uniform random contig.                     */
void dmh_sim_genome(uint8_t *out, int64_t n, uint64_t seed, const uint8_t *fam, int nfam, int flen, double div,
                    int gap_lo, int gap_hi, int tunit, int tcopies)
{ const int64_t CH = 1 << 20;
  int64_t nch = (n + CH - 1) / CH, c;
#pragma omp parallel for schedule(static)
  for (c = 0; c < nch; c++)
    { uint64_t s = stream_seed(seed, (uint64_t) c);
      int64_t  e = (c + 1) * CH < n ? (c + 1) * CH : n, i = c * CH;
      while (i < e)
        { uint64_t r = splitmix(&s);
          int k;
          for (k = 0; k < 32 && i < e; k++, i++)
            out[i] = (uint8_t) ((r >> (2 * k)) & 3);
        }
    }
  if (flen > 0 && gap_hi > gap_lo)
    { /* positions first (sequential, cheap), then the copies in parallel */
      uint64_t s = stream_seed(seed ^ 0xa0761d6478bd642full, 1);
      int64_t  cap = n / (flen + gap_lo) + 2, m = 0, p;
      int64_t *pos = (int64_t *) malloc(sizeof(int64_t) * (size_t) cap);
      p = 500 + (int64_t) (splitmix(&s) % 2500);
      while (p + flen < n && m < cap)
        { pos[m++] = p;
          p += flen + gap_lo + (int64_t) (splitmix(&s) % (uint64_t) (gap_hi - gap_lo));
        }
      uint32_t tdiv = (uint32_t) (div * 4294967296.0);
      int64_t  j;
#pragma omp parallel for schedule(static)
      for (j = 0; j < m; j++)
        { uint64_t t = stream_seed(seed ^ 0x94d049bb133111ebull, (uint64_t) j);
          uint64_t r0 = splitmix(&t);
          int      rev = (int) (r0 & 1), x;
          const uint8_t *cons = fam + (size_t) ((r0 >> 8) % (uint64_t) nfam) * (size_t) flen;
          uint8_t *o = out + pos[j];
          for (x = 0; x < flen; x++)
            { uint64_t r = splitmix(&t);
              int      b = cons[x];
              if ((uint32_t) r < tdiv) b = (b + 1 + (int) ((r >> 32) % 3)) & 3;
              if (rev) o[flen - 1 - x] = (uint8_t) (3 - b);
              else     o[x] = (uint8_t) b;
            }
        }
      free(pos);
    }
  if (tunit > 0 && tcopies > 0 && n > 4 * (int64_t) tunit * tcopies)
    { uint64_t s = stream_seed(seed ^ 0xe7037ed1a0b428dbull, 2);
      int64_t  tl = (int64_t) tunit * tcopies, q = (int64_t) (splitmix(&s) % (uint64_t) (n - tl)), i;
      uint8_t *unit = (uint8_t *) malloc((size_t) tunit);
      for (i = 0; i < tunit; i++) unit[i] = (uint8_t) (splitmix(&s) & 3);
      for (i = 0; i < tl; i++) out[q + i] = unit[i % tunit];
      free(unit);
    }
}

/* Reads sampled from SEVERAL contigs laid out one after the other in `g` (contig c = g[coff[c] .. coff[c+1])):
 * the template start is uniform over all positions that leave `rl` bases inside one contig.  Otherwise as
 * gen_read.  cum[c] = number of valid starts in contigs 0..c-1 (cum[ncontig] = total).                  */
/* Template length of read i: `rl` when sigma = 0, else log-normal with mean rl (mu = ln rl - sigma^2 / 2; "CLR-like",
 * SURVEY 8d) clipped to [rl / 10, 4 rl], from a stream of its own (so that the fixed-length reads do not change).
 * cum[] is then made for templates of 4 rl bases: every start leaves room for the longest one. */
static int template_length(int rl, double sigma, uint64_t seed, int i)
{ if (sigma <= 0.) return rl;
  uint64_t s = stream_seed(seed ^ 0x1f83d9abfb41bd6bull, (uint64_t) i);
  double u1 = ((double) (splitmix(&s) >> 11) + 1.) / 9007199254740993.;
  double u2 = (double) (splitmix(&s) >> 11) / 9007199254740992.;
  double z  = sqrt(-2. * log(u1)) * cos(6.283185307179586 * u2);
  double t  = exp(log((double) rl) - .5 * sigma * sigma + sigma * z);
  if (t < rl / 10.) t = rl / 10.;
  if (t > 4. * rl)  t = 4. * rl;
  return (int) t;
}

static int gen_read_multi(const uint8_t *g, const int64_t *coff, const int64_t *cum, int ncontig, int rl, double err,
                          uint64_t seed, int i, uint8_t *out)
{ uint64_t s = stream_seed(seed ^ 0x3c6ef372fe94f82bull, (uint64_t) i);
  int64_t  u = (int64_t) (splitmix(&s) % (uint64_t) cum[ncontig]);
  int lo = 0, hi = ncontig - 1;
  while (lo < hi)
    { int m = (lo + hi + 1) >> 1;
      if (cum[m] <= u) lo = m; else hi = m - 1;
    }
  const uint8_t *c = g + coff[lo] + (u - cum[lo]);
  /* the per-read stream continues exactly like gen_read's after its position draw */
  int      rev = (int) (splitmix(&s) & 1);
  uint32_t tins = (uint32_t) (err * 0.60 * 4294967296.0);
  uint32_t tdel = (uint32_t) (err * 0.85 * 4294967296.0);
  uint32_t tsub = (uint32_t) (err * 4294967296.0);
  int n = 0, j;
  for (j = 0; j < rl; j++)
    { uint64_t r = splitmix(&s);
      uint32_t w = (uint32_t) r;
      int b = rev ? 3 - c[rl - 1 - j] : c[j];
      if (w < tins)
        { if (out) out[n] = (uint8_t) ((r >> 32) & 3);
          n += 1;
          if (out) out[n] = (uint8_t) b;
          n += 1;
        }
      else if (w < tdel)
        continue;
      else if (w < tsub)
        { if (out) out[n] = (uint8_t) ((b + 1 + ((r >> 32) % 3)) & 3);
          n += 1;
        }
      else
        { if (out) out[n] = (uint8_t) b;
          n += 1;
        }
    }
  return n;
}

void dmh_sim_lengths_multi(const uint8_t *g, const int64_t *coff, const int64_t *cum, int ncontig, int nreads, int rl,
                           double err, uint64_t seed, int *rlen, double sigma)
{ int i;
#pragma omp parallel for schedule(static)
  for (i = 0; i < nreads; i++)
    rlen[i] = gen_read_multi(g, coff, cum, ncontig, template_length(rl, sigma, seed, i), err, seed, i, NULL);
}

void dmh_sim_fill_multi(const uint8_t *g, const int64_t *coff, const int64_t *cum, int ncontig, int nreads, int rl,
                        double err, uint64_t seed, const int64_t *boff, uint8_t *bases, double sigma)
{ int i;
  bases[0] = 4;
#pragma omp parallel for schedule(static)
  for (i = 0; i < nreads; i++)
    { int n = gen_read_multi(g, coff, cum, ncontig, template_length(rl, sigma, seed, i), err, seed, i, bases + 1 + boff[i]);
      bases[1 + boff[i] + n] = 4;
    }
}

/* The block's slice of a .bps file from its in-memory layout: read i's rlen bases at bases[1 + boff[i]] go to
 * out[coff[i] ..] four per byte, first base in the top bits (DB.c:319-338). */
void dmh_pack_bps(const uint8_t *bases, const int64_t *boff, const int64_t *coff, int nreads, uint8_t *out)
{ int i;
#pragma omp parallel for schedule(dynamic, 64)
  for (i = 0; i < nreads; i++)
    { const uint8_t *s = bases + 1 + boff[i];
      int64_t  len = boff[i + 1] - boff[i] - 1, j;
      uint8_t *o = out + coff[i];
      for (j = 0; j + 4 <= len; j += 4)
        *o++ = (uint8_t) ((s[j] << 6) | (s[j + 1] << 4) | (s[j + 2] << 2) | s[j + 3]);
      if (j < len)
        { uint8_t v = 0;
          int     k;
          for (k = 0; k < 4; k++) v = (uint8_t) ((v << 2) | (j + k < len ? s[j + k] : 0));
          *o = v;
        }
    }
}
