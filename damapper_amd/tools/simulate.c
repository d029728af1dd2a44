/* simulate.c -- host-side synthetic PacBio-like read generator for tests and bench (no GPU code).
 * Reads are sampled uniformly (position, strand) from one contig, fixed template length, i.i.d.
 * per-base errors at rate `err` split ins:del:sub = 60:25:15, inserted/substituted base uniform
 * (SURVEY.md 8d).  Each read has its own RNG stream derived from (seed, read index), so the result
 * does not depend on the thread count.  Output is the DAZZ_DB in-memory block layout
 * [4] r0 [4] r1 ... [4] (DB.c:1402-1437).
 */
#include <stdint.h>
#include <stdlib.h>
#include <omp.h>

static inline uint64_t splitmix(uint64_t *s)
{ uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

/* Generates read i; if out != NULL writes its bases there.  Returns the read length. */
static int gen_read(const uint8_t *g, int64_t glen, int rl, double err, uint64_t seed, int i, uint8_t *out)
{ uint64_t s = seed * 0x2545f4914f6cdd1dull + (uint64_t) i * 0x9e3779b97f4a7c15ull + 1;
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
