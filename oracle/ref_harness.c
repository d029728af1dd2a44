/* ref_harness.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Builds the reference's map.c as part of this translation unit (it is #included from
 * where it lies under /root/reference, nothing is copied) so that tests can reach its
 * file-static state: the sorted seed-hit list handed to chain_thread (map.c:3138-3170),
 * the per-read candidate lists (map.c:1386-1397, 1668-1766) and the -p coverage vectors.
 *
 * The only interposition is on pthread_create, which lets us observe the arguments the
 * reference passes to its own stage threads without touching its source.
 *
 * Also supplies the globals damapper.c normally defines (damapper.c:58-65) and Clean_Exit.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

static int harness_pthread_create(pthread_t *t, const pthread_attr_t *a,
                                  void *(*fn)(void *), void *arg);

#define pthread_create harness_pthread_create
#include "map.c"
#undef pthread_create

int    VERBOSE      = 0;
int    PROFILE      = 0;
int    SPACING      = 100;
double BEST_TIE     = 1.0;
char  *SORT_PATH    = "/tmp";
uint64 MEM_LIMIT    = 0;
uint64 MEM_PHYSICAL = 0;

void Clean_Exit(int val)
{ exit(val); }

/* ---- captured seed hits of the most recent Match_Filter ---- */

static SeedPair *cap_hits = NULL;
static int64     cap_nhits = 0;
static int64     cap_max = 0;

static int harness_pthread_create(pthread_t *t, const pthread_attr_t *a,
                                  void *(*fn)(void *), void *arg)
{ if (fn == chain_thread)
    { Report_Arg *r = (Report_Arg *) arg;
      if (r->hbeg == 0)
        cap_nhits = 0;
      if (r->hend > cap_max)
        { cap_max  = r->hend + r->hend/4 + 1024;
          cap_hits = (SeedPair *) realloc(cap_hits,sizeof(SeedPair)*cap_max);
        }
      if (r->hend > r->hbeg)
        memcpy(cap_hits+r->hbeg,MR_hits+r->hbeg,sizeof(SeedPair)*(r->hend-r->hbeg));
      if (r->hend > cap_nhits)
        cap_nhits = r->hend;
    }
  return (pthread_create(t,a,fn,arg));
}

void ref_set_globals(int verbose, int profile, int spacing, double best_tie,
                     const char *sort_path, uint64 mem_limit, uint64 mem_physical)
{ VERBOSE      = verbose;
  PROFILE      = profile;
  SPACING      = spacing;
  BEST_TIE     = best_tie;
  SORT_PATH    = strdup(sort_path);
  MEM_LIMIT    = mem_limit;
  MEM_PHYSICAL = mem_physical;
}

int ref_nthreads(void)
{ return (NTHREADS); }

/* Sorted seed hits {diag,apos,bread,aread} (map.c:84-89) seen by the last chain pass */
void *ref_last_seeds(int64 *n)
{ *n = cap_nhits;
  return ((void *) cap_hits);
}

void ref_free(void *p)
{ free(p); }

static Report_Arg *owner_of(int ar)
{ int i;
  for (i = 0; i < NTHREADS; i++)
    if (ar >= parmr[i].abeg && ar < parmr[i].aend)
      return (parmr+i);
  return (NULL);
}

/* Candidate list of block-local read ar, head first.  9 ints per candidate:
   score length bread comp afirst alast bfirst blast chain(unused id).  Returns the count. */
int ref_read_candidates(DAZZ_DB *ablock, int ar, int *out, int maxc)
{ Report_Arg *r = owner_of(ar);
  int d, n;

  if (r == NULL)
    return (-1);
  n = 0;
  for (d = (int) ablock->reads[ar].coff; d >= 0; d = r->cbase[d].next)
    { if (n < maxc)
        { int *o = out + 9*n;
          o[0] = r->cbase[d].score;
          o[1] = r->cbase[d].length;
          o[2] = r->cbase[d].bread;
          o[3] = r->cbase[d].comp;
          o[4] = r->cbase[d].afirst;
          o[5] = r->cbase[d].alast;
          o[6] = r->cbase[d].bfirst;
          o[7] = r->cbase[d].blast;
          o[8] = d;
        }
      n += 1;
    }
  return (n);
}

/* Jump displacements (map.c:1380-1384) of the k-th candidate (list order) of read ar,
   end -> start.  Returns the chain length. */
int ref_candidate_jumps(DAZZ_DB *ablock, int ar, int k, uint16 *adisp, uint16 *bdisp, int maxj)
{ Report_Arg *r = owner_of(ar);
  int d, j, n, m, t, x;

  if (r == NULL)
    return (-1);
  d = (int) ablock->reads[ar].coff;
  while (k-- > 0 && d >= 0)
    d = r->cbase[d].next;
  if (d < 0)
    return (-1);
  m = r->cbase[d].length;
  n = 0;
  for (j = r->cbase[d].chain; j >= 0; j = r->jbase[j].next)
    { t = (m < 5) ? m : 5;
      for (x = 0; x < t; x++)
        { if (n < maxj)
            { adisp[n] = r->jbase[j].adisp[x];
              bdisp[n] = r->jbase[j].bdisp[x];
            }
          n += 1;
        }
      m -= 5;
    }
  return (n);
}

/* -p coverage deltas of block-local read ar (map.c:1580-1604,1654-1666): its (rlen-1)/SPACING+2 int16 slots
   inside the vector of the stage thread that owns the read.  Returns the slot count, -1 without an owner. */
int ref_read_cover(DAZZ_DB *ablock, int ar, int16 *out, int maxn)
{ Report_Arg *r = owner_of(ar);
  int16 *cnt;
  int    x, n, i;

  if (r == NULL || r->cover == NULL)
    return (-1);
  cnt = r->cover;
  for (x = r->abeg; x < ar; x++)
    cnt += (ablock->reads[x].rlen-1) / SPACING + 2;
  n = (ablock->reads[ar].rlen-1) / SPACING + 2;
  for (i = 0; i < n && i < maxn; i++)
    out[i] = cnt[i];
  return (n);
}

/* -p coverage delta vector of stage thread i (map.c:1580-1587) */
void *ref_cover(int i, int *comax)
{ *comax = parmr[i].comax;
  return ((void *) parmr[i].cover);
}
