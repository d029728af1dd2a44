/* oracle_align.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's local
 * alignment (align.c:353-1946: forward_wave, reverse_wave, Local_Alignment) in the wave-parallel
 * form the HIP kernel uses: every diagonal of a wave is computed from a FROZEN copy of the
 * previous wave, and only the few order-dependent pieces are expressed as ordered reductions
 * (SURVEY.md App. E):
 *   - best/trim bookkeeping = record setters of a running max (descending k forward, align.c:819-833;
 *     ascending k reverse, align.c:1471-1485),
 *   - boundary clips = min/max diagonal that ran into a terminator (align.c:750-762, 1405-1417),
 *   - pebble allocation is order free (align.c:771-817).
 * Pinned against the compiled reference by tests/test_oracle_align.py (thousands of calls,
 * both orientations, retries and boundary cases).
 *
 * Restricted to the way damapper calls it: low == hgh == seed diagonal, lbord = hbord = -1
 * (map.c:2513), A and B are different sequences.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define PATH_LEN   60                      /* align.c:169 */
#define PATH_TOP   0x1000000000000000ull   /* align.c:173 */
#define PATH_INT   0x0fffffffffffffffull   /* align.c:174 */
#define TRIM_LEN   15                      /* align.c:163 */
#define TRIM_MASK  0x7fff
#define TRIM_MLAG  250                     /* align.c:176 */
#define WAVE_LAG   30                      /* align.c:177 */
#define DUB_TRIM   45                      /* align.c:167 */
#define BIG        0x7fffffff

typedef struct { int ptr, diag, diff, mark; } Peb;

typedef struct
  { const uint8_t *a, *b;         /* a[-1] = a[alen] = 4, same for b */
    int      alen, blen;
    int      TS, PATH_AVE, REACH;
    const int16_t *SCORE, *TABLE;
    /* wave state, indexed by diagonal + koff */
    int      koff, span;
    int     *V, *M, *HA, *HB, *HAM, *HBM, *NA, *NB;
    uint64_t *T;
    int     *nV, *nM, *nHA, *nHB, *nHAM, *nHBM;
    uint64_t *nT;
    Peb     *cells;
    int      ncell, maxcell;
  } Ctx;

static int new_peb(Ctx *c, int ptr, int diag, int diff, int mark)
{ if (c->ncell >= c->maxcell)
    { c->maxcell = c->maxcell*2 + 1024;
      c->cells = (Peb *) realloc(c->cells,sizeof(Peb)*c->maxcell);
    }
  Peb *p = c->cells + c->ncell;
  p->ptr = ptr; p->diag = diag; p->diff = diff; p->mark = mark;
  return (c->ncell++);
}

typedef struct
  { int abpos, bbpos, aepos, bepos, diffs;
    int atlen, btlen;        /* trace lengths; forward fills [0,len), reverse prepends */
    uint16_t *atrace, *btrace;   /* pointers to element 0 of the forward part */
    int aorg, borg;          /* current start offset (<= 0) relative to atrace/btrace */
  } Paths;

/* ---- forward wave (align.c:353-1011) ------------------------------------------------- */

static void forward_wave(Ctx *C, Paths *P, int *mind, int maxd, int mida, int minp, int maxp,
                         int aoff, int boff)
{ const uint8_t *aseq = C->a, *bseq = C->b;
  const int TS = C->TS, KO = C->koff;
  int *V = C->V+KO, *M = C->M+KO, *HA = C->HA+KO, *HB = C->HB+KO;
  int *HAM = C->HAM+KO, *HBM = C->HBM+KO, *NA = C->NA+KO, *NB = C->NB+KO;
  uint64_t *T = C->T+KO;
  int *nV = C->nV+KO, *nM = C->nM+KO, *nHA = C->nHA+KO, *nHB = C->nHB+KO;
  int *nHAM = C->nHAM+KO, *nHBM = C->nHBM+KO;
  uint64_t *nT = C->nT+KO;

  int hgh = maxd, low = *mind, dif = 0;
  int more = 1, aclip = BIG, bclip = -BIG;
  int besta, besty, trima, trimy, trimd, trimha, trimhb;
  int morea, morey, mored, moreha, morehb, morem, lasta;
  int k;

  C->ncell = 0;
  besta = trima = morea = lasta = mida;
  besty = trimy = morey = (mida-hgh) >> 1;
  trimd = mored = 0;
  trimha = moreha = 0;
  trimhb = morehb = 1;
  morem = -1;

  /* 0-wave (align.c:428-547) */
  for (k = hgh; k >= low; k--)
    { int y = (mida-k) >> 1, c, ha, hb, na, nb;
      na = (((y+k)+(TS-aoff))/TS-1)*TS+aoff;
      ha = new_peb(C,-1,k,0,na);
      na += TS;
      nb = ((y+(TS-boff))/TS-1)*TS+boff;
      hb = new_peb(C,-1,k,0,nb);
      nb += TS;
      while (1)
        { int cb = bseq[y], ca;
          if (cb == 4)
            { more = 0;
              if (bclip < k) bclip = k;
              break;
            }
          ca = aseq[k+y];
          if (cb != ca)
            { if (ca == 4) { more = 0; aclip = k; }
              break;
            }
          y += 1;
        }
      c = (y << 1) + k;
      while (y+k >= na) { ha = new_peb(C,ha,k,0,na); na += TS; }
      while (y >= nb)   { hb = new_peb(C,hb,k,0,nb); nb += TS; }
      if (c > besta)
        { besta = trima = lasta = c;
          besty = trimy = y;
          trimha = ha;
          trimhb = hb;
        }
      V[k] = c; T[k] = PATH_INT; M[k] = PATH_LEN;
      HA[k] = ha; HB[k] = hb; HAM[k] = C->cells[ha].mark; HBM[k] = C->cells[hb].mark;
      NA[k] = na; NB[k] = nb;
    }

#define FWD_BOUNDARY(setd)                                                   \
  if (more == 0)                                                             \
    { if (bseq[besty] != 4 && aseq[besta-besty] != 4)                        \
        more = 1;                                                            \
      if (hgh >= aclip)                                                      \
        { hgh = aclip-1;                                                     \
          if (morem <= M[aclip])                                             \
            { morem = M[aclip]; morea = V[aclip]; morey = (morea-aclip)/2;   \
              setd moreha = HA[aclip]; morehb = HB[aclip];                   \
            }                                                                \
        }                                                                    \
      if (low <= bclip)                                                      \
        { low = bclip+1;                                                     \
          if (morem <= M[bclip])                                             \
            { morem = M[bclip]; morea = V[bclip]; morey = (morea-bclip)/2;   \
              setd moreha = HA[bclip]; morehb = HB[bclip];                   \
            }                                                                \
        }                                                                    \
      aclip = BIG;                                                           \
      bclip = -BIG;                                                          \
    }

  FWD_BOUNDARY(;)

  while (more && lasta >= besta - TRIM_MLAG)
    { int run;

      low -= 1;
      hgh += 1;
      if (low >= minp)
        { NA[low] = NA[low+1]; NB[low] = NB[low+1]; V[low] = -1; }
      else
        low += 1;
      if (hgh <= maxp)
        { NA[hgh] = NA[hgh-1]; NB[hgh] = NB[hgh-1]; V[hgh] = -1; }
      else
        hgh -= 1;
      dif += 1;
      V[hgh+1] = V[low-1] = -1;

      /* phase 1: every diagonal from the frozen previous wave */
      for (k = low; k <= hgh; k++)
        { int am = V[k-1], ac = V[k], ap = V[k+1];
          int c, m, ha, hb, ham, hbm, s, y;
          uint64_t b;

          if (ac < am)
            s = (am < ap) ? k+1 : k-1;
          else
            s = (ac < ap) ? k+1 : k;
          c = (s == k) ? ac+2 : V[s]+1;
          m = M[s]; b = T[s]; ha = HA[s]; hb = HB[s]; ham = HAM[s]; hbm = HBM[s];
          if (s == k+1 && k+1 > hgh)      /* source outside the band: the reference's t/n/ua/ub */
            { m = PATH_LEN; b = PATH_INT; ha = hb = -1; ham = hbm = 0; }

          if ((b & PATH_TOP) != 0) m -= 1;
          b <<= 1;

          y = (c-k) >> 1;
          while (1)
            { int cb = bseq[y], ca;
              if (cb == 4)
                { more = 0;
                  if (bclip < k) bclip = k;
                  break;
                }
              ca = aseq[k+y];
              if (cb != ca)
                { if (ca == 4) { more = 0; if (aclip > k) aclip = k; }
                  break;
                }
              y += 1;
              if ((b & PATH_TOP) == 0) m += 1;
              b = (b << 1) | 1;
            }
          c = (y << 1) + k;

          while (y+k >= NA[k])
            { if (ham < NA[k])
                { ha = new_peb(C,ha,k,dif,NA[k]); ham = NA[k]; }
              NA[k] += TS;
            }
          while (y >= NB[k])
            { if (hbm < NB[k])
                { hb = new_peb(C,hb,k,dif,NB[k]); hbm = NB[k]; }
              NB[k] += TS;
            }
          nV[k] = c; nT[k] = b; nM[k] = m; nHA[k] = ha; nHB[k] = hb; nHAM[k] = ham; nHBM[k] = hbm;
        }

      /* phase 2: record setters of the running max, descending k (align.c:819-833) */
      run = besta;
      for (k = hgh; k >= low; k--)
        if (nV[k] > run)
          { uint64_t b = nT[k];
            run = nV[k];
            besta = nV[k];
            besty = (nV[k]-k) >> 1;
            if (nM[k] >= C->PATH_AVE)
              { lasta = nV[k];
                if (C->TABLE[b & TRIM_MASK] >= 0)
                  if (C->TABLE[(b >> TRIM_LEN) & TRIM_MASK] + C->SCORE[b & TRIM_MASK] >= 0)
                    { trima = nV[k]; trimy = besty; trimd = dif;
                      trimha = nHA[k]; trimhb = nHB[k];
                    }
              }
          }
      for (k = low; k <= hgh; k++)
        { V[k] = nV[k]; T[k] = nT[k]; M[k] = nM[k]; HA[k] = nHA[k]; HB[k] = nHB[k];
          HAM[k] = nHAM[k]; HBM[k] = nHBM[k];
        }

      FWD_BOUNDARY(mored = dif;)

      { int n = besta - WAVE_LAG;
        while (hgh >= low)
          if (V[hgh] < n)
            hgh -= 1;
          else
            { while (V[low] < n) low += 1;
              break;
            }
      }
    }

  /* trace back (align.c:895-1006) */
  { uint16_t *atrace = P->atrace + P->aorg, *btrace = P->btrace + P->borg;
    int atlen = 0, btlen = 0, trimx;
    int a, b, h, d, e, n, i;
    int *chain;

    if (morem >= 0 && C->REACH)
      { trimx = morea-morey; trimy = morey; trimd = mored; trimha = moreha; trimhb = morehb; }
    else
      trimx = trima-trimy;

    chain = (int *) malloc(sizeof(int)*(C->ncell+1));

    n = 0;
    for (h = trimha; h >= 0; h = C->cells[h].ptr) chain[n++] = h;
    k = C->cells[chain[n-1]].diag;
    b = (mida-k)/2;
    e = 0;
    for (i = n-2; i >= 0; i--)
      { h = chain[i];
        k = C->cells[h].diag;
        a = C->cells[h].mark - k;
        d = C->cells[h].diff;
        atrace[atlen++] = (uint16_t) (d-e);
        atrace[atlen++] = (uint16_t) (a-b);
        b = a; e = d;
      }
    if (b+k != trimx)
      { atrace[atlen++] = (uint16_t) (trimd-e);
        atrace[atlen++] = (uint16_t) (trimy-b);
      }
    else if (b != trimy && atlen > 0)
      { atrace[atlen-1] = (uint16_t) (atrace[atlen-1] + (trimy-b));
        atrace[atlen-2] = (uint16_t) (atrace[atlen-2] + (trimd-e));
      }

    n = 0;
    for (h = trimhb; h >= 0; h = C->cells[h].ptr) chain[n++] = h;
    k = C->cells[chain[n-1]].diag;
    b = (mida+k)/2;
    e = 0;
    low = k;
    for (i = n-2; i >= 0; i--)
      { h = chain[i];
        k = C->cells[h].diag;
        a = C->cells[h].mark + k;
        d = C->cells[h].diff;
        btrace[btlen++] = (uint16_t) (d-e);
        btrace[btlen++] = (uint16_t) (a-b);
        b = a; e = d;
      }
    if (b-k != trimy)
      { btrace[btlen++] = (uint16_t) (trimd-e);
        btrace[btlen++] = (uint16_t) (trimx-b);
      }
    else if (b != trimx && btlen > 0)
      { btrace[btlen-1] = (uint16_t) (btrace[btlen-1] + (trimx-b));
        btrace[btlen-2] = (uint16_t) (btrace[btlen-2] + (trimd-e));
      }
    free(chain);

    P->aepos = trimx;
    P->bepos = trimy;
    P->diffs = trimd;
    P->atlen = atlen;
    P->btlen = btlen;
  }
  *mind = low;
}

/* ---- reverse wave (align.c:1015-1720) ------------------------------------------------ */

static void reverse_wave(Ctx *C, Paths *P, int mind, int maxd, int mida, int minp, int maxp,
                         int aoff, int boff)
{ const uint8_t *aseq = C->a - 1, *bseq = C->b - 1;
  const int TS = C->TS, KO = C->koff;
  int *V = C->V+KO, *M = C->M+KO, *HA = C->HA+KO, *HB = C->HB+KO;
  int *HAM = C->HAM+KO, *HBM = C->HBM+KO, *NA = C->NA+KO, *NB = C->NB+KO;
  uint64_t *T = C->T+KO;
  int *nV = C->nV+KO, *nM = C->nM+KO, *nHA = C->nHA+KO, *nHB = C->nHB+KO;
  int *nHAM = C->nHAM+KO, *nHBM = C->nHBM+KO;
  uint64_t *nT = C->nT+KO;

  int hgh = maxd, low = mind, dif = 0;
  int more = 1, aclip = -BIG, bclip = BIG;
  int besta, besty, trima, trimy, trimd, trimha, trimhb;
  int morea, morey, mored, moreha, morehb, morem, lasta;
  int k;

  C->ncell = 0;
  besta = trima = morea = lasta = mida;
  besty = trimy = morey = (mida-hgh) >> 1;
  trimd = mored = 0;
  trimha = moreha = 0;
  trimhb = morehb = 1;
  morem = -1;

  for (k = low; k <= hgh; k++)
    { int y = (mida-k) >> 1, c, ha, hb, na, nb;
      na = (((y+k)+(TS-aoff)-1)/TS-1)*TS+aoff;
      ha = new_peb(C,-1,k,0,y+k);
      nb = ((y+(TS-boff)-1)/TS-1)*TS+boff;
      hb = new_peb(C,-1,k,0,y);
      while (1)
        { int cb = bseq[y], ca;
          if (cb == 4)
            { more = 0;
              if (bclip > k) bclip = k;
              break;
            }
          ca = aseq[k+y];
          if (cb != ca)
            { if (ca == 4) { more = 0; aclip = k; }
              break;
            }
          y -= 1;
        }
      c = (y << 1) + k;
      while (y+k <= na) { ha = new_peb(C,ha,k,0,na); na -= TS; }
      while (y <= nb)   { hb = new_peb(C,hb,k,0,nb); nb -= TS; }
      if (c < besta)
        { besta = trima = lasta = c;
          besty = trimy = y;
          trimha = ha;
          trimhb = hb;
        }
      V[k] = c; T[k] = PATH_INT; M[k] = PATH_LEN;
      HA[k] = ha; HB[k] = hb; HAM[k] = C->cells[ha].mark; HBM[k] = C->cells[hb].mark;
      NA[k] = na; NB[k] = nb;
    }

#define REV_BOUNDARY(setd)                                                   \
  if (more == 0)                                                             \
    { if (bseq[besty] != 4 && aseq[besta-besty] != 4)                        \
        more = 1;                                                            \
      if (low <= aclip)                                                      \
        { low = aclip+1;                                                     \
          if (morem <= M[aclip])                                             \
            { morem = M[aclip]; morea = V[aclip]; morey = (morea-aclip)/2;   \
              setd moreha = HA[aclip]; morehb = HB[aclip];                   \
            }                                                                \
        }                                                                    \
      if (hgh >= bclip)                                                      \
        { hgh = bclip-1;                                                     \
          if (morem <= M[bclip])                                             \
            { morem = M[bclip]; morea = V[bclip]; morey = (morea-bclip)/2;   \
              setd moreha = HA[bclip]; morehb = HB[bclip];                   \
            }                                                                \
        }                                                                    \
      aclip = -BIG;                                                          \
      bclip = BIG;                                                           \
    }

  REV_BOUNDARY(;)

  while (more && lasta <= besta + TRIM_MLAG)
    { int run;

      low -= 1;
      hgh += 1;
      if (low >= minp)
        { NA[low] = NA[low+1]; NB[low] = NB[low+1]; V[low] = BIG; }
      else
        low += 1;
      if (hgh <= maxp)
        { NA[hgh] = NA[hgh-1]; NB[hgh] = NB[hgh-1]; V[hgh] = BIG; }
      else
        hgh -= 1;
      dif += 1;
      V[hgh+1] = V[low-1] = BIG;

      for (k = low; k <= hgh; k++)
        { int am = V[k-1], ac = V[k], ap = V[k+1];
          int c, m, ha, hb, ham, hbm, s, y;
          uint64_t b;

          if (ac > ap)
            s = (ap > am) ? k-1 : k+1;
          else
            s = (ac > am) ? k-1 : k;
          c = (s == k) ? ac-2 : V[s]-1;
          m = M[s]; b = T[s]; ha = HA[s]; hb = HB[s]; ham = HAM[s]; hbm = HBM[s];
          if (s == k-1 && k-1 < low)
            { m = PATH_LEN; b = PATH_INT; ha = hb = -1; ham = hbm = 0; }

          if ((b & PATH_TOP) != 0) m -= 1;
          b <<= 1;

          y = (c-k) >> 1;
          while (1)
            { int cb = bseq[y], ca;
              if (cb == 4)
                { more = 0;
                  if (bclip > k) bclip = k;
                  break;
                }
              ca = aseq[k+y];
              if (cb != ca)
                { if (ca == 4) { more = 0; if (aclip < k) aclip = k; }
                  break;
                }
              y -= 1;
              if ((b & PATH_TOP) == 0) m += 1;
              b = (b << 1) | 1;
            }
          c = (y << 1) + k;

          while (y+k <= NA[k])
            { if (ham > NA[k])
                { ha = new_peb(C,ha,k,dif,NA[k]); ham = NA[k]; }
              NA[k] -= TS;
            }
          while (y <= NB[k])
            { if (hbm > NB[k])
                { hb = new_peb(C,hb,k,dif,NB[k]); hbm = NB[k]; }
              NB[k] -= TS;
            }
          nV[k] = c; nT[k] = b; nM[k] = m; nHA[k] = ha; nHB[k] = hb; nHAM[k] = ham; nHBM[k] = hbm;
        }

      run = besta;
      for (k = low; k <= hgh; k++)
        if (nV[k] < run)
          { uint64_t b = nT[k];
            run = nV[k];
            besta = nV[k];
            besty = (nV[k]-k) >> 1;
            if (nM[k] >= C->PATH_AVE)
              { lasta = nV[k];
                if (C->TABLE[b & TRIM_MASK] >= 0)
                  if (C->TABLE[(b >> TRIM_LEN) & TRIM_MASK] + C->SCORE[b & TRIM_MASK] >= 0)
                    { trima = nV[k]; trimy = besty; trimd = dif;
                      trimha = nHA[k]; trimhb = nHB[k];
                    }
              }
          }
      for (k = low; k <= hgh; k++)
        { V[k] = nV[k]; T[k] = nT[k]; M[k] = nM[k]; HA[k] = nHA[k]; HB[k] = nHB[k];
          HAM[k] = nHAM[k]; HBM[k] = nHBM[k];
        }

      REV_BOUNDARY(mored = dif;)

      { int n = besta + WAVE_LAG;
        while (hgh >= low)
          if (V[hgh] > n)
            hgh -= 1;
          else
            { while (V[low] > n) low += 1;
              break;
            }
      }
    }

  /* trace back (align.c:1554-1714): the reverse part is PREPENDED to the forward trace */
  { uint16_t *atrace = P->atrace + P->aorg, *btrace = P->btrace + P->borg;
    int atlen = 0, btlen = 0, trimx;
    int a, b, h, d, e, n, i;
    int *chain;

    if (morem >= 0 && C->REACH)
      { trimx = morea-morey; trimy = morey; trimd = mored; trimha = moreha; trimhb = morehb; }
    else
      trimx = trima-trimy;

    chain = (int *) malloc(sizeof(int)*(C->ncell+1));

    /* A view */
    n = 0;
    for (h = trimha; h >= 0; h = C->cells[h].ptr) chain[n++] = h;
    i = n-1;
    h = chain[i];
    k = C->cells[h].diag;
    b = C->cells[h].mark - k;
    e = 0;
    if ((b+k)%TS != aoff)
      { i -= 1;
        if (i < 0)
          { h = -1; a = trimy; d = trimd; }
        else
          { h = chain[i]; k = C->cells[h].diag; a = C->cells[h].mark - k; d = C->cells[h].diff; }
        if (P->atlen == 0)
          { atrace[--atlen] = (uint16_t) (b-a);
            atrace[--atlen] = (uint16_t) (d-e);
          }
        else
          { atrace[1] = (uint16_t) (atrace[1] + (b-a));
            atrace[0] = (uint16_t) (atrace[0] + (d-e));
          }
        b = a; e = d;
      }
    if (h >= 0)
      { for (i -= 1; i >= 0; i--)
          { h = chain[i];
            k = C->cells[h].diag;
            a = C->cells[h].mark - k;
            atrace[--atlen] = (uint16_t) (b-a);
            d = C->cells[h].diff;
            atrace[--atlen] = (uint16_t) (d-e);
            b = a; e = d;
          }
        if (b+k != trimx)
          { atrace[--atlen] = (uint16_t) (b-trimy);
            atrace[--atlen] = (uint16_t) (trimd-e);
          }
        else if (b != trimy && (atlen < 0 || P->atlen > 0))
          { atrace[atlen+1] = (uint16_t) (atrace[atlen+1] + (b-trimy));
            atrace[atlen]   = (uint16_t) (atrace[atlen]   + (trimd-e));
          }
      }

    /* B view */
    n = 0;
    for (h = trimhb; h >= 0; h = C->cells[h].ptr) chain[n++] = h;
    i = n-1;
    h = chain[i];
    k = C->cells[h].diag;
    b = C->cells[h].mark + k;
    e = 0;
    if ((b-k)%TS != boff)
      { i -= 1;
        if (i < 0)
          { h = -1; a = trimx; d = trimd; }
        else
          { h = chain[i]; k = C->cells[h].diag; a = C->cells[h].mark + k; d = C->cells[h].diff; }
        if (P->btlen == 0)
          { btrace[--btlen] = (uint16_t) (b-a);       /* sic: align.c:1670-1671 writes b-a twice */
            btrace[--btlen] = (uint16_t) (b-a);
          }
        else
          { btrace[1] = (uint16_t) (btrace[1] + (b-a));
            btrace[0] = (uint16_t) (btrace[0] + (d-e));
          }
        b = a; e = d;
      }
    if (h >= 0)
      { for (i -= 1; i >= 0; i--)
          { h = chain[i];
            k = C->cells[h].diag;
            a = C->cells[h].mark + k;
            btrace[--btlen] = (uint16_t) (b-a);
            d = C->cells[h].diff;
            btrace[--btlen] = (uint16_t) (d-e);
            b = a; e = d;
          }
        if (b-k != trimy)
          { btrace[--btlen] = (uint16_t) (b-trimx);
            btrace[--btlen] = (uint16_t) (trimd-e);
          }
        else if (b != trimx && (btlen < 0 || P->btlen > 0))
          { btrace[btlen+1] = (uint16_t) (btrace[btlen+1] + (b-trimx));
            btrace[btlen]   = (uint16_t) (btrace[btlen]   + (trimd-e));
          }
      }
    free(chain);

    P->abpos = trimx;
    P->bbpos = trimy;
    P->diffs = P->diffs + trimd;
    P->atlen = P->atlen - atlen;
    P->aorg  = P->aorg + atlen;
    P->btlen = P->btlen - btlen;
    P->borg  = P->borg + btlen;
  }
}

/* ---- Local_Alignment (align.c:1727-1946) as damapper calls it -------------------------- */

int oracle_local_alignment(const uint8_t *aseq, int alen, const uint8_t *bseq, int blen, int acomp,
                           int tspace, int ave_path, int reach, const int16_t *score, const int16_t *table,
                           int dg, int anti, int *apath_out, uint16_t *atrace_out,
                           int *bpath_out, uint16_t *btrace_out)
{ Ctx C;
  Paths P;
  int span = alen + blen + 16;
  int maxtp, low, hgh, minp, maxp, aoff, boff, fshort, rshort, i;
  uint16_t *abuf, *bbuf;

  memset(&C,0,sizeof(C));
  C.a = aseq; C.b = bseq; C.alen = alen; C.blen = blen;
  C.TS = tspace; C.PATH_AVE = ave_path; C.REACH = reach; C.SCORE = score; C.TABLE = table;
  C.koff = blen + 8;
  C.span = span;
#define ALLOCI(f) C.f = (int *) calloc(span,sizeof(int))
  ALLOCI(V); ALLOCI(M); ALLOCI(HA); ALLOCI(HB); ALLOCI(HAM); ALLOCI(HBM); ALLOCI(NA); ALLOCI(NB);
  ALLOCI(nV); ALLOCI(nM); ALLOCI(nHA); ALLOCI(nHB); ALLOCI(nHAM); ALLOCI(nHBM);
  C.T = (uint64_t *) calloc(span,sizeof(uint64_t));
  C.nT = (uint64_t *) calloc(span,sizeof(uint64_t));
  C.maxcell = 4096;
  C.cells = (Peb *) malloc(sizeof(Peb)*C.maxcell);

  maxtp = 2*(((alen < blen) ? blen : alen)/tspace+2);
  abuf = (uint16_t *) calloc(4*maxtp+16,sizeof(uint16_t));
  bbuf = (uint16_t *) calloc(4*maxtp+16,sizeof(uint16_t));
  memset(&P,0,sizeof(P));
  P.atrace = abuf + 2*maxtp + 8;
  P.btrace = bbuf + 2*maxtp + 8;

  low = hgh = dg;
  while (((anti-hgh) >> 1) < 0)
    hgh -= 1;
  minp = -BIG;
  maxp = BIG;
  if (acomp) { aoff = alen % tspace; boff = 0; }
  else       { aoff = 0; boff = 0; }

  forward_wave(&C,&P,&low,hgh,anti,minp,maxp,aoff,boff);
  fshort = ((P.aepos + P.bepos) - anti < DUB_TRIM);
  reverse_wave(&C,&P,low,low,anti,minp,maxp,aoff,boff);
  rshort = (anti - (P.abpos + P.bbpos) < DUB_TRIM);

  if (fshort)
    { if (rshort)
        { P.aepos = P.abpos = (P.abpos+P.aepos)/2;
          P.bepos = P.bbpos = (P.bbpos+P.bepos)/2;
          P.atlen = P.btlen = 0;
        }
      else
        { low  = P.abpos - P.bbpos;
          anti = P.abpos + P.bbpos;
          P.atlen = P.btlen = 0;
          forward_wave(&C,&P,&low,low,anti,minp,maxp,aoff,boff);
        }
    }
  else if (rshort)
    { low  = P.aepos - P.bepos;
      anti = P.aepos + P.bepos;
      P.atlen = P.btlen = 0;
      P.diffs = 0;
      reverse_wave(&C,&P,low,low,anti,minp,maxp,aoff,boff);
    }

  { uint16_t *at = P.atrace + P.aorg, *bt = P.btrace + P.borg;
    int babpos, bbbpos, baepos, bbepos;

    babpos = P.bbpos; bbbpos = P.abpos; baepos = P.bepos; bbepos = P.aepos;
    if (acomp)
      { int abp = P.abpos, bbp = P.bbpos, aep = P.aepos, bep = P.bepos;
        int j;
        /* B view keeps the complemented-frame coordinates, A view is flipped (align.c:1858-1884) */
        babpos = bbp; bbbpos = abp; baepos = bep; bbepos = aep;
        P.abpos = alen - bbepos;
        P.bbpos = blen - baepos;
        P.aepos = alen - bbbpos;
        P.bepos = blen - babpos;
        (void) aep; (void) bep;
        i = P.atlen-2;
        j = 0;
        while (j < i)
          { uint16_t p = at[i]; at[i] = at[j]; at[j] = p;
            p = at[i+1]; at[i+1] = at[j+1]; at[j+1] = p;
            i -= 2;
            j += 2;
          }
      }
    apath_out[0] = P.abpos; apath_out[1] = P.bbpos; apath_out[2] = P.aepos; apath_out[3] = P.bepos;
    apath_out[4] = P.diffs; apath_out[5] = P.atlen;
    bpath_out[0] = babpos; bpath_out[1] = bbbpos; bpath_out[2] = baepos; bpath_out[3] = bbepos;
    bpath_out[4] = P.diffs; bpath_out[5] = P.btlen;
    for (i = 0; i < P.atlen; i++) atrace_out[i] = at[i];
    for (i = 0; i < P.btlen; i++) btrace_out[i] = bt[i];
  }

  free(C.V); free(C.M); free(C.HA); free(C.HB); free(C.HAM); free(C.HBM); free(C.NA); free(C.NB);
  free(C.nV); free(C.nM); free(C.nHA); free(C.nHB); free(C.nHAM); free(C.nHBM);
  free(C.T); free(C.nT); free(C.cells);
  free(abuf); free(bbuf);
  return (0);
}
