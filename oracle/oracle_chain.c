/* oracle_chain.c -- TEST INFRASTRUCTURE ONLY.  Tree-free restatement of the chain scoring of one
 * (read, contig) group of sorted seed hits (map.c:1781-1872 per hit, 1634-1649 reporting, 1243-1260
 * chain compression), following SURVEY.md App. B: only the live set, the two nearest-neighbour
 * queries and the reporting order of the reference's splay tree are observable.
 * Quadratic in the window size -- meant for test-sized inputs. */
#include <stdint.h>
#include <stdlib.h>
#include "oracle.h"

#define MAX_GAP 1000   /* map.c:36 */

typedef struct { int apos, bpos, diag, cost, from, orig, live, expbest; } Node;

int oracle_chain_group(const int32_t *apos0, const int32_t *bpos0, int n, int kmer,
                       int32_t *out, int maxout, int32_t *from_out)
{ Node *N = (Node *) malloc(sizeof(Node)*(n+1));
  int   j, x, qhead = 0, nout = 0;
  int   hithr = 3*kmer;

  for (j = 0; j < n; j++)
    { int apos = apos0[j]+1, bpos = bpos0[j]+1, diag = apos-bpos;   /* map.c:1784-1785 */
      int l = -1, r = -1, lcost = 0, rcost = 0, p, c;

      while (qhead < j && N[qhead].apos < apos-MAX_GAP)             /* map.c:1787-1796 */
        { Node *q = N+qhead;
          if (q->live)
            { q->live = 0;
              q->expbest = (N[q->orig].orig == qhead);
            }
          qhead += 1;
        }

      /* predOf + leftmost (map.c:1262-1300): smallest larger diagonal with a live hit whose
         bpos >= bpos-MAX_GAP, then the live hit with the largest apos on that diagonal */
      { int dstar = 0x7fffffff;
        for (x = qhead; x < j; x++)
          if (N[x].live && N[x].diag > diag && N[x].bpos >= bpos-MAX_GAP && N[x].diag < dstar)
            dstar = N[x].diag;
        if (dstar != 0x7fffffff)
          for (x = qhead; x < j; x++)
            if (N[x].live && N[x].diag == dstar && (l < 0 || N[x].apos > N[l].apos))
              l = x;
      }
      /* succOf (map.c:1303-1318): largest key (diag,apos) <= own with bpos <= own bpos */
      for (x = qhead; x < j; x++)
        if (N[x].live && N[x].diag <= diag && N[x].bpos <= bpos)
          if (r < 0 || N[x].diag > N[r].diag || (N[x].diag == N[r].diag && N[x].apos > N[r].apos))
            r = x;

      if (l >= 0)
        lcost = (apos >= N[l].apos+kmer) ? N[l].cost+kmer : N[l].cost+(apos-N[l].apos);
      if (r >= 0)
        rcost = (bpos >= N[r].bpos+kmer) ? N[r].cost+kmer : N[r].cost+(bpos-N[r].bpos);
      if (lcost > rcost) rcost = 0; else lcost = 0;

      N[j].apos = apos; N[j].bpos = bpos; N[j].diag = diag; N[j].live = 1; N[j].expbest = 0;
      p = -1; c = kmer;
      if (lcost > 0) { p = l; c = lcost; }
      else if (rcost > 0) { p = r; c = rcost; }
      if (p >= 0)
        { int root = (N[p].from < 0) ? p : N[p].orig;
          N[j].from = p; N[j].cost = c; N[j].orig = root;
          if (c >= N[N[root].orig].cost)
            { int dd = abs(N[p].diag-diag);
              N[root].orig = j;
              if (dd <= .2*(apos-N[p].apos))                      /* map.c:1837,1852 */
                N[p].live = 0;
            }
        }
      else
        { N[j].from = -1; N[j].cost = kmer; N[j].orig = j; }
    }

  /* report: live best ends by (diag,apos) descending, then expired best ends newest first */
  { int *ord = (int *) malloc(sizeof(int)*(n+1)), m = 0, i, k2;
    for (x = 0; x < n; x++)
      if (N[x].live && N[x].cost >= hithr && N[x].from >= 0 && N[N[x].orig].orig == x)
        ord[m++] = x;
    for (i = 1; i < m; i++)
      { int v = ord[i];
        for (k2 = i-1; k2 >= 0; k2--)
          { int y = ord[k2];
            if (N[y].diag > N[v].diag || (N[y].diag == N[v].diag && N[y].apos > N[v].apos)) break;
            ord[k2+1] = y;
          }
        ord[k2+1] = v;
      }
    for (x = n-1; x >= 0; x--)
      if (!N[x].live && N[x].expbest && N[x].cost >= hithr && N[x].from >= 0 && N[N[x].orig].orig == x)
        ord[m++] = x;
    for (i = 0; i < m && nout < maxout; i++)
      { int h = ord[i], root = N[h].orig;
        int32_t *o = out + 6*nout;
        o[0] = N[h].cost; o[1] = N[root].apos-kmer; o[2] = N[h].apos;
        o[3] = N[root].bpos-kmer; o[4] = N[h].bpos; o[5] = h;
        nout += 1;
      }
    free(ord);
  }
  if (from_out != NULL)
    for (x = 0; x < n; x++) from_out[x] = N[x].from;
  free(N);
  return (nout);
}
