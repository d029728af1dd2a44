/* oracle_spec.c -- TEST INFRASTRUCTURE ONLY: align.c:222-269 (New_Align_Spec) restated. */
#include "oracle.h"

static const double BiasFactor[10] = { .690, .690, .690, .690, .780, .850, .900, .933, .966, 1.000 };

void oracle_align_spec(double ave_corr, const float *freq, int *ave_path, int16_t *score, int16_t *table)
{ float  fm = freq[0] + freq[3];
  double match = fm;
  int    bias, mscore, dscore, x, bit;

  if ((match <= 0.) == (match > 0.)) match = .5;
  if (match > .5) match = 1.-match;
  bias = (int) ((match+.025)*20.-1.);
  if (match < .2) bias = 3;
  *ave_path = (int) (60 * (1. - BiasFactor[bias] * (1. - ave_corr)));
  mscore = (int) (1000 * BiasFactor[bias] * (1. - ave_corr));
  dscore = 1000 - mscore;
  for (x = 0; x < 32768; x++)
    { int s = 0, max = 0;
      for (bit = 14; bit >= 0; bit--)
        { if (s > max) max = s;
          if ((x >> bit) & 1) s += mscore; else s -= dscore;
        }
      table[x] = (int16_t) (s-max);
      score[x] = (int16_t) s;
    }
}

void oracle_free(void *p)
{ extern void free(void *);
  free(p);
}
