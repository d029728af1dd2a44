// dmh_spec.hip -- alignment specification (align.c:183-287): the trim tables TABLE/SCORE over all
// 15-column match patterns and the derived thresholds the d-wave kernel needs.

#include "dmh_spec.h"

namespace dmh {

static const double Bias_Factor[10] = { .690, .690, .690, .690, .780, .850, .900, .933, .966, 1.000 };  // align.c:178

constexpr int PATH_LEN  = 60;
constexpr int TRIM_LEN  = 15;
constexpr int FRACTION  = 1000;

}

using namespace dmh;

extern "C" {

Align_Spec *New_Align_Spec(double ave_corr, int trace_space, float *freq, int reach)
{ Spec *sp = new Spec();
  sp->magic       = SPEC_MAGIC;
  sp->ave_corr    = ave_corr;
  sp->trace_space = trace_space;
  sp->reach       = reach;
  for (int i = 0; i < 4; i++) sp->freq[i] = freq[i];

  float  fm    = freq[0] + freq[3];                 // float sum, as in align.c:239
  double match = fm;
  if ((match <= 0.) == (match > 0.)) match = .5;
  if (match > .5) match = 1. - match;
  int bias = (int) ((match + .025) * 20. - 1.);
  if (match < .2)
    { fprintf(stderr, "Warning: Base bias worse than 80/20%% ! (New_Align_Spec)\n");
      fprintf(stderr, "         Capping bias at this ratio.\n");
      bias = 3;
    }
  sp->ave_path = (int) (PATH_LEN * (1. - Bias_Factor[bias] * (1. - ave_corr)));
  int mscore   = (int) (FRACTION * Bias_Factor[bias] * (1. - ave_corr));
  int dscore   = FRACTION - mscore;
  sp->mscore = mscore;
  sp->dscore = dscore;

  // align.c:207-218: columns are consumed most significant bit first; `max` is the largest
  // prefix score seen BEFORE each column (the final score does not update it).
  sp->score.resize(1 << TRIM_LEN);
  sp->table.resize(1 << TRIM_LEN);
  for (int x = 0; x < (1 << TRIM_LEN); x++)
    { int score = 0, max = 0;
      for (int bit = TRIM_LEN - 1; bit >= 0; bit--)
        { if (score > max) max = score;
          if ((x >> bit) & 1) score += mscore; else score -= dscore;
        }
      sp->table[x] = (int16_t) (score - max);
      sp->score[x] = (int16_t) score;
    }
  return (Align_Spec *) sp;
}

void Free_Align_Spec(Align_Spec *spec)
{ Spec *sp = (Spec *) spec;
  if (sp == nullptr || sp->magic != SPEC_MAGIC) return;
  spec_release_device(sp);
  delete sp;
}

int Trace_Spacing(Align_Spec *spec) { return ((Spec *) spec)->trace_space; }
double Average_Correlation(Align_Spec *spec) { return ((Spec *) spec)->ave_corr; }
float *Base_Frequencies(Align_Spec *spec) { return ((Spec *) spec)->freq; }
int Overlap_If_Possible(Align_Spec *spec) { return ((Spec *) spec)->reach; }

}
