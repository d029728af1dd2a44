#pragma once
#include "dmh_common.h"

namespace dmh {

constexpr uint64_t SPEC_MAGIC = 0x444d485350454321ull;

struct Spec
{ uint64_t magic;
  double   ave_corr;
  int      trace_space;
  int      reach;
  float    freq[4];
  int      ave_path;
  int      mscore, dscore;
  std::vector<int16_t> score, table;   // 32768 entries each (align.c:256-266)
  int16_t *d_tables = nullptr;         // device copy: score[32768] then table[32768]
};

int16_t *spec_device_tables(Spec *sp);
void     spec_release_device(Spec *sp);

}
