// dmh_report.hip -- Reporter (map.c:3227-3319) [under construction]
#include "dmh_state.h"
#include "dmh_spec.h"

namespace dmh {
int16_t *spec_device_tables(Spec *sp) { return nullptr; }
void spec_release_device(Spec *sp) { if (sp->d_tables) dev_free(sp->d_tables); sp->d_tables = nullptr; }
}
using namespace dmh;
extern "C" {
void Reporter(char *aname, DAZZ_DB *ablock, char *bname, DAZZ_DB *bblock, Align_Spec *asettings, int mflag)
{ fatal("Reporter: not implemented yet"); }
int dmh_local_alignment(Align_Spec *spec, const uint8_t *abases, int64_t abytes, const uint8_t *bbases, int64_t bbytes,
                        int njobs, const int64_t *aoff, const int *alen, const int64_t *boff, const int *blen,
                        const int *comp, const int *dg, const int *ad, int *out, uint16_t *trace_out,
                        uint16_t *btrace_out, int trace_stride)
{ fatal("dmh_local_alignment: not implemented yet"); return -1; }
}
