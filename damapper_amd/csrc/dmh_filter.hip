// dmh_filter.hip -- Match_Filter (map.c:2889-3209): merge -> seed order -> chain scoring ->
// candidate lists.  All stages are device code (dmh_merge.hip, dmh_readsort.h, dmh_chain.hip), the
// candidate lists with the domination rule and the Jump storage included; this file is the entry
// point with the reference's call contract.

#include "dmh_state.h"
#include "dmh_merge.h"
#include "dmh_chain.h"

using namespace dmh;

extern "C" {

void Match_Filter(DAZZ_DB *ablock, DAZZ_DB *bblock, void *atable, int alen, void *btable, int blen,
                  int comp, int start)
{ State &st = S();
  Params &pm = st.params;
  int64_t nhits = 0, nfilt = 0;
  report_poll();             // (a deferred Reporter whose jobs are ready launches its kernel; one that has finished is collected)
  if (start)
    { HostTimer ht("mf_reset_reads");
      chain_reset_reads(ablock);
    }

  if (alen != 0 && blen != 0 && atable != nullptr && btable != nullptr)
    { IndexDev *A = index_of(atable, "Match_Filter");
      IndexDev *B = index_of(btable, "Match_Filter");
      if (A->n != alen || B->n != blen)
        fatal("Match_Filter: table lengths (%d,%d) do not match the handles (%lld,%lld)", alen, blen,
              (long long) A->n, (long long) B->n);
      index_wait_ready(*A);
      index_wait_ready(*B);
      st.mf_calls += 1;
      nhits = merge_pass(ablock, bblock, *A, *B, pm.Kshift, comp, &nfilt);
      st.tfilt += nfilt;
    }

  // the reference consumes the REF table here (map.c:3181-3182)
  if (btable != nullptr && !st.retain_ref)
    { IndexStub *stub = (IndexStub *) btable;
      if (stub->magic == INDEX_MAGIC)
        { auto it = st.indices.find(stub->id);
          if (it != st.indices.end())
            { release_ref_index(it->second);        // into the REF cache when that is on
              st.indices.erase(it);
            }
          stub->magic = 0;
          free(stub);
        }
    }

  if (VERBOSE)
    { double atot = ablock->totlen, btot = bblock->totlen;
      printf("\n     %lld %d-mers (%e of matrix)\n     ", (long long) nhits, pm.Kmer, (1. * nhits / atot) / btot);
      if (nfilt < 0)
        printf("%lld candidates removed\n     ", (long long) -nfilt);
      else
        printf("%lld candidates added\n     ", (long long) nfilt);
      printf("%lld candidates (%e of matrix)\n", (long long) st.tfilt, (1. * st.tfilt / atot) / btot);
      fflush(stdout);
    }
}

// Pipelining hook (optional, results unchanged): the Match_Filter call AFTER the next one will be made with these two
// index handles.  The library may then queue that pass's match + count kernels -- which depend on the indices only --
// before the next call's last round trip to the host, so that the GPU's main stream does not wait while the host
// queues the rest of that call.  A wrong or missing hint costs time, never results.
void dmh_next_pass(void *atable, void *btable)
{ if (atable == nullptr || btable == nullptr) { filter_hint_next_pass(nullptr, nullptr, 0); return; }
  IndexDev *A = index_of(atable, "dmh_next_pass");
  IndexDev *B = index_of(btable, "dmh_next_pass");
  filter_hint_next_pass(A, B, S().params.Kshift);
}

int64_t dmh_last_seed_count(void) { return (int64_t) (S().last_seeds.size() / 4); }

int dmh_last_seeds(void *out)
{ State &st = S();
  memcpy(out, st.last_seeds.data(), st.last_seeds.size() * sizeof(int32_t));
  return (int) (st.last_seeds.size() / 4);
}

int dmh_last_limit(void) { return S().last_limit; }

}  // extern "C"
