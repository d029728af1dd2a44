"""Shared driver for the filter-stage parity tests: runs Sort_Kmers / Match_Filter over every REF
block in both orientations exactly as damapper.c's main() does (damapper.c:825-866), through
either the HIP library (capi) or the compiled reference (ref_lib), and collects the stage outputs."""
import numpy as np

from damapper_amd import dazzdb


def make_dataset(kind: str, seed: int, contig_lens, nreads: int, read_len: int = 10000, dropouts: int = 0,
                 err: float = 0.15):
    rng = np.random.default_rng(seed)
    if kind == "random":
        g = dazzdb.random_genome(rng, contig_lens)
    else:
        g = dazzdb.repeat_rich_genome(rng, contig_lens)
    reads, truth = dazzdb.simulate_reads(rng, g, nreads, read_len=read_len, dropouts=dropouts, err=err)
    return g, reads


def ref_blocks_of(genome, firsts):
    """Blocks as Open_DB would hand them out: contigs [firsts[k], firsts[k+1]) with tfirst."""
    out = []
    for k in range(len(firsts) - 1):
        out.append(dazzdb.make_block(genome[firsts[k]:firsts[k + 1]], tfirst=firsts[k], part=k + 1,
                                     path=b"./.REF"))
    return out


class Api:
    """Uniform facade over capi (HIP) and ref_lib (reference)."""

    def __init__(self, kind):
        self.kind = kind
        if kind == "hip":
            from damapper_amd import capi
            self.m = capi
        else:
            import ref_lib
            self.m = ref_lib

    def set_globals(self, **kw):
        self.m.set_globals(**kw)

    def set_params(self, K, supp, T):
        self.m.lib().Set_Filter_Params(K, supp, T)

    def sort_kmers(self, blk):
        if self.kind == "hip":
            return self.m.sort_kmers(blk)
        h, n, _ = self.m.sort_kmers(blk, copy=False)
        return h, n

    def match_filter(self, *a):
        self.m.match_filter(*a)

    def seeds(self):
        return self.m.last_seeds()

    def candidates(self, reads, ar):
        return self.m.read_candidates(reads, ar)

    def jumps(self, reads, ar, k):
        return self.m.candidate_jumps(reads, ar, k)

    def cover(self, reads, ar):
        return self.m.read_cover(reads, ar)

    def free_reads_index(self, h):
        if self.kind == "hip":
            self.m.lib().dmh_free_index(h)
        else:
            self.m.free(h)


def run_filter(api: Api, reads_seqs, genome, firsts, K=20, supp=0, T=4, mem_limit=8 << 30, profile=0,
               spacing=100, want_seeds=True):
    api.set_globals(verbose=0, profile=profile, spacing=spacing, best_tie=1.0, sort_path="/tmp",
                    mem_limit=mem_limit, mem_physical=mem_limit)
    api.set_params(K, supp, T)
    if api.kind == "hip":
        api.m.lib().dmh_capture_seeds(1 if want_seeds else 0)
    reads = dazzdb.make_block(reads_seqs, path=b"./.READS")
    blocks = ref_blocks_of(genome, firsts)
    rh, rn = api.sort_kmers(reads)
    seeds = []
    for k, blk in enumerate(blocks):
        for comp in (0, 1):
            if comp:
                blk.complement_inplace()
            h, n = api.sort_kmers(blk)
            api.match_filter(reads, blk, rh, rn, h, n, comp, 1 if (k == 0 and comp == 0) else 0)
            if want_seeds:
                seeds.append(api.seeds())
        blk.complement_inplace()
    cands = []
    for ar in range(reads.nreads):
        c = api.candidates(reads, ar)
        js = [api.jumps(reads, ar, k) for k in range(len(c))]
        cands.append((c[:, :8].copy(), js))
    cover = [api.cover(reads, ar).copy() for ar in range(reads.nreads)] if profile else None
    return dict(reads=reads, blocks=blocks, rh=rh, rn=rn, seeds=seeds, cands=cands, cover=cover)
