#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (oracle/): makes the driver variant INTEGRATION.md describes -- the reference's damapper.c with
its block loop (damapper.c:835-877) changed to build the REF indices once and keep them for every READS block --
WITHOUT keeping any reference text in this repository: the recipe below is line numbers of the reference file plus
the lines that replace them.  The result goes to oracle/_ref/src/ (git-ignored), is compiled there by oracle/Makefile
into oracle/_ref/damapper_hip_resident, and linked against libdamapper_hip.so like oracle/_ref/damapper_hip.

usage: resident_driver.py <reference damapper.c> <output .c>
"""
import hashlib
import sys

# sha256 of the damapper.c the line numbers refer to
EXPECT = "0056b661a221f2ea303ea913a6b7ee8434504c431a767ecd4ec50a9e50289e91"

DECLS = r'''  DAZZ_DB    *rblock = NULL;      //  resident reference: block k-1 and its two indices, built once
  void      **ridx = NULL;
  int        *rlens  = NULL;
  int         rwhole = 0;         //  ablock holds the whole reference (for Reporter)
'''

EXTERN = r'''extern void dmh_retain_ref_index(int on);     //  libdamapper_hip.so (include/damapper_hip.h)
extern void dmh_free_index(void *table);
'''

# with the reference indices retained Reporter leaves the READS index to the driver: a plain free() of the handle (as in
# damapper.c:879) would leak its 3.4 GB of HBM per READS block
FREE = r'''        dmh_free_index(bindex);
'''

LOOP = r'''        if (rblock == NULL)
          { rblock = (DAZZ_DB *) Malloc(sizeof(DAZZ_DB)*nblocks,"Allocating resident reference");
            ridx = (void **) Malloc(sizeof(void *)*2*nblocks,"Allocating resident reference");
            rlens  = (int *) Malloc(sizeof(int)*2*nblocks,"Allocating resident reference");
            if (rblock == NULL || ridx == NULL || rlens == NULL)
              Clean_Exit(1);
            dmh_retain_ref_index(1);       //  Match_Filter leaves the reference indices alone
            for (k = 1; k <= nblocks; k++)
              { if (isdam)
                  afile = Strdup(Catenate(apwd,"/",aroot,Numbered_Suffix(".",k,".dam")),
                                 "Allocating name");
                else
                  afile = Strdup(Catenate(apwd,"/",aroot,Numbered_Suffix(".",k,".db")),
                                 "Allocating name");
                read_DB(rblock+(k-1),afile,MASK,MSTAT,MTOP,KMER_LEN);
                free(afile);
                if (VERBOSE)
                  printf("\nBuilding indices for %s.%d and its complement\n",aroot,k);
                ridx[2*k-2] = Sort_Kmers(rblock+(k-1),rlens+(2*k-2));
                complement_DB(rblock+(k-1),1);
                ridx[2*k-1] = Sort_Kmers(rblock+(k-1),rlens+(2*k-1));
              }
          }

        for (k = 1; k <= nblocks; k++)
          for (j = 0; j < 2; j++)
            { if (VERBOSE)
                printf("\nComparing %s to %s%s.%d%s\n",broot,j?"c(":"",aroot,k,j?")":"");
              Match_Filter(bblock,rblock+(k-1),bindex,blen,ridx[2*k-2+j],rlens[2*k-2+j],j,(k==1 && j==0));
            }

        if (!rwhole)
          { if (isdam)
              afile = Strdup(Catenate(apwd,"/",aroot,".dam"),"Allocating name");
            else
              afile = Strdup(Catenate(apwd,"/",aroot,".db"),"Allocating name");
            read_DB(ablock,afile,MASK,MSTAT,0,KMER_LEN);
            free(afile);
            rwhole = 1;
          }

        if (VERBOSE)
          printf("\nFinding best matches for block %s\n",broot);
        Reporter(broot,bblock,aroot,ablock,settings,mflag);
'''

# (first line, last line, replacement) -- 1-based, inclusive; first > last: insert before `first`
RECIPE = [
    (51, 50, EXTERN),          # behind #include "map.h"
    (559, 558, DECLS),         # behind the declaration of ablock / bblock
    (835, 877, LOOP),          # the loop over the REF blocks, the whole-REF load, Reporter, Close_DB(ablock)
    (879, 879, FREE),          # free(bindex)
]


def main():
    src, out = sys.argv[1], sys.argv[2]
    data = open(src, "rb").read()
    if hashlib.sha256(data).hexdigest() != EXPECT:
        sys.exit("resident_driver.py: %s is not the damapper.c this recipe was written for" % src)
    lines = data.decode().splitlines(keepends=True)
    for first, last, text in sorted(RECIPE, reverse=True):
        lines[first - 1:last] = [text]
    open(out, "w").write("".join(lines))


if __name__ == "__main__":
    main()
