#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (oracle/): the reference's damapper.c with INTEGRATION.md section 3 applied IN FULL -- the C host
side that "stays" (BASELINE north_star) driving the library the way damapper_amd/driver.py does:

  * REF blocks read, complemented and indexed once (as resident_driver.py);
  * a READS block is opened and trimmed by DB.c (Open_DB, Trim_DB) but NOT decoded on the host: its slice of the .bps file
    is read into page-locked memory (dmh_host_alloc) and unpacked on the device (dmh_preload_block_2bit);
  * dmh_async_report(2): Reporter returns while the alignment kernel of its block runs next to the following block's
    filter, selection and .las writing happen on the library's worker thread;
  * two READS blocks open: Reporter of block i is called ten passes into block i + 1;
  * dmh_next_pass before every Match_Filter (one sweep of the READ keys for both orientations of a REF block);
  * the post-processing of a block (the LAsort / LAcat / LAmerge commands of damapper.c:894-910, unchanged) runs on a
    thread of the driver as soon as dmh_reports_done() says the block's files are complete.

As resident_driver.py, nothing of the reference is kept here: the recipe is line numbers of the reference file plus the
lines (ours) that replace them.  The result goes to oracle/_ref/src/ (git-ignored) and is compiled by oracle/Makefile
into oracle/_ref/damapper_hip_fast.  Blocks with mask tracks (-m) take the reference's own read_DB.

usage: fast_driver.py <reference damapper.c> <output .c>
"""
import hashlib
import sys

# sha256 of the damapper.c the line numbers refer to
EXPECT = "0056b661a221f2ea303ea913a6b7ee8434504c431a767ecd4ec50a9e50289e91"

EXTERN = r'''#include <pthread.h>
#include <sys/time.h>
extern void  dmh_retain_ref_index(int on);     //  libdamapper_hip.so (include/damapper_hip.h)
extern void  dmh_async_report(int mode);
extern void  dmh_report_wait(void);
extern int   dmh_reports_done(void);
extern void  dmh_next_pass(void *atable, void *btable);
extern void *dmh_host_alloc(size_t bytes);
extern void  dmh_host_free(void *p);
extern int   dmh_preload_block_2bit(DAZZ_DB *block, const unsigned char *bps, int64 nbytes, const int64 *coff);
extern void  dmh_release_block(DAZZ_DB *block);
extern void  dmh_free_index(void *table);
extern int   dmh_preload_block(DAZZ_DB *block);
extern int   dmh_prepare_block(DAZZ_DB *block, int reads_side);
'''

# helpers in front of main(): the packed block loader and the post-processing thread
HELPERS = r'''//  A READS block for the library without Load_All_Reads (DB.c:1389-1437): opened and trimmed as read_DB does, its
//    slice of the .bps file read into page-locked memory and handed to the device, which unpacks it.

typedef struct
  { DAZZ_DB        db;
    unsigned char *slice;      //  page-locked, grown as needed
    int64          smax;
    int64         *coff;
    int            cmax;
    int            packed;     //  loaded by fast_read_DB (else by read_DB)
    char          *root;
    void          *index;
    int            ilen;
  } Fast_Block;

static double FAST_POST;          //  seconds in the post-processing commands (their own thread)
static double FAST_T[4];          //  seconds in: block load, READS index, filter passes, Reporter (+ Close_DB)   [DMH_FAST_TIMES]

static double fast_now()
{ struct timeval tv;
  gettimeofday(&tv,NULL);
  return (tv.tv_sec + 1.e-6*tv.tv_usec);
}

static int fast_read_DB(Fast_Block *fb, char *name, int kmer)
{ DAZZ_DB *block = &(fb->db);
  int      i, isdam, nreads;
  int64    first, last, o;

  isdam = Open_DB(name,block);
  if (isdam < 0)
    Clean_Exit(1);
  Trim_DB(block);
  nreads = block->nreads;
  if (block->cutoff < kmer)
    { for (i = 0; i < nreads; i++)
        if (block->reads[i].rlen < kmer)
          { fprintf(stderr,"%s: Block %s contains reads < %dbp long !  Run DBsplit -x%d\n",
                           Prog_Name,name,kmer,kmer);
            Clean_Exit(1);
          }
    }
  if (nreads > fb->cmax)
    { fb->cmax = 1.2*nreads + 1000;
      fb->coff = (int64 *) Realloc(fb->coff,sizeof(int64)*fb->cmax,"Allocating read offsets");
      if (fb->coff == NULL)
        Clean_Exit(1);
    }
  first = (nreads > 0) ? block->reads[0].boff : 0;
  last  = first;
  o = 0;
  for (i = 0; i < nreads; i++)
    { int64 b = block->reads[i].boff;
      fb->coff[i] = b - first;
      last = b + (block->reads[i].rlen+3)/4;
      block->reads[i].boff = o;
      o += block->reads[i].rlen + 1;
    }
  block->reads[nreads].boff = o;
  if (last-first > fb->smax)
    { if (fb->slice != NULL)
        dmh_host_free(fb->slice);
      fb->smax  = 1.2*(last-first) + 4096;
      fb->slice = (unsigned char *) dmh_host_alloc(fb->smax);
      if (fb->slice == NULL)
        { fprintf(stderr,"%s: Out of page-locked memory (%lld bytes)\n",Prog_Name,fb->smax);
          Clean_Exit(1);
        }
    }
  { FILE *bases = (FILE *) block->bases;

    if (last > first)
      { fseeko(bases,first,SEEK_SET);
        if (fread(fb->slice,last-first,1,bases) != 1)
          { fprintf(stderr,"%s: Read of .bps file failed (%s)\n",Prog_Name,name);
            Clean_Exit(1);
          }
      }
    fclose(bases);
  }
  block->bases  = NULL;
  block->loaded = 0;           //  nothing for Close_DB to free: the bases live on the device only
  dmh_preload_block_2bit(block,fb->slice,last-first,fb->coff);
  fb->packed = 1;
  return (isdam);
}

//  The commands damapper.c:894-910 runs for a READS block, on a thread of their own: block i's are run when the
//    library says its files are complete (dmh_reports_done() > i)

#define POST_MAX 4

typedef struct
  { char *cmd[POST_MAX];
    int   ncmd;
  } Post_Block;

static Post_Block     *POSTS;
static int             POST_TOP;      //  blocks whose commands are queued
static int             POST_QUIT;
static int             POST_VERBOSE;
static pthread_mutex_t POST_MU = PTHREAD_MUTEX_INITIALIZER;

static void *post_thread(void *arg)
{ int done = 0;

  (void) arg;
  while (1)
    { int top, quit, ready, c;

      pthread_mutex_lock(&POST_MU);
      top  = POST_TOP;
      quit = POST_QUIT;
      pthread_mutex_unlock(&POST_MU);
      ready = dmh_reports_done();
      if (done < top && done < ready)
        { for (c = 0; c < POSTS[done].ncmd; c++)
            { if (POST_VERBOSE)
                printf("%s\n",POSTS[done].cmd[c]);
              double ts = fast_now();
              int    rc = system(POSTS[done].cmd[c]);
              FAST_POST += fast_now()-ts;
              if (rc != 0)
                { fprintf(stderr,"\n%s: Command Failed:\n%*s      %s\n",
                                 Prog_Name,(int) strlen(Prog_Name),"",POSTS[done].cmd[c]);
                  Clean_Exit(1);
                }
              free(POSTS[done].cmd[c]);
            }
          done += 1;
          continue;
        }
      if (quit && done >= top)
        break;
      usleep(1000);
    }
  return (NULL);
}

'''

REPORT = r'''//  Reporter for a block whose passes are done, then what main() does with the block (damapper.c:875-913) -- but for
//    the post-processing commands, which are only queued

static void fast_report(Fast_Block *pend, char *aroot, DAZZ_DB *ablock, Align_Spec *settings, int mflag, int map_order)
{ Post_Block *post;
  char       *command;

  if (VERBOSE)
    printf("\nFinding best matches for block %s\n",pend->root);
  Reporter(pend->root,&(pend->db),aroot,ablock,settings,mflag);
  dmh_free_index(pend->index);       //  (with the reference indices retained, Reporter leaves the READS index to the driver)
  if (pend->packed)
    dmh_release_block(&(pend->db));
  Close_DB(&(pend->db));

  pthread_mutex_lock(&POST_MU);
  post = POSTS + POST_TOP;
  post->ncmd = 0;
  command = CommandBuffer(aroot,pend->root,SORT_PATH);
  if ((mflag & FLAG_DOA) != 0)
    { sprintf(command,"LAsort %s %s %s/%s.%s.M%c.las",VERBOSE?"-v":"",
                      map_order?"-a":"",SORT_PATH,pend->root,aroot,BLOCK_SYMBOL);
      post->cmd[post->ncmd++] = Strdup(command,"Allocating command");
      sprintf(command,"LAcat %s %s/%s.%s.M%c.S >%s.%s.las",VERBOSE?"-v":"",
                      SORT_PATH,pend->root,aroot,BLOCK_SYMBOL,pend->root,aroot);
      post->cmd[post->ncmd++] = Strdup(command,"Allocating command");
    }
  if ((mflag & FLAG_DOB) != 0)
    { sprintf(command,"LAsort %s %s %s/%s.%s.R%c.las",VERBOSE?"-v":"",
                      map_order?"-a":"",SORT_PATH,aroot,pend->root,BLOCK_SYMBOL);
      post->cmd[post->ncmd++] = Strdup(command,"Allocating command");
      sprintf(command,"LAmerge %s %s %s.%s %s/%s.%s.R%c.S.las",VERBOSE?"-v":"",
                      map_order?"-a":"",aroot,pend->root,SORT_PATH,aroot,pend->root,BLOCK_SYMBOL);
      post->cmd[post->ncmd++] = Strdup(command,"Allocating command");
    }
  POST_TOP += 1;
  pthread_mutex_unlock(&POST_MU);
  free(pend->root);
}

'''

DECLS = r'''  DAZZ_DB    *rblock = NULL;      //  resident reference: block k-1 and its two indices, built once
  void      **ridx = NULL;
  int        *rlens  = NULL;
  Fast_Block  fblk[2];            //  two READS blocks open: the one being filtered, the one waiting for Reporter
  pthread_t   pthread;
'''

LOOP = r'''    memset(fblk,0,sizeof(fblk));
    POSTS = (Post_Block *) Malloc(sizeof(Post_Block)*(argc+1),"Allocating post-processing queue");
    if (POSTS == NULL)
      Clean_Exit(1);
    POST_TOP = POST_QUIT = 0;
    POST_VERBOSE = VERBOSE;
    pthread_create(&pthread,NULL,post_thread,NULL);
    dmh_async_report(2);
    dmh_retain_ref_index(1);           //  Match_Filter leaves the reference indices alone

    { Fast_Block *pend = NULL;         //  filtered, Reporter still to come
      int         npass, bdam;

      for (i = 2; i < argc; i++)
        { Fast_Block *cur = fblk + (i & 1);

          double t0 = fast_now(), t1;

          bfile = argv[i];
          cur->packed = 0;
          if (MTOP == 0)
            bdam = fast_read_DB(cur,bfile,KMER_LEN);
          else
            bdam = read_DB(&(cur->db),bfile,MASK,MSTAT,MTOP,KMER_LEN);
          cur->root = bdam ? Root(bfile,".dam") : Root(bfile,".db");
          t1 = fast_now(); FAST_T[0] += t1-t0; t0 = t1;
          if (VERBOSE)
            printf("\nBuilding index for %s\n",cur->root);
          cur->index = Sort_Kmers(&(cur->db),&(cur->ilen));
          t1 = fast_now(); FAST_T[1] += t1-t0;

          if (rblock == NULL)
            { rblock = (DAZZ_DB *) Malloc(sizeof(DAZZ_DB)*nblocks,"Allocating resident reference");
              ridx   = (void **) Malloc(sizeof(void *)*2*nblocks,"Allocating resident reference");
              rlens  = (int *) Malloc(sizeof(int)*2*nblocks,"Allocating resident reference");
              if (rblock == NULL || ridx == NULL || rlens == NULL)
                Clean_Exit(1);
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
              if (isdam)
                afile = Strdup(Catenate(apwd,"/",aroot,".dam"),"Allocating name");
              else
                afile = Strdup(Catenate(apwd,"/",aroot,".db"),"Allocating name");
              read_DB(ablock,afile,MASK,MSTAT,0,KMER_LEN);
              free(afile);
              dmh_preload_block(ablock);      //  Reporter's copy of the whole reference: on the device once, with its
              dmh_prepare_block(ablock,0);    //    2-bit packing, instead of checked (3 GB stamped) at every Reporter
            }

          npass = 0;
          t0 = fast_now();
          for (k = 1; k <= nblocks; k++)
            for (j = 0; j < 2; j++)
              { if (VERBOSE)
                  printf("\nComparing %s to %s%s.%d%s\n",cur->root,j?"c(":"",aroot,k,j?")":"");
                if (j == 0)
                  dmh_next_pass(cur->index,ridx[2*k-1]);            //  the pass that follows: this block's complement
                else if (k < nblocks)
                  dmh_next_pass(cur->index,ridx[2*k]);
                else
                  dmh_next_pass(NULL,NULL);
                Match_Filter(&(cur->db),rblock+(k-1),cur->index,cur->ilen,ridx[2*k-2+j],rlens[2*k-2+j],j,(k==1 && j==0));
                npass += 1;
                if (pend != NULL && (npass == 10 || npass == 2*nblocks))       //  the block before, ten passes late
                  { t1 = fast_now(); FAST_T[2] += t1-t0; t0 = t1;
                    fast_report(pend,aroot,ablock,settings,mflag,MAP_ORDER);
                    pend = NULL;
                    t1 = fast_now(); FAST_T[3] += t1-t0; t0 = t1;
                  }
              }
          FAST_T[2] += fast_now()-t0;
          if (getenv("DMH_FAST_TIMES") != NULL)
            fprintf(stderr,"dmh: fast driver: after %s: load %.3f index %.3f passes %.3f report %.3f (cumulative s) wall %.3f\n",
                           cur->root,FAST_T[0],FAST_T[1],FAST_T[2],FAST_T[3],fast_now());
          pend = cur;
        }
      if (pend != NULL)
        fast_report(pend,aroot,ablock,settings,mflag,MAP_ORDER);
    }

    dmh_report_wait();
    pthread_mutex_lock(&POST_MU);
    POST_QUIT = 1;
    pthread_mutex_unlock(&POST_MU);
    pthread_join(pthread,NULL);
    if (rblock != NULL)
      Close_DB(ablock);
    if (getenv("DMH_FAST_TIMES") != NULL)
      fprintf(stderr,"dmh: fast driver: block load %.3f s, READS index %.3f s, filter passes %.3f s, Reporter + Close_DB %.3f s; "
                     "post-processing commands %.3f s (own thread); wall at end %.3f\n",
                     FAST_T[0],FAST_T[1],FAST_T[2],FAST_T[3],FAST_POST,fast_now());
'''

# (first line, last line, replacement) -- 1-based, inclusive; first > last: insert before `first`
RECIPE = [
    (51, 50, EXTERN),          # behind #include "map.h"
    (556, 555, HELPERS + REPORT),       # in front of main()
    (566, 565, DECLS),         # behind the declaration of `command`
    (825, 914, LOOP),          # the loop over the READS blocks
]


def main():
    src, out = sys.argv[1], sys.argv[2]
    data = open(src, "rb").read()
    if hashlib.sha256(data).hexdigest() != EXPECT:
        sys.exit("fast_driver.py: %s is not the damapper.c this recipe was written for" % src)
    lines = data.decode().splitlines(keepends=True)
    for first, last, text in sorted(RECIPE, reverse=True):
        lines[first - 1:last] = [text]
    open(out, "w").write("".join(lines))


if __name__ == "__main__":
    main()
