#define _GNU_SOURCE
#include "../../freesasa_amd/csrc/ingest.c"
#include <glob.h>
int main(int c,char**v){ int nt=atoi(v[1]); glob_t g; glob("tests/golden/pdb/[1-9]*.pdb",0,0,&g);
 int n=g.gl_pathc*(c>2?atoi(v[2]):32); const char **paths=malloc(sizeof(char*)*n); for(int i=0;i<n;++i) paths[i]=g.gl_pathv[i%g.gl_pathc];
 for(int rep=0;rep<3;++rep){ freesasa_ingest_batch b; struct timespec a,bb; clock_gettime(CLOCK_MONOTONIC,&a);
 int rc=freesasa_ingest_pdb_files(paths,n,0,nt,&b); clock_gettime(CLOCK_MONOTONIC,&bb);
 double dt=(bb.tv_sec-a.tv_sec)+1e-9*(bb.tv_nsec-a.tv_nsec); printf("%d threads: rc %d %.3f s -> %.1f M atoms/s\n",nt,rc,dt,b.n_atoms/dt/1e6); freesasa_ingest_free(&b);} }
