// round 6: does the frame fill scale over host threads on this box, and which part of it?  Fake objects (a reference count at
// offset 0, 72 bytes apart like short str objects), the calling thread walks them first (the string packer's walk), then T
// threads fill 100k x 5 cells:  A = rows partitioned, stores + a private count per name (no object touched)
//                               B = every thread scans all cells, stores by row chunk, increments the objects it owns (by 4-KB page)
//                               C = rows partitioned, stores + atomic increments
//                               D = rows partitioned, stores only (no counts at all: the floor of the stores)
// helpers unpinned or pinned to the caller's L3 siblings (argv[1] = "pin").   gcc -O3 -msse4.1 frame_fill_mt.c -lpthread -lm
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <time.h>
typedef struct { long refcnt; long pad[8]; } Obj;
static double now(){struct timespec t; clock_gettime(CLOCK_MONOTONIC,&t); return t.tv_sec+t.tv_nsec*1e-9;}
enum { N = 100000, TOP = 5, CH = 409 };
static Obj *pool, **items, *none; static int32_t *idx; static float *val; static Obj **obj[TOP]; static double *sim[TOP];
static int32_t *cnt; static int T, variant, pin; static volatile int go, arrived;
static inline unsigned owner_of(const void *p, unsigned n){ uint32_t h=(uint32_t)((uintptr_t)p>>12)*0x9E3779B1u; return (unsigned)(((uint64_t)(h>>16)*n)>>16); }
static void run(int tid){
  if (variant=='B'){
    Obj *mine[CH*TOP];
    for (int lo=0,c=0; lo<N; lo+=CH,++c){ int hi=lo+CH<N?lo+CH:N; int st=(c%T)==tid; int nm=0;
      for (int i=lo;i<hi;++i) for (int r=0;r<TOP;++r){ int j=idx[i*TOP+r]; double t=rint((double)val[i*TOP+r]*1000.0); int keep=!(t<1.0)&(j>=0)&(j<N);
        Obj*o=items[keep?j:0]; o=keep?o:none; if(st){ sim[r][i]=keep?t/1000.0:0.0; obj[r][i]=o; }
        int m=keep&(owner_of(o,T)==(unsigned)tid); mine[nm]=o; if(m)__builtin_prefetch(o,1,1); nm+=m; }
      for (int k=0;k<nm;++k) mine[k]->refcnt++; }
  } else {
    int32_t *my=cnt+(size_t)tid*N;
    for (int lo=0,c=0; lo<N; lo+=CH,++c){ if((c%T)!=tid) continue; int hi=lo+CH<N?lo+CH:N;
      for (int r=0;r<TOP;++r) for (int i=lo;i<hi;++i){ int j=idx[i*TOP+r]; double t=rint((double)val[i*TOP+r]*1000.0); int keep=!(t<1.0)&(j>=0)&(j<N);
        Obj*o=items[keep?j:0]; o=keep?o:none; sim[r][i]=keep?t/1000.0:0.0; obj[r][i]=o;
        if (variant=='A') my[keep?j:0]+=keep; else if (variant=='C') __atomic_fetch_add(&o->refcnt,1,__ATOMIC_RELAXED); } }
  }
}
static void *worker(void*a){ int tid=(int)(intptr_t)a; while(!go) __builtin_ia32_pause(); run(tid); __atomic_add_fetch(&arrived,1,__ATOMIC_ACQ_REL); return NULL; }
static int l3cpus(int *out,int max){ int me=sched_getcpu(); char p[128],line[512]; snprintf(p,sizeof p,"/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list",me); FILE*f=fopen(p,"r"); if(!f||!fgets(line,sizeof line,f)) return 0; fclose(f);
  int n=0; for(char*q=line;*q&&*q!='\n';){ char*e; long a=strtol(q,&e,10),b=a; if(*e=='-'){q=e+1;b=strtol(q,&e,10);} for(long c=a;c<=b&&n<max;++c) if(c!=me && c<128) out[n++]=(int)c; q=*e==','?e+1:e; break; } return n; }   // (first stretch only: one thread per core)
int main(int argc,char**argv){ pin=argc>1&&!strcmp(argv[1],"pin");
  pool=malloc(sizeof(Obj)*(N+1)); items=malloc(sizeof(Obj*)*N); for(int i=0;i<N;++i){pool[i].refcnt=1;items[i]=&pool[i];} none=&pool[N];
  idx=malloc(4*N*TOP); val=malloc(4*N*TOP); srand(1); for(int i=0;i<N;++i) for(int r=0;r<TOP;++r){ int j=i+(rand()%600)-300; if(j<0)j=0; if(j>=N)j=N-1; idx[i*TOP+r]=j; val[i*TOP+r]=(float)rand()/RAND_MAX; }
  for(int r=0;r<TOP;++r){obj[r]=malloc(8*N);sim[r]=malloc(8*N);memset(obj[r],0,8*N);memset(sim[r],0,8*N);} cnt=calloc((size_t)16*N,4);
  printf("caller on cpu %d, helpers %s\n",sched_getcpu(),pin?"pinned to its L3":"unpinned");
  const char *vs="DABC"; int Ts[]={1,2,4,8};
  for(const char*v=vs;*v;++v) for(int ti=0;ti<4;++ti){ T=Ts[ti]; variant=*v; double best=1e9,med[9];
    for(int rep=0;rep<9;++rep){ for(int i=0;i<N;++i) pool[i].refcnt++;   // the caller's walk over the objects
      go=0; arrived=0; pthread_t th[16]; int cpus[16]; int nl=pin?l3cpus(cpus,16):0;
      for(int t=1;t<T;++t){ pthread_attr_t at; pthread_attr_init(&at); if(t-1<nl){cpu_set_t s;CPU_ZERO(&s);CPU_SET(cpus[t-1],&s);pthread_attr_setaffinity_np(&at,sizeof s,&s);} pthread_create(&th[t],&at,worker,(void*)(intptr_t)t); }
      struct timespec ts={0,200000}; nanosleep(&ts,NULL);   // threads are up and spinning
      double t0=now(); go=1; run(0); __atomic_add_fetch(&arrived,1,__ATOMIC_ACQ_REL); while(arrived<T) __builtin_ia32_pause(); double dt=(now()-t0)*1e3;
      for(int t=1;t<T;++t) pthread_join(th[t],NULL); med[rep]=dt; if(dt<best)best=dt; }
    for(int a=0;a<9;++a)for(int b=a+1;b<9;++b)if(med[b]<med[a]){double x=med[a];med[a]=med[b];med[b]=x;}
    printf("variant %c  T=%d  median %.3f ms  min %.3f\n",*v,T,med[4],best); }
  return 0; }
