// round 6: what bounds the frame fill (csrc_host/_pack.c fill_range) on a host CPU -- variants of the per-cell loop on fake objects
// (a reference count at offset 0): gcc -O3 -msse4.1 tools/ubench/frame_fill.c -lm && ./a.out
// prototype: frame fill from row-major (AoS) vs column-major (SoA) idx / val, without Python objects' semantics:
// objects are fake structs with a refcount at offset 0
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <time.h>
#include <string.h>
typedef struct { long refcnt; long pad[8]; } Obj;
static double now(){struct timespec t; clock_gettime(CLOCK_MONOTONIC,&t); return t.tv_sec+t.tv_nsec*1e-9;}
static void fill_aos(Obj **items, int n_names, const int32_t *idx, const float *val, int stride, Obj **obj, double *sim, int lo, int hi, Obj *none){
  for (int i=lo;i<hi;++i){
    if (i+12<hi){ int jn=idx[(i+12)*stride]; if (jn>=0&&jn<n_names) __builtin_prefetch(items[jn],1,1);}
    int j=idx[i*stride]; double s=rint((double)val[i*stride]*1000.0)/1000.0; Obj*o=none;
    if (s<0.001||j<0||j>=n_names) s=0.0; else o=items[j];
    sim[i]=s; o->refcnt++; obj[i]=o; }
}
__attribute__((target("avx2")))
static void fill_soa(Obj **items, int n_names, const int32_t *idx, const float *val, Obj **obj, double *sim, int lo, int hi, Obj *none){
  for (int i=lo;i<hi;++i){ double s=__builtin_rint((double)val[i]*1000.0)/1000.0; int j=idx[i]; sim[i]=((s<0.001)|(j<0)|(j>=n_names))?0.0:s; }
  for (int i=lo;i<hi;++i){
    if (i+12<hi){ int jn=idx[i+12]; if (jn>=0&&jn<n_names) __builtin_prefetch(items[jn],1,1);}
    Obj*o = sim[i]!=0.0? items[idx[i]]:none; o->refcnt++; obj[i]=o; }
}
static void fill_objonly(Obj **items, int n_names, const int32_t *idx, const float *val, int stride, Obj **obj, double *sim, int lo, int hi, Obj *none){
  for (int i=lo;i<hi;++i){
    if (i+12<hi){ int jn=idx[(i+12)*stride]; if (jn>=0&&jn<n_names) __builtin_prefetch(items[jn],1,1);}
    int j=idx[i*stride]; Obj*o=none;
    if ((double)val[i*stride]*1000.0<=0.5||j<0||j>=n_names) {} else o=items[j];
    o->refcnt++; obj[i]=o; }
}
static void fill_simonly(Obj **items, int n_names, const int32_t *idx, const float *val, int stride, Obj **obj, double *sim, int lo, int hi, Obj *none){
  for (int i=lo;i<hi;++i){
    int j=idx[i*stride]; double s=rint((double)val[i*stride]*1000.0)/1000.0;
    if (s<0.001||j<0||j>=n_names) s=0.0;
    sim[i]=s; }
}
static double TAB[1002];
static void fill_tab(Obj **items, int n_names, const int32_t *idx, const float *val, int stride, Obj **obj, double *sim, int lo, int hi, Obj *none){
  for (int i=lo;i<hi;++i){
    if (i+12<hi){ int jn=idx[(i+12)*stride]; if (jn>=0&&jn<n_names) __builtin_prefetch(items[jn],1,1);}
    int j=idx[i*stride]; double t=rint((double)val[i*stride]*1000.0); Obj*o=none; double s;
    if (t<1.0||j<0||j>=n_names) s=0.0; else { o=items[j]; s = t<=1001.0 ? TAB[(int)t] : t/1000.0; }
    sim[i]=s; o->refcnt++; obj[i]=o; }
}
static void fill_aos4(Obj **items, int n_names, const int32_t *idx, const float *val, int stride, Obj **obj, double *sim, int lo, int hi, Obj *none){
  int i=lo;
  for (;i+4<=hi;i+=4){
    if (i+16<hi){ for(int u=0;u<4;++u){int jn=idx[(i+12+u)*stride]; if (jn>=0&&jn<n_names) __builtin_prefetch(items[jn],1,1);} }
    int j0=idx[i*stride], j1=idx[(i+1)*stride], j2=idx[(i+2)*stride], j3=idx[(i+3)*stride];
    double s0=rint((double)val[i*stride]*1000.0)/1000.0, s1=rint((double)val[(i+1)*stride]*1000.0)/1000.0, s2=rint((double)val[(i+2)*stride]*1000.0)/1000.0, s3=rint((double)val[(i+3)*stride]*1000.0)/1000.0;
    int b0=(s0<0.001)|(j0<0)|(j0>=n_names), b1=(s1<0.001)|(j1<0)|(j1>=n_names), b2=(s2<0.001)|(j2<0)|(j2>=n_names), b3=(s3<0.001)|(j3<0)|(j3>=n_names);
    Obj *o0=b0?none:items[j0], *o1=b1?none:items[j1], *o2=b2?none:items[j2], *o3=b3?none:items[j3];
    sim[i]=b0?0.0:s0; sim[i+1]=b1?0.0:s1; sim[i+2]=b2?0.0:s2; sim[i+3]=b3?0.0:s3;
    o0->refcnt++; o1->refcnt++; o2->refcnt++; o3->refcnt++;
    obj[i]=o0; obj[i+1]=o1; obj[i+2]=o2; obj[i+3]=o3; }
  for (;i<hi;++i){ int j=idx[i*stride]; double s=rint((double)val[i*stride]*1000.0)/1000.0; Obj*o=none; if (s<0.001||j<0||j>=n_names) s=0.0; else o=items[j]; sim[i]=s; o->refcnt++; obj[i]=o; }
}
int main(){
  int n=100000, top=5; Obj *pool=malloc(sizeof(Obj)*(n+1)); Obj **items=malloc(sizeof(Obj*)*n);
  for(int i=0;i<n;++i){ pool[i].refcnt=1; items[i]=&pool[i]; } Obj *none=&pool[n];
  int32_t *idx=malloc(4*n*top), *idxT=malloc(4*n*top); float *val=malloc(4*n*top), *valT=malloc(4*n*top);
  srand(1); for(int i=0;i<n;++i) for(int r=0;r<top;++r){ int j=i+(rand()%600)-300; if(j<0)j=0; if(j>=n)j=n-1; idx[i*top+r]=j; idxT[r*n+i]=j; float v=(float)rand()/RAND_MAX; val[i*top+r]=v; valT[r*n+i]=v; }
  Obj **obj[5]; double *sim[5]; for(int r=0;r<top;++r){ obj[r]=malloc(8*n); sim[r]=malloc(8*n);} 
  for (int k=0;k<1002;++k) TAB[k]=(double)k/1000.0;
  for (int rep=0; rep<6; ++rep){
    double t0=now(); for(int r0=0;r0<n;r0+=8192){int r1=r0+8192>n?n:r0+8192; for(int r=0;r<top;++r) fill_aos(items,n,idx+r,val+r,top,obj[r],sim[r],r0,r1,none);} double t1=now();
    for(int r0=0;r0<n;r0+=8192){int r1=r0+8192>n?n:r0+8192; for(int r=0;r<top;++r) fill_soa(items,n,idxT+r*n,valT+r*n,obj[r],sim[r],r0,r1,none);} double t2=now();
    double t3=now(); for(int r0=0;r0<n;r0+=8192){int r1=r0+8192>n?n:r0+8192; for(int r=0;r<top;++r) fill_aos4(items,n,idx+r,val+r,top,obj[r],sim[r],r0,r1,none);} double t4=now();
    double t5=now(); for(int r0=0;r0<n;r0+=8192){int r1=r0+8192>n?n:r0+8192; for(int r=0;r<top;++r) fill_tab(items,n,idx+r,val+r,top,obj[r],sim[r],r0,r1,none);} double t6=now();
    double t7=now(); for(int r0=0;r0<n;r0+=8192){int r1=r0+8192>n?n:r0+8192; for(int r=0;r<top;++r) fill_objonly(items,n,idx+r,val+r,top,obj[r],sim[r],r0,r1,none);} double t8=now();
    for(int r0=0;r0<n;r0+=8192){int r1=r0+8192>n?n:r0+8192; for(int r=0;r<top;++r) fill_simonly(items,n,idx+r,val+r,top,obj[r],sim[r],r0,r1,none);} double t9=now();
    printf("full %.3f ms  soa %.3f  unroll4 %.3f  table %.3f  objects only %.3f  sims only %.3f\n",(t1-t0)*1e3,(t2-t1)*1e3,(t4-t3)*1e3,(t6-t5)*1e3,(t8-t7)*1e3,(t9-t8)*1e3);
  }
  return 0; }
