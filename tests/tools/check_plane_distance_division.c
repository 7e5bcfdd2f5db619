// The walk's plane distance (b - o) / d is computed, for rays in the "plain" ranges, as
//     y = RN(1/d) (once per ray);  q = RN(a * y);  r = fma(-q, d, a);  t = fma(r, y, q)
// (rt_kernels.hip: walk_descend).  This program checks on the CPU, with the C library's fmaf, that t is the correctly
// rounded a / d for random a and d with exponents in 2^-60..2^60 (wider than the kernel's ranges) and random or
// extreme mantissas:  gcc -O2 -fopenmp -ffp-contract=off -mfma check_plane_distance_division.c -lm && ./a.out 60000000000
// prints "bad=0" (6e10 trials, 90 s on 8 cores).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <omp.h>
static inline float u2f(uint32_t u){float f; memcpy(&f,&u,4); return f;}
static inline uint32_t f2u(float f){uint32_t u; memcpy(&u,&f,4); return u;}
static inline uint64_t rng(uint64_t *s){ *s ^= *s<<13; *s ^= *s>>7; *s ^= *s<<17; return *s; }
int main(int argc,char**argv){
  long N = argc>1? atol(argv[1]) : 200000000L;
  long bad=0, skipped=0;
  #pragma omp parallel reduction(+:bad,skipped)
  {
    uint64_t s = 0x9E3779B97F4A7C15ULL * (omp_get_thread_num()+1);
    #pragma omp for
    for(long i=0;i<N;i++){
      uint64_t r1=rng(&s), r2=rng(&s);
      // exponents limited to a window, mantissas random, plus special mantissa patterns now and then
      uint32_t ma = (uint32_t)r1 & 0x7fffff, md = (uint32_t)r2 & 0x7fffff;
      int pat = (r1>>40)&15;
      if(pat==0) md = 0x7fffff; else if(pat==1) md = 0; else if(pat==2) md = 0x7ffffe; else if (pat==3) ma = 0x7fffff; else if (pat==4) ma=0;
      int ea = 127 + (int)((r1>>24)%121) - 60, ed = 127 + (int)((r2>>24)%121) - 60;   // 2^-60 .. 2^60
      float a = u2f(((uint32_t)(r1>>63)<<31) | ((uint32_t)ea<<23) | ma);
      float d = u2f(((uint32_t)(r2>>63)<<31) | ((uint32_t)ed<<23) | md);
      float y = 1.0f/d;
      float q = a*y;
      float r = fmaf(-q,d,a);
      float q2 = fmaf(r,y,q);
      float want = a/d;
      if(f2u(q2)!=f2u(want)){ bad++; if(bad<5) printf("a=%a d=%a got %a want %a\n",a,d,q2,want);}    
    }
  }
  printf("N=%ld bad=%ld\n",N,bad);
  return 0;
}
