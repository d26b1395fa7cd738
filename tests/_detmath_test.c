
#include <quadmath.h>
#include <stdint.h>
#include <string.h>
#include "../include/dspi_detmath.h"
float t_log10f(float x){return dspi_det_log10f(x);}
float t_powf(float a,float b){return dspi_det_powf(a,b);}
void t_log10f_v(const float*x,float*y,long n){for(long i=0;i<n;i++)y[i]=dspi_det_log10f(x[i]);}
void t_powf_v(const float*a,const float*b,float*y,long n){for(long i=0;i<n;i++)y[i]=dspi_det_powf(a[i],b[i]);}
void q_log10f_v(const float*x,float*y,long n){for(long i=0;i<n;i++)y[i]=(float)log10q((__float128)x[i]);}
void q_powf_v(const float*a,const float*b,float*y,long n){for(long i=0;i<n;i++)y[i]=(float)powq((__float128)a[i],(__float128)b[i]);}
static uint32_t rs;
static uint32_t rnd(void){ rs ^= rs<<13; rs ^= rs>>17; rs ^= rs<<5; return rs; }
static float fbits(uint32_t u){ float f; memcpy(&f,&u,4); return f; }
/* out[0] = mismatches vs binary128, out[1] = calls that took step 2, out[2] = max (step-1 error / its bound), out[3] = arguments */
void sweep_log10(long n, uint32_t seed, double *out){
  rs = seed; long bad=0, slow=0; double worst=0;
  for(long i=0;i<n;i++){
    uint32_t u=rnd(); float x;
    if(i%3==0) x=fbits((u%0x7f000000u)+0x00800000u);                   /* any positive normal float */
    else if(i%3==1) x=fbits(0x3f000000u+(u&0x00ffffffu));               /* [0.5, 2): around the zero of the logarithm */
    else x=fbits(0x0da24260u+(u%(0x41200000u-0x0da24260u)));            /* 1e-30 .. 10: the leveller's rms_sq + 1e-30f */
    __float128 q=log10q((__float128)x);
    double r=dspi_dm_log((double)x)*0.43429448190325182;
    if(x!=1.0f){ double e=(double)fabsq(((__float128)r-q)/q)/1.4210854715202004e-14; if(e>worst)worst=e; }
    float f,g=dspi_det_log10f(x),ref=(float)q;
    if(!dspi_dm_unambiguous(r,1.4210854715202004e-14,&f))slow++;
    if(memcmp(&g,&ref,4))bad++;
  }
  out[0]=bad; out[1]=slow; out[2]=worst; out[3]=n;
}
void sweep_pow(long n, uint32_t seed, double *out){
  rs = seed; long bad=0, slow=0; double worst=0;
  for(long i=0;i<n;i++){
    uint32_t u=rnd(), v=rnd(); float a,b; int shape=i%4;
    if(shape==0){ a=fbits(0x3f666666u+(u%(0x3f800000u-0x3f666666u))); b=(float)(1+v%192); }          /* alpha in [0.9, 1) ^ count (leveller.c:200) */
    else if(shape==1){ a=10.0f; b=(float)((double)(int32_t)v/2147483648.0*4.0); }                     /* 10 ^ [-4, 4] (leveller.c:206: dB / 20) */
    else if(shape==2){ a=10.0f; b=fbits(0x3c000000u+(u%(0x40800000u-0x3c000000u))); if(v&1)b=-b; }     /* log-spaced exponents */
    else { a=fbits(0x3f7f0000u+(u&0xffffu)); b=(float)(1+v%192); }                                      /* alpha within 2^-8 of 1 */
    double y=(double)b*dspi_dm_log((double)a), r=dspi_dm_exp(y), ay=y<0?-y:y, bound=1.4210854715202004e-14+ay*7.105427357601002e-15;
    __float128 q=powq((__float128)a,(__float128)b);
    double e=(double)fabsq(((__float128)r-q)/q)/bound; if(e>worst)worst=e;
    float f,g=dspi_det_powf(a,b),ref=(float)q;
    if(!dspi_dm_unambiguous(r,bound,&f))slow++;
    if(memcmp(&g,&ref,4))bad++;
  }
  out[0]=bad; out[1]=slow; out[2]=worst; out[3]=n;
}
/* arguments whose step-1 value is ambiguous (they take step 2), found by scanning: up to cap of them into xs / (as, bs) */
long find_slow_log10(uint32_t seed, long tries, float *xs, long cap){
  rs=seed; long k=0; float f;
  for(long i=0;i<tries&&k<cap;i++){ float x=fbits(0x0da24260u+(rnd()%(0x41200000u-0x0da24260u)));
    if(!dspi_dm_unambiguous(dspi_dm_log((double)x)*0.43429448190325182,1.4210854715202004e-14,&f)) xs[k++]=x; }
  return k;
}
long find_slow_pow(uint32_t seed, long tries, float *as, float *bs, long cap){
  rs=seed; long k=0; float f;
  for(long i=0;i<tries&&k<cap;i++){ float a=(i&1)?10.0f:fbits(0x3f666666u+(rnd()%(0x3f800000u-0x3f666666u)));
    float b=(i&1)?(float)((double)(int32_t)rnd()/2147483648.0*4.0):(float)(1+rnd()%192);
    double y=(double)b*dspi_dm_log((double)a), ay=y<0?-y:y;
    if(!dspi_dm_unambiguous(dspi_dm_exp(y),1.4210854715202004e-14+ay*7.105427357601002e-15,&f)){ as[k]=a; bs[k]=b; k++; } }
  return k;
}
