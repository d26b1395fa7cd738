#include "../include/dspi_detmath.h"
float t_log10f(float x){return dspi_det_log10f(x);}
float t_powf(float a,float b){return dspi_det_powf(a,b);}
void t_log10f_v(const float*x,float*y,int n){for(int i=0;i<n;i++)y[i]=dspi_det_log10f(x[i]);}
void t_powf_v(const float*a,const float*b,float*y,int n){for(int i=0;i<n;i++)y[i]=dspi_det_powf(a[i],b[i]);}
