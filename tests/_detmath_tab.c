#include "../include/dspi_detmath.h"
void tab_log10(const float*x,float*y,long n){for(long i=0;i<n;i++)y[i]=dspi_det_log10f_tab(x[i]);}
void tab_exp10(const float*x,float*y,long n){for(long i=0;i<n;i++)y[i]=dspi_det_exp10f_tab(x[i]);}
void tab_pow(const float*a,const float*b,float*y,long n){for(long i=0;i<n;i++)y[i]=dspi_det_powf_tab(a[i],b[i]);}
double log_of_10(void){return dspi_dm_log(10.0);} double log_of_10_const(void){return DSPI_DM_LOG_OF_10;}
