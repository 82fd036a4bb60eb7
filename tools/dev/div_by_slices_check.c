#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
int main(){ unsigned long long bad=0, n=0; uint64_t s=88172645463325252ULL;
 for (int ns=1; ns<=1024; ++ns){ const double y=ns, r=1.0/y;
  for (int k=0;k<200000;++k){ s^=s<<13; s^=s>>7; s^=s<<17; double u=(s>>11)*(1.0/9007199254740992.0); double x = 2.0*(0.5+u*7.5); /* 2*R, R in 0.5..8 */
    if (k%7==0) x = ldexp(1.0+u, (int)(s%40)-20);
    double q=x*r; double rem=fma(-q,y,x); double q2=fma(rem,r,q); if (q2!=x/y) ++bad; ++n; } }
 printf("%llu mismatches of %llu\n", bad, n); return 0; }
