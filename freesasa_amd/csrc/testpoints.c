/*
 * testpoints.c — Shrake-Rupley unit test points, generated ON THE HOST in C.
 *
 * S&R parity is bit-exact only if the test points are the reference's, bit for bit
 * (src/sasa_sr.c:56-90: golden-section spiral, host libm cos/sin/sqrt, longitude and z
 * ACCUMULATED).  This file is therefore compiled by the same compiler family and flags as
 * the reference library (gcc -O2), not by hipcc: clang fuses the cos/sin pair into one
 * sincos() call whose results differ from glibc's separate cos()/sin() in the last bit for a
 * few points per thousand (observed: 4 of 5000), which is enough to flip a count.
 */
#include <math.h>

#include "../../include/freesasa_gpu.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

void freesasa_gpu_test_points(int n_points, double *tp)
{
    const double dlong = M_PI * (3 - sqrt(5)), dz = 2.0 / n_points;
    double longitude = 0, z = 1 - dz / 2;
    int k;
    for (k = 0; k < n_points; ++k) {
        const double r = sqrt(1 - z * z);
        tp[3 * k] = cos(longitude) * r;
        tp[3 * k + 1] = sin(longitude) * r;
        tp[3 * k + 2] = z;
        z -= dz;
        longitude += dlong;
    }
}
