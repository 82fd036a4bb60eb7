// Accuracy of the hardware seeds v_rsq_f64 / v_rcp_f64 on gfx950 (decides how many refinement steps
// sqrt_g / atan2_fast need).  hipcc --offload-arch=gfx950 -O2 seed_accuracy.hip -o seed_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *rsq, double *rcp, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { rsq[i] = __builtin_amdgcn_rsq(x[i]); rcp[i] = __builtin_amdgcn_rcp(x[i]); }
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n), a(n), b(n);
    unsigned long long s = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;
        x[i] = i & 1 ? 1e-9 + u * 4.0 : exp(40.0 * (u - 0.5));
    }
    double *dx, *da, *db;
    hipMalloc(&dx, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, da, db, n);
    hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
    long double ersq = 0, ercp = 0;
    for (int i = 0; i < n; ++i) {
        const long double tr = 1.0L / sqrtl((long double)x[i]), tc = 1.0L / (long double)x[i];
        ersq = fmaxl(ersq, fabsl((a[i] - tr) / tr));
        ercp = fmaxl(ercp, fabsl((b[i] - tc) / tc));
    }
    printf("v_rsq_f64 max rel err %.3Le (2^%.1Lf)   v_rcp_f64 max rel err %.3Le (2^%.1Lf)\n", ersq, log2l(ersq), ercp, log2l(ercp));
    return 0;
}
