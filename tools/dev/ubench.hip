// Issue cost of the VALU / LDS instructions the tile kernels are made of, on gfx950 (dev aid).
//   hipcc --offload-arch=gfx950 -O3 tools/dev/ubench.hip -o tools/dev/ubench && tools/dev/ubench
// Every kernel runs ITER x 16 independent copies of one instruction per wave, 4 waves per SIMD on every SIMD of
// the chip; cycles per wave-instruction = elapsed x clock x SIMDs / (waves x instructions), printed relative to
// v_fma_f32 = 2 cycles (SIMD-32, MI355X_MICROARCH.md) so that the figure does not depend on the clock the chip
// settles at.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ITER 4096
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL_F64(name, ASM)                                                                      \
    __global__ __launch_bounds__(256) void k_##name(double *out, double s)                         \
    {                                                                                              \
        double a[16], b = s + threadIdx.x, c = s * 0.5;                                            \
        for (int i = 0; i < 16; ++i) a[i] = s + i + threadIdx.x * 1e-3;                            \
        for (int it = 0; it < ITER; ++it) {                                                        \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "s20", "s21"); \
        }                                                                                          \
        double r = 0;                                                                              \
        for (int i = 0; i < 16; ++i) r += a[i];                                                    \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                   \
    }
#define KERNEL_B32(name, ASM)                                                                      \
    __global__ __launch_bounds__(256) void k_##name(double *out, double s)                         \
    {                                                                                              \
        unsigned a[16], b = (unsigned)s + threadIdx.x, c = (unsigned)(s * 3);                      \
        for (int i = 0; i < 16; ++i) a[i] = (unsigned)s + i + threadIdx.x;                         \
        for (int it = 0; it < ITER; ++it) {                                                        \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "s20", "s21"); \
        }                                                                                          \
        unsigned r = 0;                                                                            \
        for (int i = 0; i < 16; ++i) r += a[i];                                                    \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                   \
    }
#define KERNEL_F32(name, ASM)                                                                      \
    __global__ __launch_bounds__(256) void k_##name(double *out, double s)                         \
    {                                                                                              \
        float a[16], b = (float)s + threadIdx.x, c = (float)s * 0.5f;                              \
        for (int i = 0; i < 16; ++i) a[i] = (float)s + i + threadIdx.x * 1e-3f;                    \
        for (int it = 0; it < ITER; ++it) {                                                        \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "s20", "s21"); \
        }                                                                                          \
        float r = 0;                                                                               \
        for (int i = 0; i < 16; ++i) r += a[i];                                                    \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                   \
    }
/* mixed: a 64-bit destination fed by / feeding 32-bit values */
#define KERNEL_MIX(name, ASM)                                                                      \
    __global__ __launch_bounds__(256) void k_##name(double *out, double s)                         \
    {                                                                                              \
        double a[16];                                                                              \
        unsigned u[16];                                                                            \
        for (int i = 0; i < 16; ++i) { a[i] = s + i + threadIdx.x * 1e-3; u[i] = i + threadIdx.x; } \
        for (int it = 0; it < ITER; ++it) {                                                        \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]), "+v"(u[i]) : : "vcc", "s20", "s21"); \
        }                                                                                          \
        double r = 0;                                                                              \
        for (int i = 0; i < 16; ++i) r += a[i] + u[i];                                             \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                   \
    }

KERNEL_F32(fma_f32, "v_fma_f32 %0, %1, %2, %0")
KERNEL_F32(rsq_f32, "v_rsq_f32 %0, %0")
KERNEL_F32(sqrt_f32, "v_sqrt_f32 %0, %0")
KERNEL_F32(rcp_f32, "v_rcp_f32 %0, %0")
KERNEL_F32(pk_fma_f32x, "v_fma_f32 %0, %0, %2, %1")
KERNEL_F64(fma_f64, "v_fma_f64 %0, %1, %2, %0")
KERNEL_F64(mul_f64, "v_mul_f64 %0, %1, %0")
KERNEL_F64(add_f64, "v_add_f64 %0, %1, %0")
KERNEL_F64(min_f64, "v_min_f64 %0, %1, %0")
KERNEL_F64(max_f64, "v_max_f64 %0, %1, %0")
KERNEL_F64(cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")
KERNEL_F64(cmp_f64_sgpr, "v_cmp_lt_f64 s[20:21], %0, %1")
KERNEL_F64(rsq_f64, "v_rsq_f64 %0, %0")
KERNEL_F64(rcp_f64, "v_rcp_f64 %0, %0")
KERNEL_F64(sqrt_f64, "v_sqrt_f64 %0, %0")
KERNEL_F64(mov_b64, "v_mov_b64 %0, %1")
KERNEL_F64(lshl_b64, "v_lshlrev_b64 %0, 1, %0")
KERNEL_F64(fma_f64_abs, "v_fma_f64 %0, |%1|, -0.5, %0")
KERNEL_F64(fma_f64_lit, "v_fma_f64 %0, %0, %1, s[20:21]")
KERNEL_B32(mov_b32, "v_mov_b32 %0, %1")
KERNEL_B32(add_u32, "v_add_u32 %0, %1, %0")
KERNEL_B32(and_b32, "v_and_b32 %0, %1, %0")
__global__ __launch_bounds__(256) void k_cndmask_b32(double *out, double s)
{
    unsigned a[16], b = (unsigned)s + threadIdx.x;
    const unsigned long long m = __builtin_amdgcn_ballot_w64((threadIdx.x & 1) != 0);
    for (int i = 0; i < 16; ++i) a[i] = (unsigned)s + i + threadIdx.x;
    for (int it = 0; it < ITER; ++it) {
        _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(m));
    }
    unsigned r = 0;
    for (int i = 0; i < 16; ++i) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
/* a dependent chain of fp64 fma in one wave per SIMD: issue-to-issue latency */
__global__ __launch_bounds__(64) void k_fma_f64_chain(double *out, double s)
{
    double a = s + threadIdx.x, b = 1.0000001, c = 1e-9;
    for (int it = 0; it < ITER; ++it) {
        _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    }
    out[blockIdx.x * 64 + threadIdx.x] = a;
}
KERNEL_B32(cmp_u32, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL_B32(mul_lo_u32, "v_mul_lo_u32 %0, %1, %0")
KERNEL_B32(mul_u24, "v_mul_u32_u24 %0, %1, %0")
KERNEL_B32(mad_u24, "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL_B32(lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL_B32(ffbl, "v_ffbl_b32 %0, %0")
KERNEL_B32(bcnt, "v_bcnt_u32_b32 %0, %1, %0")
KERNEL_B32(mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL_B32(med3, "v_med3_i32 %0, %0, %1, %2")
KERNEL_B32(dpp_mov, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL_B32(dpp_add, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL_B32(readlane, "v_readlane_b32 s20, %0, 3")
KERNEL_B32(writelane, "v_writelane_b32 %0, s20, 3")
KERNEL_B32(addc, "v_addc_co_u32 %0, vcc, %0, %0, vcc")
KERNEL_MIX(cvt_f32_f64, "v_cvt_f32_f64 %1, %0")
KERNEL_MIX(cvt_f64_f32, "v_cvt_f64_f32 %0, %1")
KERNEL_MIX(cvt_i32_f64, "v_cvt_i32_f64 %1, %0")
KERNEL_MIX(cvt_f64_i32, "v_cvt_f64_i32 %0, %1")
KERNEL_MIX(frexp_exp, "v_frexp_exp_i32_f64 %1, %0")
KERNEL_MIX(ldexp, "v_ldexp_f64 %0, %0, %1")

/* LDS: throughput of independent reads / writes, conflict-free and same-address (broadcast) */
#define KERNEL_LDS(name, ASM, STRIDE)                                                        \
    __global__ __launch_bounds__(256) void k_##name(double *out, double s)                         \
    {                                                                                              \
        __shared__ double sm[4096];                                                                \
        for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = s + i;                               \
        __syncthreads();                                                                           \
        unsigned addr = (unsigned)(size_t)(sm) + (threadIdx.x & 63) * (STRIDE) + (threadIdx.x >> 6) * 4096; \
        double acc = 0;                                                                            \
        for (int it = 0; it < ITER / 4; ++it) {                                                    \
            asm volatile(ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n"  \
                         ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n s_waitcnt lgkmcnt(0)" \
                         : : "v"(addr), "v"(acc), "v"(acc), "v"(addr) : "v40", "v41", "v42", "v43", "memory");                      \
        }                                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = acc + sm[threadIdx.x];                               \
    }
KERNEL_LDS(ds_read_b64, "ds_read_b64 v[40:41], %0", 8)
KERNEL_LDS(ds_read_b64_bcast, "ds_read_b64 v[40:41], %0", 0)
KERNEL_LDS(ds_read_b64_s24, "ds_read_b64 v[40:41], %0", 24)
KERNEL_LDS(ds_read2_b64, "ds_read2_b64 v[40:43], %0 offset1:1", 16)
KERNEL_LDS(ds_read2_b64_s24, "ds_read2_b64 v[40:43], %0 offset1:1", 24)
KERNEL_LDS(ds_read_b128, "ds_read_b128 v[40:43], %0", 16)
KERNEL_LDS(ds_read_b32, "ds_read_b32 v40, %0", 4)
KERNEL_LDS(ds_read_u16, "ds_read_u16 v40, %0", 2)
KERNEL_LDS(ds_write_b64, "ds_write_b64 %0, %1", 8)
KERNEL_LDS(ds_write2_b64, "ds_write2_b64 %0, %1, %2 offset1:1", 16)
KERNEL_LDS(ds_add_u32, "ds_add_u32 %0, %3", 4)
KERNEL_LDS(ds_add_rtn_u32, "ds_add_rtn_u32 v40, %0, %3", 4)
KERNEL_LDS(ds_bpermute, "ds_bpermute_b32 v40, %0, %3", 4)

struct Case { const char *name; void (*fn)(double *, double); int per_iter; };

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, simds = cus * 4;
    const int blocks = cus * 4; /* 4 blocks of 4 waves per CU = 4 waves per SIMD */
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    std::vector<Case> cases = {
#define C(n) {#n, k_##n, 16}
        C(fma_f32), C(rsq_f32), C(sqrt_f32), C(rcp_f32), C(pk_fma_f32x),
        C(fma_f64), C(mul_f64), C(add_f64), C(min_f64), C(max_f64), C(cmp_f64), C(cmp_f64_sgpr), C(rsq_f64), C(rcp_f64), C(sqrt_f64),
        C(mov_b64), C(lshl_b64), C(fma_f64_abs), C(fma_f64_lit),
        C(mov_b32), C(add_u32), C(and_b32), C(cndmask_b32), C(cndmask_b32), C(cmp_u32), C(mul_lo_u32), C(mul_u24), C(mad_u24), C(lshl_add), C(ffbl), C(bcnt), C(mbcnt),
        C(med3), C(dpp_mov), C(dpp_add), C(readlane), C(writelane), C(addc),
        C(cvt_f32_f64), C(cvt_f64_f32), C(cvt_i32_f64), C(cvt_f64_i32), C(frexp_exp), C(ldexp),
#define L(n) {#n, k_##n, 4}
        L(ds_read_b64), L(ds_read_b64_bcast), L(ds_read_b64_s24), L(ds_read2_b64), L(ds_read2_b64_s24), L(ds_read_b128), L(ds_read_b32), L(ds_read_u16),
        L(ds_write_b64), L(ds_write2_b64), L(ds_add_u32), L(ds_add_rtn_u32), L(ds_bpermute),
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double base = 0;
    printf("%d CUs; %d blocks x 256 threads; ITER %d\n", cus, blocks, ITER);
    for (auto &c : cases) {
        c.fn<<<blocks, 256>>>(out, 1.5); /* warm */
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0);
            c.fn<<<blocks, 256>>>(out, 1.5);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        /* wave-instructions per SIMD: 4 waves x ITER x 16 (LDS: ITER/4 x 16) */
        const double n = 4.0 * (c.per_iter == 16 ? ITER * 16.0 : ITER / 4 * 16.0);
        const double ns_per = best * 1e6 / n;
        if (base == 0) base = ns_per; /* v_fma_f32 */
        printf("%-22s %8.3f ms  %7.3f ns/wave-instr/SIMD  = %6.2f cycles (v_fma_f32 = 2)   [%.2f at 2.4 GHz]\n", c.name, best, ns_per,
               2.0 * ns_per / base, ns_per * 2.4);
    }
    { /* one wave per SIMD, dependent chain */
        k_fma_f64_chain<<<cus * 4, 64>>>(out, 1.5);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k_fma_f64_chain<<<cus * 4, 64>>>(out, 1.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ns_per = ms * 1e6 / (ITER * 16.0);
        printf("fma_f64 dependent chain, 1 wave/SIMD: %.3f ns per instruction = %.2f cycles (v_fma_f32 = 2)\n", ns_per, 2.0 * ns_per / base);
    }
    (void)simds;
    return 0;
}
