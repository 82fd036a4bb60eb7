// FETCH_SIZE / WRITE_SIZE calibration (dev aid): copy kernels of KNOWN size, one with 8-byte and one with 16-byte
// accesses per lane, to be run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes):
// the ratio of the counter (KB) to the bytes really moved is what the tile kernel's roofline.traffic must be
// read with (MI355X_MICROARCH.md warns that FETCH_SIZE under-counts 16-byte-per-lane streams).
//   hipcc --offload-arch=gfx950 -O3 tools/dev/fetch_calib.hip -o tools/dev/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
struct __attribute__((aligned(16))) V4 { double a, b; };
struct __attribute__((aligned(16))) V8 { double a, b, c, d; };
__global__ void copy_b64(const double *in, double *out, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) out[i] = in[i]; }
__global__ void copy_b128(const V4 *in, V4 *out, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) out[i] = in[i]; }
/* 32-byte records, two 16-byte loads per lane: the access pattern of the tile kernels' a.sq[u] */
__global__ void copy_rec32(const V8 *in, V8 *out, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) out[i] = in[i]; }
/* the same records gathered through a permutation of 64-record blocks (every wave reads one contiguous 2 KB run somewhere else) */
__global__ void gather_rec32(const V8 *in, V8 *out, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { size_t blk = i >> 6, nb = n >> 6; size_t src = ((blk * 2654435761ull) % nb) << 6 | (i & 63); out[i] = in[src]; } }
int main()
{
    const size_t bytes = 256u << 20;
    void *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    copy_b64<<<(unsigned)(bytes / 8 / 256), 256>>>((const double *)a, (double *)b, bytes / 8);
    copy_b128<<<(unsigned)(bytes / 16 / 256), 256>>>((const V4 *)a, (V4 *)b, bytes / 16);
    copy_rec32<<<(unsigned)(bytes / 32 / 256), 256>>>((const V8 *)a, (V8 *)b, bytes / 32);
    gather_rec32<<<(unsigned)(bytes / 32 / 256), 256>>>((const V8 *)a, (V8 *)b, bytes / 32);
    hipDeviceSynchronize();
    printf("each kernel read %zu bytes and wrote %zu bytes (%.0f KB)\n", bytes, bytes, bytes / 1024.0);
    return 0;
}
