/*
 * gpu_engine.hip — __global__ wrappers around the phase functions of sasa_kernels.h, the
 * per-device workspace, the launch sequence of one batch, and the additive C-ABI of
 * include/freesasa_gpu.h.  gfx950 only; no CPU path: without a HIP device every entry
 * point fails with a message.
 */
#include <hip/hip_runtime.h>

#include <math.h>
#include <atomic>
#include <mutex>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/freesasa_gpu.h"
#include "../../include/freesasa_ingest.h"
#include "sasa_kernels.h"
#ifdef SASA_PHASE_TIMING /* dev only (tools/build_variant.sh X -DSASA_PHASE_TIMING): where a wave's time per tile goes */
__device__ unsigned long long g_phase_clock[16];
#define LR2_MARK_BEGIN unsigned long long lr2_last_ = wall_clock64(); if (lane == 0 && !a.work_items && ((p0 / a.TA) & 255) == 0) atomicAdd(&g_phase_clock[15], 1ULL)
#define LR2_MARK(k) do { if (lane == 0 && !a.work_items && ((p0 / a.TA) & 255) == 0) { const unsigned long long now_ = wall_clock64(); \
        atomicAdd(&g_phase_clock[(k)], now_ - lr2_last_); lr2_last_ = now_; } } while (0)
#endif
#include "lr2_kernels.h"

using namespace sasa;

/* ------------------------------------------------------------------ kernels */

__global__ __launch_bounds__(SASA_PIPE_B) void k_bounds(PipeArgs a)
{
    __shared__ double red[7 * SASA_PIPE_B];
    bounds_phase0(a, red, blockIdx.x, threadIdx.x, SASA_PIPE_B);
    __syncthreads();
    bounds_phase1(a, red, blockIdx.x, threadIdx.x, SASA_PIPE_B);
}

__global__ __launch_bounds__(64) void k_grid(PipeArgs a)
{
    grid_struct(a, blockIdx.x * 64 + threadIdx.x);
}

__global__ __launch_bounds__(SASA_PIPE_B) void k_cell_base(PipeArgs a)
{
    __shared__ long long part[SASA_PIPE_B];
    cellbase_phase0(a, part, threadIdx.x, SASA_PIPE_B);
    __syncthreads();
    cellbase_phase1(a, part, threadIdx.x, SASA_PIPE_B);
    __syncthreads();
    cellbase_phase2(a, part, threadIdx.x, SASA_PIPE_B);
}

/* Everything behind K2 is launched without the host having seen K2's result (no readback in the middle of the
 * pipeline: the next batch of a driver loop can be enqueued behind this one).  A batch that turned out to be in
 * error (non-finite input, grid too big) or to need a larger cell table than was allocated does nothing from here
 * on - uniformly, first thing in every kernel - and the host, which reads the status once at the end, reports the
 * error or redoes the batch with the table K2 asked for. */
#define PIPE_GATE(st) do { if ((st)[ST_ERROR] | (st)[ST_RETRY]) return; } while (0)

/* zero the histogram: cell_start[0 .. total cells + 1] (the total is on the device).  A workgroup clears 16 KB:
 * four rounds of one 16-byte store per thread, consecutive threads at consecutive addresses. */
__global__ __launch_bounds__(SASA_PIPE_B) void k_zero_cells(PipeArgs a)
{
    PIPE_GATE(a.status);
    const long long n = a.ncells[a.n_structs] + 2;
    const Int4 z = {0, 0, 0, 0};
    for (int k = 0; k < 4; ++k) {
        const long long base = (((long long)blockIdx.x * 4 + k) * SASA_PIPE_B + threadIdx.x) * 4;
        if (base + 4 <= n) *(Int4 *)(a.cell_start + base) = z;
        else for (long long i = base; i < n; ++i) a.cell_start[i] = 0;
    }
}

__global__ __launch_bounds__(SASA_PIPE_B) void k_count(PipeArgs a)
{
    __shared__ int cells[SASA_PIPE_B], base[SASA_PIPE_B];
    PIPE_GATE(a.status);
    const int i = blockIdx.x * SASA_PIPE_B + threadIdx.x;
    count_phase0(a, cells, i, threadIdx.x);
    __syncthreads();
    count_phase1(a, cells, base, threadIdx.x, SASA_PIPE_B);
    __syncthreads();
    count_phase2(a, cells, base, i, threadIdx.x);
}

static_assert(SASA_PIPE_B == SASA_SCAN_GROUP * SASA_SCAN_GROUP, "two-level combine of the scan partials");
/* n = total cells and the number of scan blocks follow from K2's total on the device; the launches are sized for
 * the table's capacity, blocks beyond the end leave at once */
__device__ __forceinline__ int scan_blocks(long long n)
{
    return (int)((n + 1 + (long long)SASA_PIPE_B * SASA_SCAN_ITEMS - 1) / ((long long)SASA_PIPE_B * SASA_SCAN_ITEMS));
}
__global__ __launch_bounds__(SASA_PIPE_B) void k_scan1(PipeArgs a)
{
    __shared__ int part[SASA_PIPE_B], part2[SASA_SCAN_GROUP];
    PIPE_GATE(a.status);
    const long long n = a.ncells[a.n_structs];
    if ((int)blockIdx.x >= scan_blocks(n)) return;
    scan1_phase0(a, n, part, blockIdx.x, threadIdx.x, SASA_PIPE_B);
    __syncthreads();
    scan_group_sums(part, part2, threadIdx.x);
    __syncthreads();
    scan1_phase2(a, part2, blockIdx.x, threadIdx.x);
}
__global__ __launch_bounds__(SASA_PIPE_B) void k_scan2(PipeArgs a)
{
    __shared__ int part[SASA_PIPE_B];
    PIPE_GATE(a.status);
    const int nblk = scan_blocks(a.ncells[a.n_structs]);
    scan2_phase0(a, nblk, part, threadIdx.x, SASA_PIPE_B);
    __syncthreads();
    scan2_phase1(part, threadIdx.x, SASA_PIPE_B);
    __syncthreads();
    scan2_phase2(a, nblk, part, threadIdx.x, SASA_PIPE_B);
}

__global__ __launch_bounds__(SASA_PIPE_B) void k_scan3(PipeArgs a)
{
    __shared__ int part[SASA_PIPE_B], part2[SASA_SCAN_GROUP];
    ScanRegs r;
    PIPE_GATE(a.status);
    const long long n = a.ncells[a.n_structs];
    if ((int)blockIdx.x >= scan_blocks(n)) return;
    scan3_phase0(a, n, part, blockIdx.x, threadIdx.x, SASA_PIPE_B, r);
    __syncthreads();
    scan3_phase1(part, part2, threadIdx.x);
    __syncthreads();
    scan3_phase2(part2, threadIdx.x);
    __syncthreads();
    scan3_phase3(a, n, part, part2, blockIdx.x, threadIdx.x, SASA_PIPE_B, r);
}

__global__ __launch_bounds__(SASA_PIPE_B) void k_scatter(PipeArgs a)
{
    PIPE_GATE(a.status);
    scatter_atom(a, blockIdx.x * SASA_PIPE_B + threadIdx.x);
}

/* ---------------------------------------------------------------------------------------------
 * K3-K5 in ONE kernel for batches of small structures (round 3): one workgroup sorts one structure in LDS.
 * The five launches it replaces (zero, count, three scan launches, scatter) stream the batch-wide cell table four
 * times — 9 cells per atom on random coils, 0.36 GB a pass for 1e7 atoms — and hand the cell and the rank of every
 * atom from one kernel to the next through HBM.  Here the table of a structure exists only as a bit per cell in
 * LDS (which cells hold atoms), a count of the occupied cells before every 32-cell word, and the atom counts of
 * the occupied cells; the first atom of any cell is  first[popcount rank of the cell]:
 *     A  bit of every atom's cell                       (cell of the atom: ref src/nb.c:74-83,137-140)
 *     B  occupied cells before each word                (block scan of the words' popcounts)
 *     C  every atom takes a place in its cell           (LDS atomic on the occupied cell's 16-bit counter)
 *     D  first atom of every occupied cell              (block scan of the counters, in place)
 *     E  atoms to their sorted places: sq, s_idx        (what k_scatter writes)
 *     F  the structure's part of cell_start[], once     (what the tile kernels' P0 reads: unchanged)
 * HBM traffic: the atoms read twice (the second time from cache), 48 B per atom and 4 B per cell written once.
 * Limits (else the batch is redone with the general pipeline, ST_RETRY = 2): SORT_ATOMS atoms and 2^26 cells per
 * structure.  The order of the atoms inside a cell is the order of arrival, as before: no result
 * depends on it (lr2_tie12). */
#define SORT_B 1024
#define SORT_APT 16
#define SORT_ATOMS (SORT_B * SORT_APT - 256) /* atoms of a structure (the 256 short of 16 threads' worth: two workgroups' LDS per CU) */
#define SORT_CELLS (1 << 18) /* cells in LDS at a time (a bit each); a structure with more is done in that many passes */
#define SORT_WORDS (SORT_CELLS / 32)
#define SORT_CELL_BITS 26    /* cells of one structure this kernel can number (6 more bits hold the border flags) */
__device__ __forceinline__ int sort_block_scan(int v, int *scratch, int tid) /* exclusive prefix of v over the workgroup; scratch[SORT_B / 64 + 1], the total in its last word */
{
    const int lane = tid & 63, wave = tid >> 6;
    int incl = v;
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    __syncthreads(); /* (scratch may still be read from the previous scan) */
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    if (tid < 64) {
        const int w = tid < SORT_B / 64 ? scratch[tid] : 0;
        int wi = w;
        for (int d = 1; d < SORT_B / 64; d <<= 1) {
            const int o = __shfl_up(wi, d, 64);
            if (tid >= d) wi += o;
        }
        if (tid < SORT_B / 64) scratch[tid] = wi - w;
        if (tid == SORT_B / 64 - 1) scratch[SORT_B / 64] = wi; /* total */
    }
    __syncthreads();
    return scratch[wave] + incl - v;
}
__global__ __launch_bounds__(SORT_B) void k_sort_struct(PipeArgs a)
{
    __shared__ unsigned bm[SORT_WORDS];              /* 32 KB: which cells hold atoms */
    __shared__ unsigned short wpre[SORT_WORDS];      /* 16 KB: occupied cells before each 32-cell word */
    __shared__ unsigned cnt[SORT_ATOMS / 2 + 2];     /* 32 KB: 16-bit counters of the occupied cells, then their first atoms (+ end) */
    __shared__ int scratch[SORT_B / 64 + 1];
    double (*const red)[SORT_B / 64] = (double (*)[SORT_B / 64])wpre; /* (bounds: before wpre is in use) */
    __shared__ GridS g_sh;
    __shared__ long long c_sh;
    /* No PIPE_GATE here: this is the first kernel behind the status memset, so the only flags it could see are those
       of sibling workgroups of the same launch - read per wave (a divergent barrier below), and a workgroup that left
       early would not add its cells to the batch total the host sizes the retry with.  Every workgroup numbers its
       cells; whether it sorts is decided uniformly through c_sh. */
    const int s = blockIdx.x, tid = threadIdx.x;
    const long long b0 = a.offsets[s];
    const int n = (int)(a.offsets[s + 1] - b0);
    if (n <= 0) { /* empty structure: as grid_struct */
        if (tid == 0) { GridS e; e.x0 = e.y0 = e.z0 = 0; e.d = 1; e.nx = e.ny = e.nz = 0; e.cell_base = 0; a.grid[s] = e; a.ncells[s] = 0; }
        return;
    }
    if (n > SORT_ATOMS) { /* (uniform) not a structure for this kernel: the host redoes the batch */
        if (tid == 0) atomicOr(&a.status[ST_RETRY], 2);
        return;
    }
    /* K1 + K2 of the general pipeline, for this structure: bounds, grid (grid_struct), and its run of the batch-wide
       cell numbering - taken from a counter, one cell more than it has (the entry behind its last cell is its own) */
    {
        double lo0 = INFINITY, lo1 = INFINITY, lo2 = INFINITY, hi0 = -INFINITY, hi1 = -INFINITY, hi2 = -INFINITY, rmax = 0; /* ref: src/nb.c:246 */
        int bad = 0;
        for (int li = tid; li < n; li += SORT_B) {
            const long long i = b0 + li;
            const double x = a.xyz[3 * i], y = a.xyz[3 * i + 1], z = a.xyz[3 * i + 2];
            lo0 = fmin(x, lo0); hi0 = fmax(x, hi0);
            lo1 = fmin(y, lo1); hi1 = fmax(y, hi1);
            lo2 = fmin(z, lo2); hi2 = fmax(z, hi2);
            const double rr = a.radii[a.shared_radii ? (long long)li : i];
            rmax = fmax(rr + a.probe, rmax);
            if (!(x - x == 0) || !(y - y == 0) || !(z - z == 0)) bad = ERR_BAD_COORD; /* (as bounds_phase0) */
            if (!(rr - rr == 0)) bad = bad ? bad : ERR_BAD_RADIUS;
        }
        if (bad) atomicMax(&a.status[ST_ERROR], bad);
        for (int d = 1; d < 64; d <<= 1) {
            lo0 = fmin(lo0, __shfl_xor(lo0, d, 64)); lo1 = fmin(lo1, __shfl_xor(lo1, d, 64)); lo2 = fmin(lo2, __shfl_xor(lo2, d, 64));
            hi0 = fmax(hi0, __shfl_xor(hi0, d, 64)); hi1 = fmax(hi1, __shfl_xor(hi1, d, 64)); hi2 = fmax(hi2, __shfl_xor(hi2, d, 64));
            rmax = fmax(rmax, __shfl_xor(rmax, d, 64));
        }
        if ((tid & 63) == 0) {
            const int w = tid >> 6;
            red[0][w] = lo0; red[1][w] = lo1; red[2][w] = lo2; red[3][w] = hi0; red[4][w] = hi1; red[5][w] = hi2; red[6][w] = rmax;
        }
        __syncthreads();
        if (tid == 0) {
            double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, rm = 0;
            for (int w = 0; w < SORT_B / 64; ++w) {
                for (int k = 0; k < 3; ++k) { lo[k] = fmin(red[k][w], lo[k]); hi[k] = fmax(red[3 + k][w], hi[k]); }
                rm = fmax(red[6][w], rm);
            }
            GridS g;
            const double d = 2 * rm; /* ref: src/nb.c:543 */
            int err = ERR_NONE;
            if (!(d > 0) || !(d < INFINITY)) err = ERR_BAD_RADIUS;
            for (int k = 0; k < 3; ++k)
                if (!(lo[k] > -INFINITY && hi[k] < INFINITY)) err = err ? err : ERR_BAD_COORD;
            long long nc = 0;
            g.d = d;
            g.x0 = lo[0] - d / 2.; g.y0 = lo[1] - d / 2.; g.z0 = lo[2] - d / 2.; /* ref: src/nb.c:61-66 */
            g.nx = g.ny = g.nz = 0; g.cell_base = 0;
            if (!err) {
                const double fx = ceil((hi[0] + d / 2. - g.x0) / d); /* ref: src/nb.c:67-69 */
                const double fy = ceil((hi[1] + d / 2. - g.y0) / d);
                const double fz = ceil((hi[2] + d / 2. - g.z0) / d);
                if (!(fx * fy * fz <= (double)a.max_cells) || !(fx >= 1 && fy >= 1 && fz >= 1)) err = ERR_GRID_TOO_BIG;
                else { g.nx = (int)fx; g.ny = (int)fy; g.nz = (int)fz; nc = (long long)g.nx * g.ny * g.nz; }
            }
            if (err) {
                atomicMax(&a.status[ST_ERROR], err);
                g.nx = g.ny = g.nz = 1; g.d = 1; g.x0 = g.y0 = g.z0 = 0;
                nc = 1; /* keep the rest of the pipeline in bounds; host discards results */
            }
            /* (compact cell table: a structure's cells start at a multiple of 32, its table words are its own) */
            const long long take = a.cell_tbl ? ((nc + 1 + 31) & ~31LL) : nc + 1;
            const long long base = (long long)atomicAdd((unsigned long long *)&a.ncells[a.n_structs], (unsigned long long)take);
            if (base + nc + 1 > a.max_cells) { atomicMax(&a.status[ST_ERROR], (int)ERR_GRID_TOO_BIG); nc = -1; }
            else if (nc > (1LL << SORT_CELL_BITS)) { atomicOr(&a.status[ST_RETRY], 2); nc = -1; }
            else if (a.cells_cap > 0 && base + nc + 1 > a.cells_cap) { atomicOr(&a.status[ST_RETRY], 1); nc = -1; }
            else if (a.status[ST_RETRY] | a.status[ST_ERROR]) nc = -1; /* the batch is redone anyway (a sibling's flag; any order is fine: this structure's cells are counted) */
            g.cell_base = (int)base;
            a.grid[s] = g;
            a.ncells[s] = nc < 0 ? 0 : nc;
            g_sh = g; c_sh = nc;
        }
        __syncthreads();
    }
    const GridS g = g_sh;
    if (c_sh < 0) return; /* (uniform) no room in the cell table, or too many cells for this kernel: the host redoes the batch */
    const int C = (int)c_sh;
    /* the cell of every atom of this thread: cell | border flags << SORT_CELL_BITS */
    unsigned cellf[SORT_APT];
    for (int k = 0; k < SORT_APT; ++k) {
        const int li = tid + k * SORT_B;
        cellf[k] = 0xffffffffu;
        if (li < n) {
            const long long i = b0 + li;
            int ix = cell_coord(a.xyz[3 * i], g.x0, g.d);
            int iy = cell_coord(a.xyz[3 * i + 1], g.y0, g.d);
            int iz = cell_coord(a.xyz[3 * i + 2], g.z0, g.d);
            if (!(ix >= 0 && ix < g.nx && iy >= 0 && iy < g.ny && iz >= 0 && iz < g.nz)) { /* as count_phase0 */
                if (a.status[ST_ERROR] == 0) atomicMax(&a.status[ST_ERROR], (int)ERR_BAD_COORD);
                ix = iy = iz = 0;
            }
            const int c = ix + g.nx * (iy + g.ny * iz); /* ref: src/nb.c:74-83 */
            const int fl = (ix == 0 ? CELL_X0 : 0) | (ix == g.nx - 1 ? CELL_X1 : 0) | (iy == 0 ? CELL_Y0 : 0) |
                           (iy == g.ny - 1 ? CELL_Y1 : 0) | (iz == 0 ? CELL_Z0 : 0) | (iz == g.nz - 1 ? CELL_Z1 : 0);
            cellf[k] = (unsigned)c | ((unsigned)fl << SORT_CELL_BITS);
        }
        if ((k & 3) == 3) __asm__ volatile("" ::: "memory"); /* (four atoms' loads in flight, not sixteen: the kernel has 128 registers) */
    }
    const unsigned cmask = (1u << SORT_CELL_BITS) - 1u;
    int base = 0; /* atoms in the cells of the passes before this one */
    int occ_done = 0; /* occupied cells of the passes before this one */
    const long long occ_base = b0 + s; /* this structure's entries of cell_first: one per occupied cell (<= n) and one behind them */
    for (int lo = 0; lo < C; lo += SORT_CELLS) { /* (once, unless the structure has more than SORT_CELLS cells) */
        const int Cp = C - lo < SORT_CELLS ? C - lo : SORT_CELLS, W = (Cp + 31) >> 5;
        __syncthreads();
        for (int w = tid; w < W; w += SORT_B) bm[w] = 0;
        for (int k = tid; k < n / 2 + 2; k += SORT_B) cnt[k] = 0;
        __syncthreads();
        /* A */
        for (int k = 0; k < SORT_APT; ++k) {
            const int cc = (int)(cellf[k] & cmask) - lo;
            if (cellf[k] != 0xffffffffu && cc >= 0 && cc < Cp) atomicOr(&bm[cc >> 5], 1u << (cc & 31));
        }
        __syncthreads();
        /* B */
        {
            int pc[SORT_WORDS / SORT_B], sum = 0;
            for (int j = 0; j < SORT_WORDS / SORT_B; ++j) {
                const int w = tid * (SORT_WORDS / SORT_B) + j;
                pc[j] = w < W ? __popc(bm[w]) : 0;
                sum += pc[j];
            }
            int run = sort_block_scan(sum, scratch, tid);
            for (int j = 0; j < SORT_WORDS / SORT_B; ++j) {
                const int w = tid * (SORT_WORDS / SORT_B) + j;
                if (w < W) wpre[w] = (unsigned short)run;
                run += pc[j];
            }
        }
        __syncthreads();
        const int occ = scratch[SORT_B / 64];
        /* C */
        unsigned place[SORT_APT]; /* occupied-cell number | place in the cell << 16 */
        for (int k = 0; k < SORT_APT; ++k) {
            place[k] = 0xffffffffu;
            const int cc = (int)(cellf[k] & cmask) - lo;
            if (cellf[k] != 0xffffffffu && cc >= 0 && cc < Cp) {
                const int oc = (int)wpre[cc >> 5] + __popc(bm[cc >> 5] & ((1u << (cc & 31)) - 1u));
                const unsigned old = atomicAdd(&cnt[oc >> 1], (oc & 1) ? 0x10000u : 1u);
                place[k] = (unsigned)oc | (((old >> ((oc & 1) * 16)) & 0xffffu) << 16);
            }
        }
        __syncthreads();
        /* D: counters -> first atoms, in place; entry occ = atoms of this pass */
        unsigned short *const c16 = (unsigned short *)cnt;
        {
            int v[SORT_APT], sum = 0;
            for (int j = 0; j < SORT_APT; ++j) {
                const int e = tid * SORT_APT + j;
                v[j] = e < occ ? (int)c16[e] : 0;
                sum += v[j];
            }
            int run = sort_block_scan(sum, scratch, tid);
            for (int j = 0; j < SORT_APT; ++j) {
                const int e = tid * SORT_APT + j;
                if (e <= occ) c16[e] = (unsigned short)run; /* (at most 16384 atoms) */
                run += v[j];
            }
        }
        __syncthreads();
        const int n_pass = scratch[SORT_B / 64];
        /* E */
        for (int k = 0; k < SORT_APT; ++k) {
            if (place[k] == 0xffffffffu) continue;
            const int li = tid + k * SORT_B;
            const long long i = b0 + li;
            const int oc = (int)(place[k] & 0xffffu);
            const long long p = b0 + base + c16[oc] + (int)(place[k] >> 16);
            Quad v;
            v.x = a.xyz[3 * i]; v.y = a.xyz[3 * i + 1]; v.z = a.xyz[3 * i + 2];
            v.w = a.radii[a.shared_radii ? (long long)li : i] + a.probe; /* ref: src/sasa_lr.c:136, src/sasa_sr.c:144 */
            a.sq[p] = v;
            SortIdx si;
            si.cell = (long long)((unsigned long long)(unsigned)(g.cell_base + (int)(cellf[k] & cmask)) |
                                  ((unsigned long long)((cellf[k] >> SORT_CELL_BITS) | cell_pack_grid(g.nx, g.ny)) << 32));
            si.orig = (int)i; si.strct = s;
            a.s_idx[p] = si;
            if (a.occ_stride > 0 && i % a.occ_stride == 0) { /* density samples, as scatter_atom */
                atomicAdd(&a.status[ST_OCC_SUM], (int)c16[oc + 1] - (int)c16[oc]);
                atomicAdd(&a.status[ST_OCC_N], 1);
            }
            if ((k & 3) == 3) __asm__ volatile("" ::: "memory");
        }
        /* F */
        if (a.cell_tbl) { /* (uniform) compact: the words of this pass, the first atoms of its occupied cells */
            unsigned long long *const tbl = a.cell_tbl + (((long long)g.cell_base + lo) >> 5);
            for (int w = tid; w < W; w += SORT_B) tbl[w] = (unsigned long long)bm[w] | ((unsigned long long)(unsigned)(occ_base + occ_done + wpre[w]) << 32);
            for (int e = tid; e < occ; e += SORT_B) a.cell_first[occ_base + occ_done + e] = (int)(b0 + base + c16[e]);
        } else {
            for (int cc = tid; cc < Cp; cc += SORT_B) {
                const int oc = (int)wpre[cc >> 5] + __popc(bm[cc >> 5] & ((1u << (cc & 31)) - 1u));
                a.cell_start[g.cell_base + lo + cc] = (int)(b0 + base + c16[oc]);
            }
        }
        base += n_pass;
        occ_done += occ;
    }
    if (tid == 0) { /* the entry behind the structure's last cell: its end */
        if (a.cell_tbl) {
            a.cell_first[occ_base + occ_done] = (int)(b0 + n);
            if ((C & 31) == 0) a.cell_tbl[((long long)g.cell_base + C) >> 5] = (unsigned long long)(unsigned)(occ_base + occ_done) << 32; /* (cell C opens a word of its own: no cell of it holds atoms) */
        } else {
            a.cell_start[g.cell_base + C] = (int)(b0 + n);
        }
    }
}

__global__ __launch_bounds__(SASA_TOT_B) void k_totals(const double *sasa, const int64_t *offsets, int n_structs, double *totals)
{
    __shared__ double part[SASA_TOT_B];
    totals_phase0(sasa, offsets, part, blockIdx.x, threadIdx.x);
    __syncthreads();
    totals_phase1(part, totals, blockIdx.x, threadIdx.x);
}

__global__ __launch_bounds__(SASA_TOT_B) void k_totals_chunks(PipeArgs a, const double *sasa, double *chunk_tot)
{
    __shared__ double part[SASA_TOT_B];
    totals_chunk_phase0(a, sasa, part, blockIdx.x, threadIdx.x);
    __syncthreads();
    totals_chunk_phase1(part, chunk_tot, blockIdx.x, threadIdx.x);
}
__global__ __launch_bounds__(256) void k_totals_structs(PipeArgs a, const double *chunk_tot, double *totals)
{
    totals_struct(a, chunk_tot, totals, blockIdx.x * 256 + threadIdx.x);
}

__global__ __launch_bounds__(256) void k_segsum_small(const double *sasa, const int64_t *seg, int n_segs, double *out)
{
    segsum_small(sasa, seg, out, blockIdx.x * 256 + threadIdx.x, n_segs);
}

__global__ __launch_bounds__(256) void k_residue_areas(const double *sasa, const unsigned char *cls, const unsigned char *bb,
                                                      const int64_t *res_first, const short *ref_row, const double *ref_table,
                                                      double *abs_out, double *rel_out, int n_res)
{
    residue_areas(sasa, cls, bb, res_first, ref_row, ref_table, abs_out, rel_out, blockIdx.x * 256 + threadIdx.x, n_res);
}

__global__ __launch_bounds__(SASA_TOT_B) void k_class_sums(const double *sasa, const unsigned char *cls, const int64_t *offsets, double *out)
{
    __shared__ double part[3 * SASA_TOT_B];
    class_phase0(sasa, cls, offsets, part, blockIdx.x, threadIdx.x);
    __syncthreads();
    class_phase1(part, out, blockIdx.x, threadIdx.x);
}

/* 4 waves per SIMD (<= 128 VGPRs): the kernel hides its LDS/global latency and the barriers of
 * one tile behind other resident tiles, so occupancy is worth a 16-byte spill (measured). */
/* TIER only names the launch (0 main, 1 second, 2 slab) so that profiles list them separately.
 * WPE = waves per SIMD the register allocation is capped for: 4 (128 VGPRs), or 5 (96 VGPRs, no
 * spill since atan2_fast) when the tile's LDS footprint lets more than 16 one-wave tiles reside. */
#ifdef SASA_PHASE_TIMING
#define PHASE_MARK(k) do { if (tid == 0 && TIER == 0 && (tile & 255) == 0) { const unsigned long long now_ = wall_clock64(); \
        atomicAdd(&g_phase_clock[k], now_ - last_); last_ = now_; } } while (0)
#else
#define PHASE_MARK(k) do { } while (0)
#endif
template <int B, bool GLOBAL, int TIER, int WPE, bool BUCKET = false>
__global__ __launch_bounds__(B) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lr_tile(TileArgs a, int items)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    PIPE_GATE(a.status);
    TileMem m = tile_carve<GLOBAL>(a, smem, items, B, blockIdx.x);
    const int n_work = a.work_tiles ? *a.work_count : ((a.n_tiles + 7) >> 3) << 3;
    int wg_max_nn = 0;
    for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
        const int tile = a.work_tiles ? a.work_tiles[w] : xcd_tile(w, a.n_tiles);
        if (tile >= a.n_tiles) continue; /* uniform per workgroup */
#ifdef SASA_PHASE_TIMING
        unsigned long long last_ = wall_clock64();
#endif
        tile_phase_load(a, m, tile, tid, B, BUCKET);
        __syncthreads();
        PHASE_MARK(0);
        tile_phase_neighbors(a, m, tile, tid, B);
        __syncthreads();
        PHASE_MARK(1);
        tile_phase_offsets(a, m, tid);
        __syncthreads();
        PHASE_MARK(2);
        tile_report<GLOBAL>(a, m, tile, tid, wg_max_nn);
        PHASE_MARK(3);
        lr_phase_beta(a, m, tid, B, BUCKET);
        __syncthreads();
        PHASE_MARK(4);
        if (BUCKET && lr_bucket_path(a, m, B)) { /* uniform per workgroup */
            RankRegs rr;
            lr_phase_prefix(a, m, tid);
            __syncthreads();
            lr_phase_scatter(a, m, tid, B);
            __syncthreads();
            lr_phase_rank2(a, m, tid, B, rr);
            __syncthreads();
            lr_phase_write(a, m, tid, B, rr);
        } else {
            lr_phase_rank(a, m, tid, B);
        }
        __syncthreads();
        PHASE_MARK(5);
        lr_phase_slices(a, m, tile, tid, B);
        __syncthreads();
        PHASE_MARK(6);
        lr_phase_store<GLOBAL>(a, m, tile, tid, B);
        __syncthreads();
        PHASE_MARK(7);
#ifdef SASA_PHASE_TIMING
        if (tid == 0 && TIER == 0 && (tile & 255) == 0) atomicAdd(&g_phase_clock[15], 1ULL);
#endif
    }
    tile_report_flush(a, tid, wg_max_nn);
}

/* Second-generation L&R kernel (lr2_kernels.h): one wave per tile.  RMAX = rounds of pair records a
 * lane keeps in registers (2-4: main launch, by the pool; 6: second launch); WPE = waves per SIMD the register
 * allocation is capped for; TIER: 0 main launch, 2 second launch, 4 the main launch of the neighbor test hooks
 * (freesasa_gpu_lr_neighbors_dev: the only build that carries their code). */
/* NOTE: Lr2Args must stay the ONLY parameter of this kernel, at offset 0 of the kernel-argument segment: the tile
 * body reads its rarely used fields from there (LR2_COLD in lr2_kernels.h). */
template <int RMAX, int TIER, int WPE, bool COVER, bool PAIRS = false, int SHAPE = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lr2_tile(Lr2Args a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    PIPE_GATE(a.status);
    Lr2Mem m = lr2_carve<SHAPE>(a, smem);
    int wg_max_nn = 0;
    lr2_wave<RMAX, COVER, PAIRS, SHAPE, (TIER & 4) != 0>(a, m, blockIdx.x, gridDim.x, lane, wg_max_nn);
    if (lane == 0 && wg_max_nn > a.status[ST_MAX_NN]) atomicMax(&a.status[ST_MAX_NN], wg_max_nn);
}
__global__ __launch_bounds__(64) void k_lr2_arc_kat(const double *arcs, const int *first, int n_sets, double *out)
{
    __shared__ Arc2 stack[8 * 64];
    const int k = threadIdx.x;
    if (k < n_sets) out[k] = lr2_arc_kat(arcs, first, k, stack + k, 8);
}

/* main launch: the instantiation is picked by the rounds of pair records the pool needs; registers are capped for
 * 4 waves per SIMD (a 5-wave build spilled and was not faster) */
static hipError_t launch_lr2_main(int rmax, int grid, size_t lds, hipStream_t st, const Lr2Args &la)
{
    /* (the cover filter is compiled into the launches over dense batches only: the sparse ones keep its registers) */
#define LR2_LAUNCH(R) do { \
        if (la.hooks) hipLaunchKernelGGL((k_lr2_tile<R, 4, 4, false>), dim3(grid), dim3(64), lds, st, la); \
        else if (lr2_pairs_shape(la.TA, la.ns)) { \
            if (la.cover > 0) hipLaunchKernelGGL((k_lr2_tile<R, 0, 4, true, true>), dim3(grid), dim3(64), lds, st, la); \
            else if (lr2_default_shape(la.TA, la.ns, la.mw, la.ds) && !getenv("FREESASA_AMD_NO_SHAPE")) hipLaunchKernelGGL((k_lr2_tile<R, 0, 4, false, true, 1>), dim3(grid), dim3(64), lds, st, la); \
            else hipLaunchKernelGGL((k_lr2_tile<R, 0, 4, false, true>), dim3(grid), dim3(64), lds, st, la); \
        } else if (la.cover > 0) hipLaunchKernelGGL((k_lr2_tile<R, 0, 4, true>), dim3(grid), dim3(64), lds, st, la); \
        else hipLaunchKernelGGL((k_lr2_tile<R, 0, 4, false>), dim3(grid), dim3(64), lds, st, la); } while (0)
    /* shapes 2-4 (lr2_shape_id), each with the pair-record rounds its workload asks for; any other combination: the generic builds below */
    if (!la.hooks && !getenv("FREESASA_AMD_NO_SHAPE")) {
        const int sid = lr2_shape_id(la.TA, la.ns, la.mw, la.ds);
        if (sid == 2 && rmax <= 2 && la.cover == 0) { hipLaunchKernelGGL((k_lr2_tile<2, 0, 4, false, false, 2>), dim3(grid), dim3(64), lds, st, la); return hipGetLastError(); }
        if (sid == 3 && rmax == 4 && la.cover > 0) { hipLaunchKernelGGL((k_lr2_tile<4, 0, 4, true, false, 3>), dim3(grid), dim3(64), lds, st, la); return hipGetLastError(); }
        if (sid == 4 && rmax == 4 && la.cover > 0) { hipLaunchKernelGGL((k_lr2_tile<4, 0, 4, true, true, 4>), dim3(grid), dim3(64), lds, st, la); return hipGetLastError(); }
    }
    if (rmax <= 2) LR2_LAUNCH(2); else if (rmax == 3) LR2_LAUNCH(3); else LR2_LAUNCH(4);
#undef LR2_LAUNCH
    return hipGetLastError();
}

template <int B, bool GLOBAL, int TIER>
__global__ __launch_bounds__(B) void k_sr_tile(TileArgs a, int items)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    PIPE_GATE(a.status);
    TileMem m = tile_carve<GLOBAL>(a, smem, items, B, blockIdx.x);
    const int n_work = a.work_tiles ? *a.work_count : ((a.n_tiles + 7) >> 3) << 3;
    int wg_max_nn = 0;
    for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
        const int tile = a.work_tiles ? a.work_tiles[w] : xcd_tile(w, a.n_tiles);
        if (tile >= a.n_tiles) continue;
        tile_phase_load(a, m, tile, tid, B);
        __syncthreads();
        tile_phase_neighbors(a, m, tile, tid, B);
        __syncthreads();
        tile_phase_offsets(a, m, tid);
        sr_phase_cursors(a, m, tid);
        __syncthreads();
        tile_report<GLOBAL>(a, m, tile, tid, wg_max_nn);
        sr_phase_pairs(a, m, tid, B);
        __syncthreads();
        sr_phase_points(a, m, tile, tid, B);
        __syncthreads();
        sr_phase_points2(a, m, tid, B);
        __syncthreads();
        sr_phase_store(a, m, tile, tid);
        __syncthreads();
    }
    tile_report_flush(a, tid, wg_max_nn);
}

/* ------------------------------------------------------------------ context */

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct freesasa_gpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool timing = false;
    bool shared_radii = false; /* d_radii holds ONE structure's radii (trajectory frames) */
    char err[512] = {0};
    freesasa_gpu_stats stats = {};
    /* Two sets of what the HOST reads of a batch (page-locked status words, stage events, end-of-batch event): a batch
       submitted with freesasa_gpu_lr_batch_dev_async leaves its set behind until it is collected, while the next one
       is enqueued with the other set.  The device side needs no second copy: the copies into a set are enqueued at
       the end of their batch, in stream order before the next batch resets the device words. */
    int slot = 0;
    hipEvent_t evs[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
    hipEvent_t done[2] = {nullptr, nullptr};
    struct Pend {
        bool active = false;
        /* the call, for the rare batch that has to be redone (cell table sizing, see RC_RETRY) */
        const double *d_xyz = nullptr, *d_radii = nullptr;
        std::vector<int64_t> offsets;
        int n_structs = 0, resolution = 0;
        double probe = 0;
        double *d_sasa = nullptr, *d_totals = nullptr;
        /* what completing it needs */
        int n = 0, TA = 0, mw = 0, ds = 0, lds = 0;
    } pend[2];
    /* workspace */
    DevBuf offsets, grid, ncells, sid, cell_of, rank, cell_start, blk_sums, cell_tbl, cell_first;
    DevBuf chunk_struct, chunk_begin, chunk_len, struct_chunk0, bpart;
    int n_chunks = 0;
    DevBuf sq, s_idx;
    DevBuf status, ovf_tiles, ovf_tiles2, ovf_atoms, unit_pts, slab, seg;
    std::vector<int64_t> offsets_host; /* last uploaded offsets */
    std::vector<double> unit_host;     /* last uploaded S&R unit points */
    /* host staging for freesasa_gpu_calc_batch */
    DevBuf h_xyz, h_radii, h_sasa, h_counts, h_totals;
    void *stage_in = nullptr, *stage_out = nullptr; /* page-locked host staging of freesasa_gpu_calc_batch_pipelined */
    size_t stage_in_cap = 0, stage_out_cap = 0;
    int *pinned = nullptr; /* page-locked host words for the small device->host readbacks: two sets of ST_WORDS + 4 */
    long long max_cells = 1LL << 30;
    long long cells_hint = 0; /* cells the last batch needed, with a margin: the table is never sized below it */
    /* adaptive neighbor-pool size, per algorithm: (resolution, TA) it was learnt for and the value */
    int hint_res[2] = {0, 0}, hint_ta[2] = {0, 0}, hint_pool[2] = {0, 0};
    double hint_probe = -1.0; /* the probe radius the hints were learnt with (another probe: other neighbor counts, so they start over) */
    bool hint_bucket = false; /* L&R: the last batch had long neighbor lists */
    bool sort_fused = true;   /* the per-structure cell sort (k_sort_struct) until a batch turns out not to fit it */
    double hint_nn = 0;       /* L&R (lr2): neighbor records per atom the main launch should hold */
    int hint_nn_max = 0;      /* ... and the longest neighbor list expected (mask words per item) */
    double hint_split2 = 0;   /* ... the share of its tiles above the 16-tiles-per-CU pool */
    int hint_pool2 = 0, hint_ta2 = 0, hint_mw2 = 0; /* ... and the pool the last batch's demand histogram asks for, for tiles of that shape */
    int *dbg_nn = nullptr, *dbg_nb = nullptr; /* test hook: freesasa_gpu_lr_neighbors_dev */
    int dbg_cap = 0;
};

static int ctx_fail(freesasa_gpu_ctx *c, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof c->err, fmt, ap);
    va_end(ap);
    return -1;
}

#define HIP_TRY(c, call)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return ctx_fail((c), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

static inline int *ctx_status_h(freesasa_gpu_ctx *c) { return c->pinned + c->slot * (ST_WORDS + 4); }
static inline hipEvent_t *ctx_ev(freesasa_gpu_ctx *c) { return c->evs[c->slot]; }
static int drain_pending(freesasa_gpu_ctx *c);

/* Test hook (include/freesasa_gpu.h, freesasa_gpu_test_fail_after): the n-th device / page-locked allocation from
 * now on fails, like the reference's interposed malloc (tests/tools.c:10-48) makes its n-th malloc fail. */
static std::atomic<int> g_fail_after(0);
extern "C" void freesasa_gpu_test_fail_after(int n) { g_fail_after.store(n > 0 ? n : 0); }
static bool alloc_fails()
{
    int v = g_fail_after.load();
    while (v > 0) {
        if (g_fail_after.compare_exchange_weak(v, v - 1)) return v == 1;
    }
    return false;
}
static hipError_t dev_malloc(void **p, size_t bytes)
{
    if (alloc_fails()) { *p = nullptr; return hipErrorOutOfMemory; }
    return hipMalloc(p, bytes);
}
static hipError_t host_malloc(void **p, size_t bytes)
{
    if (alloc_fails()) { *p = nullptr; return hipErrorOutOfMemory; }
    return hipHostMalloc(p, bytes, hipHostMallocDefault);
}

static int ensure(freesasa_gpu_ctx *c, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return 0;
    /* Batches in flight may be using the buffer about to be freed: wait for the stream, nothing more.  Their verdicts
       are already on their way into their own sets of page-locked words (enqueue_tail) and are read when they are
       collected; collecting them HERE - in the middle of another batch's set-up - could redo one of them through
       run_batch and leave this batch with status words, arguments and buffers that are no longer its own. */
    if (b.p && (c->pend[0].active || c->pend[1].active)) HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (b.p) HIP_TRY(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 4 + 256; /* slack: trajectories grow/shrink a little */
    HIP_TRY(c, dev_malloc(&b.p, want));
    b.cap = want;
    return 0;
}

extern "C" int freesasa_gpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" void freesasa_gpu_ctx_destroy(freesasa_gpu_ctx *c);
extern "C" freesasa_gpu_ctx *freesasa_gpu_ctx_create(int device, void *stream)
{
    int n = freesasa_gpu_device_count();
    if (n <= 0 || device >= n) return nullptr;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    freesasa_gpu_ctx *c = new freesasa_gpu_ctx();
    c->device = device;
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            freesasa_gpu_ctx_destroy(c);
            return nullptr;
        }
        c->own_stream = true;
    }
    for (int s_ = 0; s_ < 2; ++s_) {
        for (int k = 0; k < 4; ++k)
            if (hipEventCreate(&c->evs[s_][k]) != hipSuccess) {
                freesasa_gpu_ctx_destroy(c);
                return nullptr;
            }
        if (hipEventCreateWithFlags(&c->done[s_], hipEventDisableTiming) != hipSuccess) {
            freesasa_gpu_ctx_destroy(c);
            return nullptr;
        }
    }
    if (host_malloc((void **)&c->pinned, sizeof(int) * 2 * (ST_WORDS + 4)) != hipSuccess) {
        freesasa_gpu_ctx_destroy(c);
        return nullptr;
    }
    return c;
}

extern "C" void freesasa_gpu_ctx_destroy(freesasa_gpu_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf *all[] = {&c->chunk_struct, &c->chunk_begin, &c->chunk_len, &c->struct_chunk0, &c->bpart, &c->offsets, &c->grid, &c->ncells, &c->sid, &c->cell_of, &c->rank, &c->cell_start,
                     &c->blk_sums, &c->cell_tbl, &c->cell_first, &c->sq, &c->s_idx,
                     &c->status, &c->ovf_tiles, &c->ovf_tiles2, &c->ovf_atoms, &c->unit_pts, &c->slab, &c->seg,
                     &c->h_xyz, &c->h_radii, &c->h_sasa, &c->h_counts, &c->h_totals};
    for (DevBuf *b : all)
        if (b->p) (void)hipFree(b->p);
    for (int s_ = 0; s_ < 2; ++s_) {
        for (int k = 0; k < 4; ++k)
            if (c->evs[s_][k]) (void)hipEventDestroy(c->evs[s_][k]);
        if (c->done[s_]) (void)hipEventDestroy(c->done[s_]);
    }
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->stage_in) (void)hipHostFree(c->stage_in);
    if (c->stage_out) (void)hipHostFree(c->stage_out);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" void freesasa_gpu_ctx_set_timing(freesasa_gpu_ctx *c, int enable) { c->timing = enable != 0; }
extern "C" void freesasa_gpu_ctx_get_stats(const freesasa_gpu_ctx *c, freesasa_gpu_stats *out) { *out = c->stats; }
extern "C" const char *freesasa_gpu_ctx_last_error(const freesasa_gpu_ctx *c) { return c ? c->err : "no context"; }

/* ------------------------------------------------------------------ launch configuration */


template <bool GLOBAL, int TIER>
static hipError_t launch_lr(const TileCfg &c, const TileArgs &t, int grid, size_t lds, hipStream_t s, bool bucket = false)
{
    /* 160 KB of LDS per CU: more than 16 resident one-wave tiles only pay off with <= 96 VGPRs */
    const bool wpe5 = TIER == 0 && !GLOBAL && c.B == 64 && lds * 17 <= 160 * 1024;
    if (c.B == 320)
        hipLaunchKernelGGL((k_lr_tile<320, GLOBAL, TIER, 4>), dim3(grid), dim3(320), lds, s, t, c.items);
    else if (c.B == 256)
        hipLaunchKernelGGL((k_lr_tile<256, GLOBAL, TIER, 4>), dim3(grid), dim3(256), lds, s, t, c.items);
    else if (c.B == 128)
        hipLaunchKernelGGL((k_lr_tile<128, GLOBAL, TIER, 4>), dim3(grid), dim3(128), lds, s, t, c.items);
    else
        { if (wpe5) hipLaunchKernelGGL((k_lr_tile<64, GLOBAL, TIER, 5>), dim3(grid), dim3(64), lds, s, t, c.items);
          else if (bucket && !GLOBAL) hipLaunchKernelGGL((k_lr_tile<64, false, TIER, 4, true>), dim3(grid), dim3(64), lds, s, t, c.items);
          else hipLaunchKernelGGL((k_lr_tile<64, GLOBAL, TIER, 4>), dim3(grid), dim3(64), lds, s, t, c.items); }
    return hipGetLastError();
}
template <bool GLOBAL, int TIER>
static hipError_t launch_sr(const TileCfg &c, const TileArgs &t, int grid, size_t lds, hipStream_t s)
{
    if (c.B == 320)
        hipLaunchKernelGGL((k_sr_tile<320, GLOBAL, TIER>), dim3(grid), dim3(320), lds, s, t, c.items);
    else if (c.B == 256)
        hipLaunchKernelGGL((k_sr_tile<256, GLOBAL, TIER>), dim3(grid), dim3(256), lds, s, t, c.items);
    else if (c.B == 128)
        hipLaunchKernelGGL((k_sr_tile<128, GLOBAL, TIER>), dim3(grid), dim3(128), lds, s, t, c.items);
    else
        hipLaunchKernelGGL((k_sr_tile<64, GLOBAL, TIER>), dim3(grid), dim3(64), lds, s, t, c.items);
    return hipGetLastError();
}

static const char *err_text(int code)
{
    switch (code) {
    case ERR_BAD_RADIUS: return "cell size 2*max(radius+probe) is not positive and finite";
    case ERR_GRID_TOO_BIG: return "cell list too large for the coordinate extent (out of memory in the reference)";
    case ERR_BAD_COORD: return "non-finite coordinate";
    case ERR_NEIGHBOR_CAP: return "an atom has more neighbors than the GPU fallback path supports (4096 per atom, 16384 per tile)";
    case ERR_STACK_CAP: return "more disjoint arcs in one slice than the GPU fallback path supports";
    default: return "unknown device-side error";
    }
}

/* ------------------------------------------------------------------ L&R, second generation */

static void dump_phase_clocks()
{
#ifdef SASA_PHASE_TIMING
    {
        unsigned long long h[16];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_clock), sizeof h) == hipSuccess && h[15]) {
            static const char *names[8] = {"P0/load", "P1/neighbors", "P2/offsets", "P3/report|pairs", "P4/beta|screen", "P5/rank|queue", "P6/slices|arcs", "P7/store"};
            fprintf(stderr, "phase clocks (100 MHz ticks per sampled tile, thread 0, %llu tiles):", h[15]);
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.1f", names[k], (double)h[k] / (double)h[15]);
            fprintf(stderr, "\n");
            memset(h, 0, sizeof h);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_clock), h, sizeof h);
        }
    }
#endif
}

/* run_batch_once's third outcome: the cell table was too small, redo the batch (c->cells_hint has the size) */
#define RC_RETRY 2

static int judge_status(freesasa_gpu_ctx *c, const int *status_h, long long *total_cells);
/* Status words [0, words) and K2's cell total, read back behind everything enqueued so far.  Returns 0, RC_RETRY,
 * or -1 with the batch's error text set. */
static int collect_status(freesasa_gpu_ctx *c, int n_structs, int words, long long *total_cells)
{
    hipStream_t st = c->stream;
    int *status_h = ctx_status_h(c);
    long long *total_p = (long long *)(status_h + ST_WORDS + 2);
    HIP_TRY(c, hipMemcpyAsync(total_p, (long long *)c->ncells.p + n_structs, sizeof(long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(status_h, c->status.p, sizeof(int) * (size_t)words, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    return judge_status(c, status_h, total_cells);
}

/* the verdict of a batch's status words (already in host memory) */
static int judge_status(freesasa_gpu_ctx *c, const int *status_h, long long *total_cells)
{
    const long long *total_p = (const long long *)(status_h + ST_WORDS + 2);
    *total_cells = *total_p;
    if (status_h[ST_ERROR]) return ctx_fail(c, "%s", err_text(status_h[ST_ERROR]));
    if (*total_p <= 0 || *total_p > c->max_cells) return ctx_fail(c, "%s", err_text(ERR_GRID_TOO_BIG));
    if (status_h[ST_RETRY]) { /* bit flags (workgroups of one launch may raise both) */
        if (status_h[ST_RETRY] & 2) c->sort_fused = false; /* a structure k_sort_struct does not hold (atoms or cells): this context sorts the general way from now on */
        /* bit 0: the cell table was too small.  The total is every structure's demand (k_sort_struct: all but those it
           does not hold, which the general pipeline's own count covers on the next pass) */
        if ((status_h[ST_RETRY] & 1) && *total_p + *total_p / 16 + 1024 > c->cells_hint) c->cells_hint = *total_p + *total_p / 16 + 1024;
        return RC_RETRY;
    }
    return 0;
}

/* The tail of a batch on the stream: per-structure totals, then the status words and the cell total into the
 * current set of page-locked host words, then the set's end-of-batch event.  Nothing is waited for. */
static int enqueue_tail(freesasa_gpu_ctx *c, const PipeArgs &pa, int n_structs, double *d_sasa, double *d_totals)
{
    hipStream_t st = c->stream;
    if (d_totals) {
        /* the chunk partials reuse the bounds kernels' scratch (56 bytes per chunk, free by now) */
        hipLaunchKernelGGL(k_totals_chunks, dim3(c->n_chunks), dim3(SASA_TOT_B), 0, st, pa, (const double *)d_sasa, (double *)c->bpart.p);
        hipLaunchKernelGGL(k_totals_structs, dim3((n_structs + 255) / 256), dim3(256), 0, st, pa, (const double *)c->bpart.p, d_totals);
        HIP_TRY(c, hipGetLastError());
    }
    if (c->timing) HIP_TRY(c, hipEventRecord(ctx_ev(c)[3], st));
    int *status_h = ctx_status_h(c);
    HIP_TRY(c, hipMemcpyAsync(status_h + ST_WORDS + 2, (long long *)c->ncells.p + n_structs, sizeof(long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(status_h, c->status.p, sizeof(int) * (size_t)ST_WORDS, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipEventRecord(c->done[c->slot], st));
    return 0;
}

/* Wait for the batch whose tail went into the current set, and read its verdict and statistics there. */
static int complete_batch(freesasa_gpu_ctx *c, int n, int n_structs, int tile_atoms, int block_threads, int lds)
{
    HIP_TRY(c, hipEventSynchronize(c->done[c->slot]));
    const int *status_h = ctx_status_h(c);
    long long total_cells = 0;
    const int rcs = judge_status(c, status_h, &total_cells);
    freesasa_gpu_stats &S = c->stats;
    S.n_atoms = n; S.n_cells = total_cells; S.n_structs = n_structs;
    S.max_neighbors = status_h[ST_MAX_NN]; S.fallback_tiles = status_h[ST_OVF_TILES];
    for (int k = 0; k < 64; ++k) S.fallback_tiles += status_h[ST_SPLIT + k]; /* (L&R: tiles redone as halves) */
    S.tile_atoms = tile_atoms; S.block_threads = block_threads; S.lds_bytes = lds;
    S.ms_prep = S.ms_kernel = S.ms_total = 0;
    if (c->timing) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx_ev(c)[0], ctx_ev(c)[1]) == hipSuccess) S.ms_prep = ms;
        if (hipEventElapsedTime(&ms, ctx_ev(c)[1], ctx_ev(c)[2]) == hipSuccess) S.ms_kernel = ms;
        if (hipEventElapsedTime(&ms, ctx_ev(c)[0], ctx_ev(c)[3]) == hipSuccess) S.ms_total = ms;
    }
    if (rcs) return rcs;
    if (total_cells + total_cells / 32 > c->cells_hint) c->cells_hint = total_cells + total_cells / 32;
    dump_phase_clocks();
    return 0;
}

static int finish_batch(freesasa_gpu_ctx *c, const PipeArgs &pa, int n, int n_structs, double *d_sasa,
                        double *d_totals, int tile_atoms, int block_threads, int lds, int *)
{
    if (enqueue_tail(c, pa, n_structs, d_sasa, d_totals)) return -1;
    return complete_batch(c, n, n_structs, tile_atoms, block_threads, lds);
}

/* what a completed Lee-Richards batch teaches the context about the next one of its kind (trajectory frames, sweeps) */
static void lr2_learn(freesasa_gpu_ctx *c, const int *status_h, int TA, int ns, int mw, int ds)
{
    const int learnt = lr2_need_from_hist(status_h + ST_HIST, TA);
    if (learnt > 0) c->hint_nn = (double)(learnt - 8) / TA;
    c->hint_pool2 = lr2_pool_from_hist(status_h + ST_HIST, TA, ns, mw, ds, &c->hint_split2); /* (see there: pool vs occupancy) */
    c->hint_ta2 = TA; c->hint_mw2 = mw;
    c->hint_nn_max = status_h[ST_MAX_NN] + 4; /* the longest list of this batch, a little room */
}

/* defer: enqueue only (freesasa_gpu_lr_batch_dev_async); the caller completes the batch later (complete_pending) */
static int run_lr2(freesasa_gpu_ctx *c, const PipeArgs &pa, int n, int n_structs, int resolution, double *d_sasa,
                   double *d_totals, bool defer)
{
    hipStream_t st = c->stream;
    int *status_h = ctx_status_h(c);
    int ta_env = 0, pool_env = 0, ds_env = -1, refill_env = 0;
    if (const char *e = getenv("FREESASA_AMD_LR2")) (void)sscanf(e, "%d,%d,%d,%d", &ta_env, &pool_env, &ds_env, &refill_env); /* tuning aid: "TA,pool,ds,refill" */
    if (c->hint_res[0] != resolution) {
        /* no demand history for this resolution on this context: size the neighbor pool from the local
           density (atoms in an atom's own cell; ~3.1 neighbors per such atom on coils, globules and
           proteins alike).  One 8-byte readback, first call only. */
        long long total_cells = 0;
        const int rcs = collect_status(c, n_structs, 8, &total_cells);
        if (rcs) return rcs;
        c->hint_nn = status_h[ST_OCC_N] > 0 ? 1.25 * 3.1 * (double)status_h[ST_OCC_SUM] / (double)status_h[ST_OCC_N] + 2.0 : 0.0;
        c->hint_nn_max = (int)(1.45 * c->hint_nn); /* longest list ~ 1.6 x the mean on coils, globules and proteins alike */
        c->hint_res[0] = resolution;
        c->hint_pool2 = 0;
    }
    Lr2Cfg cfg = lr2_choose_cfg(resolution, c->hint_nn, ta_env, c->hint_nn_max, c->hint_pool2 > 0 ? c->hint_ta2 : 0, c->hint_split2);
    if (c->hint_pool2 > 0 && c->hint_ta2 == cfg.TA && c->hint_mw2 == cfg.mw) { /* same tile shape as the last batch: its demand histogram decides */
        cfg.pool = c->hint_pool2;
        cfg.rmax = (cfg.pool + LR2_LANES - 1) / LR2_LANES;
    }
    if (pool_env > 0) { cfg.pool = (pool_env + 1) & ~1; cfg.rmax = (cfg.pool + LR2_LANES - 1) / LR2_LANES; }
    if (ds_env >= 0) cfg.ds = ds_env;
    if (refill_env > 0) cfg.refill = refill_env;
    cfg.lds = lr2_layout(cfg.TA, cfg.ns, cfg.pool, cfg.mw, cfg.ds).total;
    if (getenv("FREESASA_AMD_SHOW_SHAPE")) fprintf(stderr, "lr2 shape: TA %d ns %d pool %d mw %d ds %d refill %d rmax %d lds %d cover %d\n", cfg.TA, cfg.ns, cfg.pool, cfg.mw, cfg.ds, cfg.refill, cfg.rmax, cfg.lds, c->hint_nn >= 1.35 * LR2_COVER_DENSITY ? 1 : 0); /* (dev aid) */
    const int n_tiles = (n + cfg.TA - 1) / cfg.TA;
    if (ensure(c, c->ovf_tiles, sizeof(long long) * (2 * (size_t)n_tiles + 2)) || ensure(c, c->ovf_atoms, sizeof(int) * ((size_t)n + 8)))
        return -1;

    Lr2Args la;
    memset(&la, 0, sizeof la);
    la.sq = pa.sq;
    la.s_idx = pa.s_idx;
    la.grid = pa.grid; la.cell_start = pa.cell_start; la.cell_tbl = pa.cell_tbl; la.cell_first = pa.cell_first;
    la.n_atoms = n; la.n_tiles = n_tiles; la.TA = cfg.TA; la.ns = resolution;
    la.pool = cfg.pool; la.mw = cfg.mw; la.ds = cfg.ds; la.refill = cfg.refill;
    /* (hint_nn: what all but 4 % of the last batch's tiles needed per atom, or 1.25 x the density sample: coils ~30, proteins ~58) */
    la.cover = c->hint_nn >= 1.35 * LR2_COVER_DENSITY ? LR2_COVER_DENSITY : 0;
    if (const char *e = getenv("FREESASA_AMD_COVER")) la.cover = atoi(e); /* tuning aid: neighbor records per atom from which a tile runs the cover filter; 0: never */
    la.sasa = d_sasa; la.status = (int *)c->status.p;
    la.inv_ns = 1.0 / (double)resolution;
    la.ovf_items = (long long *)c->ovf_tiles.p;
    la.ovf_count = (int *)c->status.p + ST_OVF2_TILES;
    la.split_count = (int *)c->status.p + ST_SPLIT;

    const int grid_all = ((n_tiles + 7) / 8) * 8;
    int grid_main = grid_all > 147456 ? 147456 : grid_all;
    if (const char *e = getenv("FREESASA_AMD_GRID")) { /* tuning aid */
        const int g = atoi(e);
        if (g >= 8) grid_main = g < grid_all ? (g / 8) * 8 : grid_all;
    }
    /* the last quarter of a full-size launch's workgroups takes two tiles each, the others share the rest (Lr2Args::seg_grid) */
    la.seg_grid = 0; la.seg_tiles = 0;
    if (grid_main == 147456 && !getenv("FREESASA_AMD_ONE_PART")) {
        const int g2 = grid_main / 4, t2 = 2 * g2;
        if (n_tiles >= 8 * t2) { la.seg_grid = grid_main - g2; la.seg_tiles = ((n_tiles - t2) / 8) * 8; }
    }
    la.nn_out = c->dbg_nn; la.nb_out = c->dbg_nb; la.nb_cap = c->dbg_cap;
    la.hooks = (c->dbg_nn ? 1 : 0) | (c->dbg_nb ? 2 : 0);
    hipError_t le = launch_lr2_main(cfg.rmax, grid_main, (size_t)cfg.lds, st, la);
    if (le != hipSuccess) return ctx_fail(c, "tile kernel launch failed: %s", hipGetErrorString(le));
    if (c->timing) HIP_TRY(c, hipEventRecord(ctx_ev(c)[2], st));
    if (c->dbg_nn) return finish_batch(c, pa, n, n_structs, d_sasa, nullptr, cfg.TA, 64, cfg.lds, status_h);
    /* second launch: halves that did not fit either: larger LDS lists, more registers */
    const Lr2Cfg cm = lr2_mid_cfg(cfg);
    {
        Lr2Args lm = la;
        lm.pool = cm.pool; lm.mw = cm.mw; lm.ds = cm.ds;
        if (!getenv("FREESASA_AMD_COVER")) lm.cover = LR2_COVER_DENSITY;
        lm.work_items = (const long long *)c->ovf_tiles.p;
        lm.work_count = (const int *)c->status.p + ST_OVF2_TILES;
        lm.ovf_items = nullptr;
        lm.ovf_atoms = (int *)c->ovf_atoms.p;
        lm.ovf_count = (int *)c->status.p + ST_OVF3_ATOMS;
        lm.split_count = nullptr;
        const int grid_mid = n_tiles < SASA_MID_BLOCKS ? n_tiles : SASA_MID_BLOCKS;
        hipLaunchKernelGGL((k_lr2_tile<LR2_RMAX_MID, 2, 3, true>), dim3(grid_mid), dim3(64), (size_t)cm.lds, st, lm);
        le = hipGetLastError();
        if (le != hipSuccess) return ctx_fail(c, "second tile launch failed: %s", hipGetErrorString(le));
    }
    /* last launch: whatever is left (pathological densities), atom by atom: the first-generation kernel with its
       lists in a global slab */
    {
        const TileCfg fb = fallback_cfg(lr_slab_cfg(1, resolution), true);
        const size_t stride = tile_slab_bytes(fb.TA, fb.cap_idx, fb.pool, fb.lr, fb.ds, fb.B);
        if (ensure(c, c->slab, stride * SASA_FB_BLOCKS)) return -1;
        TileArgs tf;
        memset(&tf, 0, sizeof tf);
        tf.sq = pa.sq;
        tf.s_idx = pa.s_idx;
        tf.grid = pa.grid; tf.cell_start = pa.cell_start; tf.cell_tbl = pa.cell_tbl; tf.cell_first = pa.cell_first;
        tf.n_atoms = n; tf.n_tiles = n; tf.TA = 1; tf.n_res = resolution; tf.tab = fb.tab;
        tf.sasa = d_sasa; tf.lr = 1; tf.status = (int *)c->status.p;
        tf.cap_idx = fb.cap_idx; tf.pool = fb.pool; tf.ds = fb.ds;
        tf.work_tiles = (const int *)c->ovf_atoms.p;
        tf.work_count = (const int *)c->status.p + ST_OVF3_ATOMS;
        tf.slab = (char *)c->slab.p;
        tf.slab_stride = (long long)stride;
        le = launch_lr<true, 2>(fb, tf, SASA_FB_BLOCKS, fb.lds, st);
        if (le != hipSuccess) return ctx_fail(c, "fallback kernel launch failed: %s", hipGetErrorString(le));
    }
    if (enqueue_tail(c, pa, n_structs, d_sasa, d_totals)) return -1;
    freesasa_gpu_ctx::Pend &P = c->pend[c->slot];
    P.n = n; P.TA = cfg.TA; P.mw = cfg.mw; P.ds = cfg.ds; P.lds = cfg.lds; P.n_structs = n_structs; P.resolution = resolution;
    if (defer) return 0;
    const int rc = complete_batch(c, n, n_structs, cfg.TA, 64, cfg.lds);
    if (rc) return rc;
    lr2_learn(c, status_h, cfg.TA, cfg.ns, cfg.mw, cfg.ds);
    return 0;
}

/* ------------------------------------------------------------------ one batch */

static int run_batch_once(freesasa_gpu_ctx *c, bool lr, const double *d_xyz, const double *d_radii,
                     const int64_t *offsets, int n_structs, double probe, int resolution,
                     const double *unit_points, double *d_sasa, int *d_counts, double *d_totals, bool defer = false, bool *deferred = nullptr)
{
    c->err[0] = 0;
    if (deferred) *deferred = false;
    if (!d_xyz || !d_radii || !offsets || !d_sasa) return ctx_fail(c, "null argument");
    if (n_structs <= 0) return ctx_fail(c, "n_structs must be > 0");
    if (resolution <= 0) return ctx_fail(c, "resolution must be > 0");
    if (offsets[0] != 0) return ctx_fail(c, "offsets[0] must be 0");
    for (int s = 0; s < n_structs; ++s)
        if (offsets[s + 1] < offsets[s]) return ctx_fail(c, "offsets must be non-decreasing");
    const int64_t n64 = offsets[n_structs];
    if (n64 <= 0) return ctx_fail(c, "empty batch");
    if (n64 > (int64_t)1 << 30) return ctx_fail(c, "batch too large (max 2^30 atoms per call)");
    const int n = (int)n64;

    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t nb = (size_t)n;

    /* workspace */
    if (ensure(c, c->offsets, sizeof(int64_t) * ((size_t)n_structs + 1)) || ensure(c, c->grid, sizeof(GridS) * (size_t)n_structs) ||
        ensure(c, c->ncells, sizeof(long long) * ((size_t)n_structs + 1)) || ensure(c, c->sid, 4 * nb) ||
        ensure(c, c->cell_of, 8 * nb) || ensure(c, c->rank, 4 * nb) || ensure(c, c->sq, 32 * nb) ||
        ensure(c, c->s_idx, 16 * nb) ||
        ensure(c, c->status, sizeof(int) * ST_WORDS))
        return -1;

    /* offsets and the chunk table derived from them: upload only when they changed
       (trajectory frames and repeated batches reuse them) */
    if ((int)c->offsets_host.size() != n_structs + 1 ||
        memcmp(c->offsets_host.data(), offsets, sizeof(int64_t) * ((size_t)n_structs + 1)) != 0) {
        if (c->pend[0].active || c->pend[1].active) HIP_TRY(c, hipStreamSynchronize(st)); /* (batches in flight read the tables about to be overwritten; they are collected later, from their own sets of host words: see ensure) */
        c->offsets_host.clear(); /* (set again below, once the tables derived from it are on the device) */
        std::vector<int> cs, cl, sc0((size_t)n_structs + 1);
        std::vector<int64_t> cb;
        for (int s = 0; s < n_structs; ++s) {
            sc0[s] = (int)cs.size();
            for (int64_t b = offsets[s]; b < offsets[s + 1]; b += SASA_BOUNDS_CHUNK) {
                const int64_t e = b + SASA_BOUNDS_CHUNK < offsets[s + 1] ? b + SASA_BOUNDS_CHUNK : offsets[s + 1];
                cs.push_back(s); cb.push_back(b); cl.push_back((int)(e - b));
            }
        }
        sc0[n_structs] = (int)cs.size();
        c->n_chunks = (int)cs.size();
        const size_t nc = cs.size();
        if (ensure(c, c->chunk_struct, 4 * nc) || ensure(c, c->chunk_begin, 8 * nc) || ensure(c, c->chunk_len, 4 * nc) ||
            ensure(c, c->struct_chunk0, 4 * ((size_t)n_structs + 1)) || ensure(c, c->bpart, 56 * nc))
            return -1;
        HIP_TRY(c, hipMemcpy(c->offsets.p, offsets, sizeof(int64_t) * ((size_t)n_structs + 1), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->chunk_struct.p, cs.data(), 4 * nc, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->chunk_begin.p, cb.data(), 8 * nc, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->chunk_len.p, cl.data(), 4 * nc, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->struct_chunk0.p, sc0.data(), 4 * ((size_t)n_structs + 1), hipMemcpyHostToDevice));
        c->offsets_host.assign(offsets, offsets + n_structs + 1);
    }
    HIP_TRY(c, hipMemsetAsync(c->status.p, 0, sizeof(int) * ST_WORDS, st));
    if (c->timing) HIP_TRY(c, hipEventRecord(ctx_ev(c)[0], st));

    PipeArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.xyz = d_xyz; pa.radii = d_radii; pa.offsets = (const int64_t *)c->offsets.p;
    pa.n_structs = n_structs; pa.n_atoms = n; pa.probe = probe; pa.max_cells = c->max_cells;
    pa.shared_radii = c->shared_radii ? 1 : 0;
    pa.n_chunks = c->n_chunks; pa.chunk_struct = (const int *)c->chunk_struct.p; pa.chunk_begin = (const int64_t *)c->chunk_begin.p;
    pa.chunk_len = (const int *)c->chunk_len.p; pa.struct_chunk0 = (const int *)c->struct_chunk0.p; pa.bpart = (double *)c->bpart.p;
    pa.grid = (GridS *)c->grid.p; pa.ncells = (long long *)c->ncells.p;
    pa.sid = (int *)c->sid.p; pa.cell_of = (long long *)c->cell_of.p; pa.rank = (int *)c->rank.p;
    pa.sq = (Quad *)c->sq.p;
    pa.s_idx = (SortIdx *)c->s_idx.p;
    pa.status = (int *)c->status.p;
    if (probe != c->hint_probe) { /* launch-shape history is per (resolution, probe radius) */
        c->hint_res[0] = c->hint_res[1] = 0;
        c->hint_pool2 = 0;
        c->hint_probe = probe;
    }
    pa.occ_stride = c->hint_res[lr ? 0 : 1] == resolution ? 0 : (n / 256 > 0 ? n / 256 : 1);

    /* The cell table is sized WITHOUT waiting for K2's total: for what the context has seen so far, and for a
       first batch 10 cells per atom (sparse random coils need 9, proteins 1-2) plus 256 per structure.  K2 checks
       the real total against it on the device (ST_RETRY, see PIPE_GATE); the total itself reaches the host with
       the status words at the end of the batch. */
    int *status_h = ctx_status_h(c);
    long long cells_cap = 10LL * n + 256LL * n_structs; /* (0.4 GB for 1e7 atoms; a sparser batch is redone once with K2's size) */
    if (cells_cap < c->cells_hint) cells_cap = c->cells_hint;
    if (cells_cap > c->max_cells) cells_cap = c->max_cells;
    const int nblk_scan = (int)((cells_cap + 1 + (long long)SASA_PIPE_B * SASA_SCAN_ITEMS - 1) / ((long long)SASA_PIPE_B * SASA_SCAN_ITEMS));
    const int nblk_atoms = (n + SASA_PIPE_B - 1) / SASA_PIPE_B;
    /* batches of small structures: bounds, grid and cell sort of a structure in one workgroup (k_sort_struct) */
    long long biggest = 0;
    for (int s_ = 0; s_ < n_structs; ++s_) biggest = offsets[s_ + 1] - offsets[s_] > biggest ? offsets[s_ + 1] - offsets[s_] : biggest;
    const bool fused = c->sort_fused && biggest <= SORT_ATOMS && n_structs >= 8 && !getenv("FREESASA_AMD_NO_FUSED_SORT");
    /* ... which writes the cell table in its compact form for the Lee-Richards tile kernel (PipeArgs::cell_tbl) */
    const bool compact = fused && lr && lr2_supported(resolution) && !getenv("FREESASA_AMD_LR1") && !getenv("FREESASA_AMD_DENSE_CELLS");
    if (compact) {
        if (ensure(c, c->cell_tbl, sizeof(unsigned long long) * ((size_t)(cells_cap >> 5) + 2)) || ensure(c, c->cell_first, sizeof(int) * (nb + (size_t)n_structs + 2)))
            return -1;
        pa.cell_tbl = (unsigned long long *)c->cell_tbl.p;
        pa.cell_first = (int *)c->cell_first.p;
    } else if (ensure(c, c->cell_start, sizeof(int) * ((size_t)cells_cap + 2)) || ensure(c, c->blk_sums, sizeof(int) * ((size_t)nblk_scan + 1))) {
        return -1;
    }
    pa.cell_start = (int *)c->cell_start.p;
    pa.blk_sums = (int *)c->blk_sums.p;
    pa.cells_cap = cells_cap;

    if (fused) {
        HIP_TRY(c, hipMemsetAsync((long long *)c->ncells.p + n_structs, 0, sizeof(long long), st)); /* the cell counter */
        hipLaunchKernelGGL(k_sort_struct, dim3(n_structs), dim3(SORT_B), 0, st, pa);
    } else {
    hipLaunchKernelGGL(k_bounds, dim3(c->n_chunks), dim3(SASA_PIPE_B), 0, st, pa);
    hipLaunchKernelGGL(k_grid, dim3((n_structs + 63) / 64), dim3(64), 0, st, pa);
    hipLaunchKernelGGL(k_cell_base, dim3(1), dim3(SASA_PIPE_B), 0, st, pa);
    hipLaunchKernelGGL(k_zero_cells, dim3((unsigned)((cells_cap + 2 + 16LL * SASA_PIPE_B - 1) / (16LL * SASA_PIPE_B))), dim3(SASA_PIPE_B), 0, st, pa);
    hipLaunchKernelGGL(k_count, dim3(nblk_atoms), dim3(SASA_PIPE_B), 0, st, pa);
    hipLaunchKernelGGL(k_scan1, dim3(nblk_scan), dim3(SASA_PIPE_B), 0, st, pa);
    hipLaunchKernelGGL(k_scan2, dim3(1), dim3(SASA_PIPE_B), 0, st, pa);
    hipLaunchKernelGGL(k_scan3, dim3(nblk_scan), dim3(SASA_PIPE_B), 0, st, pa);
    hipLaunchKernelGGL(k_scatter, dim3(nblk_atoms), dim3(SASA_PIPE_B), 0, st, pa);
    }
    HIP_TRY(c, hipGetLastError());
    if (c->timing) HIP_TRY(c, hipEventRecord(ctx_ev(c)[1], st));

    /* Lee & Richards at ordinary resolutions: the second-generation kernel (lr2_kernels.h) */
    if (lr && lr2_supported(resolution) && !getenv("FREESASA_AMD_LR1")) {
        const int rc2 = run_lr2(c, pa, n, n_structs, resolution, d_sasa, d_totals, defer && !c->dbg_nn);
        if (rc2 == 0 && defer && !c->dbg_nn && deferred) *deferred = true;
        return rc2;
    }

    /* fused tile kernel */
    const int hi = lr ? 0 : 1;
    if (c->hint_res[hi] != resolution) {
        /* no demand history for this resolution on this context: estimate the neighbor count from the
           local density (atoms in an atom's own cell; ~3.1 neighbors per such atom on coils, globules
           and proteins alike) so that the very first launch already has a fitting neighbor pool and,
           for dense inputs, the bucket-sort variant.  One 8-byte readback, first call only. */
        long long total_cells = 0;
        const int rcs = collect_status(c, n_structs, 8, &total_cells);
        if (rcs) return rcs;
        if (status_h[ST_OCC_N] > 0) {
            const double nn_est = 3.1 * (double)status_h[ST_OCC_SUM] / (double)status_h[ST_OCC_N];
            const TileCfg probe_cfg = choose_cfg(resolution, lr, 0);
            int pool = (int)(1.35 * nn_est * probe_cfg.TA + 16.0);
            pool = (pool + 1) & ~1;
            if (pool < 32) pool = 32;
            if (pool > 4096) pool = 4096;
            c->hint_res[hi] = resolution;
            c->hint_ta[hi] = probe_cfg.TA;
            c->hint_pool[hi] = pool;
            if (lr) c->hint_bucket = nn_est > 30.0;
        }
    }
    TileCfg cfg = choose_cfg(resolution, lr, c->hint_res[hi] == resolution ? c->hint_pool[hi] : 0);
    if (const char *e = getenv("FREESASA_AMD_CFG")) { /* tuning aid: "B,TA,pool,ds" */
        int b = 0, t = 0, pl = 0, d = 0;
        if (sscanf(e, "%d,%d,%d,%d", &b, &t, &pl, &d) == 4 && (b == 64 || b == 128 || b == 256 || b == 320) && t >= 1 && t <= b &&
            (!lr || !cfg.tab || t * resolution <= 4096)) {
            cfg.B = b; cfg.TA = t; cfg.pool = pl; cfg.ds = lr ? d : 0;
            cfg.items = lr ? (cfg.tab ? cfg.TA * resolution : cfg.B) : 1;
            cfg.lds = tile_fixed_bytes(cfg.TA, cfg.items) + tile_list_bytes(cfg.TA, cfg.cap_idx, cfg.pool, cfg.lr, cfg.ds, cfg.B);
        }
    }
    /* whatever the estimate or the tuning aid asked for: the main launch's lists must fit the CU's LDS (dense tiles
       then go to the later launches instead of failing the launch) */
    while (cfg.lds > 160 * 1024 && cfg.pool > 32) {
        cfg.pool = (cfg.pool * 3 / 4) & ~1;
        cfg.lds = tile_fixed_bytes(cfg.TA, cfg.items) + tile_list_bytes(cfg.TA, cfg.cap_idx, cfg.pool, cfg.lr, cfg.ds, cfg.B);
    }
    const int n_tiles = (n + cfg.TA - 1) / cfg.TA;
    if (ensure(c, c->ovf_tiles, sizeof(int) * ((size_t)n_tiles + 1)) || ensure(c, c->ovf_tiles2, sizeof(int) * ((size_t)n_tiles + 1))) return -1;

    TileArgs ta;
    memset(&ta, 0, sizeof ta);
    ta.sq = pa.sq;
    ta.s_idx = pa.s_idx;
    ta.grid = pa.grid; ta.cell_start = pa.cell_start;
    ta.n_atoms = n; ta.n_tiles = n_tiles; ta.TA = cfg.TA; ta.n_res = resolution; ta.tab = cfg.tab;
    ta.sasa = d_sasa; ta.counts = d_counts;
    ta.cap_idx = cfg.cap_idx; ta.pool = cfg.pool; ta.lr = cfg.lr; ta.ds = cfg.ds;
    ta.ovf_count = (int *)c->status.p + ST_OVF_TILES;
    ta.ovf_tiles = (int *)c->ovf_tiles.p;
    ta.work_tiles = nullptr;
    ta.work_count = nullptr;
    ta.status = (int *)c->status.p;
    if (!lr) {
        if (!unit_points) return ctx_fail(c, "unit_points is null");
        if (ensure(c, c->unit_pts, sizeof(double) * 3 * (size_t)resolution)) return -1;
        /* pageable host-to-device copies stall the stream: upload the points only when they change */
        if (c->unit_host.size() != 3 * (size_t)resolution ||
            memcmp(c->unit_host.data(), unit_points, sizeof(double) * 3 * (size_t)resolution) != 0) {
            c->unit_host.assign(unit_points, unit_points + 3 * (size_t)resolution);
            if (hipMemcpyAsync(c->unit_pts.p, c->unit_host.data(), sizeof(double) * 3 * (size_t)resolution, hipMemcpyHostToDevice, st) != hipSuccess) {
                c->unit_host.clear();
                return ctx_fail(c, "upload of the test points failed");
            }
        }
        ta.unit_pts = (const double *)c->unit_pts.p;
    }

    /* workgroups loop over tiles (w, w + grid, ...): ~150k workgroups measured ~2% better than
       one workgroup per tile, a grid of exactly the resident workgroups 25% worse (tiles vary) */
    int grid_main = ((n_tiles + 7) / 8) * 8;
    if (grid_main > 147456) grid_main = 147456;
    if (const char *e = getenv("FREESASA_AMD_GRID")) { /* tuning aid */
        const int g = atoi(e);
        if (g >= 8 && g < grid_main) grid_main = (g / 8) * 8;
    }
    hipError_t le;
    {
        static std::once_flag attr_once; /* allow > 64 KB of dynamic LDS */
        std::call_once(attr_once, [] {
            const void *fns[] = {(const void *)k_lr_tile<320, false, 0, 4>, (const void *)k_lr_tile<256, false, 0, 4>,
                                 (const void *)k_lr_tile<128, false, 0, 4>, (const void *)k_lr_tile<64, false, 0, 4>,
                                 (const void *)k_sr_tile<320, false, 0>, (const void *)k_sr_tile<256, false, 0>,
                                 (const void *)k_sr_tile<128, false, 0>, (const void *)k_sr_tile<64, false, 0>,
                                 (const void *)k_lr_tile<320, false, 1, 4>, (const void *)k_lr_tile<256, false, 1, 4>,
                                 (const void *)k_lr_tile<128, false, 1, 4>, (const void *)k_lr_tile<64, false, 1, 4>,
                                 (const void *)k_lr_tile<64, false, 0, 4, true>, (const void *)k_lr_tile<64, false, 1, 4, true>,
                                 (const void *)k_sr_tile<320, false, 1>, (const void *)k_sr_tile<256, false, 1>,
                                 (const void *)k_sr_tile<128, false, 1>, (const void *)k_sr_tile<64, false, 1>};
            for (const void *fn : fns) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
    }
    const bool bucket = lr && c->hint_bucket && c->hint_res[0] == resolution;
    le = lr ? launch_lr<false, 0>(cfg, ta, grid_main, cfg.lds, st, bucket) : launch_sr<false, 0>(cfg, ta, grid_main, cfg.lds, st);
    if (le != hipSuccess) return ctx_fail(c, "tile kernel launch failed: %s", hipGetErrorString(le));
    if (c->timing) HIP_TRY(c, hipEventRecord(ctx_ev(c)[2], st));

    /* second launch: the tiles whose lists did not fit the small LDS capacities (the blocks read
       the count on the device; normally a fraction of a percent of the tiles) */
    {
        const TileCfg mc = mid_cfg(cfg, lr);
        TileArgs tm = ta;
        tm.cap_idx = mc.cap_idx; tm.pool = mc.pool; tm.ds = mc.ds;
        tm.work_tiles = (const int *)c->ovf_tiles.p;
        tm.work_count = (const int *)c->status.p + ST_OVF_TILES;
        tm.ovf_tiles = (int *)c->ovf_tiles2.p;
        tm.ovf_count = (int *)c->status.p + ST_OVF2_TILES;
        const int grid_mid = n_tiles < SASA_MID_BLOCKS ? n_tiles : SASA_MID_BLOCKS;
        le = lr ? launch_lr<false, 1>(mc, tm, grid_mid, mc.lds, st, bucket) : launch_sr<false, 1>(mc, tm, grid_mid, mc.lds, st);
        if (le != hipSuccess) return ctx_fail(c, "second tile launch failed: %s", hipGetErrorString(le));
    }
    /* third launch: whatever is left (pathological densities), lists in a global slab */
    {
        const TileCfg fb = fallback_cfg(cfg, lr);
        const size_t stride = tile_slab_bytes(fb.TA, fb.cap_idx, fb.pool, fb.lr, fb.ds, fb.B);
        if (ensure(c, c->slab, stride * SASA_FB_BLOCKS)) return -1;
        TileArgs tf = ta;
        tf.cap_idx = fb.cap_idx; tf.pool = fb.pool; tf.ds = fb.ds;
        tf.work_tiles = (const int *)c->ovf_tiles2.p;
        tf.work_count = (const int *)c->status.p + ST_OVF2_TILES;
        tf.ovf_tiles = nullptr;
        tf.ovf_count = nullptr;
        tf.slab = (char *)c->slab.p;
        tf.slab_stride = (long long)stride;
        le = lr ? launch_lr<true, 2>(fb, tf, SASA_FB_BLOCKS, fb.lds, st) : launch_sr<true, 2>(fb, tf, SASA_FB_BLOCKS, fb.lds, st);
        if (le != hipSuccess) return ctx_fail(c, "fallback kernel launch failed: %s", hipGetErrorString(le));
    }

    if (d_totals) {
        /* the chunk partials reuse the bounds kernels' scratch (56 bytes per chunk, free by now) */
        hipLaunchKernelGGL(k_totals_chunks, dim3(c->n_chunks), dim3(SASA_TOT_B), 0, st, pa, (const double *)d_sasa, (double *)c->bpart.p);
        hipLaunchKernelGGL(k_totals_structs, dim3((n_structs + 255) / 256), dim3(256), 0, st, pa, (const double *)c->bpart.p, d_totals);
        HIP_TRY(c, hipGetLastError());
    }
    if (c->timing) HIP_TRY(c, hipEventRecord(ctx_ev(c)[3], st));

    long long total_cells = 0;
    const int rcs = collect_status(c, n_structs, ST_WORDS, &total_cells);

    freesasa_gpu_stats &S = c->stats;
    S.n_atoms = n; S.n_cells = total_cells; S.n_structs = n_structs;
    S.max_neighbors = status_h[ST_MAX_NN]; S.fallback_tiles = status_h[ST_OVF_TILES];
    S.tile_atoms = cfg.TA; S.block_threads = cfg.B; S.lds_bytes = (int)cfg.lds;
    S.ms_prep = S.ms_kernel = S.ms_total = 0;
    if (c->timing) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx_ev(c)[0], ctx_ev(c)[1]) == hipSuccess) S.ms_prep = ms;
        if (hipEventElapsedTime(&ms, ctx_ev(c)[1], ctx_ev(c)[2]) == hipSuccess) S.ms_kernel = ms;
        if (hipEventElapsedTime(&ms, ctx_ev(c)[0], ctx_ev(c)[3]) == hipSuccess) S.ms_total = ms;
    }
    if (rcs) return rcs;
    if (total_cells + total_cells / 32 > c->cells_hint) c->cells_hint = total_cells + total_cells / 32;
    dump_phase_clocks();
    /* learn the pool size for the next batch of this kind (trajectory frames, sweeps) */
    c->hint_res[hi] = resolution;
    c->hint_ta[hi] = cfg.TA;
    c->hint_pool[hi] = pool_from_hist(status_h + ST_HIST, cfg.TA);
    if (lr) c->hint_bucket = mean_from_hist(status_h + ST_HIST, cfg.TA) > 30.0 * cfg.TA;
    return 0;
}

static int run_batch_impl(freesasa_gpu_ctx *c, bool lr, const double *d_xyz, const double *d_radii,
                     const int64_t *offsets, int n_structs, double probe, int resolution,
                     const double *unit_points, double *d_sasa, int *d_counts, double *d_totals)
{
    int rc = run_batch_once(c, lr, d_xyz, d_radii, offsets, n_structs, probe, resolution, unit_points, d_sasa, d_counts, d_totals);
    /* once more when the cell table was too small (first batch of a very sparse kind; now with K2's size), and once more
       when a structure did not fit the per-structure cell sort (the general pipeline from then on) */
    for (int again = 0; rc == RC_RETRY && again < 3; ++again)
        rc = run_batch_once(c, lr, d_xyz, d_radii, offsets, n_structs, probe, resolution, unit_points, d_sasa, d_counts, d_totals);
    if (rc == RC_RETRY) return ctx_fail(c, "cell table sizing did not converge");
    return rc;
}

/* On failure nothing may still be running on the stream when the caller gets control back (it
 * is entitled to free its buffers right away). */
static int run_batch(freesasa_gpu_ctx *c, bool lr, const double *d_xyz, const double *d_radii,
                     const int64_t *offsets, int n_structs, double probe, int resolution,
                     const double *unit_points, double *d_sasa, int *d_counts, double *d_totals)
{
    const int rc = run_batch_impl(c, lr, d_xyz, d_radii, offsets, n_structs, probe, resolution, unit_points,
                                  d_sasa, d_counts, d_totals);
    if (rc) (void)hipStreamSynchronize(c->stream);
    return rc;
}

/* ------------------------------------------------------------------ asynchronous batches
 * freesasa_gpu_lr_batch_dev_async enqueues a batch and returns; up to two may be in flight on a context.  Each leaves
 * its host-side set (status words, events) behind; complete_pending waits for its end-of-batch event and reads the
 * verdict there.  The rare batch whose cell table was too small (RC_RETRY) is redone synchronously, with everything
 * else on the stream drained first. */
static int complete_pending(freesasa_gpu_ctx *c, int slot)
{
    freesasa_gpu_ctx::Pend &P = c->pend[slot];
    if (!P.active) return 0;
    const int keep = c->slot;
    c->slot = slot;
    int rc = complete_batch(c, P.n, P.n_structs, P.TA, 64, P.lds);
    /* (what it teaches is keyed to the resolution and probe the hints stand for NOW: a later batch of another kind may
       have reset them since this one was enqueued) */
    if (rc == 0 && P.resolution == c->hint_res[0] && P.probe == c->hint_probe) lr2_learn(c, ctx_status_h(c), P.TA, P.resolution, P.mw, P.ds);
    c->slot = keep;
    P.active = false;
    if (rc == RC_RETRY) {
        /* redo it the synchronous way (which sizes the table from this pass's count); the other batch in flight, if
           any, ran into the same table and is redone when it is collected */
        (void)hipStreamSynchronize(c->stream);
        std::vector<int64_t> offs;
        offs.swap(P.offsets);
        rc = run_batch(c, true, P.d_xyz, P.d_radii, offs.data(), P.n_structs, P.probe, P.resolution, nullptr, P.d_sasa, nullptr, P.d_totals);
    }
    return rc;
}
/* collect every batch in flight, oldest first; -1 if any failed (the context's error text is the last failure's) */
static int drain_pending(freesasa_gpu_ctx *c)
{
    int rc = 0;
    for (int k = 0; k < 2; ++k) {
        const int slot = (c->slot + k) & 1; /* c->slot is the set the next batch takes: the older one of two in flight */
        if (c->pend[slot].active && complete_pending(c, slot)) rc = -1;
    }
    return rc;
}

extern "C" int freesasa_gpu_wait(freesasa_gpu_ctx *c)
{
    if (!c) return -1;
    if (hipSetDevice(c->device) != hipSuccess) return ctx_fail(c, "hipSetDevice failed");
    return drain_pending(c);
}

extern "C" int freesasa_gpu_lr_batch_dev_async(freesasa_gpu_ctx *c, const double *d_xyz, const double *d_radii,
                                               const int64_t *offsets, int n_structs, double probe, int n_slices,
                                               double *d_sasa, double *d_totals)
{
    if (!c) return -1;
    if (hipSetDevice(c->device) != hipSuccess) return ctx_fail(c, "hipSetDevice failed");
    /* the set this batch takes may still belong to the batch submitted two calls ago: collect that one first */
    if (c->pend[c->slot].active && complete_pending(c, c->slot)) return -1;
    bool deferred = false;
    int rc = run_batch_once(c, true, d_xyz, d_radii, offsets, n_structs, probe, n_slices, nullptr, d_sasa, nullptr, d_totals, true, &deferred);
    if (rc == RC_RETRY || (rc == 0 && !deferred)) {
        /* a first batch of its kind (its density is read back before the tile kernel is shaped) that has to be
           redone, or a path without a deferred tail: finish it the synchronous way */
        if (rc == RC_RETRY) rc = run_batch(c, true, d_xyz, d_radii, offsets, n_structs, probe, n_slices, nullptr, d_sasa, nullptr, d_totals);
        return rc;
    }
    if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
    freesasa_gpu_ctx::Pend &P = c->pend[c->slot];
    P.active = true;
    P.d_xyz = d_xyz; P.d_radii = d_radii; P.offsets.assign(offsets, offsets + n_structs + 1);
    P.probe = probe; P.d_sasa = d_sasa; P.d_totals = d_totals;
    c->slot ^= 1;
    return 0;
}

extern "C" int freesasa_gpu_lr_batch_dev(freesasa_gpu_ctx *c, const double *d_xyz, const double *d_radii,
                                         const int64_t *offsets, int n_structs, double probe, int n_slices,
                                         double *d_sasa, double *d_totals)
{
    if (!c) return -1;
    if (freesasa_gpu_wait(c)) return -1; /* (batches submitted asynchronously come first) */
    return run_batch(c, true, d_xyz, d_radii, offsets, n_structs, probe, n_slices, nullptr, d_sasa, nullptr, d_totals);
}

extern "C" int freesasa_gpu_sr_batch_dev(freesasa_gpu_ctx *c, const double *d_xyz, const double *d_radii,
                                         const int64_t *offsets, int n_structs, double probe, int n_points,
                                         const double *unit_points, double *d_sasa, int *d_counts, double *d_totals)
{
    if (!c) return -1;
    if (freesasa_gpu_wait(c)) return -1;
    return run_batch(c, false, d_xyz, d_radii, offsets, n_structs, probe, n_points, unit_points, d_sasa, d_counts, d_totals);
}

extern "C" int freesasa_gpu_segment_sums_dev(freesasa_gpu_ctx *c, const double *d_sasa, const int64_t *seg,
                                             int n_segs, double *d_out)
{
    if (!c) return -1;
    c->err[0] = 0;
    if (!d_sasa || !seg || !d_out || n_segs <= 0) return ctx_fail(c, "bad argument");
    for (int k = 0; k < n_segs; ++k)
        if (seg[k + 1] < seg[k]) return ctx_fail(c, "segment offsets must be non-decreasing");
    HIP_TRY(c, hipSetDevice(c->device));
    if (ensure(c, c->seg, sizeof(int64_t) * ((size_t)n_segs + 1))) return -1;
    HIP_TRY(c, hipMemcpyAsync(c->seg.p, seg, sizeof(int64_t) * ((size_t)n_segs + 1), hipMemcpyHostToDevice, c->stream));
    /* residues: a thread per segment (strict atom order); long segments: a workgroup each */
    if (seg[n_segs] - seg[0] < (int64_t)64 * n_segs)
        hipLaunchKernelGGL(k_segsum_small, dim3((n_segs + 255) / 256), dim3(256), 0, c->stream, d_sasa, (const int64_t *)c->seg.p, n_segs, d_out);
    else
        hipLaunchKernelGGL(k_totals, dim3(n_segs), dim3(SASA_TOT_B), 0, c->stream, d_sasa, (const int64_t *)c->seg.p, n_segs, d_out);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int freesasa_gpu_class_sums_dev(freesasa_gpu_ctx *c, const double *d_sasa, const unsigned char *d_class,
                                           const int64_t *offsets, int n_structs, double *d_out)
{
    if (!c) return -1;
    c->err[0] = 0;
    if (!d_sasa || !d_class || !offsets || !d_out || n_structs <= 0) return ctx_fail(c, "bad argument");
    for (int k = 0; k < n_structs; ++k)
        if (offsets[k + 1] < offsets[k]) return ctx_fail(c, "structure offsets must be non-decreasing");
    HIP_TRY(c, hipSetDevice(c->device));
    if (ensure(c, c->seg, sizeof(int64_t) * ((size_t)n_structs + 1))) return -1;
    HIP_TRY(c, hipMemcpyAsync(c->seg.p, offsets, sizeof(int64_t) * ((size_t)n_structs + 1), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_class_sums, dim3(n_structs), dim3(SASA_TOT_B), 0, c->stream, d_sasa, d_class, (const int64_t *)c->seg.p, d_out);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int freesasa_gpu_residue_areas_dev(freesasa_gpu_ctx *c, const double *d_sasa, const unsigned char *d_class,
                                              const unsigned char *d_backbone, const int64_t *res_first, int n_res,
                                              const short *ref_row, const double *ref_table, int ref_rows,
                                              double *d_abs, double *d_rel)
{
    if (!c) return -1;
    c->err[0] = 0;
    if (!d_sasa || !d_class || !d_backbone || !res_first || !d_abs || n_res <= 0) return ctx_fail(c, "bad argument");
    if (d_rel && (!ref_row || !ref_table || ref_rows <= 0)) return ctx_fail(c, "relative areas need the reference rows and table");
    for (int k = 0; k < n_res; ++k) {
        if (res_first[k + 1] < res_first[k]) return ctx_fail(c, "residue offsets must be non-decreasing");
        if (d_rel && ref_row[k] >= ref_rows) return ctx_fail(c, "reference row out of range");
    }
    HIP_TRY(c, hipSetDevice(c->device));
    /* one staging buffer: offsets, reference table, reference rows */
    const size_t b_first = sizeof(int64_t) * ((size_t)n_res + 1);
    const size_t b_table = d_rel ? sizeof(double) * 5 * (size_t)ref_rows : 0;
    const size_t b_rows = d_rel ? sizeof(short) * (size_t)n_res : 0;
    if (ensure(c, c->seg, b_first + b_table + b_rows)) return -1;
    char *base = (char *)c->seg.p;
    HIP_TRY(c, hipMemcpyAsync(base, res_first, b_first, hipMemcpyHostToDevice, c->stream));
    if (d_rel) {
        HIP_TRY(c, hipMemcpyAsync(base + b_first, ref_table, b_table, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(base + b_first + b_table, ref_row, b_rows, hipMemcpyHostToDevice, c->stream));
    }
    hipLaunchKernelGGL(k_residue_areas, dim3((n_res + 255) / 256), dim3(256), 0, c->stream, d_sasa, d_class, d_backbone,
                       (const int64_t *)base, d_rel ? (const short *)(base + b_first + b_table) : nullptr,
                       d_rel ? (const double *)(base + b_first) : nullptr, d_abs, d_rel, n_res);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

/* ------------------------------------------------------------------ test hooks of the L&R kernel's integer parts */

/* The neighbor sets the Lee-Richards kernel finds (ref: freesasa_nb_new with radii + probe, src/nb.c:524-557, what
 * tests/test_nb.c checks): per atom, in original order, the number of neighbors and, if d_nb is given, the first
 * nb_cap of them (original atom indices, in order of discovery).  Device pointers; d_nb may be NULL. */
extern "C" int freesasa_gpu_lr_neighbors_dev(freesasa_gpu_ctx *c, const double *d_xyz, const double *d_radii, const int64_t *offsets,
                                             int n_structs, double probe, int *d_nn, int *d_nb, int nb_cap)
{
    if (!c) return -1;
    if (!d_nn || (d_nb && nb_cap <= 0)) return ctx_fail(c, "bad argument");
    const int64_t n = offsets && n_structs > 0 ? offsets[n_structs] : 0;
    if (n <= 0) return ctx_fail(c, "empty batch");
    if (freesasa_gpu_wait(c)) return -1; /* (batches submitted asynchronously come first, as for every synchronous entry) */
    if (hipSetDevice(c->device) != hipSuccess || ensure(c, c->h_sasa, 8 * (size_t)n)) return -1;
    c->dbg_nn = d_nn; c->dbg_nb = d_nb; c->dbg_cap = nb_cap;
    const int rc = run_batch(c, true, d_xyz, d_radii, offsets, n_structs, probe, 20, nullptr,
                             (double *)c->h_sasa.p, nullptr, nullptr);
    c->dbg_nn = c->dbg_nb = nullptr; c->dbg_cap = 0;
    return rc;
}

/* The exposed arc length of n_sets sets of arcs (start, end pairs in [0, 2 pi], set k = arcs first[k] .. first[k+1]),
 * computed on the device by the arc union and sweep of the Lee-Richards kernel (ref: exposed_arc_length,
 * src/sasa_lr.c:389-408, and its KATs :455-475).  Host arrays; at most 64 sets. */
extern "C" int freesasa_gpu_arc_union_dev(freesasa_gpu_ctx *c, const double *arcs, const int *first, int n_sets, double *out)
{
    if (!c) return -1;
    if (!arcs || !first || !out || n_sets <= 0 || n_sets > 64) return ctx_fail(c, "bad argument");
    const int total = first[n_sets];
    /* the arc pass feeds the union in the order of the arcs' mid-points (the neighbors' directions) */
    std::vector<double> sorted(2 * (size_t)(total > 0 ? total : 1));
    for (int k = 0; k < n_sets; ++k) {
        std::vector<std::pair<double, double>> v;
        for (int i = first[k]; i < first[k + 1]; ++i) v.emplace_back(arcs[2 * i], arcs[2 * i + 1]);
        std::stable_sort(v.begin(), v.end(), [](const std::pair<double, double> &x, const std::pair<double, double> &y) {
            return x.first + x.second < y.first + y.second; });
        for (size_t i = 0; i < v.size(); ++i) { sorted[2 * (first[k] + i)] = v[i].first; sorted[2 * (first[k] + i) + 1] = v[i].second; }
    }
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t b_arcs = sizeof(double) * sorted.size(), b_first = sizeof(int) * ((size_t)n_sets + 1);
    if (ensure(c, c->seg, b_arcs + b_first + 8 * 64 + 64)) return -1;
    char *base = (char *)c->seg.p;
    HIP_TRY(c, hipMemcpyAsync(base, sorted.data(), b_arcs, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(base + b_arcs, first, b_first, hipMemcpyHostToDevice, c->stream));
    double *d_out = (double *)(base + ((b_arcs + b_first + 15) & ~(size_t)15));
    hipLaunchKernelGGL(k_lr2_arc_kat, dim3(1), dim3(64), 0, c->stream, (const double *)base, (const int *)(base + b_arcs), n_sets, d_out);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, d_out, sizeof(double) * (size_t)n_sets, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

/* ------------------------------------------------------------------ host-pointer batch */

/* A small pool of contexts so that concurrent host threads (the reference library is
 * re-entrant, doc/doxy-main.md:741-756) each get their own stream and workspace. */
static std::mutex g_pool_mu;
static std::vector<freesasa_gpu_ctx *> g_pool;

static freesasa_gpu_ctx *pool_get(int device)
{
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t k = 0; k < g_pool.size(); ++k)
            if (device < 0 || g_pool[k]->device == device) {
                freesasa_gpu_ctx *c = g_pool[k];
                g_pool.erase(g_pool.begin() + k);
                return c;
            }
    }
    return freesasa_gpu_ctx_create(device, nullptr);
}
static void pool_put(freesasa_gpu_ctx *c)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool.push_back(c);
}

/* Destroy the idle contexts of the pool (their streams, workspaces and staging buffers): device memory goes back
 * to the runtime; the next host-pointer call builds what it needs again. */
extern "C" void freesasa_gpu_release_pool(void)
{
    std::vector<freesasa_gpu_ctx *> idle;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        idle.swap(g_pool);
    }
    for (freesasa_gpu_ctx *c : idle) freesasa_gpu_ctx_destroy(c);
}

static int set_err(char *out, int len, const char *msg)
{
    if (out && len > 0) snprintf(out, (size_t)len, "%s", msg);
    return -1;
}

extern "C" int freesasa_gpu_calc_batch(const double *xyz, const double *radii, const int64_t *offsets, int n_structs,
                                       int alg, double probe, int resolution, double *sasa_out, int *counts_out,
                                       double *totals_out, int device, char *err_out, int err_len)
{
    if (err_out && err_len > 0) err_out[0] = 0;
    if (!xyz || !radii || !offsets || !sasa_out) return set_err(err_out, err_len, "null argument");
    if (freesasa_gpu_device_count() <= 0)
        return set_err(err_out, err_len, "no HIP device available: libfreesasa_amd has no CPU path");
    freesasa_gpu_ctx *c = pool_get(device);
    if (!c) return set_err(err_out, err_len, "could not create a GPU context");
    int ret = -1;
    do {
        if (n_structs <= 0 || offsets[n_structs] <= 0) { ctx_fail(c, "empty batch"); break; }
        const size_t n = (size_t)offsets[n_structs];
        if (hipSetDevice(c->device) != hipSuccess) { ctx_fail(c, "hipSetDevice failed"); break; }
        if (ensure(c, c->h_xyz, 24 * n) || ensure(c, c->h_radii, 8 * n) || ensure(c, c->h_sasa, 8 * n) ||
            ensure(c, c->h_counts, 4 * n) || ensure(c, c->h_totals, 8 * (size_t)n_structs))
            break;
        if (hipMemcpyAsync(c->h_xyz.p, xyz, 24 * n, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(c->h_radii.p, radii, 8 * n, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
            ctx_fail(c, "host-to-device copy failed");
            break;
        }
        if (alg == 0) {
            ret = run_batch(c, true, (double *)c->h_xyz.p, (double *)c->h_radii.p, offsets, n_structs, probe, resolution,
                            nullptr, (double *)c->h_sasa.p, nullptr, totals_out ? (double *)c->h_totals.p : nullptr);
        } else if (alg == 1) {
            std::vector<double> tp(3 * (size_t)(resolution > 0 ? resolution : 1));
            if (resolution > 0) freesasa_gpu_test_points(resolution, tp.data());
            ret = run_batch(c, false, (double *)c->h_xyz.p, (double *)c->h_radii.p, offsets, n_structs, probe, resolution,
                            tp.data(), (double *)c->h_sasa.p, counts_out ? (int *)c->h_counts.p : nullptr,
                            totals_out ? (double *)c->h_totals.p : nullptr);
        } else {
            ctx_fail(c, "unknown algorithm %d", alg);
        }
        if (ret) break;
        ret = -1;
        if (hipMemcpyAsync(sasa_out, c->h_sasa.p, 8 * n, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { ctx_fail(c, "device-to-host copy failed"); break; }
        if (counts_out && alg == 1 &&
            hipMemcpyAsync(counts_out, c->h_counts.p, 4 * n, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { ctx_fail(c, "device-to-host copy failed"); break; }
        if (totals_out &&
            hipMemcpyAsync(totals_out, c->h_totals.p, 8 * (size_t)n_structs, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { ctx_fail(c, "device-to-host copy failed"); break; }
        if (hipStreamSynchronize(c->stream) != hipSuccess) { ctx_fail(c, "stream synchronize failed"); break; }
        ret = 0;
    } while (0);
    if (ret) {
        (void)hipStreamSynchronize(c->stream); /* the caller's arrays must not be read after we return */
        set_err(err_out, err_len, c->err[0] ? c->err : "GPU batch failed");
    }
    pool_put(c);
    return ret;
}

/* ------------------------------------------------------------------ several GPUs, one process */

/* Independent structures shard with no exchange: the batch is cut into contiguous runs of
 * structures with about equal atom counts, one run per device of the mask, each run handled by its
 * own host thread through freesasa_gpu_calc_batch (its own pooled context, stream and workspace).
 * Contiguous runs need no gather: every device reads and writes its slice of the caller's arrays. */
/* cuts[k] = first structure of shard k (cuts[n_parts] = n_structs): where the running atom count passes
 * k/n_parts of the total; shards may be empty when there are fewer structures than parts */
extern "C" void freesasa_gpu_shard_cuts(const int64_t *offsets, int n_structs, int n_parts, int *cuts)
{
    cuts[0] = 0;
    const int64_t base = offsets[0], total = offsets[n_structs] - base;
    for (int k = 1, s = 0; k < n_parts; ++k) {
        const int64_t want = base + total * k / n_parts;
        while (s < n_structs && offsets[s] < want) ++s;
        cuts[k] = s < cuts[k - 1] ? cuts[k - 1] : s;
    }
    cuts[n_parts] = n_structs;
}

extern "C" int freesasa_gpu_calc_batch_devices(const double *xyz, const double *radii, const int64_t *offsets, int n_structs,
                                               int alg, double probe, int resolution, double *sasa_out, int *counts_out,
                                               double *totals_out, const int *devices, int n_devices, char *err_out, int err_len)
{
    if (err_out && err_len > 0) err_out[0] = 0;
    if (!xyz || !radii || !offsets || !sasa_out || n_structs <= 0 || !devices || n_devices <= 0)
        return set_err(err_out, err_len, "bad argument");
    const int n_dev = freesasa_gpu_device_count();
    if (n_dev <= 0) return set_err(err_out, err_len, "no HIP device available: libfreesasa_amd has no CPU path");
    for (int k = 0; k < n_devices; ++k)
        if (devices[k] < 0 || devices[k] >= n_dev) return set_err(err_out, err_len, "device index out of range");
    const int nd = n_devices;
    std::vector<int> cut(nd + 1);
    freesasa_gpu_shard_cuts(offsets, n_structs, nd, cut.data());
    std::vector<int> rc(nd, 0);
    std::vector<std::vector<char>> errs(nd, std::vector<char>(256, 0));
    auto run = [&](int k) {
        const int s0 = cut[k], ns = cut[k + 1] - cut[k];
        if (ns <= 0 || offsets[s0 + ns] == offsets[s0]) return;
        std::vector<int64_t> off(ns + 1); /* the shard's own CSR offsets start at 0 */
        for (int i = 0; i <= ns; ++i) off[i] = offsets[s0 + i] - offsets[s0];
        const int64_t a0 = offsets[s0];
        rc[k] = freesasa_gpu_calc_batch(xyz + 3 * a0, radii + a0, off.data(), ns, alg, probe, resolution, sasa_out + a0,
                                        counts_out ? counts_out + a0 : nullptr, totals_out ? totals_out + s0 : nullptr,
                                        devices[k], errs[k].data(), (int)errs[k].size());
    };
    std::vector<std::thread> th;
    for (int k = 1; k < nd; ++k) th.emplace_back(run, k);
    run(0);
    for (auto &t : th) t.join();
    for (int k = 0; k < nd; ++k)
        if (rc[k]) return set_err(err_out, err_len, errs[k].data()[0] ? errs[k].data() : "a device shard failed");
    return 0;
}

extern "C" int freesasa_gpu_calc_batch_multi(const double *xyz, const double *radii, const int64_t *offsets, int n_structs,
                                             int alg, double probe, int resolution, double *sasa_out, int *counts_out,
                                             double *totals_out, unsigned device_mask, char *err_out, int err_len)
{
    if (err_out && err_len > 0) err_out[0] = 0;
    const int n_dev = freesasa_gpu_device_count();
    if (n_dev <= 0) return set_err(err_out, err_len, "no HIP device available: libfreesasa_amd has no CPU path");
    std::vector<int> devs;
    for (int d = 0; d < 32 && d < n_dev; ++d)
        if (device_mask & (1u << d)) devs.push_back(d);
    if (devs.empty()) return set_err(err_out, err_len, "device mask selects no available device");
    return freesasa_gpu_calc_batch_devices(xyz, radii, offsets, n_structs, alg, probe, resolution, sasa_out, counts_out,
                                           totals_out, devs.data(), (int)devs.size(), err_out, err_len);
}

/* ------------------------------------------------------------------ host arrays in, host arrays out, pipelined */

/* One host pointer: page-locked already (hipHostMalloc / hipHostRegister, e.g. a pinned tensor)? */
static bool host_pinned(const void *p)
{
    hipPointerAttribute_t at;
    if (!p || hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

/* Grow a context's page-locked staging buffer (for callers whose arrays are pageable). */
static int ensure_pinned(freesasa_gpu_ctx *c, void **p, size_t *cap, size_t bytes)
{
    if (bytes <= *cap) return 0;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr; *cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (host_malloc(p, want) != hipSuccess) return ctx_fail(c, "out of page-locked host memory (%zu bytes)", want);
    *cap = want;
    return 0;
}

/* The batch is cut into chunks of whole structures (about chunk_atoms atoms each) that n_lanes host threads take
 * from a shared counter; every lane owns a pooled context (stream, workspace, staging) and runs
 *     host -> device copy,  cell sort + tile kernels,  device -> host copy
 * for its chunk while the other lanes are in a different stage: PCIe in, kernels and PCIe out of different
 * chunks overlap.  Page-locked caller arrays are copied by DMA in place; pageable ones go through the lane's
 * page-locked staging buffers (the memcpy of one lane overlaps the DMA of another). */
extern "C" int freesasa_gpu_calc_batch_pipelined(const double *xyz, const double *radii, const int64_t *offsets, int n_structs,
                                                 int alg, double probe, int resolution, double *sasa_out, int *counts_out,
                                                 double *totals_out, int device, int n_lanes, long long chunk_atoms,
                                                 char *err_out, int err_len)
{
    if (err_out && err_len > 0) err_out[0] = 0;
    if (!xyz || !radii || !offsets || !sasa_out || n_structs <= 0) return set_err(err_out, err_len, "bad argument");
    if (alg != 0 && alg != 1) return set_err(err_out, err_len, "unknown algorithm");
    if (freesasa_gpu_device_count() <= 0)
        return set_err(err_out, err_len, "no HIP device available: libfreesasa_amd has no CPU path");
    const bool pin_in = host_pinned(xyz) && host_pinned(radii);
    const bool pin_out = host_pinned(sasa_out) && (!counts_out || host_pinned(counts_out)) && (!totals_out || host_pinned(totals_out));
    if (n_lanes <= 0) n_lanes = pin_in && pin_out ? 3 : 4; /* (measured, 1e7 atoms: 1 / 2 / 3 / 4 / 6 lanes with DMA in place 24.0 / 22.4 /
                                                              15.9 / 17.1 / 16.6 ms - one lane each in PCIe in, kernels, PCIe out;
                                                              staging through page-locked buffers also spends host memcpy time) */
    if (n_lanes > 8) n_lanes = 8;
    if (chunk_atoms <= 0) chunk_atoms = 1250000;
    std::vector<int> cut(1, 0);
    for (int s = 0; s < n_structs; ++s)
        if (offsets[s + 1] - offsets[cut.back()] >= chunk_atoms && s + 1 < n_structs) cut.push_back(s + 1);
    cut.push_back(n_structs);
    const int n_chunks = (int)cut.size() - 1;
    if (n_lanes > n_chunks) n_lanes = n_chunks;
    std::vector<double> tp;
    if (alg == 1) { tp.resize(3 * (size_t)(resolution > 0 ? resolution : 1)); if (resolution > 0) freesasa_gpu_test_points(resolution, tp.data()); }
    std::atomic<int> next(0), failed(0);
    std::vector<std::vector<char>> errs(n_lanes, std::vector<char>(256, 0));
    auto lane = [&](int id) {
        freesasa_gpu_ctx *c = pool_get(device);
        if (!c) { snprintf(errs[id].data(), 256, "could not create a GPU context"); failed = 1; return; }
        std::vector<int64_t> off;
        for (;;) {
            const int k = next.fetch_add(1);
            if (k >= n_chunks || failed.load()) break;
            const int s0 = cut[k], ns = cut[k + 1] - cut[k];
            const int64_t a0 = offsets[s0];
            const size_t n = (size_t)(offsets[s0 + ns] - a0);
            if (n == 0) { if (totals_out) for (int i = 0; i < ns; ++i) totals_out[s0 + i] = 0; continue; }
            off.resize((size_t)ns + 1);
            for (int i = 0; i <= ns; ++i) off[i] = offsets[s0 + i] - a0;
            int rc = -1;
            do {
                if (hipSetDevice(c->device) != hipSuccess) { ctx_fail(c, "hipSetDevice failed"); break; }
                if (ensure(c, c->h_xyz, 24 * n) || ensure(c, c->h_radii, 8 * n) || ensure(c, c->h_sasa, 8 * n) ||
                    ensure(c, c->h_counts, 4 * n) || ensure(c, c->h_totals, 8 * (size_t)ns))
                    break;
                const double *src_xyz = xyz + 3 * a0, *src_r = radii + a0;
                if (!pin_in) {
                    if (ensure_pinned(c, &c->stage_in, &c->stage_in_cap, 32 * n)) break;
                    memcpy(c->stage_in, src_xyz, 24 * n);
                    memcpy((char *)c->stage_in + 24 * n, src_r, 8 * n);
                    src_xyz = (const double *)c->stage_in;
                    src_r = (const double *)((char *)c->stage_in + 24 * n);
                }
                if (hipMemcpyAsync(c->h_xyz.p, src_xyz, 24 * n, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                    hipMemcpyAsync(c->h_radii.p, src_r, 8 * n, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
                    ctx_fail(c, "host-to-device copy failed");
                    break;
                }
                if (run_batch(c, alg == 0, (double *)c->h_xyz.p, (double *)c->h_radii.p, off.data(), ns, probe, resolution,
                              alg == 1 ? tp.data() : nullptr, (double *)c->h_sasa.p, counts_out && alg == 1 ? (int *)c->h_counts.p : nullptr,
                              totals_out ? (double *)c->h_totals.p : nullptr))
                    break;
                const bool want_counts = counts_out && alg == 1;
                double *dst_sasa = sasa_out + a0, *dst_tot = totals_out ? totals_out + s0 : nullptr;
                int *dst_cnt = want_counts ? counts_out + a0 : nullptr;
                const size_t out_bytes = 8 * n + (want_counts ? 4 * n : 0) + (dst_tot ? 8 * (size_t)ns : 0);
                if (!pin_out) {
                    if (ensure_pinned(c, &c->stage_out, &c->stage_out_cap, out_bytes)) break;
                    dst_sasa = (double *)c->stage_out;
                    dst_cnt = want_counts ? (int *)((char *)c->stage_out + 8 * n) : nullptr;
                    dst_tot = totals_out ? (double *)((char *)c->stage_out + 8 * n + (want_counts ? 4 * n : 0)) : nullptr;
                }
                bool ok = hipMemcpyAsync(dst_sasa, c->h_sasa.p, 8 * n, hipMemcpyDeviceToHost, c->stream) == hipSuccess;
                if (ok && want_counts) ok = hipMemcpyAsync(dst_cnt, c->h_counts.p, 4 * n, hipMemcpyDeviceToHost, c->stream) == hipSuccess;
                if (ok && dst_tot) ok = hipMemcpyAsync(dst_tot, c->h_totals.p, 8 * (size_t)ns, hipMemcpyDeviceToHost, c->stream) == hipSuccess;
                if (!ok) { ctx_fail(c, "device-to-host copy failed"); break; }
                if (hipStreamSynchronize(c->stream) != hipSuccess) { ctx_fail(c, "stream synchronize failed"); break; }
                if (!pin_out) {
                    memcpy(sasa_out + a0, dst_sasa, 8 * n);
                    if (want_counts) memcpy(counts_out + a0, dst_cnt, 4 * n);
                    if (totals_out) memcpy(totals_out + s0, dst_tot, 8 * (size_t)ns);
                }
                rc = 0;
            } while (0);
            if (rc) {
                (void)hipStreamSynchronize(c->stream); /* nothing may still read the caller's arrays when we return */
                snprintf(errs[id].data(), 256, "%s", c->err[0] ? c->err : "GPU batch failed");
                failed = 1;
                break;
            }
        }
        pool_put(c);
    };
    std::vector<std::thread> th;
    for (int k = 1; k < n_lanes; ++k) th.emplace_back(lane, k);
    lane(0);
    for (auto &t : th) t.join();
    if (failed.load())
        for (int k = 0; k < n_lanes; ++k)
            if (errs[k][0]) return set_err(err_out, err_len, errs[k].data());
    return failed.load() ? set_err(err_out, err_len, "GPU batch failed") : 0;
}

/* ------------------------------------------------------------------ structure sweep */

static bool pread_all(int fd, void *buf, size_t bytes, long long off)
{
    char *p = (char *)buf;
    while (bytes) {
        const ssize_t r = pread(fd, p, bytes, (off_t)off);
        if (r <= 0) return false;
        p += r; off += r; bytes -= (size_t)r;
    }
    return true;
}
static bool pwrite_all(int fd, const void *buf, size_t bytes, long long off)
{
    const char *p = (const char *)buf;
    while (bytes) {
        const ssize_t r = pwrite(fd, p, bytes, (off_t)off);
        if (r <= 0) return false;
        p += r; off += r; bytes -= (size_t)r;
    }
    return true;
}



/* Files -> per-structure totals: the loader (host threads, include/freesasa_ingest.h) reads batch
 * k+1 while this thread has batch k on the GPU.  Inputs that fail to load get total 0 and their
 * loader status; the call only fails for GPU errors. */
/* done_path (may be NULL): the sweep's done-list, same idea and format as the trajectory driver's — a first line
 * with the run's parameters, then "shard <batch> <first file> <files>" per finished batch — next to a result file
 * <done_path>.bin that holds, per file, total | class sums (3) | atoms | status (fixed 48-byte records), written
 * before the batch is listed.  A call that finds the done-list of the same run takes the listed batches' results
 * from the result file and only computes the others.  Returns 0 done, 1 stopped after max_new_batches, -1 error. */
static int sweep_impl(const char *const *paths, int n_paths, int ingest_options, int n_threads,
                      int alg, double probe, int resolution, long long batch_atoms,
                      double *totals_out, double *class_sums_out, long long *atoms_out, int *status_out,
                      const char *done_path, long long max_new_batches, int device, char *err_out, int err_len)
{
    if (err_out && err_len > 0) err_out[0] = 0;
    if (!paths || n_paths < 0 || !totals_out || !status_out) return set_err(err_out, err_len, "null argument");
    if (alg != 0 && alg != 1) return set_err(err_out, err_len, "unknown algorithm");
    if (freesasa_gpu_device_count() <= 0)
        return set_err(err_out, err_len, "no HIP device available: libfreesasa_amd has no CPU path");
    if (n_paths == 0) return 0;
    if (batch_atoms <= 0) batch_atoms = 2000000;
    /* batches of roughly batch_atoms atoms, estimated from the file sizes (~81 bytes per ATOM line) */
    std::vector<int> cut(1, 0);
    {
        long long bytes = 0;
        for (int k = 0; k < n_paths; ++k) {
            struct stat st;
            bytes += (paths[k] && stat(paths[k], &st) == 0) ? (long long)st.st_size : 0;
            if (bytes >= batch_atoms * 81 && k + 1 < n_paths) { cut.push_back(k + 1); bytes = 0; }
        }
        cut.push_back(n_paths);
    }
    const int n_batches = (int)cut.size() - 1;
    /* done-list and result file */
    struct Rec { double total, cls[3]; long long atoms; int status, pad; };
    static_assert(sizeof(Rec) == 48, "result record");
    std::vector<char> done((size_t)n_batches, 0);
    int fd_done = -1, fd_res = -1;
    if (done_path) {
        unsigned long long h = 1469598103934665603ULL; /* FNV-1a over the files' names, sizes and modification times: the done-list belongs to THESE files as they are now */
        for (int k = 0; k < n_paths; ++k) {
            for (const char *q = paths[k] ? paths[k] : ""; ; ++q) { h = (h ^ (unsigned char)*q) * 1099511628211ULL; if (!*q) break; }
            struct stat st;
            long long id[3] = {-1, -1, -1};
            if (paths[k] && stat(paths[k], &st) == 0) { id[0] = (long long)st.st_size; id[1] = (long long)st.st_mtim.tv_sec; id[2] = (long long)st.st_mtim.tv_nsec; }
            for (size_t q = 0; q < sizeof id; ++q) h = (h ^ ((const unsigned char *)id)[q]) * 1099511628211ULL;
        }
        char head[256];
        snprintf(head, sizeof head, "freesasa_amd sweep done-list v2 n_files=%d batches=%d files=%016llx options=%d alg=%d resolution=%d probe=%.17g\n",
                 n_paths, n_batches, h, ingest_options, alg, resolution, probe);
        const std::string res_path = std::string(done_path) + ".bin";
        bool resume = false;
        if (FILE *fp = fopen(done_path, "r")) {
            char line[256];
            if (fgets(line, sizeof line, fp)) {
                if (strcmp(line, head) != 0) { fclose(fp); return set_err(err_out, err_len, "the done-list belongs to a sweep with other parameters or other (changed) input files"); }
                resume = true;
                int b, first, count;
                while (fgets(line, sizeof line, fp))
                    if (sscanf(line, "shard %d %d %d", &b, &first, &count) == 3 && b >= 0 && b < n_batches && first == cut[b] &&
                        count == cut[b + 1] - cut[b] && line[strlen(line) - 1] == '\n')
                        done[(size_t)b] = 1;
            }
            fclose(fp);
        }
        fd_res = open(res_path.c_str(), resume ? O_RDWR | O_CREAT : O_RDWR | O_CREAT | O_TRUNC, 0644);
        fd_done = open(done_path, resume ? O_WRONLY | O_APPEND : O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd_res < 0 || fd_done < 0 || (!resume && write(fd_done, head, strlen(head)) != (ssize_t)strlen(head))) {
            if (fd_res >= 0) close(fd_res);
            if (fd_done >= 0) close(fd_done);
            return set_err(err_out, err_len, "cannot open the done-list or its result file");
        }
        for (int b = 0; b < n_batches; ++b) { /* results of the batches already done */
            if (!done[(size_t)b]) continue;
            std::vector<Rec> recs((size_t)(cut[b + 1] - cut[b]));
            if (!pread_all(fd_res, recs.data(), sizeof(Rec) * recs.size(), (long long)sizeof(Rec) * cut[b])) { done[(size_t)b] = 0; continue; }
            for (size_t k = 0; k < recs.size(); ++k) {
                const int f = cut[b] + (int)k;
                totals_out[f] = recs[k].total; status_out[f] = recs[k].status;
                if (atoms_out) atoms_out[f] = recs[k].atoms;
                if (class_sums_out) for (int q = 0; q < 3; ++q) class_sums_out[3 * f + q] = recs[k].cls[q];
            }
        }
    }
    std::vector<int> todo;
    for (int b = 0; b < n_batches; ++b)
        if (!done[(size_t)b]) todo.push_back(b);
    bool stopped = false;
    if (max_new_batches > 0 && (long long)todo.size() > max_new_batches) { todo.resize((size_t)max_new_batches); stopped = true; }
    if (todo.empty()) {
        if (fd_res >= 0) close(fd_res);
        if (fd_done >= 0) close(fd_done);
        return stopped ? 1 : 0;
    }
    freesasa_gpu_ctx *c = pool_get(device);
    if (!c) {
        if (fd_res >= 0) close(fd_res);
        if (fd_done >= 0) close(fd_done);
        return set_err(err_out, err_len, "could not create a GPU context");
    }
    const bool want_cls = class_sums_out != nullptr || done_path != nullptr;
    std::vector<double> cls_tmp;
    freesasa_ingest_batch cur, next;
    int cur_rc = 0, next_rc = 0;
    memset(&cur, 0, sizeof cur);
    memset(&next, 0, sizeof next);
    auto load = [&](int b, freesasa_ingest_batch *out, int *rc) {
        *rc = freesasa_ingest_pdb_files(paths + cut[b], cut[b + 1] - cut[b], ingest_options, n_threads, out);
    };
    std::vector<double> tp;
    if (alg == 1) { tp.resize(3 * (size_t)(resolution > 0 ? resolution : 1)); if (resolution > 0) freesasa_gpu_test_points(resolution, tp.data()); }
    load(todo[0], &cur, &cur_rc);
    int ret = 0;
    for (size_t ti = 0; ti < todo.size() && !ret; ++ti) {
        const int b = todo[ti];
        std::thread loader;
        if (ti + 1 < todo.size()) loader = std::thread(load, todo[ti + 1], &next, &next_rc);
        const int first = cut[b], ns = cut[b + 1] - cut[b];
        do {
            if (cur_rc) { ctx_fail(c, "loader failed with code %d", cur_rc); ret = -1; break; }
            for (int k = 0; k < ns; ++k) {
                status_out[first + k] = cur.status[k];
                totals_out[first + k] = 0;
                if (atoms_out) atoms_out[first + k] = cur.offsets[k + 1] - cur.offsets[k];
                if (class_sums_out) class_sums_out[3 * (first + k)] = class_sums_out[3 * (first + k) + 1] = class_sums_out[3 * (first + k) + 2] = 0;
            }
            const size_t n = (size_t)cur.n_atoms;
            if (n == 0) break;
            ret = -1;
            if (hipSetDevice(c->device) != hipSuccess) { ctx_fail(c, "hipSetDevice failed"); break; }
            if (ensure(c, c->h_xyz, 24 * n) || ensure(c, c->h_radii, 8 * n) || ensure(c, c->h_sasa, 8 * n) ||
                ensure(c, c->h_counts, n) || ensure(c, c->h_totals, 8 * 4 * (size_t)ns))
                break;
            if (hipMemcpyAsync(c->h_xyz.p, cur.xyz, 24 * n, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                hipMemcpyAsync(c->h_radii.p, cur.radii, 8 * n, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
                ctx_fail(c, "host-to-device copy failed");
                break;
            }
            double *d_tot = (double *)c->h_totals.p, *d_cls = d_tot + ns;
            if (run_batch(c, alg == 0, (double *)c->h_xyz.p, (double *)c->h_radii.p, cur.offsets, ns, probe, resolution,
                          alg == 1 ? tp.data() : nullptr, (double *)c->h_sasa.p, nullptr, d_tot))
                break;
            double *cls_dst = class_sums_out ? class_sums_out + 3 * (size_t)first : nullptr;
            if (want_cls) {
                if (!cls_dst) { cls_tmp.resize(3 * (size_t)ns); cls_dst = cls_tmp.data(); }
                if (hipMemcpyAsync(c->h_counts.p, cur.atom_class, n, hipMemcpyHostToDevice, c->stream) != hipSuccess) { ctx_fail(c, "host-to-device copy failed"); break; }
                if (freesasa_gpu_class_sums_dev(c, (double *)c->h_sasa.p, (const unsigned char *)c->h_counts.p, cur.offsets, ns, d_cls)) break;
                if (hipMemcpyAsync(cls_dst, d_cls, 8 * 3 * (size_t)ns, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { ctx_fail(c, "device-to-host copy failed"); break; }
            }
            if (hipMemcpyAsync(totals_out + first, d_tot, 8 * (size_t)ns, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { ctx_fail(c, "device-to-host copy failed"); break; }
            if (hipStreamSynchronize(c->stream) != hipSuccess) { ctx_fail(c, "stream synchronize failed"); break; }
            ret = 0;
        } while (0);
        if (ret) (void)hipStreamSynchronize(c->stream); /* no copy may still read the batch when it is freed */
        if (!ret && fd_done >= 0) { /* the batch's results to the result file, then its line in the done-list */
            std::vector<Rec> recs((size_t)ns);
            const bool have_cls = cur.n_atoms > 0;
            const double *cls_src = class_sums_out ? class_sums_out + 3 * (size_t)first : (have_cls ? cls_tmp.data() : nullptr);
            for (int k = 0; k < ns; ++k) {
                Rec &r = recs[(size_t)k];
                memset(&r, 0, sizeof r);
                r.total = totals_out[first + k]; r.status = status_out[first + k];
                r.atoms = cur.offsets ? cur.offsets[k + 1] - cur.offsets[k] : 0;
                if (cls_src) for (int q = 0; q < 3; ++q) r.cls[q] = cls_src[3 * k + q];
            }
            char line[96];
            const int len = snprintf(line, sizeof line, "shard %d %d %d\n", b, first, ns);
            if (!pwrite_all(fd_res, recs.data(), sizeof(Rec) * recs.size(), (long long)sizeof(Rec) * first) || fdatasync(fd_res) != 0 ||
                write(fd_done, line, (size_t)len) != len || fdatasync(fd_done) != 0) {
                ctx_fail(c, "could not record the finished batch in the done-list");
                ret = -1;
            }
        }
        if (loader.joinable()) loader.join();
        freesasa_ingest_free(&cur);
        cur = next;
        cur_rc = next_rc;
        memset(&next, 0, sizeof next);
    }
    freesasa_ingest_free(&cur);
    if (ret) set_err(err_out, err_len, c->err[0] ? c->err : "GPU sweep failed");
    pool_put(c);
    if (fd_res >= 0) close(fd_res);
    if (fd_done >= 0) close(fd_done);
    return ret ? ret : (stopped ? 1 : 0);
}

extern "C" int freesasa_gpu_sweep_files(const char *const *paths, int n_paths, int ingest_options, int n_threads,
                                        int alg, double probe, int resolution, long long batch_atoms,
                                        double *totals_out, double *class_sums_out, long long *atoms_out, int *status_out,
                                        int device, char *err_out, int err_len)
{
    return sweep_impl(paths, n_paths, ingest_options, n_threads, alg, probe, resolution, batch_atoms, totals_out, class_sums_out,
                      atoms_out, status_out, nullptr, 0, device, err_out, err_len);
}

extern "C" int freesasa_gpu_sweep_files_resumable(const char *const *paths, int n_paths, int ingest_options, int n_threads,
                                                  int alg, double probe, int resolution, long long batch_atoms,
                                                  double *totals_out, double *class_sums_out, long long *atoms_out, int *status_out,
                                                  const char *done_path, long long max_new_batches, int device, char *err_out, int err_len)
{
    return sweep_impl(paths, n_paths, ingest_options, n_threads, alg, probe, resolution, batch_atoms, totals_out, class_sums_out,
                      atoms_out, status_out, done_path, max_new_batches, device, err_out, err_len);
}


/* ------------------------------------------------------------------ trajectory driver */

/* Frames of ONE system (same atoms, same radii) are independent structures: a SHARD is a run of frames_per_batch
 * frames that goes through the engine as one batch.  A few host lanes take shards from a shared counter; a lane
 * owns a pooled context (stream, workspace, page-locked staging) and does, for its shard,
 *     read (memory or frame file) -> host-to-device -> [fp32 frames widened to fp64 on the device: an INPUT format,
 *     the arithmetic stays fp64] -> cell sort + tile kernels -> device-to-host -> write (memory or files)
 * while the other lanes are in another stage.  The radii live once per device context (shared by every frame of
 * a batch).  With a done-list file every finished shard is recorded after its results are on disk; a later call
 * with the same parameters skips the recorded shards: an interrupted run resumes. */
__global__ __launch_bounds__(256) void k_widen_f32(const float *in, double *out, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}

namespace {
struct TrajIO {
    const double *mem_in = nullptr; /* frames in host memory (fp64) ... */
    int fd_in = -1;                 /* ... or in a file of raw frames */
    int in_f32 = 0;
    long long in_header = 0;
    double *totals_mem = nullptr, *sasa_mem = nullptr;
    int fd_totals = -1, fd_sasa = -1;
    int fd_done = -1;               /* done-list (append) */
    std::vector<char> done;         /* shards already recorded */
};

/* returns 0: all shards done, 1: stopped after max_new shards (more left), -1: error */
static int traj_run(TrajIO &io, const double *radii, int n_atoms, long long n_frames, int alg, double probe, int resolution,
                    int frames_per_batch, int n_lanes, long long max_new, int device, char *err_out, int err_len)
{
    const size_t n = (size_t)n_atoms, FB = (size_t)frames_per_batch;
    const long long n_shards = (n_frames + frames_per_batch - 1) / frames_per_batch;
    if (io.done.size() < (size_t)n_shards) io.done.resize((size_t)n_shards, 0);
    if (n_lanes <= 0) {
        n_lanes = 3;
        if (const char *e = getenv("FREESASA_AMD_TRAJ_LANES")) n_lanes = atoi(e) > 0 ? atoi(e) : 3; /* tuning aid */
    }
    if (n_lanes > 8) n_lanes = 8;
    if (n_lanes > n_shards) n_lanes = (int)n_shards;
    std::vector<double> tp;
    if (alg == 1) { tp.resize(3 * (size_t)resolution); freesasa_gpu_test_points(resolution, tp.data()); }
    std::vector<int64_t> offs(FB + 1);
    for (size_t k = 0; k <= FB; ++k) offs[k] = (int64_t)(k * n);
    const bool in_pinned = io.mem_in && host_pinned(io.mem_in);
    const bool out_pinned = io.totals_mem && host_pinned(io.totals_mem) && (!io.sasa_mem || host_pinned(io.sasa_mem));
    const bool want_sasa = io.sasa_mem || io.fd_sasa >= 0;
    std::atomic<long long> next(0), fresh(0);
    std::atomic<int> failed(0), stopped(0);
    std::mutex done_mu;
    std::vector<std::vector<char>> errs(n_lanes, std::vector<char>(256, 0));
    auto lane = [&](int id) {
        freesasa_gpu_ctx *c = pool_get(device);
        if (!c) { snprintf(errs[id].data(), 256, "could not create a GPU context"); failed = 1; return; }
        bool radii_up = false;
        for (;;) {
            const long long k = next.fetch_add(1);
            if (k >= n_shards || failed.load()) break;
            if (io.done[(size_t)k]) continue;
            if (max_new > 0 && fresh.fetch_add(1) >= max_new) { stopped = 1; break; }
            const long long f0 = k * frames_per_batch;
            const int nf = (int)(n_frames - f0 < frames_per_batch ? n_frames - f0 : frames_per_batch);
            const size_t na = n * (size_t)nf;
            const size_t in_bytes = (io.in_f32 ? 12 : 24) * na;
            int rc = -1;
            do {
                if (hipSetDevice(c->device) != hipSuccess) { ctx_fail(c, "hipSetDevice failed"); break; }
                if (ensure(c, c->h_xyz, 24 * n * FB) || ensure(c, c->h_radii, 8 * n) || ensure(c, c->h_sasa, 8 * n * FB) ||
                    ensure(c, c->h_totals, 8 * FB) || (io.in_f32 && ensure(c, c->h_counts, 12 * n * FB)))
                    break;
                if (!radii_up) { /* once per lane: the radii of the system */
                    if (hipMemcpyAsync(c->h_radii.p, radii, 8 * n, hipMemcpyHostToDevice, c->stream) != hipSuccess) { ctx_fail(c, "radii upload failed"); break; }
                    radii_up = true;
                }
                const void *src;
                if (io.mem_in && in_pinned) {
                    src = io.mem_in + 3 * n * (size_t)f0;
                } else {
                    if (ensure_pinned(c, &c->stage_in, &c->stage_in_cap, in_bytes)) break;
                    if (io.mem_in) memcpy(c->stage_in, io.mem_in + 3 * n * (size_t)f0, in_bytes);
                    else if (!pread_all(io.fd_in, c->stage_in, in_bytes, io.in_header + (long long)(io.in_f32 ? 12 : 24) * (long long)n * f0)) {
                        ctx_fail(c, "could not read frames %lld..%lld of the frame file", f0, f0 + nf - 1);
                        break;
                    }
                    src = c->stage_in;
                }
                if (io.in_f32) {
                    if (hipMemcpyAsync(c->h_counts.p, src, in_bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) { ctx_fail(c, "host-to-device copy failed"); break; }
                    hipLaunchKernelGGL(k_widen_f32, dim3((unsigned)((3 * na + 255) / 256)), dim3(256), 0, c->stream,
                                       (const float *)c->h_counts.p, (double *)c->h_xyz.p, (long long)(3 * na));
                } else if (hipMemcpyAsync(c->h_xyz.p, src, in_bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
                    ctx_fail(c, "host-to-device copy failed");
                    break;
                }
                c->shared_radii = true;
                const int rb = run_batch(c, alg == 0, (double *)c->h_xyz.p, (double *)c->h_radii.p, offs.data(), nf, probe, resolution,
                                         alg == 1 ? tp.data() : nullptr, (double *)c->h_sasa.p, nullptr, (double *)c->h_totals.p);
                c->shared_radii = false;
                if (rb) break;
                double *dst_tot = io.totals_mem ? io.totals_mem + f0 : nullptr, *dst_sasa = io.sasa_mem ? io.sasa_mem + n * (size_t)f0 : nullptr;
                const bool staged = !(io.totals_mem && out_pinned);
                if (staged) {
                    if (ensure_pinned(c, &c->stage_out, &c->stage_out_cap, 8 * (size_t)nf + (want_sasa ? 8 * na : 0))) break;
                    dst_tot = (double *)c->stage_out;
                    dst_sasa = want_sasa ? (double *)c->stage_out + nf : nullptr;
                }
                bool ok = hipMemcpyAsync(dst_tot, c->h_totals.p, 8 * (size_t)nf, hipMemcpyDeviceToHost, c->stream) == hipSuccess;
                if (ok && want_sasa) ok = hipMemcpyAsync(dst_sasa, c->h_sasa.p, 8 * na, hipMemcpyDeviceToHost, c->stream) == hipSuccess;
                if (!ok) { ctx_fail(c, "device-to-host copy failed"); break; }
                if (hipStreamSynchronize(c->stream) != hipSuccess) { ctx_fail(c, "stream synchronize failed"); break; }
                if (staged) {
                    if (io.totals_mem) memcpy(io.totals_mem + f0, dst_tot, 8 * (size_t)nf);
                    if (io.sasa_mem) memcpy(io.sasa_mem + n * (size_t)f0, dst_sasa, 8 * na);
                    if (io.fd_totals >= 0 && !pwrite_all(io.fd_totals, dst_tot, 8 * (size_t)nf, 8 * f0)) { ctx_fail(c, "could not write the totals file"); break; }
                    if (io.fd_sasa >= 0 && !pwrite_all(io.fd_sasa, dst_sasa, 8 * na, 8 * (long long)n * f0)) { ctx_fail(c, "could not write the per-atom file"); break; }
                }
                if (io.fd_done >= 0) { /* results first, then the record: a shard is listed only when its numbers are on disk */
                    if ((io.fd_totals >= 0 && fdatasync(io.fd_totals) != 0) || (io.fd_sasa >= 0 && fdatasync(io.fd_sasa) != 0)) {
                        ctx_fail(c, "could not flush the result files: the shard is not listed as done"); break;
                    }
                    char line[96];
                    const int len = snprintf(line, sizeof line, "shard %lld %lld %d\n", k, f0, nf);
                    std::lock_guard<std::mutex> lk(done_mu);
                    if (write(io.fd_done, line, (size_t)len) != len || fdatasync(io.fd_done) != 0) { ctx_fail(c, "could not append to the done-list"); break; }
                }
                io.done[(size_t)k] = 1;
                rc = 0;
            } while (0);
            if (rc) {
                c->shared_radii = false;
                (void)hipStreamSynchronize(c->stream);
                snprintf(errs[id].data(), 256, "%s", c->err[0] ? c->err : "trajectory shard failed");
                failed = 1;
                break;
            }
        }
        pool_put(c);
    };
    std::vector<std::thread> th;
    for (int k = 1; k < n_lanes; ++k) th.emplace_back(lane, k);
    lane(0);
    for (auto &t : th) t.join();
    if (failed.load()) {
        for (int k = 0; k < n_lanes; ++k)
            if (errs[k][0]) return set_err(err_out, err_len, errs[k].data());
        return set_err(err_out, err_len, "trajectory run failed");
    }
    return stopped.load() ? 1 : 0;
}
}

extern "C" int freesasa_gpu_trajectory(const double *xyz_frames, const double *radii, int n_atoms, int n_frames,
                                       int alg, double probe, int resolution, int frames_per_batch,
                                       double *totals_out, double *sasa_out, int device, char *err_out, int err_len)
{
    if (err_out && err_len > 0) err_out[0] = 0;
    if (!xyz_frames || !radii || !totals_out) return set_err(err_out, err_len, "null argument");
    if (n_atoms <= 0 || n_frames <= 0) return set_err(err_out, err_len, "n_atoms and n_frames must be > 0");
    if (alg != 0 && alg != 1) return set_err(err_out, err_len, "unknown algorithm");
    if (resolution <= 0) return set_err(err_out, err_len, "resolution must be > 0");
    if (freesasa_gpu_device_count() <= 0)
        return set_err(err_out, err_len, "no HIP device available: libfreesasa_amd has no CPU path");
    if (frames_per_batch <= 0) frames_per_batch = (int)(1250000 / n_atoms) + 1;
    if (frames_per_batch > n_frames) frames_per_batch = n_frames;
    if ((long long)frames_per_batch * n_atoms > (1LL << 30)) return set_err(err_out, err_len, "batch too large");
    TrajIO io;
    io.mem_in = xyz_frames; io.totals_mem = totals_out; io.sasa_mem = sasa_out;
    return traj_run(io, radii, n_atoms, n_frames, alg, probe, resolution, frames_per_batch, 0, 0, device, err_out, err_len) < 0 ? -1 : 0;
}

/* Frame file -> result files, resumable.  frames_path: raw little-endian frames, frame f = 3 * n_atoms values
 * (x1, y1, z1, x2, ...) of type double (frames_f32 = 0) or float (1) at byte header_bytes + f * frame size.
 * totals_path: one double per frame at byte 8 * f; sasa_path (may be NULL): n_atoms doubles per frame at byte
 * 8 * n_atoms * f.  done_path (may be NULL): the done-list — a text file, first line = the run's parameters, then
 * one line "shard <index> <first frame> <frames>" per finished shard, appended after the shard's results are on
 * disk.  A call that finds a done-list with the same parameters skips its shards (results stay as they are in the
 * result files); with different parameters it fails rather than mix two runs.  max_new_shards > 0 stops after
 * that many shards (returns 1: incomplete; used by the tests to interrupt a run).
 * n_frames <= 0: as many whole frames as the file holds.  Returns 0 done, 1 incomplete, -1 error. */
extern "C" int freesasa_gpu_trajectory_file(const char *frames_path, int frames_f32, long long header_bytes, const double *radii,
                                            int n_atoms, long long n_frames, int alg, double probe, int resolution,
                                            int frames_per_batch, const char *totals_path, const char *sasa_path,
                                            const char *done_path, long long max_new_shards, int device,
                                            long long *frames_total_out, char *err_out, int err_len)
{
    if (err_out && err_len > 0) err_out[0] = 0;
    if (!frames_path || !radii || !totals_path) return set_err(err_out, err_len, "null argument");
    if (n_atoms <= 0 || header_bytes < 0) return set_err(err_out, err_len, "bad argument");
    if (alg != 0 && alg != 1) return set_err(err_out, err_len, "unknown algorithm");
    if (resolution <= 0) return set_err(err_out, err_len, "resolution must be > 0");
    if (freesasa_gpu_device_count() <= 0)
        return set_err(err_out, err_len, "no HIP device available: libfreesasa_amd has no CPU path");
    TrajIO io;
    int ret = -1;
    do {
        io.fd_in = open(frames_path, O_RDONLY);
        if (io.fd_in < 0) { set_err(err_out, err_len, "cannot open the frame file"); break; }
        struct stat st;
        if (fstat(io.fd_in, &st) != 0) { set_err(err_out, err_len, "cannot stat the frame file"); break; }
        const long long frame_bytes = (frames_f32 ? 12LL : 24LL) * n_atoms;
        const long long in_file = ((long long)st.st_size - header_bytes) / frame_bytes;
        if (n_frames <= 0) n_frames = in_file;
        if (n_frames <= 0 || n_frames > in_file) { set_err(err_out, err_len, "the frame file holds fewer frames than asked for"); break; }
        if (frames_total_out) *frames_total_out = n_frames;
        if (frames_per_batch <= 0) frames_per_batch = (int)(1250000 / n_atoms) + 1;
        if (frames_per_batch > n_frames) frames_per_batch = (int)n_frames;
        if ((long long)frames_per_batch * n_atoms > (1LL << 30)) { set_err(err_out, err_len, "batch too large"); break; }
        io.in_f32 = frames_f32 ? 1 : 0; io.in_header = header_bytes;
        const long long n_shards = (n_frames + frames_per_batch - 1) / frames_per_batch;
        io.done.assign((size_t)n_shards, 0);
        unsigned long long hr = 1469598103934665603ULL; /* FNV-1a over the radii */
        for (size_t q = 0; q < 8 * (size_t)n_atoms; ++q) hr = (hr ^ ((const unsigned char *)radii)[q]) * 1099511628211ULL;
        char head[384];
        snprintf(head, sizeof head, "freesasa_amd trajectory done-list v2 n_atoms=%d n_frames=%lld frames_per_batch=%d alg=%d resolution=%d probe=%.17g f32=%d "
                 "header_bytes=%lld frames_size=%lld frames_mtime=%lld.%09ld radii=%016llx\n",
                 n_atoms, n_frames, frames_per_batch, alg, resolution, probe, io.in_f32, header_bytes, (long long)st.st_size,
                 (long long)st.st_mtim.tv_sec, (long)st.st_mtim.tv_nsec, hr);
        bool resume = false;
        if (done_path) {
            FILE *fp = fopen(done_path, "r");
            if (fp) {
                char line[384];
                if (fgets(line, sizeof line, fp)) {
                    if (strcmp(line, head) != 0) { fclose(fp); set_err(err_out, err_len, "the done-list belongs to a run with other parameters, radii or frame file"); break; }
                    resume = true;
                    long long k, f0; int nf;
                    while (fgets(line, sizeof line, fp))
                        if (sscanf(line, "shard %lld %lld %d", &k, &f0, &nf) == 3 && k >= 0 && k < n_shards && f0 == k * frames_per_batch &&
                            line[strlen(line) - 1] == '\n') /* (a record cut short by a crash does not count) */
                            io.done[(size_t)k] = 1;
                }
                fclose(fp);
            }
        }
        io.fd_totals = open(totals_path, resume ? O_WRONLY | O_CREAT : O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (io.fd_totals < 0) { set_err(err_out, err_len, "cannot open the totals file"); break; }
        if (sasa_path) {
            io.fd_sasa = open(sasa_path, resume ? O_WRONLY | O_CREAT : O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (io.fd_sasa < 0) { set_err(err_out, err_len, "cannot open the per-atom file"); break; }
        }
        if (done_path) {
            io.fd_done = open(done_path, resume ? O_WRONLY | O_APPEND : O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (io.fd_done < 0) { set_err(err_out, err_len, "cannot open the done-list"); break; }
            if (!resume && write(io.fd_done, head, strlen(head)) != (ssize_t)strlen(head)) { set_err(err_out, err_len, "cannot write the done-list"); break; }
        }
        ret = traj_run(io, radii, n_atoms, n_frames, alg, probe, resolution, frames_per_batch, 0, max_new_shards, device, err_out, err_len);
    } while (0);
    if (io.fd_in >= 0) close(io.fd_in);
    if (io.fd_totals >= 0) close(io.fd_totals);
    if (io.fd_sasa >= 0) close(io.fd_sasa);
    if (io.fd_done >= 0) close(io.fd_done);
    return ret;
}
