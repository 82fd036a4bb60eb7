/*
 * gpu_kernels.hip — every __global__ wrapper around the phase functions of sasa_kernels.h and lr2_kernels.h, and the
 * kl_* launchers the host side of the engine calls (engine_internal.h).  The only translation unit of the library
 * that holds device code.  gfx950 only.
 */
#include <hip/hip_runtime.h>

#include <math.h>
#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sasa_kernels.h"
#ifdef SASA_PHASE_TIMING /* dev only (tools/build_variant.sh X -DSASA_PHASE_TIMING): where a wave's time per tile goes */
__device__ unsigned long long g_phase_clock[16];
#define LR2_MARK_BEGIN unsigned long long lr2_last_ = wall_clock64(); if (lane == 0 && !a.work_items && ((p0 / a.TA) & 255) == 0) atomicAdd(&g_phase_clock[15], 1ULL)
#define LR2_MARK(k) do { if (lane == 0 && !a.work_items && ((p0 / a.TA) & 255) == 0) { const unsigned long long now_ = wall_clock64(); \
        atomicAdd(&g_phase_clock[(k)], now_ - lr2_last_); lr2_last_ = now_; } } while (0)
#endif
#include "engine_internal.h"

using namespace sasa;

/* ------------------------------------------------------------------ kernels */

/* K1 + K2 of the general cell sort in ONE launch (round 6; until then bounds, grids, cell bases and the clearing of
 * the histogram were four): workgroups [0, n_chunks) reduce the bounds of their chunk of <= SASA_BOUNDS_CHUNK atoms,
 * every workgroup - these and the extra ones the launch adds for the purpose - clears its share of the cell histogram
 * (the whole capacity of the table: its used length is only known at the end of this kernel), and the workgroup that
 * finishes LAST (a ticket behind a device-scope fence) derives every structure's grid from the chunk bounds and scans
 * the cells per structure into the structures' cell bases. */
__global__ __launch_bounds__(SASA_PIPE_B) void k_prep_general(PipeArgs a)
{
    __shared__ double red[7 * SASA_PIPE_B], red2[7 * 16];
    __shared__ int last;
    {
        const Int4 z = {0, 0, 0, 0};
        const long long words = a.zero_n >> 2; /* (whole 16-byte words; the tail below) */
        for (long long w = (long long)blockIdx.x * SASA_PIPE_B + threadIdx.x; w < words; w += (long long)gridDim.x * SASA_PIPE_B)
            ((Int4 *)a.cell_start)[w] = z;
        if (blockIdx.x == 0 && threadIdx.x < (a.zero_n & 3)) a.cell_start[(words << 2) + threadIdx.x] = 0;
    }
    if ((int)blockIdx.x < a.n_chunks) {
        bounds_phase0(a, red, blockIdx.x, threadIdx.x, SASA_PIPE_B);
        __syncthreads();
        bounds_phase1(red, red2, threadIdx.x, SASA_PIPE_B);
        __syncthreads();
        bounds_phase2(a, red2, blockIdx.x, threadIdx.x);
    }
    __threadfence(); /* this workgroup's chunk bounds and zeros before its ticket */
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&a.status[ST_TICKET_PREP], 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence(); /* every other workgroup's bounds after their tickets */
    for (int s0 = 0; s0 < a.n_structs; s0 += SASA_PIPE_B / SASA_GRID_GROUP) { /* sixteen structures at a time, sixteen threads each */
        __syncthreads();
        grid_phase0(a, red, s0, threadIdx.x);
        __syncthreads();
        grid_phase1(a, red, s0, threadIdx.x);
    }
    __syncthreads();
    long long *part = (long long *)red, *part2 = (long long *)red2;
    cellbase_phase0(a, part, threadIdx.x, SASA_PIPE_B);
    __syncthreads();
    cellbase_phase1(part, part2, threadIdx.x, SASA_PIPE_B);
    __syncthreads();
    cellbase_phase1b(a, part2, threadIdx.x);
    __syncthreads();
    cellbase_phase2(a, part, part2, threadIdx.x, SASA_PIPE_B);
}

/* Everything behind K2 is launched without the host having seen K2's result (no readback in the middle of the
 * pipeline: the next batch of a driver loop can be enqueued behind this one).  A batch that turned out to be in
 * error (non-finite input, grid too big) or to need a larger cell table than was allocated does nothing from here
 * on - uniformly, first thing in every kernel - and the host, which reads the status once at the end, reports the
 * error or redoes the batch with the table K2 asked for. */
#define PIPE_GATE(st) do { if ((st)[ST_ERROR] | (st)[ST_RETRY]) return; } while (0)

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
/* n = total cells and the number of scan blocks follow from K2's total on the device; the launch is sized for
 * the table's capacity, workgroups whose ticket lies beyond the end leave at once */
__device__ __forceinline__ int scan_blocks(long long n)
{
    return (int)((n + 1 + (long long)SASA_PIPE_B * SASA_SCAN_ITEMS - 1) / ((long long)SASA_PIPE_B * SASA_SCAN_ITEMS));
}
__global__ __launch_bounds__(SASA_PIPE_B) void k_scan(PipeArgs a)
{
    __shared__ int part[SASA_PIPE_B], part2[SASA_SCAN_GROUP], blk_s, before;
    ScanRegs r;
    PIPE_GATE(a.status);
    const long long n = *cell_total(a);
    if (threadIdx.x == 0) blk_s = scan_take_block(a);
    __syncthreads();
    const int blk = blk_s;
    if (blk >= scan_blocks(n)) return;
    scan_phase0(a, n, part, blk, threadIdx.x, SASA_PIPE_B, r);
    __syncthreads();
    scan_phase1(part, part2, threadIdx.x);
    __syncthreads();
    scan_phase2(a, part2, &before, blk, threadIdx.x);
    __syncthreads();
    scan_phase3(a, n, part, part2, &before, blk, threadIdx.x, SASA_PIPE_B, r);
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
            /* (the counter lives in the status words, sasa_kernels.h ST_CELLS - addressed as an element of ncells[], the form this
               kernel had until round 6: through a pointer of its own the compiler, at the 128-register limit of a 1024-thread
               workgroup, put 125 registers into scratch and the kernel went from 0.28 to 0.50 ms per 1e7 atoms) */
            const long long base = (long long)atomicAdd((unsigned long long *)&a.ncells[a.cells_total_at], (unsigned long long)take);
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
    /* (TIER 2, the second launch, and TIER 1, the main launch of batches that lie far from the origin: the builds that walk to
       the slice planes of atoms beyond LR2_WALK_Z as the reference walks to them) */
    lr2_wave<RMAX, COVER, PAIRS, SHAPE, (TIER & 4) != 0, TIER == 2 || TIER == 1>(a, m, blockIdx.x, gridDim.x, lane, wg_max_nn);
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
hipError_t kl_lr2_main(int rmax, int grid, size_t lds, hipStream_t st, const Lr2Args &la)
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
    /* a batch most of whose tiles lie beyond LR2_WALK_Z (the context learnt it from the batch before: ST_FAR): the walking
       builds, generic shape, four rounds of pair records (they hold any pool of the main launch) */
    if (!la.hooks && la.walk) {
        const bool pr = lr2_pairs_shape(la.TA, la.ns);
        if (la.cover > 0) { if (pr) hipLaunchKernelGGL((k_lr2_tile<4, 1, 4, true, true>), dim3(grid), dim3(64), lds, st, la); else hipLaunchKernelGGL((k_lr2_tile<4, 1, 4, true, false>), dim3(grid), dim3(64), lds, st, la); }
        else { if (pr) hipLaunchKernelGGL((k_lr2_tile<4, 1, 4, false, true>), dim3(grid), dim3(64), lds, st, la); else hipLaunchKernelGGL((k_lr2_tile<4, 1, 4, false, false>), dim3(grid), dim3(64), lds, st, la); }
        return hipGetLastError();
    }
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

#ifndef SR_STOP_AFTER /* dev only (tools/build_variant.sh): leave the tile after phase k, for instruction / time attribution */
#define SR_STOP_AFTER 99
#endif
template <int B, bool GLOBAL, int TIER, bool CAPS = false>
#ifndef SR_WPE
#define SR_WPE 7 /* waves per SIMD the S&R kernel's registers are capped for (72 registers: seven 256-thread tiles per CU, what their LDS allows; the kernel is latency-bound - 67 % of its issue slots used - and measured on the MI355X, round 5, PDB entries x 251 / coil batch: uncapped (92 registers, 5 waves) 4.52 / 12.2 ms, 6 waves 4.16 / 11.2, 7 waves 4.04 / 11.0, 8 waves 4.49 / 11.4) */
#endif
#ifndef SR_CAPS_WPE
#define SR_CAPS_WPE 5 /* ... and the third arrangement's (sr_caps.h; measured, round 6, PDB entries x 251 / coil batch at 128 threads x 6 atoms: 4 waves 3.48 / 8.8 ms, 5 waves 3.24 / 7.7, 6 waves 3.48 / 8.5, 7 waves 4.6 / 10.9) */
#endif
__global__ __launch_bounds__(B) __attribute__((amdgpu_waves_per_eu(CAPS ? SR_CAPS_WPE : SR_WPE, CAPS ? SR_CAPS_WPE : SR_WPE))) void k_sr_tile(TileArgs a, int items)
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
        sr_phase_load(a, m, tile, tid, B);
        if constexpr (CAPS) sr_caps_clear(a, m, tile, tid, B);
        __syncthreads();
        if (SR_STOP_AFTER == 0) continue;
        sr_phase_neighbors(a, m, tile, tid, B);
        __syncthreads();
        if (SR_STOP_AFTER == 1) continue;
        if constexpr (CAPS) sr_caps_lists(a, m, tid); else sr_phase_lists(a, m, tid);
        __syncthreads();
        sr_report<GLOBAL>(a, m, tile, tid, wg_max_nn);
        if constexpr (CAPS) {
            /* third arrangement (sr_caps.h): a thread per neighbor record looks the record's cap masks up and ORs what it
               covers for sure into its atom's words; then the (neighbor, point) pairs that have to be asked the
               reference's way are listed, and asked */
            SrCapRegs regs;
            sr_caps_lookup_pass(a, m, tid, B, regs);
            __syncthreads();
            if (SR_STOP_AFTER == 12) continue;
            sr_caps_todo_pass(a, m, tid, B, items, regs);
            __syncthreads();
            if (SR_STOP_AFTER == 2) continue;
            sr_caps_exact(a, m, tid, B, items);
            __syncthreads();
            if (SR_STOP_AFTER == 3) continue;
            sr_caps_store(a, m, tile, tid);
            __syncthreads();
        } else {
            if constexpr (GLOBAL) sr_order_serial(a, m, tid); /* (the slab launch: segments of 4096 records) */
            else sr_order_wave(a, m, tid, B);                 /* (LDS launches: C <= 64 * SR_ORDER_RECS, see choose_cfg / mid_cfg) */
            __syncthreads();
            if (SR_STOP_AFTER == 2) continue;
            sr_phase_points(a, m, tile, tid, B);
            __syncthreads();
            if (SR_STOP_AFTER == 3) continue;
            sr_phase_points2(a, m, tid, B);
            __syncthreads();
            sr_phase_store(a, m, tile, tid);
            __syncthreads();
        }
    }
    tile_report_flush(a, tid, wg_max_nn);
}

/* ------------------------------------------------------------------ launchers (engine_internal.h) */

hipError_t kl_lr2_mid(int grid, size_t lds, hipStream_t st, const Lr2Args &la)
{
    hipLaunchKernelGGL((k_lr2_tile<LR2_RMAX_MID, 2, 3, true>), dim3(grid), dim3(64), lds, st, la);
    return hipGetLastError();
}

/* The first-generation tile kernels (Shrake-Rupley; Lee-Richards above 256 slices and as the last, slab-backed launch
   of lr2): workgroups of 64, 128 or 256 threads - what choose_cfg picks from.  Only the builds a launch path reaches are
   instantiated (until round 4 every (B, tier) pair also carried the 5-waves and the bucket-ranking builds). */
template <bool GLOBAL, int TIER>
static hipError_t launch_lr(const TileCfg &c, const TileArgs &t, int grid, size_t lds, hipStream_t s, bool bucket)
{
    if (c.B == 256)
        hipLaunchKernelGGL((k_lr_tile<256, GLOBAL, TIER, 4>), dim3(grid), dim3(256), lds, s, t, c.items);
    else if (c.B == 128)
        hipLaunchKernelGGL((k_lr_tile<128, GLOBAL, TIER, 4>), dim3(grid), dim3(128), lds, s, t, c.items);
    else if (c.B == 64) {
        if constexpr (!GLOBAL) {
            /* 160 KB of LDS per CU: more than 16 resident one-wave tiles only pay off with <= 96 VGPRs */
            if (TIER == 0 && lds * 17 <= 160 * 1024) { if constexpr (TIER == 0) hipLaunchKernelGGL((k_lr_tile<64, false, 0, 5>), dim3(grid), dim3(64), lds, s, t, c.items); }
            else if (bucket) hipLaunchKernelGGL((k_lr_tile<64, false, TIER, 4, true>), dim3(grid), dim3(64), lds, s, t, c.items);
            else hipLaunchKernelGGL((k_lr_tile<64, false, TIER, 4>), dim3(grid), dim3(64), lds, s, t, c.items);
        } else {
            hipLaunchKernelGGL((k_lr_tile<64, true, TIER, 4>), dim3(grid), dim3(64), lds, s, t, c.items);
        }
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}
template <bool GLOBAL, int TIER>
static hipError_t launch_sr(const TileCfg &c, const TileArgs &t, int grid, size_t lds, hipStream_t s)
{
    if constexpr (!GLOBAL) {
        /* the third arrangement (sr_caps.h) where its table exists and a wave's lanes hold an atom's list (C <= 128) */
        if (t.captab && t.tab && t.n_res <= SR_CAP_POINTS_MAX && sr_order_in_wave(c.cap_idx) && c.B >= 64) {
            if (c.B == 256)
                hipLaunchKernelGGL((k_sr_tile<256, false, TIER, true>), dim3(grid), dim3(256), lds, s, t, c.items);
            else if (c.B == 128)
                hipLaunchKernelGGL((k_sr_tile<128, false, TIER, true>), dim3(grid), dim3(128), lds, s, t, c.items);
            else if (c.B == 64)
                hipLaunchKernelGGL((k_sr_tile<64, false, TIER, true>), dim3(grid), dim3(64), lds, s, t, c.items);
            else return hipErrorInvalidValue;
            return hipGetLastError();
        }
    }
    if (c.B == 256)
        hipLaunchKernelGGL((k_sr_tile<256, GLOBAL, TIER>), dim3(grid), dim3(256), lds, s, t, c.items);
    else if (c.B == 128)
        hipLaunchKernelGGL((k_sr_tile<128, GLOBAL, TIER>), dim3(grid), dim3(128), lds, s, t, c.items);
    else if (c.B == 64)
        hipLaunchKernelGGL((k_sr_tile<64, GLOBAL, TIER>), dim3(grid), dim3(64), lds, s, t, c.items);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

static void allow_large_lds()
{
    static std::once_flag attr_once; /* allow > 64 KB of dynamic LDS */
    std::call_once(attr_once, [] {
        const void *fns[] = {(const void *)k_lr_tile<256, false, 0, 4>, (const void *)k_lr_tile<128, false, 0, 4>, (const void *)k_lr_tile<64, false, 0, 4>,
                             (const void *)k_lr_tile<256, false, 1, 4>, (const void *)k_lr_tile<128, false, 1, 4>, (const void *)k_lr_tile<64, false, 1, 4>,
                             (const void *)k_lr_tile<64, false, 0, 4, true>, (const void *)k_lr_tile<64, false, 1, 4, true>,
                             (const void *)k_sr_tile<256, false, 0>, (const void *)k_sr_tile<128, false, 0>, (const void *)k_sr_tile<64, false, 0>,
                             (const void *)k_sr_tile<256, false, 1>, (const void *)k_sr_tile<128, false, 1>, (const void *)k_sr_tile<64, false, 1>,
                             (const void *)k_sr_tile<256, false, 0, true>, (const void *)k_sr_tile<128, false, 0, true>, (const void *)k_sr_tile<64, false, 0, true>,
                             (const void *)k_sr_tile<256, false, 1, true>, (const void *)k_sr_tile<128, false, 1, true>, (const void *)k_sr_tile<64, false, 1, true>};
        for (const void *fn : fns) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
}

hipError_t kl_lr_tile(int tier, const TileCfg &c, const TileArgs &t, int grid, size_t lds, hipStream_t st, bool bucket)
{
    allow_large_lds();
    if (tier == 0) return launch_lr<false, 0>(c, t, grid, lds, st, bucket);
    if (tier == 1) return launch_lr<false, 1>(c, t, grid, lds, st, bucket);
    return launch_lr<true, 2>(c, t, grid, lds, st, false);
}
hipError_t kl_sr_tile(int tier, const TileCfg &c, const TileArgs &t, int grid, size_t lds, hipStream_t st)
{
    allow_large_lds();
    if (tier == 0) return launch_sr<false, 0>(c, t, grid, lds, st);
    if (tier == 1) return launch_sr<false, 1>(c, t, grid, lds, st);
    return launch_sr<true, 2>(c, t, grid, lds, st);
}

hipError_t kl_prep_fused(const PipeArgs &pa, hipStream_t st)
{
    hipLaunchKernelGGL(k_sort_struct, dim3(pa.n_structs), dim3(SORT_B), 0, st, pa);
    return hipGetLastError();
}
hipError_t kl_prep_general(const PipeArgs &pa, long long cells_cap, hipStream_t st)
{
    const int nblk_scan = (int)((cells_cap + 1 + (long long)SASA_PIPE_B * SASA_SCAN_ITEMS - 1) / ((long long)SASA_PIPE_B * SASA_SCAN_ITEMS));
    const int nblk_atoms = (pa.n_atoms + SASA_PIPE_B - 1) / SASA_PIPE_B;
    /* four launches (nine until round 6): bounds + grids + cell bases + clearing, count, chained scan, scatter.  The first
       gets one extra workgroup per 64 KB of histogram to clear (at most 2048) beside its one per chunk of atoms */
    long long zb = (pa.zero_n * 4 + 65535) / 65536;
    if (zb > 2048) zb = 2048;
    hipLaunchKernelGGL(k_prep_general, dim3((unsigned)(pa.n_chunks + zb)), dim3(SASA_PIPE_B), 0, st, pa);
    hipLaunchKernelGGL(k_count, dim3(nblk_atoms), dim3(SASA_PIPE_B), 0, st, pa);
    hipLaunchKernelGGL(k_scan, dim3(nblk_scan), dim3(SASA_PIPE_B), 0, st, pa);
    hipLaunchKernelGGL(k_scatter, dim3(nblk_atoms), dim3(SASA_PIPE_B), 0, st, pa);
    return hipGetLastError();
}

hipError_t kl_totals(const PipeArgs &pa, int n_chunks, int n_structs, const double *d_sasa, double *bpart, double *d_totals, hipStream_t st)
{
    hipLaunchKernelGGL(k_totals_chunks, dim3(n_chunks), dim3(SASA_TOT_B), 0, st, pa, d_sasa, bpart);
    hipLaunchKernelGGL(k_totals_structs, dim3((n_structs + 255) / 256), dim3(256), 0, st, pa, (const double *)bpart, d_totals);
    return hipGetLastError();
}
hipError_t kl_segment_sums(const double *d_sasa, const int64_t *d_seg, int n_segs, bool short_segments, double *d_out, hipStream_t st)
{
    /* residues: a thread per segment (strict atom order); long segments: a workgroup each */
    if (short_segments) hipLaunchKernelGGL(k_segsum_small, dim3((n_segs + 255) / 256), dim3(256), 0, st, d_sasa, d_seg, n_segs, d_out);
    else hipLaunchKernelGGL(k_totals, dim3(n_segs), dim3(SASA_TOT_B), 0, st, d_sasa, d_seg, n_segs, d_out);
    return hipGetLastError();
}
hipError_t kl_class_sums(const double *d_sasa, const unsigned char *d_class, const int64_t *d_offsets, int n_structs, double *d_out, hipStream_t st)
{
    hipLaunchKernelGGL(k_class_sums, dim3(n_structs), dim3(SASA_TOT_B), 0, st, d_sasa, d_class, d_offsets, d_out);
    return hipGetLastError();
}
hipError_t kl_residue_areas(const double *d_sasa, const unsigned char *d_class, const unsigned char *d_backbone, const int64_t *d_res_first,
                            const short *d_ref_row, const double *d_ref_table, double *d_abs, double *d_rel, int n_res, hipStream_t st)
{
    hipLaunchKernelGGL(k_residue_areas, dim3((n_res + 255) / 256), dim3(256), 0, st, d_sasa, d_class, d_backbone, d_res_first, d_ref_row,
                       d_ref_table, d_abs, d_rel, n_res);
    return hipGetLastError();
}
hipError_t kl_arc_kat(const double *d_arcs, const int *d_first, int n_sets, double *d_out, hipStream_t st)
{
    hipLaunchKernelGGL(k_lr2_arc_kat, dim3(1), dim3(64), 0, st, d_arcs, d_first, n_sets, d_out);
    return hipGetLastError();
}

/* fp32 trajectory frames widened on the device: an INPUT format, the arithmetic stays fp64 */
__global__ __launch_bounds__(256) void k_widen_f32(const float *in, double *out, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}
hipError_t kl_widen_f32(const float *d_in, double *d_out, long long n, hipStream_t st)
{
    hipLaunchKernelGGL(k_widen_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_in, d_out, n);
    return hipGetLastError();
}
/* ... and per-atom areas narrowed to fp32 for the trajectory drivers' compact output format (an OUTPUT format: computed in fp64) */
__global__ __launch_bounds__(256) void k_narrow_f64(const double *in, float *out, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}
hipError_t kl_narrow_f64(const double *d_in, float *d_out, long long n, hipStream_t st)
{
    hipLaunchKernelGGL(k_narrow_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_in, d_out, n);
    return hipGetLastError();
}

void kl_dump_phase_clocks(void)
{
#ifdef SASA_PHASE_TIMING
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_clock), sizeof h) == hipSuccess && h[15]) {
        static const char *names[8] = {"P0/load", "P1/neighbors", "P2/offsets", "P3/report|pairs", "P4/beta|screen", "P5/rank|queue", "P6/slices|arcs", "P7/store"};
        fprintf(stderr, "phase clocks (100 MHz ticks per sampled tile, thread 0, %llu tiles):", h[15]);
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.1f", names[k], (double)h[k] / (double)h[15]);
        fprintf(stderr, "\n");
        memset(h, 0, sizeof h);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_clock), h, sizeof h);
    }
#endif
}
