/*
 * sasa_kernels.h — device code of the MI355X SASA hot path (gfx950 / CDNA4).
 *
 * Every kernel is written as a sequence of barrier-separated PHASE functions
 *     phase(args, smem-view, tile/block id, tid)
 * so that the very same source is
 *   (a) wrapped in __global__ kernels by gpu_engine.hip (the product), and
 *   (b) driven thread-by-thread on a CPU by tests/emu/ (tests only, -DSASA_EMU) to check
 *       the kernel LOGIC in the GPU-less build container.  The emulation is never linked
 *       into libfreesasa_amd.so; the product has no CPU path.
 *
 * Pipeline for one batch of independent structures (xyz AoS fp64 + radii, CSR offsets):
 *   K1 bounds/grid per 4096-atom chunk: bounding box, max(R+probe); per structure -> cell grid (ref: src/nb.c:43-72, 543)
 *   K2 cell_base   exclusive scan of cells-per-structure
 *   K3 count       per atom: cell id, rank within cell (atomic)              (ref: src/nb.c:133-175)
 *   K4 scan        exclusive scan of the cell histogram (3 launches)
 *   K5 scatter     cell-sorted SoA copy x[],y[],z[],R[] (+ original index)
 *   K6 lr_tile / sr_tile   fused: neighbor discovery from the 27 surrounding cells into LDS,
 *                  then Lee-Richards slices or Shrake-Rupley test points   (ref: src/nb.c:458-522,
 *                  src/sasa_lr.c:270-408, src/sasa_sr.c:276-338); results scattered back.
 *                  Launched up to three times over one tiling (small LDS lists, larger LDS lists
 *                  for the tiles that overflowed, global slab for the rest).
 *   K7 totals      per-structure (or per-residue) sums
 *
 * Arithmetic contract: fp64 everywhere, compiled with -ffp-contract=off (fma only where written
 * explicitly).  The neighbor predicate and the whole S&R path keep the reference's operand order:
 * S&R counts and areas are bit-exact.  L&R reproduces the reference's arc union exactly for equal
 * arc end points, but computes cos(alpha) with reciprocals and acos/atan2/sqrt with device
 * routines: per-atom differences ~1e-13 A^2 (tolerance 1e-4 in north_star, 1e-9 in the tests).
 */
#ifndef SASA_KERNELS_H
#define SASA_KERNELS_H

#include <math.h>
#include <stdint.h>
#include <vector> /* (host side: sr_captab_build) */

#ifdef SASA_EMU
#define SASA_D inline
#define SASA_HD inline
namespace sasa_emu {
inline int atomic_add(int *p, int v) { int o = *p; *p = o + v; return o; }
inline int atomic_max(int *p, int v) { int o = *p; if (v > o) *p = v; return o; }
}
#define SASA_ATOMIC_ADD_LDS(p, v) sasa_emu::atomic_add((p), (v))
#define SASA_ATOMIC_ADD_GLB(p, v) sasa_emu::atomic_add((p), (v))
#define SASA_ATOMIC_MAX_GLB(p, v) sasa_emu::atomic_max((p), (v))
#define SASA_ATOMIC_MAX_LDS(p, v) sasa_emu::atomic_max((p), (v))
#define SASA_LOAD_ACQ64(p) (*(p))
#define SASA_STORE_REL64(p, v) (*(p) = (v))
#define SASA_RSQ(x) (1.0 / sqrt(x))
#define SASA_FMA_K(p, z, k) fma((p), (z), (k))
#define SASA_RCP(x) (1.0 / (x))
#define SASA_MIN(a, b) fmin((a), (b))
#define SASA_MAX(a, b) fmax((a), (b))
#define SASA_MIN_INTO(x, y) ((x) = fmin((x), (y)))
#define SASA_MAX_INTO(x, y) ((x) = fmax((x), (y)))
#define SASA_SHIFT_IN_LT1(w, c) (((w) << 1) | ((c) < 1.0 ? 1u : 0u))
#define SASA_OPAQUE(v) ((void)0)
#else
#define SASA_D __device__ __forceinline__
#define SASA_HD __host__ __device__ __forceinline__
#define SASA_ATOMIC_ADD_LDS(p, v) atomicAdd((p), (v))
#define SASA_ATOMIC_ADD_GLB(p, v) atomicAdd((p), (v))
#define SASA_ATOMIC_MAX_GLB(p, v) atomicMax((p), (v))
#define SASA_ATOMIC_MAX_LDS(p, v) atomicMax((p), (v))
/* a 64-bit word handed from one workgroup to another (the chained scan's block descriptors): device scope */
#define SASA_LOAD_ACQ64(p) __hip_atomic_load((p), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
#define SASA_STORE_REL64(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)
#define SASA_RSQ(x) __builtin_amdgcn_rsq(x)
/* p*z + k with the constant k held in an SGPR pair: one v_fma_f64 per Horner step and no VGPRs
   spent on coefficients (hipcc otherwise copies each coefficient into a VGPR pair and issues
   v_mov + v_fmac) */
__device__ __forceinline__ double sasa_fma_k(double p, double z, double k)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(z), "s"(k));
    return r;
}
#define SASA_FMA_K(p, z, k) sasa_fma_k((p), (z), (k))
#define SASA_RCP(x) __builtin_amdgcn_rcp(x)
/* min/max of two numbers that are never NaN: one v_min_f64 / v_max_f64.  fmin()/fmax() cost three —
   hipcc first quiets both operands (v_max_f64 x, x, x) because it cannot rule out signalling NaNs */
__device__ __forceinline__ double sasa_min(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double sasa_max(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#define SASA_MIN(a, b) sasa_min((a), (b))
#define SASA_MAX(a, b) sasa_max((a), (b))
/* x = min(x, y) / max(x, y) in x's own registers (a loop-carried value updated this way needs no copy at the loop's end) */
#define SASA_MIN_INTO(x, y) asm("v_min_f64 %0, %0, %1" : "+v"(x) : "v"(y))
#define SASA_MAX_INTO(x, y) asm("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(y))
/* (w << 1) | (c < 1.0): the compare leaves the per-lane result in vcc and an add-with-carry of w to
   itself shifts it in — two VALU instructions per screened neighbor instead of four */
__device__ __forceinline__ unsigned sasa_shift_in_lt1(unsigned w, double c)
{
    unsigned r;
    asm("v_cmp_gt_f64 vcc, 1.0, %1\n\tv_addc_co_u32 %0, vcc, %2, %2, vcc" : "=v"(r) : "v"(c), "v"(w) : "vcc");
    return r;
}
#define SASA_SHIFT_IN_LT1(w, c) sasa_shift_in_lt1((w), (c))
/* (w << 1) | (v < lim) */
__device__ __forceinline__ unsigned sasa_shift_in_lt(unsigned w, double v, double lim)
{
    unsigned r;
    asm("v_cmp_gt_f64 vcc, %3, %1\n\tv_addc_co_u32 %0, vcc, %2, %2, vcc" : "=v"(r) : "v"(v), "v"(w), "v"(lim) : "vcc");
    return r;
}
/* keeps a constant in a VGPR the optimizer cannot fold */
#define SASA_OPAQUE(v) asm("" : "+v"(v))
#endif

namespace sasa {

#define SASA_PI 3.14159265358979323846
#define SASA_TWOPI (2 * SASA_PI) /* ref: src/sasa_lr.c:25 */

/* status[] slots (device -> host) */
enum {
    ST_ERROR = 0,      /* first error code, 0 = ok */
    ST_OVF_TILES = 1,  /* tiles handed to the second (large-LDS) launch */
    ST_MAX_NN = 2,     /* max neighbors/atom seen */
    ST_OVF2_TILES = 3, /* tiles handed on to the third (slab) launch */
    ST_OCC_SUM = 4,    /* sum over sampled atoms of the number of atoms in their own cell ... */
    ST_OCC_N = 5,      /* ... and how many atoms were sampled: local density -> first launch shape */
    ST_OVF3_ATOMS = 6, /* L&R (lr2_kernels.h): atoms handed to the last (slab) launch */
    ST_RETRY = 7,      /* the batch needs more cells than the table was sized for: nothing after K2 ran, the host
                          redoes the batch with a table of status[ST_CELLS] cells (gpu_engine.hip) */
    ST_HIST = 8,       /* [64] tiles by neighbor records needed, bins of hist_bin_width(TA) */
    ST_SPLIT = 72,     /* [64] L&R (lr2_kernels.h): tiles redone as two halves, counted in 64 buckets */
    ST_FAR = 136,      /* L&R (lr2_kernels.h): tiles of the main launch with an atom beyond LR2_WALK_Z (slice planes walked as the reference walks them) */
    ST_TICKET_PREP = 137, /* general cell sort: workgroups of k_prep_general that are through (the last one derives the grids) */
    ST_TICKET_SCAN = 138, /* ... and the order in which the workgroups of the chained scan took their blocks */
    ST_CELLS = 140,    /* [2] one long long (8-byte aligned): cells of the whole batch - the general cell sort's total, the fused
                          sort's running counter - read back with the other words in ONE copy (round 6; until then a slot
                          behind ncells[] with a clearing fill and a device-to-host copy of its own per batch) */
    ST_WORDS = 142
};
enum {
    ERR_NONE = 0,
    ERR_BAD_RADIUS = 1,   /* cell size 2*max(R+probe) not > 0 (ref asserts, src/nb.c:544) */
    ERR_GRID_TOO_BIG = 2, /* nx*ny*nz over the limit (the reference would fail its malloc) */
    ERR_BAD_COORD = 3,    /* non-finite coordinate */
    ERR_NEIGHBOR_CAP = 4, /* an atom has more neighbors than the last (slab) launch holds */
    ERR_STACK_CAP = 5     /* more disjoint arcs in one slice than the last launch's stack holds */
};

struct GridS {
    double x0, y0, z0; /* lower corner = min - d/2          (ref: src/nb.c:61-66) */
    double d;          /* cell edge = 2*max(R+probe)        (ref: src/nb.c:543)   */
    int nx, ny, nz;    /* cells per axis                    (ref: src/nb.c:67-69) */
    int cell_base;     /* first cell of this structure in the batch-wide cell arrays */
};

/* ------------------------------------------------------------------------------------
 * K1..K5: cell-sort pipeline
 * ---------------------------------------------------------------------------------- */
struct __attribute__((aligned(16))) Quad { double x, y, z, w; };
/* what else a cell-sorted atom carries, one 16-byte record: its cell (batch-wide index | grid-border flags << 32,
   as cell_of), its index in the caller's order, its structure */
struct __attribute__((aligned(16))) SortIdx { long long cell; int orig, strct; };
struct PipeArgs {
    const double *xyz;      /* [3*n_atoms] x1,y1,z1,...   (ref layout: src/coord.h:26-38) */
    const double *radii;    /* [n_atoms] atom radii WITHOUT probe; shared_radii: ONE structure's radii, used by every structure */
    int shared_radii;       /* 1: radii[k] belongs to atom k of every structure (trajectory frames: one copy per device) */
    const int64_t *offsets; /* [n_structs+1] first atom of each structure */
    int n_structs;
    int n_atoms;
    double probe;
    long long max_cells; /* capacity of the cell arrays */
    /* per chunk of <= SASA_BOUNDS_CHUNK atoms (host-built from offsets) */
    int n_chunks;
    const int *chunk_struct;
    const int64_t *chunk_begin;
    const int *chunk_len;
    const int *struct_chunk0; /* [n_structs+1] first chunk of each structure */
    double *bpart;            /* [n_chunks*7] per-chunk min xyz, max xyz, max radius */
    /* per structure */
    GridS *grid;
    long long *ncells; /* [n_structs] cells of each structure (the batch's total: status[ST_CELLS], see cell_total) */
    long long cells_cap; /* > 0: cells the table cell_start[] has room for (K2 raises ST_RETRY beyond it) */
    /* per atom, original order */
    int *sid;     /* structure of atom i */
    long long *cell_of; /* batch-wide cell index | grid-border flags << 32 (CELL_*) */
    int occ_stride;     /* > 0: every occ_stride-th atom reports its cell's occupancy (ST_OCC_*) */
    int *rank;    /* arrival rank within the cell */
    /* per cell */
    int *cell_start; /* [total_cells+1]: histogram, then exclusive scan */
    /* The cell table in its compact form (k_sort_struct for the Lee-Richards tile kernel, round 4): per 32 cells one
       word {which of them hold atoms, occupied cells before them}, and the first atom of every OCCUPIED cell - 2 + 1.3
       bytes per atom on random coils where the dense table is 36 (9 cells per atom, 4 bytes each), written once and
       gathered from by every tile.  null: the dense table. */
    unsigned long long *cell_tbl; /* [total cells / 32 + 1] bits | occupied cells before the word << 32; a structure's cells start at a multiple of 32 */
    int *cell_first;              /* [n_atoms + n_structs] first atom of an occupied cell; structure s uses entries offsets[s] + s ...; behind its last: its end */
    int *blk_sums_unused_; /* (keeps the layout the fused sort kernel was tuned with: see the end of the struct) */
    /* per atom, cell-sorted order */
    Quad *sq; /* (x, y, z, radius + probe) (ref: src/sasa_lr.c:136, sasa_sr.c:144): one 32-byte record per atom - two
                 16-byte accesses where four arrays took four, one pointer where they took four */
    SortIdx *s_idx;
    int *status;
    /* (round 6; behind everything else: k_sort_struct sits at the 128-register limit of its 1024-thread workgroups, and
       with these in the middle of the struct the compiler put 125 registers into scratch - 0.28 -> 0.50 ms per 1e7 atoms) */
    long long *cells_total; /* = (long long *)(status + ST_CELLS): the batch's cell total / the fused sort's running counter */
    long long cells_total_at; /* ... and where it lies seen from ncells[]: &ncells[cells_total_at] == cells_total (k_sort_struct addresses it this way: see there) */
    unsigned long long *scan_desc; /* [scan blocks] the chained scan's block descriptors: value | state << 32 | epoch << 34 */
    int scan_epoch;  /* 1 .. 2^30 - 1, another one for every batch of a context: descriptors of earlier batches read as "not yet" */
    long long zero_n; /* words of cell_start[] the first kernel of the general pipeline clears (the table's capacity + 2) */
};

#define SASA_PIPE_B 256
SASA_D long long *cell_total(const PipeArgs &a) { return a.cells_total; } /* = (long long *)(status + ST_CELLS) */

/* K1a: one workgroup per CHUNK of at most SASA_BOUNDS_CHUNK atoms of one structure (a 200k-atom
 * structure is 49 chunks, not one serial workgroup).  red = LDS doubles [7][B]; the chunk's
 * min/max/max-radius go to bpart[chunk*7 ..]. */
#define SASA_BOUNDS_CHUNK 4096
SASA_D void bounds_phase0(const PipeArgs &a, double *red, int chunk, int tid, int B)
{
    const int s = a.chunk_struct[chunk];
    const int64_t b = a.chunk_begin[chunk], e = a.chunk_begin[chunk] + a.chunk_len[chunk];
    const int64_t b0 = a.shared_radii ? a.offsets[s] : 0; /* first atom of the chunk's structure */
    double lo0 = INFINITY, lo1 = INFINITY, lo2 = INFINITY;
    double hi0 = -INFINITY, hi1 = -INFINITY, hi2 = -INFINITY, rmax = 0; /* ref: src/nb.c:246 */
    int bad = 0;
    /* four atoms per trip, their sixteen loads in flight together (round 6: one atom per trip made a chunk of 4096 atoms
       sixteen dependent round trips to memory, ~25 us of the first kernel of the general cell sort) */
    for (int64_t i0 = b + tid; i0 < e; i0 += 4 * (int64_t)B) {
        double x[4], y[4], z[4], rr[4];
        for (int k = 0; k < 4; ++k) {
            const int64_t i = i0 + k * (int64_t)B < e ? i0 + k * (int64_t)B : i0; /* (beyond the chunk: the first atom again - harmless) */
            x[k] = a.xyz[3 * i]; y[k] = a.xyz[3 * i + 1]; z[k] = a.xyz[3 * i + 2];
            rr[k] = a.radii[a.shared_radii ? i - b0 : i];
        }
        for (int k = 0; k < 4; ++k) {
            lo0 = fmin(x[k], lo0); hi0 = fmax(x[k], hi0);
            lo1 = fmin(y[k], lo1); hi1 = fmax(y[k], hi1);
            lo2 = fmin(z[k], lo2); hi2 = fmax(z[k], hi2);
            rmax = fmax(rr[k] + a.probe, rmax);
            /* fmin/fmax drop NaN and the float-to-int conversion of a NaN is 0 on gfx950 (INT_MIN on x86): a
               non-finite coordinate or radius is flagged here, explicitly (v - v is NaN for NaN and +-inf) */
            if (!(x[k] - x[k] == 0) || !(y[k] - y[k] == 0) || !(z[k] - z[k] == 0)) bad = ERR_BAD_COORD;
            if (!(rr[k] - rr[k] == 0)) bad = bad ? bad : ERR_BAD_RADIUS;
        }
        for (int k = 0; k < 4; ++k)
            if (i0 + k * (int64_t)B < e) a.sid[i0 + k * (int64_t)B] = s;
    }
    if (bad) SASA_ATOMIC_MAX_GLB(&a.status[ST_ERROR], bad);
    red[0 * B + tid] = lo0; red[1 * B + tid] = lo1; red[2 * B + tid] = lo2;
    red[3 * B + tid] = hi0; red[4 * B + tid] = hi1; red[5 * B + tid] = hi2;
    red[6 * B + tid] = rmax;
}

/* the B values of each of the 7 quantities in two levels of 16 (B = 256): 7 x 16 threads fold 16 values each into
   red2[7][16], then 7 threads fold those (round 6: one level - 7 threads walking 255 values each, a chain of dependent
   LDS reads - took 25 of the 39 us the first kernel of the general cell sort ran on a 200 000-atom structure) */
SASA_D void bounds_phase1(const double *red, double *red2, int tid, int B)
{
    if (tid >= 7 * 16) return;
    const int q = tid >> 4, g = tid & 15, per = B / 16;
    const double *r = red + q * B + g * per;
    double v = r[0];
    for (int t = 1; t < per; ++t) v = q < 3 ? fmin(r[t], v) : fmax(r[t], v);
    red2[tid] = v;
}
SASA_D void bounds_phase2(const PipeArgs &a, const double *red2, int chunk, int tid)
{
    if (tid >= 7) return;
    double v = red2[tid * 16];
    for (int t = 1; t < 16; ++t) v = tid < 3 ? fmin(red2[tid * 16 + t], v) : fmax(red2[tid * 16 + t], v);
    a.bpart[(size_t)chunk * 7 + tid] = v;
}

/* K1b: a structure's chunks combined, its cell grid derived.  SIXTEEN threads share a structure (round 6: one thread
 * walked all of a structure's chunks - 49 dependent round trips for 200 000 atoms, the other ~30 us of that kernel):
 * thread g of the group folds chunks c0 + g, c0 + g + 16, ... into gpart[slot][g][7] (LDS, slot = the structure's
 * place among the 16 the workgroup does at a time), then one thread per structure folds the 16 and derives the grid. */
#define SASA_GRID_GROUP 16
SASA_D void grid_phase0(const PipeArgs &a, double *gpart, int s_first, int tid)
{
    const int slot = tid / SASA_GRID_GROUP, g = tid % SASA_GRID_GROUP, s = s_first + slot;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, rmax = 0;
    if (s < a.n_structs) {
        const int c0 = a.struct_chunk0[s], c1 = a.struct_chunk0[s + 1];
        for (int c = c0 + g; c < c1; c += SASA_GRID_GROUP) {
            const double *q = a.bpart + (size_t)c * 7;
            for (int k = 0; k < 3; ++k) { lo[k] = fmin(q[k], lo[k]); hi[k] = fmax(q[3 + k], hi[k]); }
            rmax = fmax(q[6], rmax);
        }
    }
    double *o = gpart + (size_t)tid * 7;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2]; o[6] = rmax;
}
SASA_D void grid_phase1(const PipeArgs &a, const double *gpart, int s_first, int tid)
{
    const int s = s_first + tid;
    if (tid >= SASA_PIPE_B / SASA_GRID_GROUP || s >= a.n_structs) return;
    GridS g;
    const int c0 = a.struct_chunk0[s], c1 = a.struct_chunk0[s + 1];
    if (c1 <= c0) { /* empty structure */
        g.x0 = g.y0 = g.z0 = 0; g.d = 1; g.nx = g.ny = g.nz = 0; g.cell_base = 0;
        a.grid[s] = g;
        a.ncells[s] = 0;
        return;
    }
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, rmax = 0;
    for (int c = 0; c < SASA_GRID_GROUP; ++c) {
        const double *q = gpart + ((size_t)tid * SASA_GRID_GROUP + c) * 7;
        for (int k = 0; k < 3; ++k) { lo[k] = fmin(q[k], lo[k]); hi[k] = fmax(q[3 + k], hi[k]); }
        rmax = fmax(q[6], rmax);
    }
    const double d = 2 * rmax; /* ref: src/nb.c:543 */
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
        if (!(fx * fy * fz <= (double)a.max_cells) || !(fx >= 1 && fy >= 1 && fz >= 1)) {
            err = ERR_GRID_TOO_BIG;
        } else {
            g.nx = (int)fx; g.ny = (int)fy; g.nz = (int)fz;
            nc = (long long)g.nx * g.ny * g.nz;
        }
    }
    if (err) {
        SASA_ATOMIC_MAX_GLB(&a.status[ST_ERROR], err);
        g.nx = g.ny = g.nz = 1; g.d = 1; g.x0 = g.y0 = g.z0 = 0;
        nc = 1; /* keep the rest of the pipeline in bounds; host discards results */
    }
    a.grid[s] = g;
    a.ncells[s] = nc;
}

/* K2: one workgroup; exclusive scan of ncells[] -> grid[].cell_base, total -> status[ST_CELLS].
 * part = LDS long long [B]. */
SASA_D void cellbase_phase0(const PipeArgs &a, long long *part, int tid, int B)
{
    const int per = (a.n_structs + B - 1) / B;
    long long s = 0;
    for (int k = tid * per; k < (tid + 1) * per && k < a.n_structs; ++k) s += a.ncells[k];
    part[tid] = s;
}
/* (the B partial sums in two levels of 16, as everywhere: one thread walking all 256 through LDS took ~15 us) */
SASA_D void cellbase_phase1(long long *part, long long *part2, int tid, int B)
{
    const int G = B / 16;
    if (tid >= 16) return;
    long long run = 0;
    for (int t = 0; t < G; ++t) { const long long v = part[tid * G + t]; part[tid * G + t] = run; run += v; }
    part2[tid] = run;
}
SASA_D void cellbase_phase1b(const PipeArgs &a, long long *part2, int tid)
{
    if (tid != 0) return;
    long long run = 0;
    for (int t = 0; t < 16; ++t) { const long long v = part2[t]; part2[t] = run; run += v; }
    *cell_total(a) = run;
    if (run > a.max_cells) SASA_ATOMIC_MAX_GLB(&a.status[ST_ERROR], (int)ERR_GRID_TOO_BIG);
    else if (a.cells_cap > 0 && run > a.cells_cap) a.status[ST_RETRY] = 1;
}
SASA_D void cellbase_phase2(const PipeArgs &a, const long long *part, const long long *part2, int tid, int B)
{
    const int per = (a.n_structs + B - 1) / B;
    long long run = part2[tid / (B / 16)] + part[tid];
    for (int k = tid * per; k < (tid + 1) * per && k < a.n_structs; ++k) {
        a.grid[k].cell_base = (int)run;
        run += a.ncells[k];
    }
}

/* which faces of its structure's grid an atom's cell lies on (saves the tile kernels the
 * divisions that would recover ix, iy, iz from the cell index) */
enum { CELL_X0 = 1, CELL_X1 = 2, CELL_Y0 = 4, CELL_Y1 = 8, CELL_Z0 = 16, CELL_Z1 = 32 };
SASA_D int cell_coord(double v, double v0, double d) { return (int)((v - v0) / d); } /* ref: src/nb.c:137-140 */
/* first cell-sorted atom of the first occupied cell >= x: what cell_start[x] holds in the dense table */
SASA_D int cell_rank(unsigned long long w, int x)
{
#ifdef SASA_EMU
    return (int)(w >> 32) + __builtin_popcount((unsigned)w & ((1u << (x & 31)) - 1u));
#else
    return (int)(w >> 32) + __popc((unsigned)w & ((1u << (x & 31)) - 1u));
#endif
}
SASA_D int cell_first_atom(const int *cell_start, const unsigned long long *tbl, const int *first, int x)
{
    if (!tbl) return cell_start[x];
    return first[cell_rank(tbl[x >> 5], x)];
}
/* nx and ny of the structure's grid, beside the six border flags in the high word of a sort record's cell (13 bits
 * each; 0: a grid of 8192 cells or more along x or y, look them up in grid[]): the L&R tile kernel finds an atom's
 * candidate rows from the record alone, one dependent load less at the start of every tile (lr2_pre_b) */
SASA_HD unsigned cell_pack_grid(int nx, int ny) { return (nx < 8192 && ny < 8192) ? ((unsigned)nx << 6) | ((unsigned)ny << 19) : 0u; }

/* K3: one thread per atom (original order): its cell, and its rank among the atoms of that cell.
 * Consecutive atoms of a chain mostly share a cell, so a RUN of equal cells inside the workgroup
 * takes one atomic (by its first thread, for the whole run) instead of one per atom.
 * cells/base = LDS int [B]. */
SASA_D void count_phase0(const PipeArgs &a, int *cells, int i, int tid)
{
    if (i >= a.n_atoms) { cells[tid] = -1; return; }
    const GridS g = a.grid[a.sid[i]];
    int ix = cell_coord(a.xyz[3 * i], g.x0, g.d);
    int iy = cell_coord(a.xyz[3 * i + 1], g.y0, g.d);
    int iz = cell_coord(a.xyz[3 * i + 2], g.z0, g.d);
    if (!(ix >= 0 && ix < g.nx && iy >= 0 && iy < g.ny && iz >= 0 && iz < g.nz)) {
        /* NaN/inf coordinate (or an errored grid): park the atom in cell 0, flag */
        if (a.status[ST_ERROR] == 0) SASA_ATOMIC_MAX_GLB(&a.status[ST_ERROR], (int)ERR_BAD_COORD);
        ix = iy = iz = 0;
    }
    const int c = g.cell_base + ix + g.nx * (iy + g.ny * iz); /* ref: src/nb.c:74-83 */
    const int fl = (ix == 0 ? CELL_X0 : 0) | (ix == g.nx - 1 ? CELL_X1 : 0) | (iy == 0 ? CELL_Y0 : 0) |
                   (iy == g.ny - 1 ? CELL_Y1 : 0) | (iz == 0 ? CELL_Z0 : 0) | (iz == g.nz - 1 ? CELL_Z1 : 0);
    a.cell_of[i] = (long long)((unsigned long long)(unsigned)c | ((unsigned long long)((unsigned)fl | cell_pack_grid(g.nx, g.ny)) << 32));
    cells[tid] = c;
}
SASA_D void count_phase1(const PipeArgs &a, const int *cells, int *base, int tid, int B)
{
    const int c = cells[tid];
    if (c < 0 || (tid > 0 && cells[tid - 1] == c)) return; /* not the head of a run */
    int len = 1;
    while (tid + len < B && cells[tid + len] == c) ++len;
    base[tid] = SASA_ATOMIC_ADD_GLB(&a.cell_start[c], len);
}
SASA_D void count_phase2(const PipeArgs &a, const int *cells, const int *base, int i, int tid)
{
    const int c = cells[tid];
    if (c < 0) return;
    int h = tid;
    while (h > 0 && cells[h - 1] == c) --h;
    a.rank[i] = base[h] + (tid - h);
}

/* K4: exclusive scan of cell_start[0..n) in place, n = total cells; cell_start[n] = total.
 * ONE launch (round 6; until then three: block sums, scan of the block sums by one workgroup, apply): the blocks are
 * chained ("decoupled look-back"): a workgroup takes the next block of SCAN_ITEMS x B cells from a ticket counter -
 * so a block's predecessors are always held by workgroups that are already running -, adds its cells up, publishes
 * the sum in the block's descriptor, then walks back over its predecessors' descriptors - a sum, or an inclusive
 * prefix, which ends the walk - and publishes its own inclusive prefix.  A descriptor is one 64-bit word (value |
 * state << 32 | epoch << 34) stored and loaded with release / acquire at device scope; the epoch is the context's
 * batch number, so the words need no clearing between batches (a word of an earlier batch reads as "not yet").
 * Each thread owns SCAN_ITEMS consecutive cells, moved as four 16-byte words and kept in registers between the
 * phases (a sparse batch has ~10 cells per atom: this kernel streams more bytes than the rest of the cell sort
 * together).  The B partial sums of a block are combined in two levels of SASA_SCAN_GROUP. */
#define SASA_SCAN_ITEMS 16
#define SASA_SCAN_GROUP 16 /* B == GROUP * GROUP */
struct __attribute__((aligned(16))) Int4 { int x, y, z, w; };
struct ScanRegs { Int4 v[SASA_SCAN_ITEMS / 4]; };
SASA_D int scan_load(const PipeArgs &a, long long n, int blk, int tid, int B, ScanRegs &r)
{
    const long long base = ((long long)blk * B + tid) * SASA_SCAN_ITEMS;
    if (base + SASA_SCAN_ITEMS <= n) {
        const Int4 *p = (const Int4 *)(a.cell_start + base);
        for (int k = 0; k < SASA_SCAN_ITEMS / 4; ++k) r.v[k] = p[k];
    } else {
        int *w = (int *)r.v;
        for (int k = 0; k < SASA_SCAN_ITEMS; ++k) w[k] = base + k < n ? a.cell_start[base + k] : 0;
    }
    int s = 0;
    for (int k = 0; k < SASA_SCAN_ITEMS / 4; ++k) s += (r.v[k].x + r.v[k].y) + (r.v[k].z + r.v[k].w);
    return s;
}
/* the block this workgroup scans (thread 0 takes the ticket; the kernel hands it to the others through LDS) */
SASA_D int scan_take_block(const PipeArgs &a) { return SASA_ATOMIC_ADD_GLB(&a.status[ST_TICKET_SCAN], 1); }
SASA_D void scan_phase0(const PipeArgs &a, long long n, int *part, int blk, int tid, int B, ScanRegs &r)
{
    part[tid] = scan_load(a, n, blk, tid, B, r);
}
/* exclusive scan inside each group, group totals to part2 */
SASA_D void scan_phase1(int *part, int *part2, int tid)
{
    if (tid >= SASA_SCAN_GROUP) return;
    int run = 0;
    for (int t = 0; t < SASA_SCAN_GROUP; ++t) {
        const int v = part[tid * SASA_SCAN_GROUP + t];
        part[tid * SASA_SCAN_GROUP + t] = run;
        run += v;
    }
    part2[tid] = run;
}
#define SASA_SCAN_SUM 1ULL  /* descriptor states: the block's own sum is known ... */
#define SASA_SCAN_INCL 2ULL /* ... the sum of every cell up to the block's end is known */
/* thread 0: exclusive scan of the group totals; the block's sum into its descriptor; the cells before the block
   (*before, LDS) from the predecessors' descriptors; the block's inclusive prefix into its descriptor */
SASA_D void scan_phase2(const PipeArgs &a, int *part2, int *before, int blk, int tid)
{
    if (tid != 0) return;
    int run = 0;
    for (int t = 0; t < SASA_SCAN_GROUP; ++t) { const int v = part2[t]; part2[t] = run; run += v; }
    const unsigned long long ep = (unsigned long long)(unsigned)a.scan_epoch << 34;
    int excl = 0;
    if (blk > 0) {
        SASA_STORE_REL64(&a.scan_desc[blk], ep | (SASA_SCAN_SUM << 32) | (unsigned long long)(unsigned)run);
        for (int p = blk - 1;; --p) {
            unsigned long long d;
            do { d = SASA_LOAD_ACQ64(&a.scan_desc[p]); } while ((d >> 34) != (ep >> 34) || ((d >> 32) & 3ULL) == 0); /* (the predecessor's workgroup is running: it took its ticket first) */
            excl += (int)(unsigned)d;
            if (((d >> 32) & 3ULL) == SASA_SCAN_INCL) break;
        }
    }
    SASA_STORE_REL64(&a.scan_desc[blk], ep | (SASA_SCAN_INCL << 32) | (unsigned long long)(unsigned)(excl + run));
    *before = excl;
}
SASA_D void scan_phase3(const PipeArgs &a, long long n, const int *part, const int *part2, const int *before, int blk, int tid, int B,
                        ScanRegs &r)
{
    const long long base = ((long long)blk * B + tid) * SASA_SCAN_ITEMS;
    int run = *before + part2[tid / SASA_SCAN_GROUP] + part[tid];
    if (base + SASA_SCAN_ITEMS <= n) {
        Int4 *p = (Int4 *)(a.cell_start + base);
        for (int k = 0; k < SASA_SCAN_ITEMS / 4; ++k) {
            const Int4 v = r.v[k];
            Int4 o;
            o.x = run; o.y = o.x + v.x; o.z = o.y + v.y; o.w = o.z + v.z;
            run = o.w + v.w;
            p[k] = o;
        }
    } else {
        const int *w = (const int *)r.v;
        for (int k = 0; k < SASA_SCAN_ITEMS; ++k)
            if (base + k < n) { a.cell_start[base + k] = run; run += w[k]; }
    }
    if (base <= n && n < base + SASA_SCAN_ITEMS) a.cell_start[n] = run; /* sentinel = n_atoms */
}

/* K5: one thread per atom: scatter into cell-sorted SoA. */
SASA_D void scatter_atom(const PipeArgs &a, int i)
{
    if (i >= a.n_atoms) return;
    const long long cf = a.cell_of[i];
    const int c = (int)(cf & 0xffffffffLL);
    const int p = a.cell_start[c] + a.rank[i];
    Quad v;
    v.x = a.xyz[3 * i]; v.y = a.xyz[3 * i + 1]; v.z = a.xyz[3 * i + 2];
    v.w = a.radii[a.shared_radii ? i - a.offsets[a.sid[i]] : i] + a.probe; /* ref: src/sasa_lr.c:136, src/sasa_sr.c:144 */
    a.sq[p] = v;
    if (a.occ_stride > 0 && i % a.occ_stride == 0) { /* ~256 density samples, only while the context has no demand history */
        SASA_ATOMIC_ADD_GLB(&a.status[ST_OCC_SUM], a.cell_start[c + 1] - a.cell_start[c]);
        SASA_ATOMIC_ADD_GLB(&a.status[ST_OCC_N], 1);
    }
    SortIdx si;
    si.cell = cf; si.orig = i; si.strct = a.sid[i];
    a.s_idx[p] = si;
}

/* ------------------------------------------------------------------------------------
 * K6: fused tile kernels
 * ---------------------------------------------------------------------------------- */
struct Arc { double s, e; };

struct TileArgs {
    const Quad *sq;
    const SortIdx *s_idx;
    const GridS *grid;
    const int *cell_start;
    const unsigned long long *cell_tbl; /* compact cell table (see PipeArgs), or null */
    const int *cell_first;
    int n_atoms;
    int n_tiles;
    int TA;      /* atoms per tile */
    int n_res;   /* L&R: slices per atom; S&R: test points */
    int tab;     /* 1: per-slice areas kept in LDS and summed in slice order */
    const double *unit_pts; /* S&R: [3*n_res] unit test points (host libm, ref: src/sasa_sr.c:56-90) */
    double *sasa; /* [n_atoms] original order */
    int *counts;  /* S&R: exposed points per atom (original order), may be null */
    /* capacities of the per-tile lists */
    int cap_idx; /* neighbor indices per atom */
    int pool;    /* neighbor records per tile */
    int lr;      /* 1: Lee-Richards records (z, D - Rj^2, 1/dij, beta); 0: Shrake-Rupley (x, y, z, Rj^2) */
    int ds;      /* spilled stack levels per thread (L&R) */
    /* overflow hand-off: a tile that does not fit this launch's capacities is appended to the
       next launch's work list (null in the last launch: error) */
    int *ovf_count;
    int *ovf_tiles;
    const int *work_tiles; /* tile ids to (re)do; null in the main launch (all tiles) */
    const int *work_count;
    /* third launch: lists live in a global slab, one slice per workgroup */
    char *slab;
    long long slab_stride;
    int *status;
    /* S&R, third arrangement (sr_caps.h): the table of cap masks for unit_pts, or null (the second arrangement) */
    const void *captab;
    int cap_n, cap_l; /* cube-map cells per face edge, intervals of g */
};

/* One neighbor record, 32 B, read with two ds_read_b128:
 *   L&R: x = z_j, y = D - R_j^2 (D = xd^2+yd^2), z = 1/sqrt(D), w = beta     (lr_record)
 *   S&R: x = x_j, y = y_j,                       z = z_j,       w = R_j^2
 * In the L&R pool every atom's list is padded to an even number of records with a dummy whose
 * cos(alpha) is huge (lr_padding: it never cuts an arc) so the screening loop runs 2 neighbors per trip. */
struct TileMem {
    double *ax, *ay, *az, *aR; /* [TA] tile atoms */
    int *acnt;                 /* [TA] neighbors found */
    int *aoff;                 /* [TA+1] offsets into pool */
    int *aexp;                 /* [TA] S&R exposed-point counters */
    int *flags;                /* [4] 0: tile overflow, 1: stack overflow, 2: max neighbor count */
    int *rowlo, *rowcnt;       /* [TA*9] candidate runs: first sorted position, length */
    double *contrib;           /* [TA*n_res] slice areas (tab mode) / [B] partials */
    int *idx;                  /* [TA*cap_idx] neighbor candidates (sorted positions) */
    double *tb;                /* [pool] beta of each pair before ranking */
    unsigned short *ki2;       /* [pool] L&R: list position of the pair behind each bucket-sorted beta */
    int *hist;                 /* [TA*LR_NBUCKET] L&R: beta histogram -> bucket cursors */
    Quad *pq;                  /* [pool] neighbor records */
    double *sb;                /* L&R bucket sort: betas in bucket order (scratch inside pq) */
    Arc *stack;                /* [ds][B] spilled components */
};

SASA_HD size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

/* Bytes of LDS needed.  In the LDS variant idx+tb (phases N..P) alias the stack (phase L). */
SASA_HD size_t tile_fixed_bytes(int TA, int items)
{
    return align16(sizeof(double) * 4 * TA) + align16(sizeof(int) * (3 * TA + 1 + 4 + 18 * TA)) +
           align16(sizeof(double) * items);
}
#ifndef LR_NBUCKET
#define LR_NBUCKET 32
#endif
SASA_HD size_t tile_union_bytes(int TA, int cap_idx, int pool, int ds, int B, bool lr)
{
    size_t u1 = lr ? align16(sizeof(int) * (size_t)TA * cap_idx) + align16(sizeof(double) * (size_t)pool) : 0; /* (S&R: no index lists since round 5) */
    size_t u2 = sizeof(Arc) * (size_t)ds * B;
    return u1 > u2 ? u1 : u2;
}
SASA_HD size_t tile_list_bytes(int TA, int cap_idx, int pool, int lr, int ds, int B)
{
    return tile_union_bytes(TA, cap_idx, pool, ds, B, lr != 0) + sizeof(Quad) * (size_t)pool;
}
/* slab of the third launch (no aliasing) */
SASA_HD size_t tile_slab_bytes(int TA, int cap_idx, int pool, int lr, int ds, int B)
{
    return align16(sizeof(int) * (size_t)TA * cap_idx) + align16(sizeof(double) * (size_t)pool) +
           align16(sizeof(Arc) * (size_t)ds * B) + sizeof(Quad) * (size_t)pool;
}

template <bool GLOBAL>
SASA_D TileMem tile_carve(const TileArgs &a, char *smem, int items, int B, int blk)
{
    TileMem m;
    char *p = smem;
    m.ax = (double *)p; m.ay = m.ax + a.TA; m.az = m.ay + a.TA; m.aR = m.az + a.TA;
    p += align16(sizeof(double) * 4 * a.TA);
    m.acnt = (int *)p; m.aoff = m.acnt + a.TA; m.aexp = m.aoff + a.TA + 1; m.flags = m.aexp + a.TA;
    m.rowlo = m.flags + 4; m.rowcnt = m.rowlo + 9 * a.TA;
    p += align16(sizeof(int) * (3 * a.TA + 1 + 4 + 18 * a.TA));
    m.contrib = (double *)p; p += align16(sizeof(double) * items);
    const size_t pw = align16(sizeof(double) * (size_t)a.pool);
    char *q;
    if (GLOBAL) {
        q = a.slab + (long long)blk * a.slab_stride;
        m.idx = (int *)q; q += align16(sizeof(int) * (size_t)a.TA * a.cap_idx);
        m.tb = (double *)q; q += pw;
        m.stack = (Arc *)q; q += align16(sizeof(Arc) * (size_t)a.ds * B);
    } else {
        m.idx = (int *)p;
        m.tb = (double *)(p + align16(sizeof(int) * (size_t)a.TA * a.cap_idx));
        m.stack = (Arc *)p;
        q = p + tile_union_bytes(a.TA, a.cap_idx, a.pool, a.ds, B, a.lr != 0);
    }
    m.pq = (Quad *)q;
    /* the bucket sort's scratch lives in the record array, which stays unwritten until the sort's
     * last phase (which no longer reads it): 2*pool + 128*TA + 8*pool bytes of the 32*pool */
    m.ki2 = (unsigned short *)q;
    m.hist = (int *)(q + align16(sizeof(unsigned short) * (size_t)a.pool));
    m.sb = (double *)((char *)m.hist + align16(sizeof(int) * (size_t)LR_NBUCKET * a.TA));
    return m;
}

SASA_D int tile_first_atom(const TileArgs &a, int tile) { return tile * a.TA; }
SASA_D int tile_atoms(const TileArgs &a, int tile)
{
    int n = a.n_atoms - tile * a.TA;
    return n < a.TA ? n : a.TA;
}

/* phase A: load the tile's atoms and, one thread per (atom, dy, dz), the 9 runs of
 * cell-sorted atoms that cover its 27 surrounding cells (x-adjacent cells are contiguous).
 * Every thread has one short chain s_struct/s_cell -> grid -> cell_start, and the chains of
 * all threads run concurrently, so the whole tile pays about three memory round trips. */
SASA_D void tile_phase_load(const TileArgs &a, TileMem &m, int tile, int tid, int B, bool bucket = false)
{
    const int na = tile_atoms(a, tile), p0 = tile_first_atom(a, tile);
    if (tid < a.TA) {
        if (tid < na) {
            const Quad v = a.sq[p0 + tid];
            m.ax[tid] = v.x; m.ay[tid] = v.y; m.az[tid] = v.z; m.aR[tid] = v.w;
        } else {
            m.ax[tid] = m.ay[tid] = m.az[tid] = 0; m.aR[tid] = 1;
        }
        m.acnt[tid] = 0;
        m.aexp[tid] = 0;
    }
    if (tid < 4) m.flags[tid] = 0;
    if (bucket)
        for (int t = tid; t < LR_NBUCKET * a.TA; t += B) m.hist[t] = 0;
    for (int t = tid; t < 9 * a.TA; t += B) {
        const int la = t / 9, r = t - 9 * la;
        int lo = 0, cnt = 0;
        if (la < na) {
            /* branch-free up to the last loads, so that s_struct and s_cell are fetched together and
               the chain is three round trips (s_struct -> grid -> cell_start), not four */
            const int p = p0 + la;
            const SortIdx si = a.s_idx[p];
            const long long cf = si.cell;
            const int c = (int)(cf & 0xffffffffLL), fl = (int)(cf >> 32);
            /* nx and ny ride in the record's flag word (cell_pack_grid): the chain is s_idx -> cell table, without the
               grid in between (round 6, as the L&R kernel since round 4); 0: a grid of 8192 cells or more along x or y */
            int nx = (fl >> 6) & 8191, ny = (fl >> 19) & 8191;
            if (nx == 0) { nx = a.grid[si.strct].nx; ny = a.grid[si.strct].ny; }
            const int dy = (r % 3) - 1, dz = (r / 3) - 1;
            const bool out = (dy < 0 && (fl & CELL_Y0)) || (dy > 0 && (fl & CELL_Y1)) ||
                             (dz < 0 && (fl & CELL_Z0)) || (dz > 0 && (fl & CELL_Z1));
            const int row = out ? c : c + nx * (dy + ny * dz); /* same ix, neighbouring (iy, iz) */
            const int x_lo = row - ((fl & CELL_X0) ? 0 : 1), x_hi = row + ((fl & CELL_X1) ? 0 : 1);
            const int s0 = cell_first_atom(a.cell_start, a.cell_tbl, a.cell_first, x_lo), s1 = cell_first_atom(a.cell_start, a.cell_tbl, a.cell_first, x_hi + 1);
            lo = out ? 0 : s0;
            cnt = out ? 0 : s1 - s0;
        }
        m.rowlo[t] = lo;
        m.rowcnt[t] = cnt;
    }
}

/* Contact test of the reference, operand for operand (ref: src/nb.c:483-492). */
SASA_D void nb_test(const TileArgs &a, TileMem &m, int la, int p, int q, double xi, double yi,
                    double zi, double ri, double xq, double yq, double zq, double rq)
{
    if (q == p) return;
    const double cut2 = (ri + rq) * (ri + rq);
    const double dx = xq - xi, dy = yq - yi, dz = zq - zi;
    if (dx * dx + dy * dy + dz * dz < cut2) {
        const int slot = SASA_ATOMIC_ADD_LDS(&m.acnt[la], 1);
        if (slot < a.cap_idx) m.idx[la * a.cap_idx + slot] = q;
    }
}

/* phase N: neighbor discovery.  SUB = B/TA lanes share one atom and stride over the
 * concatenation of its 9 candidate runs, four candidates per trip so that sixteen loads are in
 * flight before the first test (a coil atom's ~70 candidates then take one round trip to memory). */
#ifndef SASA_NB_UNROLL
#define SASA_NB_UNROLL 3
#endif
SASA_D void tile_phase_neighbors(const TileArgs &a, TileMem &m, int tile, int tid, int B)
{
    const int na = tile_atoms(a, tile), p0 = tile_first_atom(a, tile);
    const int SUB = B / a.TA;
    const int la = tid / SUB, sub = tid - la * SUB;
    if (la >= na) return;
#ifdef SASA_ABLATE_NB
    return;
#endif
    const int p = p0 + la;
    const double xi = m.ax[la], yi = m.ay[la], zi = m.az[la], ri = m.aR[la];
    const int *rl = m.rowlo + 9 * la, *rc = m.rowcnt + 9 * la;
    int total = 0;
    for (int r = 0; r < 9; ++r) total += rc[r];
    int r = 0, base = 0, cnt = rc[0];
    for (int f = sub; f < total; f += SASA_NB_UNROLL * SUB) {
        int q[SASA_NB_UNROLL];
        double x[SASA_NB_UNROLL], y[SASA_NB_UNROLL], z[SASA_NB_UNROLL], rq[SASA_NB_UNROLL];
        for (int j = 0; j < SASA_NB_UNROLL; ++j) {
            const int fj = f + j * SUB;
            if (fj < total) {
                while (fj >= base + cnt) { base += cnt; ++r; cnt = rc[r]; }
                q[j] = rl[r] + (fj - base);
            } else {
                q[j] = p; /* the atom itself: never a neighbor */
            }
        }
        for (int j = 0; j < SASA_NB_UNROLL; ++j) {
            const unsigned u = (unsigned)q[j]; /* 32-bit offset from a uniform base: no 64-bit address per load */
            { const Quad v = a.sq[u]; x[j] = v.x; y[j] = v.y; z[j] = v.z; rq[j] = v.w; }
        }
        for (int j = 0; j < SASA_NB_UNROLL; ++j) nb_test(a, m, la, p, q[j], xi, yi, zi, ri, x[j], y[j], z[j], rq[j]);
    }
}

/* phase O: offsets into the pool (each of the first TA threads sums the counts before it) */
SASA_D int pad4(int c, int lr) { return lr ? (c + 1) & ~1 : c; } /* L&R lists: even length */
SASA_D void tile_phase_offsets(const TileArgs &a, TileMem &m, int tid)
{
    if (tid >= a.TA) return;
    int off = 0;
    for (int k = 0; k < tid; ++k) off += pad4(m.acnt[k], a.lr);
    const int c = m.acnt[tid];
    m.aoff[tid] = off;
    if (c > a.cap_idx) m.flags[0] = 1;
    if (tid == a.TA - 1) {
        m.aoff[a.TA] = off + pad4(c, a.lr);
        if (off + pad4(c, a.lr) > a.pool) m.flags[0] = 1;
    }
    SASA_ATOMIC_MAX_LDS(&m.flags[2], c);
}

SASA_HD int hist_bin_width(int TA) { return 2 * TA + 2; }

/* first thing after the barrier that follows phase O: overflow -> next launch's work list */
template <bool GLOBAL>
SASA_D void tile_report(const TileArgs &a, TileMem &m, int tile, int tid, int &wg_max_nn)
{
    if (tid != 0) return;
    /* statistics must not serialise half a million tiles on one L2 atomic unit, nor put a global
       round trip into every tile: the largest neighbor count is kept in a register of thread 0 and
       pushed once per workgroup (tile_report_flush), the demand histogram samples 1 tile in 32 */
    if (m.flags[2] > wg_max_nn) wg_max_nn = m.flags[2];
    if (!a.work_tiles && (tile & 31) == 0) {
        const int need = m.aoff[a.TA] / hist_bin_width(a.TA);
        SASA_ATOMIC_ADD_GLB(&a.status[ST_HIST + (need < 63 ? need : 63)], 1);
    }
    if (m.flags[0]) {
        if (!a.ovf_tiles) {
            SASA_ATOMIC_MAX_GLB(&a.status[ST_ERROR], (int)ERR_NEIGHBOR_CAP);
        } else {
            const int w = SASA_ATOMIC_ADD_GLB(a.ovf_count, 1);
            a.ovf_tiles[w] = tile;
        }
    }
}

/* after a workgroup's last tile */
SASA_D void tile_report_flush(const TileArgs &a, int tid, int wg_max_nn)
{
    /* pushed only when it beats the value already there: same-address atomics serialise in L2 */
    if (tid == 0 && wg_max_nn > a.status[ST_MAX_NN]) SASA_ATOMIC_MAX_GLB(&a.status[ST_MAX_NN], wg_max_nn);
}

/* ---------------------------------------------------------------- fp64 helpers */

/* g ~ sqrt(x), h ~ 0.5/sqrt(x) for normal positive x: hardware reciprocal-sqrt seed, one
 * coupled Goldschmidt step, two residual corrections (~1 ulp; not correctly rounded). */
SASA_D void sqrt_rh(double x, double &g, double &h)
{
    const double y = SASA_RSQ(x);
    g = x * y;
    h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    r = fma(-h, g, 0.5);
    h = fma(h, r, h);
    d = fma(-g, g, x);
    g = fma(d, h, g);
}

/* g ~ sqrt(x) for normal positive x, relative error <= 4.1e-15: the hardware seed (v_rsq_f64 is
 * good to 2^-24.2 on gfx950, measured by tools/dev/seed_accuracy.hip) and one coupled Goldschmidt
 * step, which squares that error (x 1.5).  sqrt_rh spends seven more instructions on the last
 * bits and on h. */
SASA_D double sqrt_g(double x)
{
    const double y = SASA_RSQ(x);
    const double g = x * y, h = 0.5 * y;
    return fma(g, fma(-h, g, 0.5), g);
}

/* acos on (-1,1): fdlibm's range reduction (|x| <= 0.5: pi/2 - asin x; else 2 asin sqrt((1-|x|)/2),
 * reflected for x < 0), asin u = u + u z P(z), z = u^2 <= 0.25, P = degree-9 interpolant at
 * Chebyshev nodes (max relative error of asin 1.4e-14, fitted with mpmath; degree 11 reaches 5.6e-17
 * for two more instructions per arc, which the 1e-12-level conditioning of tangent arcs makes moot).  Branch-free; the
 * square root carries the sign of x, so both outer cases are 2 asin(u) (+ pi).
 * Relative error <= 2e-14 (polynomial 1.4e-14, square root 4e-15) plus 1e-16 absolute (the low
 * word of pi/2 is not carried). */
SASA_D double acos_fast(double x)
{
    const double ax = fabs(x);
    const bool big = ax > 0.5;
    const double zb = fma(ax, -0.5, 0.5); /* (1 - |x|)/2, exact; in (0, 0.25) for 0.5 < |x| < 1 */
    const double z = big ? zb : x * x;
    double p = 0x1.c93a92d53b4f1p-6;
    p = SASA_FMA_K(p, z, -0x1.815314c864b09p-9);
    p = SASA_FMA_K(p, z, 0x1.00d47e7966d94p-6);
    p = SASA_FMA_K(p, z, 0x1.b02442413f6bap-7);
    p = SASA_FMA_K(p, z, 0x1.1dc2ef640046fp-6);
    p = SASA_FMA_K(p, z, 0x1.6e72146fda29ep-6);
    p = SASA_FMA_K(p, z, 0x1.f1c81c59ea536p-6);
    p = SASA_FMA_K(p, z, 0x1.6db6d8e71341bp-5);
    p = SASA_FMA_K(p, z, 0x1.33333335a9cd6p-4);
    p = SASA_FMA_K(p, z, 0x1.5555555554f05p-3);
    const double s = copysign(sqrt_g(zb), x); /* only used when big */
    const double u = big ? s : x;             /* asin argument */
    const double t = fma(u * z, p, u);        /* asin(u) */
    const double pio2 = 0x1.921fb54442d18p+0, pi = 0x1.921fb54442d18p+1;
    /* |x| <= 0.5: pi/2 - t;  x > 0.5: 2 t = 2 asin(sqrt zb);  x < -0.5: pi + 2 t = pi - 2 asin(sqrt zb).
       One fma with selected operands (the products -1 t and 2 t are exact: the same single rounding as the sums),
       not three results behind two branches */
    const double m = big ? 2.0 : -1.0;
    const double off = big ? (x > 0 ? 0.0 : pi) : pio2;
    return fma(m, t, off);
}

/* acos on (-1,1) without a case distinction (the arc pass of lr2_kernels.h): acos|x| = 2 asin sqrt((1-|x|)/2) for
 * every |x| < 1, reflected for x < 0.  asin u = u + u z P(z), z = u^2 = (1-|x|)/2 <= 0.5: P an interpolant at
 * Chebyshev nodes on [0, 0.5] (fitted with mpmath), degree 14: max relative error of asin 1.5e-14 — what the
 * two-range acos_fast reaches with degree 9 on z <= 0.25, for four more fma and eleven instructions of selecting
 * operands less (31 -> 25 VALU instructions per arc).  ACOS2_DEG 13 / 12: 9.7e-14 / 6.3e-13, one / two fma less.
 * The sign of x enters as the factor +-2 of (asin u - pi/4) in one fma with pi/2. */
#ifndef ACOS2_DEG
#define ACOS2_DEG 12 /* 6.3e-13: with LR2_FAST_H2 the areas stay within ~1e-11 A^2 of the reference (measured: DESIGN.md; contract: 1e-4) */
#endif
/* z = (1 - |x|)/2 in (0, 0.5] and any number with the sign of x */
SASA_D double acos_fast2_z(double z, double x)
{
#if ACOS2_DEG == 14
    double p = 0x1.5983ba6d23362p-2;
    p = SASA_FMA_K(p, z, -0x1.cda34edad75c6p-1);
    p = SASA_FMA_K(p, z, 0x1.2a403f79f9e33p+0);
    p = SASA_FMA_K(p, z, -0x1.c6e3b97cdbe9fp-1);
    p = SASA_FMA_K(p, z, 0x1.dd61d178e44a1p-2);
    p = SASA_FMA_K(p, z, -0x1.3d430158670dbp-3);
    p = SASA_FMA_K(p, z, 0x1.a07b72b875c6fp-5);
    p = SASA_FMA_K(p, z, 0x1.159bde5b9fe00p-8);
    p = SASA_FMA_K(p, z, 0x1.e7736601fb531p-7);
    p = SASA_FMA_K(p, z, 0x1.1b0b34172c6c3p-6);
    p = SASA_FMA_K(p, z, 0x1.6e9d66a108da2p-6);
    p = SASA_FMA_K(p, z, 0x1.f1c686ee47a13p-6);
    p = SASA_FMA_K(p, z, 0x1.6db6dcb595fe6p-5);
    p = SASA_FMA_K(p, z, 0x1.3333333218b17p-4);
    p = SASA_FMA_K(p, z, 0x1.55555555557d9p-3);
#elif ACOS2_DEG == 10
    double p = 0x1.36a2e9ecec19dp-3;
    p = SASA_FMA_K(p, z, -0x1.fd83d62a27701p-3);
    p = SASA_FMA_K(p, z, 0x1.c42ef3b38479fp-3);
    p = SASA_FMA_K(p, z, -0x1.67afdf6ca0003p-4);
    p = SASA_FMA_K(p, z, 0x1.5ebf5d159b43ep-5);
    p = SASA_FMA_K(p, z, 0x1.8ca7e177a09d9p-7);
    p = SASA_FMA_K(p, z, 0x1.782204b9c3370p-6);
    p = SASA_FMA_K(p, z, 0x1.f129697301e6fp-6);
    p = SASA_FMA_K(p, z, 0x1.6db96e15f3a8ap-5);
    p = SASA_FMA_K(p, z, 0x1.33332f0b36fd2p-4);
    p = SASA_FMA_K(p, z, 0x1.555555566f3ffp-3);
#elif ACOS2_DEG == 11
    double p = 0x1.76020a8746ae9p-3;
    p = SASA_FMA_K(p, z, -0x1.63634279a6629p-2);
    p = SASA_FMA_K(p, z, 0x1.5b1ba556b0131p-2);
    p = SASA_FMA_K(p, z, -0x1.5cd68bf0c924cp-3);
    p = SASA_FMA_K(p, z, 0x1.2da9683adb436p-4);
    p = SASA_FMA_K(p, z, 0x1.7c628cf410d04p-15);
    p = SASA_FMA_K(p, z, 0x1.3d624404189a5p-6);
    p = SASA_FMA_K(p, z, 0x1.6b8acd1365601p-6);
    p = SASA_FMA_K(p, z, 0x1.f1efe6cd87ef6p-6);
    p = SASA_FMA_K(p, z, 0x1.6db64d758973dp-5);
    p = SASA_FMA_K(p, z, 0x1.333333f35cdcfp-4);
    p = SASA_FMA_K(p, z, 0x1.555555552a8ffp-3);
#elif ACOS2_DEG == 13
    double p = 0x1.174d39e43815ep-2;
    p = SASA_FMA_K(p, z, -0x1.5167956168ca9p-1);
    p = SASA_FMA_K(p, z, 0x1.8ec93aa6263cdp-1);
    p = SASA_FMA_K(p, z, -0x1.0f49ec92348f7p-1);
    p = SASA_FMA_K(p, z, 0x1.037f12ba48b93p-2);
    p = SASA_FMA_K(p, z, -0x1.10756f3cf669bp-4);
    p = SASA_FMA_K(p, z, 0x1.ca53e50c8b27dp-6);
    p = SASA_FMA_K(p, z, 0x1.79ec6d7065e80p-7);
    p = SASA_FMA_K(p, z, 0x1.20458b1d083b7p-6);
    p = SASA_FMA_K(p, z, 0x1.6e4abceb01895p-6);
    p = SASA_FMA_K(p, z, 0x1.f1c994c2364ffp-6);
    p = SASA_FMA_K(p, z, 0x1.6db6d52f65e8fp-5);
    p = SASA_FMA_K(p, z, 0x1.33333339625fap-4);
    p = SASA_FMA_K(p, z, 0x1.5555555554529p-3);
#elif defined(ACOS2_HORNER)
    double p = 0x1.c70b84f2604a4p-3;
    p = SASA_FMA_K(p, z, -0x1.eb330d405c37dp-2);
    p = SASA_FMA_K(p, z, 0x1.084859012fcc8p-1);
    p = SASA_FMA_K(p, z, -0x1.3a4ea43eb48adp-2);
    p = SASA_FMA_K(p, z, 0x1.1661c4e1fecccp-3);
    p = SASA_FMA_K(p, z, -0x1.6b69eecac872ep-6);
    p = SASA_FMA_K(p, z, 0x1.482102e85b292p-6);
    p = SASA_FMA_K(p, z, 0x1.1080015ff5fd3p-6);
    p = SASA_FMA_K(p, z, 0x1.6f7002236e00bp-6);
    p = SASA_FMA_K(p, z, 0x1.f1bcecaaa9600p-6);
    p = SASA_FMA_K(p, z, 0x1.6db6f971a2415p-5);
    p = SASA_FMA_K(p, z, 0x1.33333310a8a5ep-4);
    p = SASA_FMA_K(p, z, 0x1.555555555be1fp-3);
#else
    /* the same degree-12 polynomial, even and odd powers apart: two chains of six and five fma that run side by side
       where Horner's rule is one chain of twelve (a dependent fp64 instruction waits two issue slots for its operand,
       profiles/r03_ubench.txt: the arc pass is as much a matter of a wave's own critical path as of instruction count
       once fewer than all four waves of a SIMD are in it); one multiplication and one fma more */
    const double w = z * z;
    double pe = 0x1.c70b84f2604a4p-3, po = -0x1.eb330d405c37dp-2;
    pe = SASA_FMA_K(pe, w, 0x1.084859012fcc8p-1);  po = SASA_FMA_K(po, w, -0x1.3a4ea43eb48adp-2);
    pe = SASA_FMA_K(pe, w, 0x1.1661c4e1fecccp-3);  po = SASA_FMA_K(po, w, -0x1.6b69eecac872ep-6);
    pe = SASA_FMA_K(pe, w, 0x1.482102e85b292p-6);  po = SASA_FMA_K(po, w, 0x1.1080015ff5fd3p-6);
    pe = SASA_FMA_K(pe, w, 0x1.6f7002236e00bp-6);  po = SASA_FMA_K(po, w, 0x1.f1bcecaaa9600p-6);
    pe = SASA_FMA_K(pe, w, 0x1.6db6f971a2415p-5);  po = SASA_FMA_K(po, w, 0x1.33333310a8a5ep-4);
    pe = SASA_FMA_K(pe, w, 0x1.555555555be1fp-3);
    const double p = fma(po, z, pe);
#endif
    /* asin u = u (1 + z P(z)) with u = sqrt z from the seed's coupled Goldschmidt step (sqrt_g): u = g (1 + e), so
       asin u - pi/4 = g ((1 + e) q) - pi/4, q = 1 + z P: seven instructions behind the polynomial where forming u and
       u z first took eight (round 4; the two differ by a rounding or two of 1e-16) */
    const double q = fma(z, p, 1.0);
    const double y = SASA_RSQ(z);
    const double g = z * y, h = 0.5 * y;
    const double e = fma(-h, g, 0.5);
    const double pio4 = 0x1.921fb54442d18p-1, pio2 = 0x1.921fb54442d18p+0;
    const double t = SASA_FMA_K(g, fma(e, q, q), -pio4); /* asin(u) - pi/4 (absolute error 1e-16: of no account for an angle) */
    return SASA_FMA_K(copysign(2.0, x), t, pio2); /* x >= 0: 2 asin u;  x < 0: pi - 2 asin u */
}
SASA_D double acos_fast2(double x)
{
    return acos_fast2_z(fma(fabs(x), -0.5, 0.5), x); /* (1 - |x|)/2, exact up to the rounding of the fma */
}

/* atan2(y, x) for finite arguments: ONE division (hardware reciprocal seed + two Newton steps + a
 * residual correction) shared by both reductions — u = mn/mx, or (mn-mx)/(mn+mx) with a pi/4
 * offset when mn/mx > tan(pi/8) — then atan u = u + u s P(s), s = u^2 <= 0.1716, P a degree-10
 * interpolant at Chebyshev nodes (max relative error 5e-18, fitted with mpmath).  ~2 ulp.
 * Replaces ocml's atan2, whose register footprint (not its speed) capped the kernel at 4 waves
 * per SIMD. */
SASA_D double atan2_fast(double y, double x)
{
    const double ax = fabs(x), ay = fabs(y);
    const bool swap = ay > ax;
    const double mx = swap ? ay : ax, mn = swap ? ax : ay;
    if (!(mx > 0)) return 0.0;
    const bool red = mn > 0x1.a827999fcef32p-2 * mx; /* tan(pi/8) */
    const double num = red ? mn - mx : mn;
    const double den = red ? mn + mx : mx;
    double r = SASA_RCP(den);
    r = fma(fma(-den, r, 1.0), r, r);
    r = fma(fma(-den, r, 1.0), r, r);
    double u = num * r;
    u = fma(fma(-den, u, num), r, u);
    const double s2 = u * u;
    double p = -0x1.3a31b1c0fd3b7p-6;
    p = SASA_FMA_K(p, s2, 0x1.4162c02b1dda3p-5);
    p = SASA_FMA_K(p, s2, -0x1.a0999c632b6edp-5);
    p = SASA_FMA_K(p, s2, 0x1.dfe6497e96323p-5);
    p = SASA_FMA_K(p, s2, -0x1.10fa77b1a6d57p-4);
    p = SASA_FMA_K(p, s2, 0x1.3b1263064f6b9p-4);
    p = SASA_FMA_K(p, s2, -0x1.745d0b28a7e37p-4);
    p = SASA_FMA_K(p, s2, 0x1.c71c71853d7fap-4);
    p = SASA_FMA_K(p, s2, -0x1.2492492436201p-3);
    p = SASA_FMA_K(p, s2, 0x1.999999999934cp-3);
    p = SASA_FMA_K(p, s2, -0x1.5555555555555p-2);
    double a = fma(u * s2, p, u);
    const double pio4 = 0x1.921fb54442d18p-1, pio2 = 0x1.921fb54442d18p+0, pi = 0x1.921fb54442d18p+1;
    a = red ? a + pio4 : a;
    a = swap ? pio2 - a : a;
    a = x < 0 ? pi - a : a;
    return y < 0 ? -a : a;
}

/* beta = atan2(y, x) of a neighbor whose 1/sqrt(x^2 + y^2) = inv the pair record has already paid for (lr2_record):
 * no division.  With mx = max(|x|, |y|), mn = min: phi = atan(mn / mx) has sin(phi) = mn inv; for phi > pi/8
 * (mn > tan(pi/8) mx) take phi - pi/4, whose sine is (mn - mx) inv / sqrt 2.  asin s = s + s z P(z), z = s^2 <=
 * sin^2(pi/8) = 0.1464: degree-8 interpolant at Chebyshev nodes (max relative error 6e-16, fitted with mpmath).
 * ~3 ulp; 33 VALU instructions where atan2_fast takes 45 and a reciprocal. */
SASA_D double atan2_inv(double y, double x, double inv)
{
    const double ax = fabs(x), ay = fabs(y);
    const bool swap = ay > ax;
    const double mx = swap ? ay : ax, mn = swap ? ax : ay;
    /* (x = y = 0 comes with inv = 0: s = 0 and every select below leaves 0, as atan2(0, 0) - no branch for it) */
    const bool red = mn > 0x1.a827999fcef32p-2 * mx; /* tan(pi/8) */
    const double num = red ? mn - mx : mn;
    const double k = red ? inv * 0x1.6a09e667f3bcdp-1 : inv; /* 1/sqrt 2 */
    const double s = num * k;
    const double z = s * s;
    double p = 0x1.22f1f7591b110p-6;
    p = SASA_FMA_K(p, z, 0x1.2a3eff9205476p-7);
    p = SASA_FMA_K(p, z, 0x1.d6236bde6a953p-7);
    p = SASA_FMA_K(p, z, 0x1.1bbfea905019bp-6);
    p = SASA_FMA_K(p, z, 0x1.6e930536f6b5fp-6);
    p = SASA_FMA_K(p, z, 0x1.f1c6e5efcb3d9p-6);
    p = SASA_FMA_K(p, z, 0x1.6db6dbd204301p-5);
    p = SASA_FMA_K(p, z, 0x1.33333332ec66bp-4);
    p = SASA_FMA_K(p, z, 0x1.55555555555d9p-3);
    double a = fma(s * z, p, s);
    const double pio4 = 0x1.921fb54442d18p-1, pio2 = 0x1.921fb54442d18p+0, pi = 0x1.921fb54442d18p+1;
    a = red ? a + pio4 : a;
    a = swap ? pio2 - a : a;
    a = x < 0 ? pi - a : a;
    return y < 0 ? -a : a;
}

/* ---------------------------------------------------------------- Lee & Richards */

/* Angular bucket of beta in [0, 2pi]: monotone in beta, so bucket order is beta order. */
SASA_D int lr_bucket(double beta)
{
    const int b = (int)(beta * (LR_NBUCKET / SASA_TWOPI));
    return b < 0 ? 0 : (b > LR_NBUCKET - 1 ? LR_NBUCKET - 1 : b);
}

/* phase P1: beta = atan2(yd, xd) + pi for every (atom, neighbor) pair (ref: src/sasa_lr.c:337;
 * the reference recomputes it per slice, it only depends on the pair). */
SASA_D void lr_phase_beta(const TileArgs &a, TileMem &m, int tid, int B, bool bucket = false)
{
    if (m.flags[0]) return;
#ifdef SASA_ABLATE_PAIRS
    return;
#endif
    const int total = m.aoff[a.TA];
    for (int gp = tid; gp < total; gp += B) {
        int la = 0;
        while (m.aoff[la + 1] <= gp) ++la;
        const int k = gp - m.aoff[la];
        if (k >= m.acnt[la]) { m.tb[gp] = INFINITY; continue; } /* padding slot: ranks behind everything */
        const int q = m.idx[la * a.cap_idx + k];
        const double xd = a.sq[q].x - m.ax[la], yd = a.sq[q].y - m.ay[la]; /* ref: src/nb.c:445-448 */
        const double beta = atan2_fast(yd, xd) + SASA_PI;
        m.tb[gp] = beta;
        if (bucket) SASA_ATOMIC_ADD_LDS(&m.hist[la * LR_NBUCKET + lr_bucket(beta)], 1);
    }
}

/* The L&R record of neighbor q of an atom at (xi, yi): what the slice loop needs of the pair,
 * independent of the slice.  With A = Ri'^2, B = Rj'^2 = Rj^2 - (zj - z)^2, D = dij^2 the
 * reference's three geometric tests (src/sasa_lr.c:320-331) are exactly c >= 1 (no contact, j's
 * circle inside i's, or j does not reach the plane: B <= 0 gives c >= (A + D)/(2 Ri' dij) >= 1)
 * and c <= -1 (i inside j: slice buried) for c = (A + D - B)/(2 Ri' dij) (ref: :335)
 *   = ((zj - z)^2 + (A + E)) * (1/dij) * (1/(2 Ri')),  E = D - Rj^2,
 * so neither sqrt(Rj'^2) nor a division is needed per (pair, slice).  dij == 0: 1/dij = inf makes
 * c = +-inf, deciding inside/buried like the reference; 0*inf = NaN (the reference's acos(0/0),
 * two coincident equal circles) compares false -> no arc. */
SASA_D Quad lr_record_of(double xq, double yq, double zq, double rj, double xi, double yi, double beta)
{
    const double xd = xq - xi, yd = yq - yi; /* ref: src/nb.c:445-448 */
    const double D = xd * xd + yd * yd; /* = d_ij^2 (ref: src/nb.c:438) */
    double g = 0, h = 0;
    if (D > 0) sqrt_rh(D, g, h);
    Quad rec;
    rec.x = zq;
    rec.y = D - rj * rj;
    rec.z = D > 0 ? 2.0 * h : INFINITY; /* 1/d_ij */
    rec.w = beta;
    return rec;
}
SASA_D Quad lr_record(const TileArgs &a, int q, double xi, double yi, double beta)
{
    const Quad v = a.sq[q];
    return lr_record_of(v.x, v.y, v.z, v.w, xi, yi, beta);
}
/* padding slot of an odd-length list: a record whose c is huge, so it never cuts an arc */
SASA_D Quad lr_padding() { Quad d; d.x = 0; d.y = 1e300; d.z = 1; d.w = 0; return d; }

/* phase P2: rank each pair by beta inside its atom's list and write the pair record at its
 * sorted position.  Sorting by the arc mid-angle is what lets the slice loop merge arcs with
 * a stack instead of the reference's per-slice insertion sort. */
/* beta with its low 12 mantissa bits replaced by a list position (see lr_phase_rank); only the low
 * word changes.  `low` = 0xfff: handed in from a register the compiler cannot see through, so that
 * (lo & ~low) | pos is one v_and_or_b32 instead of an and + or with a literal. */
SASA_D double lr_rank_key(double beta, unsigned pos, unsigned low)
{
    unsigned w[2];
    memcpy(w, &beta, 8);
    w[0] = (w[0] & ~low) | pos; /* pos < 4096 */
    double key;
    memcpy(&key, w, 8);
    return key;
}
SASA_D void lr_phase_rank(const TileArgs &a, TileMem &m, int tid, int B)
{
    if (m.flags[0]) return;
#ifdef SASA_ABLATE_PAIRS
    return;
#endif
    const int total = m.aoff[a.TA];
    for (int gp = tid; gp < total; gp += B) {
        int la = 0;
        while (m.aoff[la + 1] <= gp) ++la;
        const int o = m.aoff[la], nn = m.acnt[la], k = gp - o;
        if (k >= nn) { m.pq[gp] = lr_padding(); continue; }
        /* the neighbor's coordinates are requested before the ranking loop and used after it */
        const unsigned q = (unsigned)m.idx[la * a.cap_idx + k];
        const Quad vq = a.sq[q];
        const double xq = vq.x, yq = vq.y, zq = vq.z, rq = vq.w;
        const double beta = m.tb[gp];
        int rank = 0;
        /* Order by (beta, list position) with ONE comparison per element: the list position replaces
           the low 12 bits of beta's mantissa (positions < 4096 = SASA_FB_CAP), which makes the keys
           distinct; two betas that agree in everything above (relative difference < 1e-12) belong to
           arcs that overlap around a common midpoint (alpha > 1e-8), and such arcs commute in the
           union.  Two betas per LDS read (o is even, so the pair is 16-byte aligned); the slot after
           an odd-length list holds +inf, whose key is a NaN and never compares below. */
        unsigned low = 0xfffu;
        SASA_OPAQUE(low);
        const double kme = lr_rank_key(beta, (unsigned)k, low);
        for (int t = 0; t < nn; t += 2) {
            const Arc bb = *(const Arc *)(m.tb + o + t);
            rank += lr_rank_key(bb.s, (unsigned)t, low) < kme ? 1 : 0;
            rank += lr_rank_key(bb.e, (unsigned)t + 1u, low) < kme ? 1 : 0;
        }
        m.pq[o + rank] = lr_record_of(xq, yq, zq, rq, m.ax[la], m.ay[la], beta);
    }
}

/* Bucketed ranking (O(nn) per atom instead of the O(nn^2) of lr_phase_rank), used when the
 * tile's pairs fit LR_RANK_ROUNDS rounds of the workgroup:
 *   P1  histogram of beta over LR_NBUCKET angular buckets per atom (in lr_phase_beta)
 *   Pa  exclusive prefix per atom                       -> bucket cursors
 *   Pb  scatter (beta, list position) by bucket         -> sb / ki2   (order inside a bucket: arrival)
 *   Pc  exact rank inside the pair's own bucket (mean occupancy ~1)  -> final position, in registers
 *   Pd  write the records at their final positions.
 * The result is the same permutation as lr_phase_rank's: ascending beta, ties by list position. */
#define LR_RANK_ROUNDS 3
struct RankRegs { /* carried across the barrier between Pc and Pd */
    int fin[LR_RANK_ROUNDS]; /* final position of this slot's pair */
    int q[LR_RANK_ROUNDS];   /* its neighbor atom */
};
/* worth its fixed cost (three more phases) only for long lists: measured -15 % kernel time at 47
 * neighbors/atom, +13 % at 19 when applied unconditionally */
SASA_D bool lr_bucket_path(const TileArgs &a, const TileMem &m, int B)
{
    const int total = m.aoff[a.TA];
    return total <= LR_RANK_ROUNDS * B && total > 30 * a.TA; /* (scratch fits: 10*total + 128*TA < 32*total) */
}

SASA_D void lr_phase_prefix(const TileArgs &a, TileMem &m, int tid)
{
    if (m.flags[0] || tid >= a.TA) return;
    int run = 0;
    for (int b = 0; b < LR_NBUCKET; ++b) {
        const int v = m.hist[tid * LR_NBUCKET + b];
        m.hist[tid * LR_NBUCKET + b] = run;
        run += v;
    }
}

SASA_D void lr_phase_scatter(const TileArgs &a, TileMem &m, int tid, int B)
{
    if (m.flags[0]) return;
    const int total = m.aoff[a.TA];
    for (int gp = tid; gp < total; gp += B) {
        int la = 0;
        while (m.aoff[la + 1] <= gp) ++la;
        const int o = m.aoff[la], k = gp - o;
        if (k >= m.acnt[la]) continue;
        const double beta = m.tb[gp];
        const int slot = SASA_ATOMIC_ADD_LDS(&m.hist[la * LR_NBUCKET + lr_bucket(beta)], 1);
        m.sb[o + slot] = beta;
        m.ki2[o + slot] = (unsigned short)k;
    }
}

SASA_D void lr_phase_rank2(const TileArgs &a, TileMem &m, int tid, int B, RankRegs &rr)
{
    if (m.flags[0]) return;
    const int total = m.aoff[a.TA];
    for (int r = 0; r < LR_RANK_ROUNDS; ++r) {
        const int gp = tid + r * B;
        rr.fin[r] = -1;
        if (gp >= total) continue;
        int la = 0;
        while (m.aoff[la + 1] <= gp) ++la;
        const int o = m.aoff[la], s = gp - o;
        if (s >= m.acnt[la]) { rr.fin[r] = -2; continue; } /* padding slot */
        const double beta = m.sb[gp];
        const int kme = m.ki2[gp], b = lr_bucket(beta);
        const int lo = b ? m.hist[la * LR_NBUCKET + b - 1] : 0, hi = m.hist[la * LR_NBUCKET + b]; /* cursors now = bucket ends */
        int rank = lo;
        for (int t = lo; t < hi; ++t) {
            const double bt = m.sb[o + t];
            rank += (bt < beta || (bt == beta && m.ki2[o + t] < kme)) ? 1 : 0;
        }
        rr.fin[r] = o + rank;
        rr.q[r] = m.idx[la * a.cap_idx + kme];
        m.tb[gp] = beta; /* tb is dead since Pb: park this slot's beta where Pd cannot clobber it */
    }
}

SASA_D void lr_phase_write(const TileArgs &a, TileMem &m, int tid, int B, const RankRegs &rr)
{
    if (m.flags[0]) return;
    for (int r = 0; r < LR_RANK_ROUNDS; ++r) {
        const int gp = tid + r * B;
        if (rr.fin[r] == -1) continue;
        if (rr.fin[r] == -2) { m.pq[gp] = lr_padding(); continue; }
        int la = 0;
        while (m.aoff[la + 1] <= gp) ++la;
        m.pq[rr.fin[r]] = lr_record(a, rr.q[r], m.ax[la], m.ay[la], m.tb[gp]);
    }
}

/* cos(alpha) of the arc record q cuts out of circle i (A = Ri'^2, h2 = 1/(2 Ri')) at height z,
 * or a value outside (-1, 1): see lr_record */
SASA_D double lr_cos(const Quad q, double A, double h2, double z)
{
    const double dj = q.x - z;
    return fma(dj, dj, A + q.y) * (q.z * h2);
}

/* Screening pass over up to 32 neighbors (lim is even, lists are padded): bit k of the result
 * is set when neighbor k may cut an arc out of circle i (cos alpha < 1; NaN never does).  cmin
 * collects the smallest cos alpha: at or below -1 some neighbor's circle contains circle i entirely
 * (ref: src/sasa_lr.c:327-330) and the slice is buried.  The list is walked from its end so that
 * shifting the results in from the right leaves neighbor k at bit k. */
SASA_D unsigned lr_screen32(const Quad *PQ, int lim, double A, double h2, double z, double &cmin)
{
    unsigned w = 0;
    for (int k = lim - 2; k >= 0; k -= 2) {
        const double c0 = lr_cos(PQ[k], A, h2, z), c1 = lr_cos(PQ[k + 1], A, h2, z);
        cmin = SASA_MIN(cmin, c0);
        cmin = SASA_MIN(cmin, c1);
        w = SASA_SHIFT_IN_LT1(w, c1);
        w = SASA_SHIFT_IN_LT1(w, c0);
    }
    return w;
}

/* Raw end points beta -+ alpha of the arc neighbor record q buries on circle i
 * (ref: src/sasa_lr.c:335-339).  The arc passes the origin iff inf < 0 or sup > 2pi (never
 * both: alpha < pi), which is when the reference's normalisation (:340-341) makes sup < inf. */
SASA_D void lr_arc(const Quad q, double A, double h2, double z, double &inf, double &sup)
{
    const double alpha = acos_fast(lr_cos(q, A, h2, z)); /* the screening's value, bit for bit */
    inf = q.w - alpha;
    sup = q.w + alpha;
}

/* Sum of the exposed gaps given the disjoint components in ascending order (component c is
 * comp(c)), the covered prefix [0,W] and suffix [V,2pi] of the arcs that wrap.
 * ref: src/sasa_lr.c:396-407 with sum = arc[0] == (0 < arc[0] ? arc[0] - 0 : 0). */
#define LR_SWEEP(depth, wrap, W, V, COMP_S, COMP_E, result)                                   \
    do {                                                                                     \
        double sum_ = 0, sup_ = (W);                                                         \
        for (int c_ = 0; c_ < (depth); ++c_) {                                               \
            const double cs_ = COMP_S(c_), ce_ = COMP_E(c_);                                  \
            if ((wrap) && cs_ >= (V)) break; /* sorted behind the [V,2pi] piece: covered */   \
            if (sup_ < cs_) sum_ += cs_ - sup_;                                              \
            if (ce_ > sup_) sup_ = ce_;                                                      \
        }                                                                                    \
        if (wrap) {                                                                          \
            if (sup_ < (V)) sum_ += (V) - sup_;                                              \
            sup_ = SASA_TWOPI;                                                               \
        }                                                                                    \
        (result) = sum_ + SASA_TWOPI - sup_; /* ref: :407 */                                 \
    } while (0)

/* One slice of one atom: exposed arc length of circle i at height z, or -1 if the
 * slice is buried.  The atom's neighbor records are sorted by beta.
 *
 * Arc union: arcs arrive ordered by mid-angle beta, so disjoint components form a stack —
 * a new arc either overlaps the top component (merge, then keep popping while the merged
 * start reaches the next one down) or lies entirely to its right (push).  Arcs that wrap
 * through 0 only extend a covered prefix [0,W] / suffix [V,2pi].  End points are only ever
 * compared and copied, and gaps are summed in ascending order, so given the same inf/sup
 * values the result equals the reference's sort + sweep (src/sasa_lr.c:367-408) bit for bit. */
struct UnionState {
    double W, V;   /* covered prefix [0,W] and suffix [V,2pi] of the arcs that pass the origin */
    double ts, te; /* top component (in registers); the ones below it are in the LDS stack */
    int depth;
};

/* Feed the arcs of the set bits of w (neighbors PQ[k]) to the union, ascending k = ascending beta. */
SASA_D void lr_arcs32(unsigned w, const Quad *PQ, double A, double h2, double z,
                      UnionState &u, Arc *stk, int stride, int ds, int *err)
{
    while (w) {
        const int k = __builtin_ctz(w);
        w &= w - 1;
        double inf, sup;
        lr_arc(PQ[k], A, h2, z, inf, sup);
        if (inf < 0 || sup > SASA_TWOPI) {           /* ref: :340-351 arc passes the origin */
            const double wi = inf < 0 ? inf + SASA_TWOPI : inf;
            const double ws = sup > SASA_TWOPI ? sup - SASA_TWOPI : sup;
            u.W = SASA_MAX(u.W, ws);
            u.V = SASA_MIN(u.V, wi);
        } else { /* inf <= beta <= sup: alpha < pi because the screening kept only c > -1 */
            /* written with selects and min/max rather than one branch per case: the cases differ
               from lane to lane, so every branch would be executed anyway */
            const bool fresh = inf > u.te; /* the arc starts a new top component (te = -inf while there is none) */
            if (fresh && u.depth > 0) {
                if (u.depth - 1 < ds) {
                    Arc t; t.s = u.ts; t.e = u.te;
                    stk[(u.depth - 1) * stride] = t;
                } else {
                    *err = 1;
                }
            }
            u.ts = fresh ? inf : SASA_MIN(u.ts, inf);
            u.te = fresh ? sup : SASA_MAX(u.te, sup);
            u.depth += fresh ? 1 : 0;
            if (!fresh)
                while (u.depth > 1) { /* the merged component may now reach the ones below it */
                    const Arc lo = stk[(u.depth - 2) * stride];
                    if (lo.e < u.ts) break;
                    u.ts = SASA_MIN(u.ts, lo.s);
                    u.te = SASA_MAX(u.te, lo.e);
                    --u.depth;
                }
        }
    }
}

SASA_D double lr_union_exact(const TileMem &m, int o, int nn, double A, double h2, double z,
                             Arc *stk, int stride, int ds, int *err)
{
    UnionState u;
    u.W = 0; u.V = SASA_TWOPI; u.ts = 0; u.te = -INFINITY; u.depth = 0;
    for (int base = 0; base < nn; base += 64) { /* nn is even (padded) */
        const int lim = nn - base < 64 ? nn - base : 64;
        const Quad *PQ = m.pq + o + base;
        double cmin = 1.0;
        unsigned lo = lr_screen32(PQ, lim < 32 ? lim : 32, A, h2, z, cmin), hi = 0;
        if (lim > 32) hi = lr_screen32(PQ + 32, lim - 32, A, h2, z, cmin);
        if (cmin <= -1.0) return -1;
#ifdef SASA_ABLATE_ARCS /* timing attribution only: tools/build_variant.sh, never in the product */
        lo = hi = 0;
#endif
        lr_arcs32(lo, PQ, A, h2, z, u, stk, stride, ds, err);
        if (hi) lr_arcs32(hi, PQ + 32, A, h2, z, u, stk, stride, ds, err);
    }
    double res;
    const double ts = u.ts, te = u.te;
    const int depth = u.depth;
#define CS_(c) ((c) == depth - 1 ? ts : stk[(c) * stride].s)
#define CE_(c) ((c) == depth - 1 ? te : stk[(c) * stride].e)
    const bool wrap = u.V < SASA_TWOPI; /* every arc through the origin starts below 2pi */
    LR_SWEEP(depth, wrap, u.W, u.V, CS_, CE_, res);
#undef CS_
#undef CE_
    return res;
}

/* One slice of one atom: exposed arc length, or a negative value if it contributes nothing. */
SASA_D double lr_slice(const TileMem &m, int o, int nn, double zi, double Ri, double z,
                       Arc *stk, int stride, int ds, int *err)
{
    const double di = fabs(zi - z);                 /* ref: src/sasa_lr.c:308 */
    const double A = Ri * Ri - di * di;             /* Ri'^2 */
    if (!(A > 0)) return -1;                        /* ref: :310-312 */
    double Rip, h2;
    sqrt_rh(A, Rip, h2);                            /* h2 = 1/(2 Ri') */
#ifdef SASA_ABLATE_SLICES
    return h2;
#endif
    return lr_union_exact(m, o, nn, A, h2, z, stk, stride, ds, err);
}

/* phase L: work items = (atom, slice) */
SASA_D void lr_phase_slices(const TileArgs &a, TileMem &m, int tile, int tid, int B)
{
    if (m.flags[0]) return;
    const int na = tile_atoms(a, tile);
    const int ns = a.n_res;
    Arc *stk = m.stack + tid;
    int err = 0;
    if (a.tab) {
        /* slice-major items: a wave holds a few ADJACENT slices of every atom of the tile, so
           its lanes have similar arc counts (polar slices cut few arcs, equatorial many) */
        const int items = na * ns;
        const float inv_na = 1.0f / (float)na;
        for (int it = tid; it < items; it += B) {
#ifdef SASA_ATOM_MAJOR
            const int la = it / ns, s = it - la * ns;
#else
            int s = (int)(((float)it + 0.5f) * inv_na), la = it - s * na; /* it / na without the */
            if (la < 0) { --s; la += na; }                              /* integer-division sequence */
            else if (la >= na) { ++s; la -= na; }
#endif
            const double Ri = m.aR[la], zi = m.az[la];
            const double delta = 2 * Ri / ns;       /* ref: src/sasa_lr.c:304 */
            double z = zi - Ri - 0.5 * delta;
            int k = s + 1;                          /* accumulated like the reference, :307: s + 1 */
            for (; k >= 2; k -= 2) { z += delta; z += delta; } /* separately rounded additions    */
            if (k) z += delta;
            const int o = m.aoff[la];
            const double ex = lr_slice(m, o, m.aoff[la + 1] - o, zi, Ri, z, stk, B, a.ds, &err); /* padded count */
            m.contrib[la * ns + s] = ex < 0 ? 0.0 : delta * Ri * ex; /* ref: :360 */
        }
    } else {
        /* many slices per atom (TA == 1): strided partial sums, z by direct formula */
        const double Ri = m.aR[0], zi = m.az[0];
        const double delta = 2 * Ri / ns;
        const double z0 = zi - Ri - 0.5 * delta;
        double part = 0;
        for (int s = tid; s < ns; s += B) {
            const double z = z0 + (double)(s + 1) * delta;
            const double ex = lr_slice(m, 0, m.aoff[1], zi, Ri, z, stk, B, a.ds, &err);
            if (!(ex < 0)) part += delta * Ri * ex;
        }
        m.contrib[tid] = part;
    }
    if (err) m.flags[1] = 1;
}

/* phase S: per-atom sum in slice order (ref: src/sasa_lr.c:305-361 accumulates sasa += ...) */
template <bool GLOBAL>
SASA_D void lr_phase_store(const TileArgs &a, TileMem &m, int tile, int tid, int B)
{
    if (m.flags[0]) return;
    const int na = tile_atoms(a, tile), p0 = tile_first_atom(a, tile);
    if (m.flags[1]) { /* a stack overflowed: redo the tile in the next launch */
        if (tid == 0) {
            if (!a.ovf_tiles) {
                SASA_ATOMIC_MAX_GLB(&a.status[ST_ERROR], (int)ERR_STACK_CAP);
            } else {
                const int w = SASA_ATOMIC_ADD_GLB(a.ovf_count, 1);
                a.ovf_tiles[w] = tile;
            }
        }
        return;
    }
    if (a.tab) {
        if (tid < na) {
            double s = 0;
            for (int k = 0; k < a.n_res; ++k) s += m.contrib[tid * a.n_res + k];
            a.sasa[a.s_idx[p0 + tid].orig] = s;
        }
    } else if (tid == 0) {
        double s = 0;
        for (int t = 0; t < B; ++t) s += m.contrib[t];
        a.sasa[a.s_idx[p0].orig] = s;
    }
}

/* ---------------------------------------------------------------- Shrake & Rupley */

/* Wave-aggregated LDS counters (device): the lanes of a wavefront that want to bump the same counter are found
 * with a ballot and counted with a popcount; ONE lane issues the atomic for all of them.
 *   sasa_wave_slot(ctr, pred)       pred lanes get consecutive slots of *ctr (the order inside the wave is the lane order)
 *   sasa_wave_count(ctrs, key, pred) ctrs[key] += number of pred lanes with that key (a wave's lanes hold one or two
 *                                    different atoms: one atomic per atom and wave instead of one per exposed point)
 * The CPU emulation (tests only) runs the lanes one after the other and keeps the plain atomics. */
#ifdef SASA_EMU
SASA_D int sasa_wave_slot(int *ctr, bool pred) { return pred ? SASA_ATOMIC_ADD_LDS(ctr, 1) : -1; }
SASA_D void sasa_wave_count(int *ctrs, int key, bool pred) { if (pred) SASA_ATOMIC_ADD_LDS(&ctrs[key], 1); }
#else
SASA_D int sasa_wave_slot(int *ctr, bool pred)
{
    const unsigned long long m = __builtin_amdgcn_ballot_w64(pred);
    if (m == 0) return -1;
    const int leader = __builtin_ctzll(m);
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    int base = 0;
    if (lane == leader) base = atomicAdd(ctr, __popcll(m));
    base = __builtin_amdgcn_readlane(base, leader);
    return pred ? base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)) : -1;
}
SASA_D void sasa_wave_count(int *ctrs, int key, bool pred)
{
    unsigned long long pending = __builtin_amdgcn_ballot_w64(pred);
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    while (pending) { /* (uniform) one trip per different key among the pred lanes */
        const int leader = __builtin_ctzll(pending);
        const int k0 = __builtin_amdgcn_readlane(key, leader);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(pred && key == k0);
        if (lane == leader) atomicAdd(&ctrs[k0], __popcll(same));
        pending &= ~same;
    }
}
#endif

/* ---- Shrake & Rupley, second arrangement (round 5) ----------------------------------------------------------
 * Until round 4 the S&R tile went  neighbors (candidate INDICES into per-atom lists) -> offsets -> pairs (every
 * neighbor fetched from global memory a second time, its record (x, y, z, Rj^2) written to the packed pool, big caps
 * to the front) -> points.  The profile of round 5 (profiles/r05_sr_phase_ablation.txt) put 23 % of the kernel on the
 * reference's PDB entries into that second fetch: a dependent global load per pair behind LDS atomics.  Now the
 * neighbor phase itself keeps what it has in registers the moment the contact test passes - (x, y, z, Rj) of the
 * candidate - in a FIXED segment of C = cap_idx records per atom, in order of discovery (one LDS atomic and one
 * 32-byte store in the sparse hit path: nothing else belongs there, a wave runs it for one lane in seven).  A dense
 * pass (sr_order_wave: one wave per atom, every lane a record) then squares the radius and moves the neighbors with a
 * large cap on the atom's sphere to the front of the list, the others to its end, in place.  No index lists, no offsets phase, no
 * second fetch.  An atom with more than C neighbors sends its tile to the next launch (larger C), as a list overflow
 * did before; C for the next batch follows the sampled histogram of the tiles' longest lists (sr_cap_from_hist).
 * Counts are unchanged bit for bit: the point test is an OR over the same neighbor set. */

SASA_D void sr_phase_load(const TileArgs &a, TileMem &m, int tile, int tid, int B)
{
    tile_phase_load(a, m, tile, tid, B);
    if (tid < a.TA) m.aoff[tid] = 0; /* (front cursor of sr_phase_order) */
}

/* (Round 6, measured and not kept: the NEXT tile's load phase fetched link by link between the phases of this one, as the
 * L&R kernel does - lr2_pre_a .. c.  The load phase is 1.6 of 7.1 ms on the coil batch, but a 128-thread tile that carries
 * six more registers through its neighbor phase spills at the five-waves cap: 2.77 -> 3.05 ms on the PDB entries, 7.1 -> 7.5
 * on the coils; the second arrangement at its seven-waves cap 4.0 -> 4.95.  Also measured and not kept: the atoms' candidate rows
 * written by a kernel of their own before the tile kernel (72 bytes per atom; the tile's load phase then one read instead of a
 * chain of three): the row kernel costs more than the tiles gain - 2.77 -> 2.90 ms on the PDB entries, 6.55 -> 6.9 on the coils.
 * The chain's latency is hidden by the other resident tiles; what the cumulative "load phase only" build shows is the tile loop
 * running empty, not a cost the full kernel pays.) */
/* contact test of the reference, operand for operand (src/nb.c:483-492, as nb_test); a neighbor's (x, y, z, R) is kept */
SASA_D void sr_nb_test(const TileArgs &a, TileMem &m, int la, int p, int q, double xi, double yi,
                       double zi, double ri, double xq, double yq, double zq, double rq)
{
    if (q == p) return;
    const double cut2 = (ri + rq) * (ri + rq);
    const double dx = xq - xi, dy = yq - yi, dz = zq - zi;
    if (dx * dx + dy * dy + dz * dz < cut2) {
        const int s = SASA_ATOMIC_ADD_LDS(&m.acnt[la], 1);
        if (s < a.cap_idx) { Quad rec; rec.x = xq; rec.y = yq; rec.z = zq; rec.w = rq; m.pq[la * a.cap_idx + s] = rec; }
    }
}

/* phase N: neighbor discovery.  SUB = B/TA lanes share one atom and stride over the concatenation of its 9 candidate
 * runs, three candidates per trip (tile_phase_neighbors' loop, with the neighbor kept at once) */
SASA_D void sr_phase_neighbors(const TileArgs &a, TileMem &m, int tile, int tid, int B)
{
    const int na = tile_atoms(a, tile), p0 = tile_first_atom(a, tile);
    const int SUB = B / a.TA;
    const int la = tid / SUB, sub = tid - la * SUB;
    if (la >= na) return;
    const int p = p0 + la;
    const double xi = m.ax[la], yi = m.ay[la], zi = m.az[la], ri = m.aR[la];
    const int *rl = m.rowlo + 9 * la, *rc = m.rowcnt + 9 * la;
    int total = 0;
    for (int r = 0; r < 9; ++r) total += rc[r];
    int r = 0, base = 0, cnt = rc[0];
    for (int f = sub; f < total; f += SASA_NB_UNROLL * SUB) {
        int q[SASA_NB_UNROLL];
        double x[SASA_NB_UNROLL], y[SASA_NB_UNROLL], z[SASA_NB_UNROLL], rq[SASA_NB_UNROLL];
        for (int j = 0; j < SASA_NB_UNROLL; ++j) {
            const int fj = f + j * SUB;
            if (fj < total) {
                while (fj >= base + cnt) { base += cnt; ++r; cnt = rc[r]; }
                q[j] = rl[r] + (fj - base);
            } else {
                q[j] = p; /* the atom itself: never a neighbor */
            }
        }
        for (int j = 0; j < SASA_NB_UNROLL; ++j) {
            const unsigned u = (unsigned)q[j];
            { const Quad v = a.sq[u]; x[j] = v.x; y[j] = v.y; z[j] = v.z; rq[j] = v.w; }
        }
        for (int j = 0; j < SASA_NB_UNROLL; ++j) sr_nb_test(a, m, la, p, q[j], xi, yi, zi, ri, x[j], y[j], z[j], rq[j]);
    }
}

/* behind the barrier that follows phase N: does every list fit its segment?  The longest list of the tile. */
SASA_D void sr_phase_lists(const TileArgs &a, TileMem &m, int tid)
{
    if (tid >= a.TA) return;
    const int c = m.acnt[tid];
    if (c > a.cap_idx) m.flags[0] = 1;
    SASA_ATOMIC_MAX_LDS(&m.flags[2], c);
}

/* The ordering pass over the neighbors found squares the radius (ref: src/sasa_sr.c:146) and moves every record to its
 * place in the atom's list: neighbors whose sphere hides a large cap of atom i from the FRONT, the others from the END
 * of the list.  The point test is an OR over neighbors (any order gives the same counts), and with the big caps first
 * almost every covered point is rejected by the first SR_FIRST tests (measured at protein density: 2 % of the points
 * survive instead of 21 %).  Cap of half-angle theta: cos(theta) = (Ri^2 + d^2 - Rj^2)/(2 Ri d) < 0.6, without the root. */
#define SR_ORDER_RECS 2 /* records of one atom a lane of the ordering wave holds: C <= 64 * SR_ORDER_RECS in the LDS launches */
#ifndef SASA_EMU
/* ONE WAVE orders one atom's list in place (wave w the atoms w, w + waves, ...): lane l takes records l and l + 64 into
 * registers, the wave's ballots say how many big caps stand before each, and every record goes to its place - no
 * atomics, and no workgroup barrier inside: the LDS executes a wave's instructions in order, so all of the list is in
 * registers before the first record is written back (the compiler is kept from moving the stores up). */
SASA_D void sr_order_wave(const TileArgs &a, TileMem &m, int tid, int B)
{
    if (m.flags[0]) return;
    const int lane = tid & 63, C = a.cap_idx;
    for (int la = tid >> 6; la < a.TA; la += B >> 6) { /* (uniform per wave) */
        const int nn = m.acnt[la];
        const double ri = m.aR[la], xi = m.ax[la], yi = m.ay[la], zi = m.az[la];
        Quad *L = m.pq + la * C;
        Quad rec[SR_ORDER_RECS];
        bool big[SR_ORDER_RECS];
#pragma unroll
        for (int j = 0; j < SR_ORDER_RECS; ++j) {
            const int k = lane + 64 * j;
            big[j] = false;
            if (k < nn) {
                rec[j] = L[k];
                const double dx = rec[j].x - xi, dy = rec[j].y - yi, dz = rec[j].z - zi;
                const double d2 = dx * dx + dy * dy + dz * dz;
                rec[j].w = rec[j].w * rec[j].w; /* ref: src/sasa_sr.c:146 */
                const double num = ri * ri + d2 - rec[j].w;
                big[j] = num < 0 || num * num < 1.44 * (ri * ri) * d2;
            }
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST); __builtin_amdgcn_wave_barrier(); __atomic_signal_fence(__ATOMIC_SEQ_CST);
        int nf = 0, nb = 0; /* big / small caps in the chunks before this one (uniform) */
#pragma unroll
        for (int j = 0; j < SR_ORDER_RECS; ++j) {
            const int k = lane + 64 * j;
            const unsigned long long mb = __builtin_amdgcn_ballot_w64(big[j]), mv = __builtin_amdgcn_ballot_w64(k < nn);
            const int before = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0u));
            const unsigned long long ms = mv & ~mb;
            const int sbefore = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ms >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ms, 0u));
            if (k < nn) L[big[j] ? nf + before : nn - 1 - (nb + sbefore)] = rec[j];
            nf += __popcll(mb); nb += __popcll(ms);
        }
    }
}
#endif
/* launches whose segments are longer than the lanes can hold across a barrier (the slab launch: C = 4096): one lane
   per atom orders its list in place - squares every radius, then swaps big caps forward */
SASA_D void sr_order_serial(const TileArgs &a, TileMem &m, int tid)
{
    if (m.flags[0] || tid >= a.TA) return;
    const int C = a.cap_idx, nn = m.acnt[tid];
    const double ri = m.aR[tid], xi = m.ax[tid], yi = m.ay[tid], zi = m.az[tid];
    Quad *L = m.pq + tid * C;
    int nf = 0;
    for (int k = 0; k < nn; ++k) {
        Quad rec = L[k];
        const double dx = rec.x - xi, dy = rec.y - yi, dz = rec.z - zi;
        const double d2 = dx * dx + dy * dy + dz * dz;
        rec.w = rec.w * rec.w;
        const double num = ri * ri + d2 - rec.w;
        const bool big = num < 0 || num * num < 1.44 * (ri * ri) * d2;
        if (big) { /* (L[nf .. k) are small caps, already squared: the first of them makes room) */
            if (k != nf) { const double mx = L[nf].x, my = L[nf].y, mz = L[nf].z, mw = L[nf].w; L[k].x = mx; L[k].y = my; L[k].z = mz; L[k].w = mw; }
            L[nf] = rec;
            ++nf;
        } else {
            L[k] = rec;
        }
    }
}
SASA_HD bool sr_order_in_wave(int C) { return C <= 64 * SR_ORDER_RECS; }
SASA_HD int sr_items(int TA, int n_points) { return (TA * n_points + 1) / 2; } /* S&R: the tile's slice-area table (unused) holds the survivor list, an entry per (atom, point) */
/* ... while it fits: 4 bytes per (atom, point) of the tile.  The reference takes any number of test points
   (src/sasa_sr.c:56-90, :168-224); from a table of 32 KB on (one atom per tile and more than 8192 points) the tile runs
   without the survivor list (TileArgs::tab = 0) instead of asking for more LDS than a CU has (round-5 advisor: 39.9k
   points and more failed the launch, and a narrow range below that overran the shrunk segment) */
#define SR_SURVIVOR_BYTES_MAX (32 * 1024)
SASA_HD bool sr_survivors_fit(int TA, int n_points) { return (long long)TA * n_points * 4 <= SR_SURVIVOR_BYTES_MAX; }
SASA_HD int sr_tile_items(int TA, int n_points, bool caps) /* ... or the third arrangement's masks and list (sr_caps.h): 4 copies of 4 DEF words + 4 COV words per atom, 256 list entries (dwords / 2) */
{
    return caps ? 10 * TA + 128 : sr_items(TA, n_points);
}
SASA_HD int sr_hist_bin(int longest) { return longest >> 1; } /* demand histogram of S&R batches: tiles by their longest neighbor list, bins of 2 */
template <bool GLOBAL>
SASA_D void sr_report(const TileArgs &a, TileMem &m, int tile, int tid, int &wg_max_nn)
{
    if (tid != 0) return;
    if (m.flags[2] > wg_max_nn) wg_max_nn = m.flags[2];
    /* one tile in 32, picked by a hash of its number: tiles are in cell order, so "every 32nd tile" samples the same
       places of every structure of a batch of equal structures (round 5: it put the capacity of the coil batch at 48
       records where the 98.5th percentile of all tiles needs 56, and a tenth of the tiles went to the second launch) */
    if (!a.work_tiles && (((unsigned)tile * 2654435761u) >> 27) == 0) {
        const int need = sr_hist_bin(m.flags[2]);
        SASA_ATOMIC_ADD_GLB(&a.status[ST_HIST + (need < 63 ? need : 63)], 1);
    }
    if (m.flags[0]) {
        if (!a.ovf_tiles) {
            SASA_ATOMIC_MAX_GLB(&a.status[ST_ERROR], (int)ERR_NEIGHBOR_CAP);
        } else {
            const int w = SASA_ATOMIC_ADD_GLB(a.ovf_count, 1);
            a.ovf_tiles[w] = tile;
        }
    }
}

/* phase L1: work items = (atom, test point).  A point is exposed iff no neighbor covers it
 * (ref: src/sasa_sr.c:311-330; the reference's "last hit first" order does not change the
 * outcome, which is an OR over neighbors).  About 93 % of the points of a protein are covered,
 * most of them by one of the first few neighbors, while an exposed point must be tested against
 * all of them: with one lane per point a wave would run the full neighbor loop with a quarter
 * of its lanes alive.  So L1 tests only the first SR_FIRST neighbors — the ones with the largest
 * caps, see sr_nb_test — and appends the survivors (still uncovered, neighbors left) to a
 * compact LDS list; L2 finishes them with dense lanes. */
#ifndef SR_FIRST
#define SR_FIRST 6 /* (round 5, MI355X, coil batch / PDB entries x 251: 6 neighbors 10.39 / 3.92 ms, 8: 10.63 / 3.95, 10: 10.97 / 3.98, 4: slower on the PDB entries) */
#endif
SASA_D bool sr_inside(const Quad q, double tx, double ty, double tz)
{
    const double dx = tx - q.x, dy = ty - q.y, dz = tz - q.z;
    return dx * dx + dy * dy + dz * dz <= q.w; /* ref: src/sasa_sr.c:324 */
}
/* is the point inside one of the neighbors k0 .. k1-1 of the list at `o`?  One neighbor per look: most covered points are
 * inside the first or second (grouping the tests four by four, with one exit per group, was 35 % slower: round 5, measured) */
SASA_D bool sr_covered(const TileMem &m, int o, int k0, int k1, double tx, double ty, double tz)
{
    for (int k = k0; k < k1; ++k)
        if (sr_inside(m.pq[o + k], tx, ty, tz)) return true;
    return false;
}
/* the same for the SURVIVORS of the first look (sr_phase_points2).  Most of them are exposed points, whose loop never leaves
 * early, so reading two or four records together before one exit looked right - and measured wrong on the MI355X (round 5,
 * coil batch / PDB entries x 251, kernel ms: one record per look 10.30 / 3.89, two 10.50 / 3.98, four 12.99 / 4.90: under
 * the 72-register cap of seven waves the wider loop body spills).  One per look it stays; the variants remain for the record. */
#ifndef SR_GROUP2
#define SR_GROUP2 1
#endif
SASA_D bool sr_covered_rest(const TileMem &m, int o, int k0, int k1, double tx, double ty, double tz)
{
    int k = k0;
#if SR_GROUP2 == 4
    for (; k + 4 <= k1; k += 4) {
        const Quad q0 = m.pq[o + k], q1 = m.pq[o + k + 1], q2 = m.pq[o + k + 2], q3 = m.pq[o + k + 3];
        const bool c0 = sr_inside(q0, tx, ty, tz), c1 = sr_inside(q1, tx, ty, tz), c2 = sr_inside(q2, tx, ty, tz), c3 = sr_inside(q3, tx, ty, tz);
        if (c0 | c1 | c2 | c3) return true;
    }
#elif SR_GROUP2 == 2
    for (; k + 2 <= k1; k += 2) {
        const Quad q0 = m.pq[o + k], q1 = m.pq[o + k + 1];
        if (sr_inside(q0, tx, ty, tz) | sr_inside(q1, tx, ty, tz)) return true;
    }
#endif
    for (; k < k1; ++k)
        if (sr_inside(m.pq[o + k], tx, ty, tz)) return true;
    return false;
}
SASA_D void sr_point(const TileArgs &a, const TileMem &m, int la, int pt, double &tx, double &ty, double &tz)
{
    const double ri = m.aR[la];
    /* test point = unit * ri, then + centre: two rounded steps (ref: src/coord.c:331-342, 314-329) */
    tx = a.unit_pts[3 * pt] * ri; ty = a.unit_pts[3 * pt + 1] * ri; tz = a.unit_pts[3 * pt + 2] * ri;
    tx += m.ax[la]; ty += m.ay[la]; tz += m.az[la];
}
SASA_D void sr_phase_points(const TileArgs &a, TileMem &m, int tile, int tid, int B)
{
    if (m.flags[0]) return;
#ifdef SASA_ABLATE_POINTS
    return;
#endif
    const int na = tile_atoms(a, tile);
    const int np = a.n_res, items = na * np, C = a.cap_idx;
    /* (a.tab, S&R launches: the tile has a survivor table - an entry per (atom, point), sr_items() - in LDS.  Point
       counts whose table would not fit (sr_survivors_fit) run without: every point meets its atom's whole list in this
       first look, as until round 4, and the second look finds nothing to do) */
    const bool compact = a.tab != 0 && a.TA <= 64 && np <= 65536;
    unsigned *surv = (unsigned *)m.contrib; /* an entry per (atom, point): sr_items() */
    for (int it0 = 0; it0 < items; it0 += B) { /* (trip count uniform over the workgroup: the wave operations below need every lane) */
        const int it = it0 + tid;
        const bool have = it < items;
        const int la = have ? it / np : 0, pt = have ? it - la * np : 0;
        bool exposed = false, survivor = false;
        if (have) {
            double tx, ty, tz;
            sr_point(a, m, la, pt, tx, ty, tz);
            const int nn = m.acnt[la];
            const int k1 = compact && nn > SR_FIRST ? SR_FIRST : nn;
            if (!sr_covered(m, la * C, 0, k1, tx, ty, tz)) {
                exposed = k1 == nn;
                survivor = !exposed;
            }
        }
        /* exposed points are counted per wavefront with ballot + popcount, survivors get their slots the same way */
        sasa_wave_count(m.aexp, la, exposed);
        const int w = sasa_wave_slot(&m.flags[3], survivor);
        if (survivor) surv[w] = ((unsigned)la << 16) | (unsigned)pt;
    }
}

/* phase L2: the survivors of L1 against the rest of their atom's neighbors */
SASA_D void sr_phase_points2(const TileArgs &a, TileMem &m, int tid, int B)
{
    if (m.flags[0]) return;
    const unsigned *surv = (const unsigned *)m.contrib;
    const int C = a.cap_idx;
    const int ns = m.flags[3];
    for (int s0 = 0; s0 < ns; s0 += B) {
        const int s = s0 + tid;
        const bool have = s < ns;
        const int la = have ? (int)(surv[s] >> 16) : 0, pt = have ? (int)(surv[s] & 0xffffu) : 0;
        bool exposed = false;
        if (have) {
            double tx, ty, tz;
            sr_point(a, m, la, pt, tx, ty, tz);
            exposed = !sr_covered_rest(m, la * C, SR_FIRST, m.acnt[la], tx, ty, tz);
        }
        sasa_wave_count(m.aexp, la, exposed);
    }
}

SASA_D void sr_phase_store(const TileArgs &a, TileMem &m, int tile, int tid)
{
    if (m.flags[0]) return;
    const int na = tile_atoms(a, tile), p0 = tile_first_atom(a, tile);
    if (tid < na) {
        const double ri = m.aR[tid];
        const int n_surface = m.aexp[tid];
        const int i = a.s_idx[p0 + tid].orig;
        a.sasa[i] = (4.0 * SASA_PI * ri * ri * n_surface) / a.n_res; /* ref: src/sasa_sr.c:337 */
        if (a.counts) a.counts[i] = n_surface;
    }
}

#include "sr_caps.h" /* the third arrangement (round 6): cap masks looked up, the reference's test for the doubtful points only */

/* ---------------------------------------------------------------- per-structure totals */
/* One workgroup of SASA_TOT_B threads per structure: each thread sums a contiguous chunk in
 * atom order (4 loads in flight), thread 0 adds the partials in order.  Deterministic; equals the
 * reference's sequential sum (src/freesasa.c:113-116) up to fp64 reassociation (the drop-in
 * freesasa_calc sums on the host in exact reference order). */
#define SASA_TOT_B 256
SASA_D void totals_phase0(const double *sasa, const int64_t *offsets, double *part, int s, int tid)
{
    const int64_t b = offsets[s], e = offsets[s + 1];
    const int64_t per = (e - b + SASA_TOT_B - 1) / SASA_TOT_B;
    const int64_t lo = b + tid * per, hi = lo + per < e ? lo + per : e;
    double t = 0;
    int64_t i = lo;
    for (; i + 4 <= hi; i += 4) {
        const double v0 = sasa[i], v1 = sasa[i + 1], v2 = sasa[i + 2], v3 = sasa[i + 3];
        t += v0; t += v1; t += v2; t += v3;
    }
    for (; i < hi; ++i) t += sasa[i];
    part[tid] = t;
}
SASA_D void totals_phase1(const double *part, double *totals, int s, int tid)
{
    if (tid != 0) return;
    double t = 0;
    for (int k = 0; k < SASA_TOT_B; ++k) t += part[k];
    totals[s] = t;
}

/* Per-structure totals in two levels, so that one 200k-atom structure is not summed by a single
 * workgroup: (1) one workgroup per chunk of the bounds chunk table (<= SASA_BOUNDS_CHUNK atoms of one
 * structure): every thread adds a contiguous run in atom order, thread 0 adds the runs in order;
 * (2) one thread per structure adds its chunks in order.  Deterministic. */
SASA_D void totals_chunk_phase0(const PipeArgs &a, const double *sasa, double *part, int chunk, int tid)
{
    const int64_t b = a.chunk_begin[chunk];
    const int len = a.chunk_len[chunk];
    double t = 0;
    for (int i = tid; i < len; i += SASA_TOT_B) t += sasa[b + i];
    part[tid] = t;
}
SASA_D void totals_chunk_phase1(const double *part, double *chunk_tot, int chunk, int tid)
{
    if (tid != 0) return;
    double t = 0;
    for (int k = 0; k < SASA_TOT_B; ++k) t += part[k];
    chunk_tot[chunk] = t;
}
SASA_D void totals_struct(const PipeArgs &a, const double *chunk_tot, double *totals, int s)
{
    if (s >= a.n_structs) return;
    double t = 0;
    for (int k = a.struct_chunk0[s]; k < a.struct_chunk0[s + 1]; ++k) t += chunk_tot[k];
    totals[s] = t;
}

/* Short segments (residues: ~8 atoms): one thread per segment, strictly sequential in atom order,
 * i.e. exactly the sum the reference's result tree forms (ref: src/node.c:150-176). */
SASA_D void segsum_small(const double *sasa, const int64_t *seg, double *out, int k, int n_segs)
{
    if (k >= n_segs) return;
    double t = 0;
    for (int64_t i = seg[k]; i < seg[k + 1]; ++i) t += sasa[i];
    out[k] = t;
}

/* The reference's per-residue node area (ref: freesasa_atom_nodearea + freesasa_add_nodearea,
 * src/node.c:717-764): total, main chain, side chain, polar, apolar, unknown, each a sequential sum
 * in atom order, and the relative values 100 * abs / reference (ref: src/rsa.c:14-25) for total,
 * main chain, side chain, polar, apolar.  One thread per residue.  ref_row < 0: no reference
 * values for that residue (rel = NaN, the reference prints N/A). */
SASA_D void residue_areas(const double *sasa, const unsigned char *cls, const unsigned char *bb, const int64_t *res_first,
                          const short *ref_row, const double *ref_table, double *abs_out, double *rel_out, int r, int n_res)
{
    if (r >= n_res) return;
    double total = 0, mc = 0, sc = 0, polar = 0, apolar = 0, unknown = 0;
    for (int64_t i = res_first[r]; i < res_first[r + 1]; ++i) {
        const double a = sasa[i];
        total += a;
        if (bb[i]) mc += a; else sc += a;
        const int c = cls[i];
        if (c == 0) apolar += a;
        else if (c == 1) polar += a;
        else unknown += a;
    }
    double *o = abs_out + 6 * (int64_t)r;
    o[0] = total; o[1] = mc; o[2] = sc; o[3] = polar; o[4] = apolar; o[5] = unknown;
    if (rel_out) {
        double *q = rel_out + 5 * (int64_t)r;
        const int row = ref_row ? ref_row[r] : -1;
        if (row < 0) {
            q[0] = q[1] = q[2] = q[3] = q[4] = NAN;
        } else {
            const double *t = ref_table + 5 * row;
            q[0] = 100. * total / t[0]; q[1] = 100. * mc / t[1]; q[2] = 100. * sc / t[2];
            q[3] = 100. * polar / t[3]; q[4] = 100. * apolar / t[4];
        }
    }
}

/* Per-structure sums by atom class (0 apolar, 1 polar, 2 unknown; ref: freesasa_result_classes,
 * src/classifier.c:830-866): like the totals, SASA_TOT_B threads per structure take contiguous
 * chunks in atom order and thread 0 adds the partials in order.  out[3*s + c]. */
SASA_D void class_phase0(const double *sasa, const unsigned char *cls, const int64_t *offsets, double *part, int s, int tid)
{
    const int64_t b = offsets[s], e = offsets[s + 1];
    const int64_t per = (e - b + SASA_TOT_B - 1) / SASA_TOT_B;
    const int64_t lo = b + tid * per, hi = lo + per < e ? lo + per : e;
    double t0 = 0, t1 = 0, t2 = 0;
    for (int64_t i = lo; i < hi; ++i) {
        const double v = sasa[i];
        const int c = cls[i];
        if (c == 0) t0 += v;
        else if (c == 1) t1 += v;
        else t2 += v;
    }
    part[3 * tid] = t0; part[3 * tid + 1] = t1; part[3 * tid + 2] = t2;
}
SASA_D void class_phase1(const double *part, double *out, int s, int tid)
{
    if (tid >= 3) return;
    double t = 0;
    for (int k = 0; k < SASA_TOT_B; ++k) t += part[3 * k + tid];
    out[3 * s + tid] = t;
}

/* Workgroup b runs on XCD b % 8 (observed dispatch order).  Give each XCD one contiguous
 * eighth of the cell-sorted tiles so that the candidate cells a tile reads were, with high
 * probability, last touched through the same XCD's L2.  Speed only; any mapping is correct.
 * The main launch uses a grid of 8*ceil(n_tiles/8) workgroups. */
SASA_D int xcd_tile(int b, int n_tiles)
{
    const int per = (n_tiles + 7) >> 3;
    return (b & 7) * per + (b >> 3);
}

/* ------------------------------------------------------------------------------------
 * launch configuration (host side; shared by gpu_engine.hip and the test emulation)
 * ---------------------------------------------------------------------------------- */
struct TileCfg {
    int B, TA, tab, items, cap_idx, pool, lr, ds;
    size_t lds;
};

#define SR_CAP_DEFAULT 64 /* S&R: records per atom's segment (main launch) without history */
#define SR_CAP_MIN 32
#define SR_CAP_MAX 112
#define SR_CAP_MID 128     /* ... of the second launch (two records per lane of the ordering wave: sr_order_in_wave) */
#define SASA_ITEMS_CAP 640
#define SASA_TA_MAX 16
/* Three launches share one tiling.  1: small LDS lists sized for the typical tile (most
 * resident tiles per CU).  2: the tiles that overflowed, with 2x-4x larger LDS lists.  3: what
 * still overflows, lists in a global slab, far larger capacities. */
#define SASA_MID_BLOCKS 2048
#define SASA_FB_BLOCKS 64
#define SASA_FB_CAP 4096   /* neighbors per atom */
#define SASA_FB_POOL 16384 /* neighbors per tile */
#define SASA_FB_DS 96      /* spilled stack levels */

/* Pick workgroup size and atoms per tile so that TA*resolution work items fill whole rounds
 * of B threads (resolution 20 -> 16 atoms x 20 slices = 320 threads, one round). */
static inline TileCfg choose_cfg(int resolution, bool lr, int pool_hint = 0, bool sr_caps = false, int sr_last_ta = 0)
{
    TileCfg c;
    c.tab = 1;
    /* Small tiles win (measured, profiles/): with two-wave workgroups many independent tiles
       are resident per CU, so one tile's memory round trips and barriers overlap another's
       arithmetic.  TA*resolution work items should fill whole rounds of B threads
       (resolution 20 -> 6 atoms x 20 slices = 120 of 128 threads). */
    const int items_cap = lr ? SASA_ITEMS_CAP : 4096;
    const int ta_max = lr ? SASA_TA_MAX : 16;
    if (resolution > items_cap) {
        c.B = 256;
        c.TA = 1;
        if (lr) c.tab = 0;
    } else {
        double best = -1;
        c.B = 128; c.TA = 1;
        const int Bs[3] = {64, 128, 256};
        for (int bi = 0; bi < 3; ++bi)
            for (int ta = 1; ta <= ta_max && ta * resolution <= items_cap; ++ta) {
                const int items = ta * resolution, B = Bs[bi];
                if (ta > B / 4) continue; /* keep >= 4 lanes per atom for the neighbor search */
                const int rounds = (items + B - 1) / B;
                if (rounds > 8 && ta > 1) continue;
                const double eff = (double)items / (double)(rounds * B);
                /* one-wave tiles first (no barrier waits), more rounds are cheap, more atoms cost LDS;
                   measured at 100 slices: 64x3 6.4 ms, 64x2 6.7, 128x5 6.8, 128x3 7.4, 64x1 7.6, 256x5 7.6 */
                const double score = eff - 0.01 * (rounds - 1) - 0.02 * ta - (B == 256 ? 0.20 : 0.0) - (B == 128 ? 0.10 : 0.0);
                if (score > best + 1e-9) { best = score; c.B = B; c.TA = ta; }
            }
    }
    if (!lr && resolution >= 50 && resolution <= items_cap) {
        /* S&R: the per-tile overhead and the neighbor search amortise over more atoms; measured at 100
           points (profiles/r01_secondary.md): 256 threads x 8 atoms 0.33 ms per 2e5 atoms, x 5 atoms
           0.40, x 12 atoms 0.39, 128 threads x 5 atoms 0.36 */
        int ta = 1024 / resolution;
        c.B = 256;
        c.TA = ta < 1 ? 1 : (ta > 8 ? 8 : ta);
        /* up to 128 points the third arrangement runs (sr_caps.h): the point tests are all but gone and with them the reason
           for 256 threads (sr_caps: the caller has the table for these points); measured on the MI355X, round 6 (PDB entries x 251 / coil batch, kernel ms): 256 x 8 3.75 / 9.7,
           256 x 12 3.93 / 8.0, 128 x 4 3.50 / 9.4, 128 x 6 3.24 / 7.7, 128 x 8 3.55 / 8.0, 64 x 3 3.40 / 8.8, 64 x 4 3.37 / 8.2 */
        /* ... and the per-tile phases (a third of the kernel on sparse input) amortise over more atoms where the lists are
           short: the final build, 128 x 5 / 6 / 8 / 10: PDB entries 2.78 / 2.85 / 3.19 / 3.59, coils 7.5 / 6.9 / 6.41 / 7.3 */
        if (sr_caps) { c.B = 128; c.TA = pool_hint > 0 && pool_hint <= (sr_last_ta == 8 ? 56 : 48) ? 8 : 6; } /* (the longest list of a tile of 8 is a little longer than of 6: a batch stays with 8 up to 56) */
    }
    if (!lr) c.tab = sr_survivors_fit(c.TA, resolution) ? 1 : 0; /* (S&R: tab = the tile has a survivor table) */
    c.items = lr ? (c.tab ? c.TA * resolution : c.B) : (c.tab ? sr_tile_items(c.TA, resolution, sr_caps) : 1);
    c.lr = lr ? 1 : 0;
    c.cap_idx = 128;
    c.pool = 64 * c.TA < 128 ? 128 : 64 * c.TA;
    if (pool_hint > 0) c.pool = pool_hint;
    if (!lr) { /* S&R (second arrangement): a fixed segment of cap_idx records per atom; pool_hint = that capacity (sr_cap_from_hist) */
        c.cap_idx = pool_hint > 0 ? pool_hint : SR_CAP_DEFAULT;
        c.pool = c.TA * c.cap_idx;
    }
    c.ds = lr ? 3 : 0; /* deeper arc stacks are rare: those tiles go to the second launch */
    c.lds = tile_fixed_bytes(c.TA, c.items) + tile_list_bytes(c.TA, c.cap_idx, c.pool, c.lr, c.ds, c.B);
    if (pool_hint > 0 && lr) {
        /* occupancy comes in steps of whole workgroups per CU (160 KB of LDS): spend the slack of
           the current step on a larger pool instead of leaving it unused */
        const size_t cu_lds = 160 * 1024;
        size_t nblk = cu_lds / c.lds;
        const size_t wave_cap = (size_t)(20 * 64 / c.B); /* L&R: registers allow 5 waves per SIMD at best */
        if (lr && nblk > wave_cap) nblk = wave_cap; /* (the S&R kernel is lighter) */
        if (nblk >= 1 && nblk < 32) {
            const size_t slack = cu_lds / nblk - c.lds;
            c.pool += (int)(slack / (sizeof(Quad) + (lr ? 8 : 0))) & ~1;
            size_t lds2 = tile_fixed_bytes(c.TA, c.items) + tile_list_bytes(c.TA, c.cap_idx, c.pool, c.lr, c.ds, c.B);
            while (cu_lds / lds2 < nblk && c.pool > pool_hint) { /* aliasing/rounding pushed it over */
                c.pool -= 2;
                lds2 = tile_fixed_bytes(c.TA, c.items) + tile_list_bytes(c.TA, c.cap_idx, c.pool, c.lr, c.ds, c.B);
            }
            c.lds = lds2;
        }
    }
    return c;
}

/* S&R: records per atom's segment for the NEXT batch on this context, from the sampled histogram of the tiles' longest
 * neighbor lists (sr_hist_bin): what all but ~0.5 % of the tiles need, a multiple of 8 in [SR_CAP_MIN, SR_CAP_MAX] -
 * the few tiles above go to the second launch, and every 8 records less are 2 KB of LDS per 8-atom tile. */
static inline int sr_cap_from_hist(const int *hist)
{
    long long total = 0;
    for (int k = 0; k < 64; ++k) total += hist[k];
    if (total <= 0) return 0;
    long long allowed = total / 200, acc = 0; /* (measured on the coil batch, round 5: 1 % of the tiles in the second launch - 48 records - costs more than 8 records more per atom: step 11.8 ms at 48, 11.7 at 56 and 64) */
    int k = 63;
    for (; k > 0; --k) {
        acc += hist[k];
        if (acc > allowed) break;
    }
    int cap = ((2 * (k + 1) + 7) / 8) * 8;
    if (cap < SR_CAP_MIN) cap = SR_CAP_MIN;
    if (cap > SR_CAP_MAX) cap = SR_CAP_MAX;
    return cap;
}

/* Pool size for the NEXT batch on this context from the demand histogram of the last one: the
 * smallest size that would have kept all but ~1.5% of the tiles out of the second launch (which
 * re-does them at a fraction of the main launch's cost).  Less LDS per tile = more resident
 * tiles; choose_cfg then grows the pool again up to the next occupancy step. */
static inline int pool_from_hist(const int *hist, int TA)
{
    long long total = 0;
    for (int k = 0; k < 64; ++k) total += hist[k];
    if (total <= 0) return 0;
    long long allowed = total / 64, acc = 0; /* (the histogram is a 1-in-32 sample of the tiles) */
    int k = 63;
    for (; k > 0; --k) {
        acc += hist[k];
        if (acc > allowed) break;
    }
    if (k >= 63) return 0; /* demand beyond the histogram: keep the default */
    int pool = (k + 1) * hist_bin_width(TA);
    pool = (pool + 1) & ~1;
    if (pool < 32) pool = 32;
    return pool;
}

/* Mean neighbor records per tile of the last batch (same sampled histogram): long lists switch
 * the next batch to the kernel variant with bucketed beta ranking. */
static inline double mean_from_hist(const int *hist, int TA)
{
    long long total = 0;
    double acc = 0;
    for (int k = 0; k < 64; ++k) { total += hist[k]; acc += (double)hist[k] * (k + 0.5) * hist_bin_width(TA); }
    return total > 0 ? acc / (double)total : 0.0;
}

static inline TileCfg mid_cfg(const TileCfg &main_cfg, bool lr)
{
    TileCfg c = main_cfg;
    if (!lr) { /* S&R: segments of SR_CAP_MID records per atom */
        c.cap_idx = SR_CAP_MID > main_cfg.cap_idx ? SR_CAP_MID : main_cfg.cap_idx;
        c.pool = c.TA * c.cap_idx;
        c.lds = tile_fixed_bytes(c.TA, c.items) + tile_list_bytes(c.TA, c.cap_idx, c.pool, c.lr, c.ds, c.B);
        return c;
    }
    c.cap_idx = 256;
    c.pool = 2 * main_cfg.pool < 64 * c.TA ? 64 * c.TA : 2 * main_cfg.pool; /* twice the (adaptive) main pool */
    if (c.pool > 3072) c.pool = 3072;
    if (c.pool < main_cfg.pool) c.pool = main_cfg.pool; /* never smaller than the launch it backs up */
    c.ds = lr ? 8 : 0;
    c.lds = tile_fixed_bytes(c.TA, c.items) + tile_list_bytes(c.TA, c.cap_idx, c.pool, c.lr, c.ds, c.B);
    while (c.lds > 160 * 1024 && c.pool > 64) { /* (the CU's LDS) */
        c.pool = (c.pool * 3 / 4) & ~1;
        c.lds = tile_fixed_bytes(c.TA, c.items) + tile_list_bytes(c.TA, c.cap_idx, c.pool, c.lr, c.ds, c.B);
    }
    return c;
}

static inline TileCfg fallback_cfg(const TileCfg &main_cfg, bool lr)
{
    TileCfg fb = main_cfg;
    fb.cap_idx = SASA_FB_CAP;
    fb.pool = lr ? SASA_FB_POOL : fb.TA * SASA_FB_CAP; /* (S&R: a segment of SASA_FB_CAP records per atom, in the slab) */
    fb.ds = lr ? SASA_FB_DS : 0;
    fb.lds = tile_fixed_bytes(fb.TA, fb.items);
    return fb;
}

} /* namespace sasa */
#endif
