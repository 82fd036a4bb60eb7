/*
 * lr2_kernels.h — the Lee & Richards tile kernel, second generation (gfx950 / CDNA4).
 *
 * Same job as the L&R phases of sasa_kernels.h (neighbor discovery from the cell-sorted atoms,
 * pair records, slices, arc union; ref: src/nb.c:458-522, src/sasa_lr.c:270-408), reorganised
 * around what the round-1 counters showed: the kernel is bound by VALU issue, and more than half
 * of the issued lane slots did nothing.  One WAVE owns one tile of TA consecutive cell-sorted
 * atoms (TA = 6 at 20 slices: 120 (atom, slice) items for 64 lanes):
 *
 *   P0 load        tile atoms, their 9 candidate runs, the slice heights of every atom
 *                  (accumulated exactly like the reference, src/sasa_lr.c:304-307)
 *   P1 neighbors   atoms of the tile that share a cell share their candidates: every candidate is
 *                  loaded ONCE and tested against all of them (ref predicate, src/nb.c:483-492);
 *                  hits keep (xd, yd, zd, Rj) in LDS — no second fetch of a neighbor
 *   P2 offsets     the atoms' lists in the pool; whether the tile fits the launch
 *   (caps)         round 6, "P1.5" in the tables: hits whose cap on the atom's sphere lies inside another hit's cap are
 *                  dropped - their arcs lie inside the other's on every slice (lr2_prune_contained); offsets again
 *   P3 pairs       per (atom, neighbor): beta = atan2(yd, xd) + pi and the two coefficients of
 *                  2 Ri' cos(alpha) = b' + a' t, LINEAR in the slice height t (see lr2_record);
 *                  lists sorted by beta
 *   P4 screening   per (atom, slice): which neighbors cut an arc (bit mask), buried or not
 *   P5 queue       items with arcs, heaviest first (counting sort by arc count)
 *   P6 arcs        64 lanes drain the queue: a lane that has finished its item takes the next one
 *                  when enough lanes wait (refills are batched: the item switch is divergent code);
 *                  arc union with the beta-ordered stack of sasa_kernels.h, on raw (un-normalised)
 *                  end points
 *   P7 store       per-atom sum in slice order (ref: src/sasa_lr.c:360)
 *
 * A tile that does not fit the LDS capacities of the launch is redone at once as two halves; what still does not
 * fit goes to the next launch's work list (larger LDS lists, then the slab-backed first-generation kernel, atom by atom).
 *
 * Arithmetic: fp64 only (the contained-caps test is a sufficient one in fp32: it decides what is dropped, never a value);
 * -ffp-contract=off; fma only where written.
 */
#ifndef LR2_KERNELS_H
#define LR2_KERNELS_H

#include "sasa_kernels.h"

#ifdef SASA_EMU
namespace sasa_emu { /* tests/emu/emu.cpp: the 64 lanes of a wave run as fibers in lock step */
unsigned long long wave_ballot(bool p);
void wave_sync();
long long wave_exchange(long long v, int src);
}
#define LR2_BALLOT(p) sasa_emu::wave_ballot(p)
#define LR2_SYNC() sasa_emu::wave_sync()
#define LR2_SHFL(v, src) ((int)sasa_emu::wave_exchange((long long)(v), (src)))
#define LR2_POPC64(m) __builtin_popcountll(m)
#define LR2_POPC32(m) __builtin_popcount(m)
#define LR2_RANK(m, lane) __builtin_popcountll((m) & ((1ull << (lane)) - 1ull))
#define LR2_SHIFT_IN_LT1(w, c) (((w) << 1) | ((c) < 1.0 ? 1u : 0u))
#define LR2_SHIFT_IN_LT(w, v, lim) (((w) << 1) | ((v) < (lim) ? 1u : 0u))
#define LR2_UNIFORM(v) (v)
#define LR2_READLANE(v, src) LR2_SHFL((v), (src))
#define LR2_MUL24(a, b) ((int)(a) * (int)(b))
#define LR2_UMUL24(a, b) ((unsigned)(a) * (unsigned)(b))
#define LR2_RCPF(x) (1.0f / (x))
#define LR2_RSQF(x) (1.0f / sqrtf(x))
#define LR2_SQRTF(x) sqrtf(x)
#define LR2_ADD64_LDS(p, v) (*(p) += (v))
#define LR2_OR_LDS(p, v) (*(p) |= (v))
#define LR2_INC_LDS(p) (++*(p))
namespace sasa_emu { extern long long lr2_count[16]; } /* wave-level trip counts (lane 0 counts): 0 tiles, 1 arc iterations, 2 refills, 3 P1 test rounds, 4 rank trips, 5 screening trips, 6 P3 rounds */
#define LR2_COUNT(k, n) do { if (lane == 0) sasa_emu::lr2_count[(k)] += (n); } while (0)
/* (every lane of the wave: the largest and the sum of the lanes' trip counts of a divergent loop - what the wave pays and what the lanes use) */
#define LR2_COUNT_LANES(kmax, ksum, v) do { int mx_ = (v), sm_ = (v); for (int d_ = 1; d_ < 64; d_ <<= 1) { const int om_ = LR2_SHFL(mx_, lane ^ d_), os_ = LR2_SHFL(sm_, lane ^ d_); mx_ = mx_ > om_ ? mx_ : om_; sm_ += os_; } \
        if (lane == 0) { sasa_emu::lr2_count[(kmax)] += mx_; sasa_emu::lr2_count[(ksum)] += sm_; } } while (0)
#else
#define LR2_COUNT(k, n) do { } while (0)
#define LR2_COUNT_LANES(kmax, ksum, v) do { } while (0)
#define LR2_BALLOT(p) __builtin_amdgcn_ballot_w64(p)
/* One wave per workgroup: the LDS executes a wave's instructions in the order they were issued, so a read that
   follows a write in program order sees it, whichever lane wrote.  What is needed between phases is only that
   the COMPILER keeps the order: no s_barrier, and no s_waitcnt for accesses that are still in flight (a full
   workgroup fence waits for every outstanding global access, e.g. the store of the previous tile's result) */
#ifdef LR2_FULL_FENCE
#define LR2_SYNC() __syncthreads()
#else
#define LR2_SYNC() do { __atomic_signal_fence(__ATOMIC_SEQ_CST); __builtin_amdgcn_wave_barrier(); __atomic_signal_fence(__ATOMIC_SEQ_CST); } while (0)
#endif
#define LR2_SHFL(v, src) __shfl((v), (src), 64)
#define LR2_POPC64(m) __popcll(m)
#define LR2_POPC32(m) __popc(m)
#define LR2_RANK(m, lane) ((int)__builtin_amdgcn_mbcnt_hi((unsigned)((m) >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)(m), 0u)))
#define LR2_SHIFT_IN_LT1(w, c) sasa_shift_in_lt1((w), (c))
#define LR2_SHIFT_IN_LT(w, v, lim) sasa_shift_in_lt((w), (v), (lim))
/* a value every lane of the wave holds alike, moved to a scalar register: loops and branches on it become scalar
   control flow instead of exec-mask bookkeeping (the compiler cannot see that an LDS read or a shuffle is uniform) */
#define LR2_UNIFORM(v) __builtin_amdgcn_readfirstlane(v)
#define LR2_READLANE(v, src) __builtin_amdgcn_readlane((v), (src)) /* src wave-uniform */
/* index arithmetic on small non-negative numbers: v_mul_u32_u24 runs at full rate, the 32-bit v_mul_lo_u32 /
   v_mul_hi at a quarter of it (what the compiler emits when it cannot see that an index is small) */
#define LR2_MUL24(a, b) ((int)__umul24((unsigned)(a), (unsigned)(b)))
/* (the product's low 32 bits AS AN UNSIGNED NUMBER: the header's __umul24 returns int, and a right shift of a product
   with bit 31 set was arithmetic - v_ashrrev_i32 - until round 5: P1 decoded i / hc wrongly from i * 2^17 >= 2^31, i.e.
   for the 16 384th candidate of a row onwards; found by tests/test_adversarial.py's giant-cell case on the MI355X) */
#define LR2_UMUL24(a, b) ((unsigned)__umul24((unsigned)(a), (unsigned)(b)))
#define LR2_RCPF(x) __builtin_amdgcn_rcpf(x)
#define LR2_RSQF(x) __builtin_amdgcn_rsqf(x)
#define LR2_SQRTF(x) __builtin_amdgcn_sqrtf(x)
#define LR2_ADD64_LDS(p, v) atomicAdd((p), (v))
#define LR2_OR_LDS(p, v) atomicOr((p), (v))
#define LR2_INC_LDS(p) ((void)__hip_atomic_fetch_add((p), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) /* result unused: ds_add_u32, nothing to wait for */
#endif

namespace sasa {

/* Inclusive prefix sum / running maximum over the lanes of the wave (values >= 0).  On the device: data-parallel
 * primitives, one VALU instruction per step and no trip through the LDS crossbar (a shuffle is a ds_bpermute plus
 * its address arithmetic): shifts by 1, 2, 4, 8 inside the rows of 16 lanes (zeros shifted in), then the last lane
 * of row 0 / 2 into row 1 / 3 (row_bcast15) and lane 31 into the upper half (row_bcast31). */
#ifdef SASA_EMU
SASA_D int lr2_scan_add(int v, int lane)
{
    for (int d = 1; d < 64; d <<= 1) {
        const int o = LR2_SHFL(v, lane >= d ? lane - d : lane);
        if (lane >= d) v += o;
    }
    return v;
}
SASA_D int lr2_scan_add16(int v, int lane) /* inclusive prefix sum inside each row of 16 lanes */
{
    for (int d = 1; d < 16; d <<= 1) {
        const int o = LR2_SHFL(v, (lane & 15) >= d ? lane - d : lane);
        if ((lane & 15) >= d) v += o;
    }
    return v;
}
SASA_D int lr2_scan_max16(int v, int lane) /* running maximum inside each row of 16 lanes */
{
    for (int d = 1; d < 16; d <<= 1) {
        const int o = LR2_SHFL(v, (lane & 15) >= d ? lane - d : lane);
        if ((lane & 15) >= d) v = v > o ? v : o;
    }
    return v;
}
#else
#define LR2_DPP(v, ctrl, rows) __builtin_amdgcn_update_dpp(0, (v), (ctrl), (rows), 0xf, true)
SASA_D int lr2_scan_add(int v, int)
{
    v += LR2_DPP(v, 0x111, 0xf); /* row_shr:1 */
    v += LR2_DPP(v, 0x112, 0xf); /* row_shr:2 */
    v += LR2_DPP(v, 0x114, 0xf); /* row_shr:4 */
    v += LR2_DPP(v, 0x118, 0xf); /* row_shr:8 */
    v += LR2_DPP(v, 0x142, 0xa); /* row_bcast15 into rows 1 and 3 */
    v += LR2_DPP(v, 0x143, 0xc); /* row_bcast31 into rows 2 and 3 */
    return v;
}
SASA_D int lr2_scan_add16(int v, int)
{
    v += LR2_DPP(v, 0x111, 0xf); /* row_shr:1 */
    v += LR2_DPP(v, 0x112, 0xf); /* row_shr:2 */
    v += LR2_DPP(v, 0x114, 0xf); /* row_shr:4 */
    v += LR2_DPP(v, 0x118, 0xf); /* row_shr:8 */
    return v;
}
SASA_D int lr2_scan_max16(int v, int)
{
    int o;
    o = LR2_DPP(v, 0x111, 0xf); v = v > o ? v : o;
    o = LR2_DPP(v, 0x112, 0xf); v = v > o ? v : o;
    o = LR2_DPP(v, 0x114, 0xf); v = v > o ? v : o;
    o = LR2_DPP(v, 0x118, 0xf); v = v > o ? v : o;
    return v;
}
#endif

SASA_D double lr2_shfl_f64(double v, int src) /* v of lane src */
{
    unsigned long long b;
    memcpy(&b, &v, 8);
    const unsigned lo = (unsigned)LR2_SHFL((int)(unsigned)b, src), hi = (unsigned)LR2_SHFL((int)(unsigned)(b >> 32), src);
    b = (unsigned long long)lo | ((unsigned long long)hi << 32);
    memcpy(&v, &b, 8);
    return v;
}

/* x / y, correctly rounded, from r = RN(1 / y): q = RN(x r), then one correction with the exact remainder
 * (Markstein).  Checked against the division for every y = 1 .. 1024 on 2e8 operands (tools/dev/div_by_slices_check.c):
 * identical.  Four instructions where the hardware division sequence takes about thirty. */
SASA_D double lr2_div_ns(double x, double y, double r)
{
    const double q = x * r;
    return fma(fma(-q, y, x), r, q);
}
SASA_D int lr2_div9(int v) { return LR2_MUL24(v, 57) >> 9; }  /* v / 9 for 0 <= v < 69 */
SASA_D int lr2_div3(int v) { return LR2_MUL24(v, 43) >> 7; }  /* v / 3 for 0 <= v < 44 */

#ifndef LR2_STOP_AFTER /* dev only (tools/build_variant.sh): return after phase k, for instruction attribution */
#define LR2_STOP_AFTER 99
#endif
#define LR2_STOP(k) do { if (LR2_STOP_AFTER == (k)) { lr2_pre_none(pre); return 0; } } while (0) /* (whatever was fetched ahead for the next tile is incomplete: it walks its chain itself) */
#ifndef LR2_MARK /* dev only (-DSASA_PHASE_TIMING in gpu_engine.hip): wall clock of lane 0 at the phase boundaries */
#define LR2_MARK(k) do { } while (0)
#define LR2_MARK_BEGIN do { } while (0)
#endif

/* h = 1/(2 sqrt x) for the screening and the arc pass (the same bits in both): one coupled Goldschmidt step on
 * the hardware seed (relative error 4e-15; -DLR2_EXACT_H2: sqrt_rh's two and a half steps, 1 ulp).  With the
 * degree-12 acos_fast2 this spends two of the eight orders of magnitude between the reference-level 1e-12 and the
 * 1e-4 A^2 contract: -2.5 % kernel time (round 3, measured). */
#ifndef LR2_EXACT_H2
#define LR2_H2(x, h) do { const double x_ = (x), y_ = SASA_RSQ(x_), g_ = x_ * y_, h_ = 0.5 * y_; (h) = fma(h_, fma(-h_, g_, 0.5), h_); } while (0)
#else
#define LR2_H2(x, h) do { double g_; sqrt_rh((x), g_, (h)); } while (0)
#endif
/* The screening's limit for v = b' + a' t (lr2_record): a neighbor cuts an arc when |v| h < 1, h = 1/(2 Ri') as LR2_H2
   made it.  Comparing v with T = (1 - 2^-49) / h spares the product per (neighbor, slice); T from 4 A h (= 1/h up to
   h's own 4e-15) and one Newton step (T h = 1 to 2^-52), so |v| < T guarantees |v| h < 1 - 2^-50: the arc pass never
   takes the root of a negative number (lr2_arc_alpha), and the two passes agree on what an arc is because both read
   the mask.  Against the exact rule (|v| / (2 Ri') < 1) a decision can differ only where cos(alpha) is within 2e-15 of
   +-1: an arc of half-width < 6e-8 rad, or one that leaves that much of its circle (tests/test_adversarial.py). */
SASA_D double lr2_arc_limit(double A, double h)
{
    const double T0 = (4.0 * A) * h;
    const double T1 = fma(T0, fma(-T0, h, 1.0), T0);
    return T1 * (1.0 - 0x1p-49);
}
#define LR2_LANES 64
/* The tile shape of the library's default parameters on sparse input - 6 atoms x 20 slices, two mask words, two
   spilled stack levels (what lr2_choose_cfg gives coils and most proteins at 20 slices) - has a build of its own in
   which these four are compile-time constants (template parameter SHAPE = 1): LDS addresses become immediates, the
   index arithmetic folds, and the kernel spills 44 scalar registers instead of 104 (round 4: wave instructions -4 %,
   kernel -1 %).  SHAPE = 0 reads them from the arguments. */
#define LR2_SHAPE_TA 6
#define LR2_SHAPE_NS 20
#define LR2_SHAPE_MW 2
#define LR2_SHAPE_DS 2
/* Three more shapes have builds of their own since the end of round 4, when the kernel had become bound by its
   instruction count alone: 2 = 3 atoms x 100 slices, two mask words (coils at 100 slices: BASELINE
   configs[2] as written), 3 = 3 x 20 with three mask words (protein density: the lattice globules and, since the chooser
   learnt at the end of round 5 that three atoms beat the four that just fit, the reference's PDB entries), 4 = 4 x 20
   with three (inputs a little less dense); all with two spilled stack levels. */
SASA_HD constexpr int lr2_shape_ta(int s) { return s == 1 ? LR2_SHAPE_TA : (s == 4 ? 4 : 3); }
SASA_HD constexpr int lr2_shape_ns(int s) { return s == 2 ? 100 : LR2_SHAPE_NS; }
SASA_HD constexpr int lr2_shape_mw(int s) { return s <= 2 ? 2 : 3; }
SASA_HD constexpr int lr2_shape_ds(int s) { return (void)s, LR2_SHAPE_DS; }
#define LR2_A_TA(a) (SHAPE ? lr2_shape_ta(SHAPE) : (a).TA)
#define LR2_A_NS(a) (SHAPE ? lr2_shape_ns(SHAPE) : (a).ns)
#define LR2_A_MW(a) (SHAPE ? lr2_shape_mw(SHAPE) : (a).mw)
#define LR2_A_DS(a) (SHAPE ? lr2_shape_ds(SHAPE) : (a).ds)
#define LR2_NONE 0xffff

struct Lr2Args {
    const Quad *sq;
    const SortIdx *s_idx;
    const GridS *grid;
    const int *cell_start;
    const unsigned long long *cell_tbl; /* compact cell table (sasa_kernels.h, PipeArgs), or null */
    const int *cell_first;
    int n_atoms, n_tiles;
    int TA;     /* atoms per tile, 9*TA <= 64 */
    int ns;     /* slices per atom */
    int pool;   /* neighbor records per tile (even) */
    int mw;     /* 32-bit mask words per item: at most 32*mw neighbors per atom */
    int ds;     /* spilled levels of the arc stack */
    int refill; /* waiting lanes that trigger a refill of the arc pass */
    int cover;  /* cover filter (see lr2_cover_*): on for tiles with at least this many neighbor records per atom; 0: off */
    double *sasa;
    /* Work that does not fit this launch's LDS capacities: the wave redoes the tile as two halves on the spot; a
       half (or a single atom's tile) that still does not fit is appended to the next launch's work list as
       (first atom | atoms << 32), or — ovf_atoms — atom by atom for the last launch.  No list: an error. */
    int *ovf_count;
    long long *ovf_items;
    int *ovf_atoms;               /* instead of ovf_items: a list of single atoms (the last launch works atom by atom) */
    int *split_count;             /* [64] tiles split in place, counted in 64 buckets (statistics) */
    const long long *work_items;  /* (first atom | atoms << 32) to (re)do; null in the main launch (all tiles) */
    const int *work_count;
    /* test hook (freesasa_gpu_lr_neighbors_dev): stop after the neighbor discovery and report, in original atom
       order, every atom's neighbor count and (optionally, nb_cap per atom) its neighbors */
    int *nn_out;
    int *nb_out;
    int nb_cap;
    int *status;
    double inv_ns; /* 1.0 / ns, correctly rounded (host division): lr2_div_ns */
    /* The main launch in two parts, so that its last workgroups are short ones: workgroups [0, seg_grid) share the tiles
       [0, seg_tiles) (a dozen each), the others share the rest (two or three each).  Workgroups start in the order of
       their numbers; those that start last decide how ragged the end of the launch is: a third of a workgroup's
       duration on average (measured: kernel time = 9.45 ms + 0.34 x that duration over grids of 4 096 .. 147 456
       workgroups, 1e7 atoms).  seg_grid 0: one part (work lists, small launches). */
    int seg_grid, seg_tiles;
    int hooks; /* bit 0: nn_out is set, bit 1: nb_out is set (what the tile body tests; the pointers themselves are cold) */
    int walk;  /* host side only (kl_lr2_main): launch the main launch's walking build - most tiles of the last batch had an atom beyond LR2_WALK_Z */
    int prune; /* contained caps (lr2_prune_contained): caps wanted in an atom's list | the list's capacity << 8 (lr2_prune_arg); 0: off */
};
#include <stddef.h>
/* Arguments only rare paths need (overflow lists, statistics, the test hooks): read from the kernel-argument
 * segment where they are used, so that they do not hold scalar registers through the whole tile loop (the kernel is
 * short of them: ~130 spilled to VGPR lanes, and where the reloads land decides about 3 % of its speed).  Lr2Args is
 * the kernel's only parameter. */
#ifdef SASA_EMU
#define LR2_COLD(a, field) ((a).field)
#define LR2_WARM(a, field) ((a).field)
#else
#define LR2_COLD(a, field) (*(const decltype(sasa::Lr2Args::field) *)lr2_cold_arg(offsetof(sasa::Lr2Args, field)))
/* (P0's pointers and the result pointer, used once or twice per tile, stay cold in the build for the default tile shape
   too, although it has scalar registers to spare: as ordinary arguments they cost 32 more spilled scalar registers and
   0.7 % of the kernel's time - round 4, measured) */
#define LR2_WARM(a, field) LR2_COLD(a, field)
__device__ __forceinline__ const char *lr2_cold_arg(size_t off)
{
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (const char *)p + off;
}
#endif

/* One neighbor record of P3..P6, 24 B: cos(alpha) of the arc the neighbor cuts at slice height t is
 * (b + a t) * 1/(2 Ri') (see lr2_record); beta = direction of the neighbor in the slice plane.  Kept as two
 * arrays: the coefficient pair (16 B, one ds_read_b128: 4 LDS cycles where the ds_read2_b64 a 24-byte record
 * needs takes 8; MI355X_MICROARCH.md, LDS) and beta (8 B, read by the arc pass only). */
struct __attribute__((aligned(16))) Ab16 { double a, b; };
/* a component of the arc union, or two neighboring sort keys: 16 bytes at a 16-byte boundary (one ds_read_b128) */
struct __attribute__((aligned(16))) Arc2 { double s, e; };

/* LDS layout of one tile (byte offsets), shared by the host (launch size) and the device */
struct Lr2Layout {
    int o_atoms, o_ints, o_rec, o_mask, o_queue, o_tc, o_r2, o_tag, total;
};
SASA_HD int lr2_a16(int v) { return (v + 15) & ~15; }
SASA_HD int lr2_n_ints(int TA, int mw) { return 6 * TA + 1 + 8 + TA * mw; }
SASA_HD int lr2_n_row_ints(int TA) { return 9 * TA + 9 * TA + (9 * TA + 2); } /* rowlo, rinfo, cpre: P0 and P1 only */
SASA_HD Lr2Layout lr2_layout(int TA, int ns, int pool, int mw, int ds)
{
    Lr2Layout L;
    const int items = TA * ns;
    int p = 0;
    L.o_atoms = p; p += 32 * TA + lr2_a16(8 * TA);
    L.o_ints = p;  p += lr2_a16(4 * lr2_n_ints(TA, mw));
    /* records (24 B each; P3..P6), masks and queue (P4..P6), slice heights / areas (P4..P7).  The hits of P1
       (32 B each) lie over all of them until P3 has taken every hit into registers */
    L.o_rec = p;   p += lr2_a16(24 * pool);
    L.o_mask = p;  p += lr2_a16(4 * items * mw);
    L.o_queue = p; p += lr2_a16(2 * items);
    L.o_tc = p;    p += lr2_a16(8 * items);
    if (p - L.o_rec < 32 * pool) p = L.o_rec + 32 * pool;
    /* R2: the candidate rows of P0/P1 where the sort keys of P3 will be, and the hit tags (P1..P3); then queue
       scratch (P4, P5), then the arc stack (P6) */
    int r2k = lr2_a16(8 * pool);
    if (r2k < lr2_a16(4 * lr2_n_row_ints(TA))) r2k = lr2_a16(4 * lr2_n_row_ints(TA));
    int r2 = r2k + lr2_a16(pool); /* (a byte per hit: its atom) */
    const int r2q = lr2_a16(2 * items) + 256, r2s = 16 * LR2_LANES * (ds > 0 ? ds : 1); /* (one column level even when ds == 0: see lr2_union_step) */
    if (r2q > r2) r2 = r2q;
    if (r2s > r2) r2 = r2s;
    L.o_r2 = p;
    L.o_tag = p + r2k;
    p += r2;
    L.total = p;
    return L;
}

struct Lr2Mem {
    Quad *atom;   /* [TA] x, y, z, R + probe of the tile atoms */
    double *adel; /* [TA] 2 Ri / ns (ref: src/sasa_lr.c:304) */
    int *acell, *lead, *gsz, *acnt, *aoff, *sorig, *flags, *hist;
    unsigned *cmask; /* [TA*mw] cover filter: the atom's neighbors with the largest caps, as bits of its (beta-sorted) list */
    int *rowlo, *rinfo, *cpre; /* [9 TA], [9 TA], [9 TA + 2]: of the rows that have work: first candidate, leading atom | group size << 4 | items per candidate << 8, prefix of P1's work items (over the keys until P3) */
    double *it_tc;  /* [items] 1/(4 Ri') of an item with arcs until its slice is done, then (and for the others from P4 on) its area */
    unsigned *it_mask; /* [items*mw] neighbors that cut an arc */
    unsigned short *queue; /* [items] items with arcs, heaviest first: item | atom << 10 */
    unsigned short *qtmp;  /* [items] bin and arrival order of an item before the bins are laid out */
    Quad *hits;     /* [pool] (xd, yd, zd, Rj) of the neighbors found, in order of discovery */
    Ab16 *ab;       /* [pool] records, sorted by beta inside each atom's list: coefficients ... */
    double *beta;   /* [pool] ... and direction */
    double *keys;   /* [pool] beta with the list position in its low mantissa bits */
    unsigned char *tag; /* [pool] atom of a hit */
    Arc2 *stack;     /* [ds][64] */
};
/* flags: 0 tile overflow, 1 stack overflow, 2 max neighbor count, 6 atoms with a coincident neighbor of equal radius (bits) */

template <int SHAPE = 0>
SASA_D Lr2Mem lr2_carve(const Lr2Args &a, char *smem)
{
    const Lr2Layout L = lr2_layout(LR2_A_TA(a), LR2_A_NS(a), a.pool, LR2_A_MW(a), LR2_A_DS(a));
    const int TA = LR2_A_TA(a);
    Lr2Mem m;
    m.atom = (Quad *)(smem + L.o_atoms); m.adel = (double *)(smem + L.o_atoms + 32 * TA);
    int *q = (int *)(smem + L.o_ints);
    m.acell = q; q += TA; m.lead = q; q += TA; m.gsz = q; q += TA; m.acnt = q; q += TA; m.aoff = q; q += TA + 1;
    m.sorig = q; q += TA; m.flags = q; q += 8;
    m.cmask = (unsigned *)q;
    m.rowlo = (int *)(smem + L.o_r2); m.rinfo = m.rowlo + 9 * TA; m.cpre = m.rinfo + 9 * TA;
    m.it_tc = (double *)(smem + L.o_tc);
    m.it_mask = (unsigned *)(smem + L.o_mask);
    m.queue = (unsigned short *)(smem + L.o_queue);
    m.hits = (Quad *)(smem + L.o_rec);
    m.ab = (Ab16 *)(smem + L.o_rec);
    m.beta = (double *)(smem + L.o_rec + 16 * a.pool); /* (the pool is even) */
    m.keys = (double *)(smem + L.o_r2);
    m.tag = (unsigned char *)(smem + L.o_tag);
    m.hist = (int *)(smem + L.o_r2);
    m.qtmp = (unsigned short *)(smem + L.o_r2 + 256);
    m.stack = (Arc2 *)(smem + L.o_r2);
    return m;
}

/* The record of neighbor j of atom i.  With t = z - zi (slice height above the centre of i),
 * A = Ri'^2 = Ri^2 - t^2, B = Rj'^2 = Rj^2 - (zd - t)^2, D = dij^2 (zd = zj - zi):
 *     A + D - B = (Ri^2 - Rj^2 + D + zd^2) - 2 zd t            — linear in t, the t^2 cancel —
 * so the reference's acos argument (src/sasa_lr.c:335) is
 *     cos(alpha) = (A + D - B) / (2 Ri' dij) = (b' + a' t) * 1/(2 Ri'),
 *     a' = -2 zd / dij,  b' = (Ri^2 - Rj^2 + dij^2 + zd^2) / dij.
 * As in sasa_kernels.h (lr_record_of) the reference's three geometric tests (:320-331) are
 * cos(alpha) >= 1 (no arc) and cos(alpha) <= -1 (slice buried).  dij == 0 (centres on one vertical
 * line): the sign of A + D - B decides like the reference's tests; a' and b' then carry the factor
 * 2^500 instead of 1/dij (cos(alpha) = +-huge, or 0 * huge = 0 in the one case where the reference
 * divides 0 by 0; two coincident atoms of equal radius get b' = NaN: no arc, and P7 returns NaN for the atom as
 * the reference does — duplicate atom records do not pass for areas). */
SASA_D void lr2_record(double xd, double yd, double zd, double rj, double ri, double &ap, double &bp, double &Kout, double &d3sq, double &inv_d)
{
    const double D = xd * xd + yd * yd; /* ref: src/nb.c:438 */
    double h;
    LR2_H2(D, h); /* 1/(2 dij) to 4e-15, as the slices' 1/(2 Ri'); every lane computes (D == 0: not a number, replaced below) */
    const double inv = D > 0 ? 2.0 * h : 0x1p500; /* 1/dij */
    inv_d = D > 0 ? inv : 0.0;
    d3sq = D + zd * zd;
    const double K = (ri * ri - rj * rj) + d3sq;
    Kout = K;
    ap = -2.0 * zd * inv;
    bp = K * inv;
    if (!(D > 0) && zd == 0 && K == 0) bp = NAN;
}

/* Half-width of the arc a neighbor cuts at slice height t: acos((b' + a' t) / (2 Ri')) (lr2_record).  The arc pass gets
 * hh = 1/(4 Ri') from the screening and forms acos_fast2's argument z = (1 - |c|)/2 = 1/2 - |b' + a' t| hh in ONE fma, the
 * product with 1/(2 Ri') never rounded on its own (round 4: an instruction per arc less).  The screening has decided
 * RN(|b' + a' t| / (2 Ri')) < 1, i.e. <= 1 - 2^-53, so the exact product is below 1 - 2^-54 and z > 0 after its single
 * rounding: no arc reaches the square root with a negative argument. */
SASA_D double lr2_arc_alpha(double t, const Ab16 &ab, double hh)
{
    const double v = fma(t, ab.a, ab.b);
    return acos_fast2_z(fma(fabs(v), -hh, 0.5), v);
}

/* Arc union on raw end points (inf may be negative, sup may exceed 2 pi: every arc contains its beta
 * in [0, 2 pi], so only the lowest component can start below 0 and only the highest can end above
 * 2 pi; lr2_sweep folds them back).  Arcs arrive ordered by beta, so the disjoint components form a
 * stack (sasa_kernels.h, lr_arcs32): the two highest live in registers (ts,te above bs,be; be = -inf
 * while there is no second one), the ones below them in the lane's LDS column. */
struct Lr2Union {
    double ts, te, bs, be;
    int depth;
};
SASA_D void lr2_union_reset(Lr2Union &u) { u.ts = 0; u.te = -INFINITY; u.bs = 0; u.be = -INFINITY; u.depth = 0; }
SASA_D int lr2_med3(int v, int lo, int hi)
{
#ifdef SASA_EMU
    return v < lo ? lo : (v > hi ? hi : v);
#else
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "s"(lo), "v"(hi));
    return r;
#endif
}
/* the level of the lane's LDS column that holds component (depth - 2) from below, kept inside the column: level 0 for
   fewer than two components, the last level beyond the column's end (maxd tells).  The "- 2" is folded into the
   column's address (one clamp and one shift-add per push or pop) */
SASA_D Arc2 &lr2_level(Arc2 *stk, int depth, int ds)
{
    return (stk - 2 * LR2_LANES)[lr2_med3(depth, 2, ds > 0 ? ds + 1 : 2) * LR2_LANES];
}
/* maxd: the largest depth the lane has seen (a tile whose stack column was too short is redone) */
SASA_D void lr2_union_step(double inf, double sup, Lr2Union &u, Arc2 *stk, int ds, int &maxd)
{
#ifdef LR2_UNION_SELECT /* (round 3's form: selects around one short branch; measured against the if / else form in round 4) */
    const bool fresh = inf > u.te; /* te = -inf while there is no component */
    const double mts = SASA_MIN(u.ts, inf), mte = SASA_MAX(u.te, sup);
    if (fresh) {
        Arc2 t; t.s = u.bs; t.e = u.be;
        lr2_level(stk, u.depth, ds) = t;
        u.bs = u.ts; u.be = u.te;
    }
    u.ts = fresh ? inf : mts;
    u.te = fresh ? sup : mte;
    u.depth += fresh ? 1 : 0;
    maxd = maxd > u.depth ? maxd : u.depth;
    while (u.be >= u.ts) {
        u.ts = SASA_MIN(u.ts, u.bs);
        --u.depth;
        if (u.depth >= 2) {
            const Arc2 lo = lr2_level(stk, u.depth, ds);
            u.bs = lo.s; u.be = lo.e;
        } else {
            u.be = -INFINITY;
        }
    }
#else
    if (inf > u.te) { /* te = -inf while there is no component */
        /* a new component on top: the second goes to the LDS column, the top one becomes the second.  With fewer than
           two components the store lands in level 0, which is not in use then (it is written again, properly, by
           the push that makes a third component) */
        Arc2 t; t.s = u.bs; t.e = u.be;
        lr2_level(stk, u.depth, ds) = t;
        u.bs = u.ts; u.be = u.te;
        u.ts = inf; u.te = sup;
        ++u.depth;
        maxd = maxd > u.depth ? maxd : u.depth;
    } else {
        u.ts = SASA_MIN(u.ts, inf);
        u.te = SASA_MAX(u.te, sup);
        while (u.be >= u.ts) { /* the merged component reaches the one below it (never after a push: be = old te < inf) */
            u.ts = SASA_MIN(u.ts, u.bs);
            --u.depth;
            if (u.depth >= 2) {
                const Arc2 lo = lr2_level(stk, u.depth, ds);
                u.bs = lo.s; u.be = lo.e;
            } else {
                u.be = -INFINITY;
            }
        }
    }
#endif
}

/* Exposed arc length from the final components C_0 < C_1 < ... (disjoint, ascending, raw end points).
 * ref: src/sasa_lr.c:340-351 (arcs through the origin) and :389-408 (sweep).
 * Every component contains a direction in [0, 2 pi], so on the circle only two of them can reach into the others'
 * gaps: the lowest, whose part below 0 comes back as [s_0 + 2 pi, 2 pi], and the highest, whose part above 2 pi comes
 * back as [0, e_top - 2 pi].  With hi = s_0 + 2 pi and lo = e_top - 2 pi the exposed length is therefore, in closed form,
 *     max(0, hi - e_top)  +  sum_k max(0, min(s_k+1, hi) - max(e_k, lo))
 * (the gap across the origin, then what the two wrapped pieces leave of every gap between neighbors): 16 instructions
 * for the usual one or two components where the reference's sweep over normalised pieces (rounds 1-3: restated with
 * selects, ~65 instructions per item switch) walks a sorted list.  A covered circle is exactly 0 here as there (every
 * term is clamped at 0); otherwise the two differ by the rounding of one or two additions of 2 pi. */
SASA_D double lr2_sweep(const Lr2Union &u, const Arc2 *stk, int ds)
{
    const int depth = u.depth;
    const bool two = depth >= 2;
    double s0 = two ? u.bs : u.ts;      /* start of the lowest component */
    double e_below = two ? u.be : INFINITY; /* end of the component below the top one (none: no gap to count) */
    double sum = 0;
    if (depth > 2) { /* (rare) components in the LDS column, ascending */
        const Arc2 k0 = stk[0];
        s0 = k0.s;
        const double hi_ = s0 + SASA_TWOPI, lo_ = u.te - SASA_TWOPI;
        double pe = k0.e;
        for (int c = 1; c < depth - 2; ++c) {
            const Arc2 k = stk[(c < ds ? c : 0) * LR2_LANES];
            sum += SASA_MAX(SASA_MIN(k.s, hi_) - SASA_MAX(pe, lo_), 0.0);
            pe = k.e;
        }
        sum += SASA_MAX(SASA_MIN(u.bs, hi_) - SASA_MAX(pe, lo_), 0.0);
    }
    const double hi = s0 + SASA_TWOPI, lo = u.te - SASA_TWOPI;
    sum += SASA_MAX(SASA_MIN(u.ts, hi) - SASA_MAX(e_below, lo), 0.0);
    sum += SASA_MAX(hi - u.te, 0.0);
    return depth == 0 ? SASA_TWOPI : sum; /* ref: :392 */
}

/* test hook (freesasa_gpu_arc_union_dev): the exposed length of set `k`'s arcs, given sorted by their mid-points,
 * through the arc union and the sweep of the arc pass (ref KATs: src/sasa_lr.c:455-475).  One lane per set. */
SASA_D double lr2_arc_kat(const double *arcs, const int *first, int k, Arc2 *stk, int ds)
{
    Lr2Union u;
    lr2_union_reset(u);
    int maxd = 0;
    for (int i = first[k]; i < first[k + 1]; ++i) lr2_union_step(arcs[2 * i], arcs[2 * i + 1], u, stk, ds, maxd);
    return maxd - 2 > ds ? NAN : lr2_sweep(u, stk, ds);
}

#ifndef LR2_NB_UNROLL
#ifndef LR2_ARC_STEPS
#define LR2_ARC_STEPS 2 /* arc steps between two looks at how many lanes wait for a refill */
#endif
#define LR2_NB_UNROLL 1 /* candidates a lane has in flight per round of P1.  Round 2 measured 1 / 2 / 3 on coils: 3.59 / 3.63 / 3.58 ms per 3e6 atoms, and 3 it was; the items of a tile are a multiple of 64 x 3 only by chance, and since the kernel's time became its instruction count the padded rounds cost what they compute: round 5, MI355X, kernel ms on the coil batch / the PDB entries, 1: 8.84 / 6.36, 2: 8.86 / 6.38, 3: 8.90 / 6.43, 4: 9.22 / 6.55; with one per round and the NEXT item's candidate fetched while this one is tested (built, measured): no difference - four waves per SIMD hide the load */
#endif
#define LR2_P1_G 3 /* atoms of a cell group one work item of P1 tests its candidate against */
/* ... in the build for tiles of four atoms at protein density (SHAPE 4) all four: there the four atoms of a tile mostly
   share their cell (1a0q: 563 of 1062 groups), and with three to an item the fourth atom costs every candidate a second
   item - its fetch, its decoding and two idle test slots (round 5: the phase -19 % by count) */
#ifndef LR2_P1_G_COILS
#define LR2_P1_G_COILS LR2_P1_G /* (the coil shape, 6 x 20: groups of every size from one to six; round 5, MI355X, kernel ms with 2 / 3 / 4 atoms to an item: 8.79 / 8.80 / 9.01 - as the count of instructions says) */
#endif
#define LR2_P1_G_OF(shape) ((shape) == 4 ? 4 : ((shape) == 1 ? LR2_P1_G_COILS : LR2_P1_G))
/* items per candidate of a group of gs atoms, g atoms to an item (gs <= 7) */
#define LR2_P1_HC(gs, g) ((g) == 4 ? ((gs) + 3) >> 2 : ((g) == 2 ? ((gs) + 1) >> 1 : lr2_div3((gs) + 2)))
#ifndef LR2_P1_ITEMS_MAX
#define LR2_P1_ITEMS_MAX 0x8000 /* P1's work items per tile, exclusive: the decode of an item's place in its row is exact below 2^15 (tests build with less to walk the hand-on path) */
#endif

/* a work item that fits no capacity of this launch, even halved */
SASA_D void lr2_overflow(const Lr2Args &a, int p0, int na, int err_code)
{
    int *const ovf_atoms = LR2_COLD(a, ovf_atoms), *const ovf_count = LR2_COLD(a, ovf_count);
    long long *const ovf_items = LR2_COLD(a, ovf_items);
    if (ovf_atoms) {
        const int w = SASA_ATOMIC_ADD_GLB(ovf_count, na);
        for (int k = 0; k < na; ++k) ovf_atoms[w + k] = p0 + k;
    } else if (ovf_items) {
        const int w = SASA_ATOMIC_ADD_GLB(ovf_count, 1);
        ovf_items[w] = (long long)p0 | ((long long)na << 32);
    } else {
        SASA_ATOMIC_MAX_GLB(&LR2_COLD(a, status)[ST_ERROR], err_code); /* the last launch: nothing left to hand the work to */
    }
}


/* Height of the mid-plane of slice s above the atom's centre: (s + 1/2) delta - Ri, delta = 2 Ri / ns.  The reference
 * walks there (z = zi - Ri - delta/2, then z += delta per slice, src/sasa_lr.c:304-307) and is off the exact plane by
 * up to ~20 roundings of |z| (1e-13 A for coordinates of hundreds of A); this closed form (one fma) is the exact plane
 * to half an ulp.  The two differ by less than the reference's own drift, the areas by ~1e-13 A^2 - a hundredth of what
 * the arc pass spends on its acos - and the slice table (a 20-step dependent loop on 6 of 64 lanes) is gone, its LDS
 * word per item free to carry 1/(2 Ri') from the screening to the arc pass. */
SASA_D double lr2_slice_height(int s, double delta, double Ri) { return fma((double)s + 0.5, delta, -Ri); }
/* ... and for atoms FAR from the origin the reference's own plane: it walks in absolute coordinates, z = zi - Ri - delta/2,
 * then z += delta per slice (src/sasa_lr.c:304-307), and every addition rounds to half an ulp of |z|, so at |z| = 1.6e4 A
 * its plane sits ~1e-11 A off the exact one - nothing for an ordinary arc, but where a neighbor's circle is TANGENT to
 * the atom's the arc's half-width is the square root of that (round 4, tests/test_adversarial.py: 2.9e-5 A^2 at 1.6e4 A
 * against 1.7e-7 at the origin, and growing with |z|).  Beyond LR2_WALK_Z an atom's planes are therefore walked exactly as
 * the reference walks them, t = z - zi (exact: the two are within a factor of two): the reference's plane bit for bit at
 * any distance from the origin.  The check is NOT in the ordinary tile (measured, round 5: a per-item test of |zi| cost
 * the 100-slice kernel 6 % of its instructions and 9 % of its time, the 20-slice one 0.9 %): the builds of the main launch
 * (WALK = false) hand a tile with such an atom on, like a tile whose lists do not fit, and the second launch's build
 * (WALK = true) walks - per ATOM, by its own z, so no area depends on which atoms share a tile.
 * Two paths outside this kernel do not follow the rule, and are documented limits: the last (slab) launch - the
 * first-generation kernel, atom by atom, for what overflowed the second launch's lists: pathological densities only - walks
 * to EVERY atom's planes as the reference does (sasa_kernels.h, lr_phase_slices), and resolutions above LR2_NS_MAX slices per
 * atom in its strided form take z0 + (s + 1) delta everywhere; an atom that ends up there differs from this kernel's
 * value by the plane's ~1e-13 A, i.e. ~1e-13 A^2 (tangent constructions: tests/test_adversarial.py). */
#ifndef LR2_WALK_Z
#define LR2_WALK_Z 1024.0 /* (closed form inside: the reference's drift is <= 2.3e-12 A there; tangent constructions at |z| = 1000 A: tests/test_adversarial.py) */
#endif
template <bool WALK>
SASA_D double lr2_slice_height_at(int s, double delta, double Ri, double zi)
{
    if (!WALK || !(fabs(zi) > LR2_WALK_Z)) return lr2_slice_height(s, delta, Ri);
    double z = (zi - Ri) - 0.5 * delta; /* ref: src/sasa_lr.c:305 */
    for (int k = 0; k <= s; ++k) z += delta; /* ref: :307 */
    return z - zi;
}

/* The order of two neighbors whose beta agree in their leading 40 bits: by 12 bits of the pair's own geometry (the
 * low mantissa words of xd, yd, zd), not by when the neighbor was found — the order, and with it every bit of the
 * result, is then the same whatever the tile shape (atoms per tile, halves, the second launch), which is what lets
 * a resumed run reproduce its files byte for byte.  Two neighbors of one atom with equal keys (beta equal to 40 bits
 * AND equal tie bits: ~1e-16 per atom) would rank alike; P3 checks that the ranks of a list are a permutation (their
 * sum) and hands such a tile on, in the end to the slab launch, whose ranking goes by list position. */
SASA_D unsigned lr2_tie12(double xd, double yd, double zd)
{
    unsigned long long a, b, c;
    memcpy(&a, &xd, 8); memcpy(&b, &yd, 8); memcpy(&c, &zd, 8);
    const unsigned h = (unsigned)a ^ ((unsigned)b * 0x9E3779B1u) ^ ((unsigned)c >> 7) ^ ((unsigned)c << 11);
    return (h ^ (h >> 13)) & 0xfffu;
}

/* ---------------------------------------------------------------- cover filter (dense inputs)
 * At protein density 9 of 10 arcs belong to slices whose circle ends up fully covered (area 0; tools/dev/
 * cover_stats.cpp), and the arcs of the ~10 neighbors with the largest caps on the atom's sphere already cover 95 %
 * of those circles.  So, on tiles that are dense enough, P4 runs a cheap CONSERVATIVE test on the arcs of these
 * neighbors before an item is queued: half-widths from a polynomial lower bound of acos (no root; less 1e-9 rad),
 * one running component in beta order.  A circle this proves covered is covered with room to spare, and its area is
 * exactly 0 in the reference as well (arcs through the origin are cut at 0 and 2 pi: src/sasa_lr.c:340-351, so a
 * covered circle sums to sum = 0, sup = 2 pi exactly, :407); everything it cannot prove goes through the exact arc
 * pass as before.  The filter decides which slices are skipped, never what an area is.
 * Which neighbors: cos(theta_j) = (Ri^2 - Rj^2 + d^2) / (2 Ri d), the cap neighbor j cuts out of sphere i, in
 * 8 bins of 0.1 from 0.2 (a byte counter each, one 64-bit LDS word per atom); the bins that hold the
 * LR2_COVER_WANT largest caps, LR2_COVER_MAX neighbors at most. */
#ifndef LR2_COVER_WANT
#define LR2_COVER_WANT 12 /* (round 4, by the instruction counter on globules: 8 / 10 / 12 / 14 / 16 wanted with caps of 12 ... 32: 12 of at most 20 is the least, 2 % under round 3's 10 of 16) */
#endif
#define LR2_COVER_DENSITY 30 /* neighbor records per atom of a tile from which the filter pays (coils: ~20, proteins: 40-60) */
#ifndef LR2_COVER_MAX
#define LR2_COVER_MAX 20
#endif
SASA_D int lr2_cover_bin(double K, double d3sq, double ri)
{
    const float ct = (float)K * LR2_RSQF((float)d3sq) * LR2_RCPF(2.0f * (float)ri);
    const int b = (int)fmaf(ct, 10.0f, -2.0f);
    return b < 0 ? 0 : (b > 7 ? 7 : b);
}
/* the last bin needed for LR2_COVER_WANT neighbors, from the atom's byte counters */
SASA_D int lr2_cover_last_bin(unsigned long long h)
{
    int cum = 0, tb = 7;
    for (int b = 0; b < 8; ++b) {
        cum += (int)((h >> (8 * b)) & 255u);
        if (cum >= LR2_COVER_WANT && b < tb) tb = b;
    }
    return tb;
}
/* a lower bound of acos(c) for c <= 0.9, less a margin of 1e-9: asin x <= x + x^3/6 + 0.17 x^5 on [0, 0.9] and
 * asin x >= x + x^3/6 + 3/40 x^5 for x >= 0 (its series has positive terms only) */
SASA_D double lr2_acos_lower(double c)
{
    const double c2 = c * c, k5 = c < 0 ? 0.075 : 0.17;
    return (1.5707963267948966 - 1e-9) - fma(c * c2, fma(k5, c2, 1.0 / 6.0), c);
}

/* ---------------------------------------------------------------- contained caps (P1.5, round 6)
 * Neighbor j cuts the cap { u : u . n_j >= cos(theta_j) } out of sphere i (n_j: direction of j, cos(theta_j) = (Ri^2 - Rj^2
 * + d^2) / (2 Ri d), as the cover filter's bins).  Where cap j lies INSIDE the cap of another neighbor k - angle(n_j, n_k) +
 * theta_j <= theta_k - the arc of j lies inside the arc of k on every slice that j cuts, and a slice that j buries k
 * buries too: the union of the slice's arcs, whose components are minima and maxima of the arcs' end points
 * (lr2_union_step), is the same with and without j, bit for bit, and so is every area.  On random coils and on the
 * reference's PDB entries alike 45 % of all neighbor records are of this kind (42 % against the eight largest caps of
 * their atom), and they carry a third of all arcs.  So the hits of P1 are tested against the largest caps of their atom,
 * and those that lie inside one are dropped before the records are made: P3 ranks shorter lists, P4 screens fewer
 * records per item and the arc pass unites fewer arcs, for ~10 instructions per (hit, cap) pair here.
 *    The test is a SUFFICIENT one, in fp32: with c = cos(theta), s >= sin(theta) (from 1 - c^2 + 4e-6), cap j is dropped
 * for k only if c_k <= c_j - 4e-6 (k is the larger cap) and n_j . n_k >= c_k c_j + s_k s_j + 8e-6, i.e. cos(angle) >=
 * cos(theta_k - theta_j) + 8e-6.  The operands carry relative errors of a few 1e-7 (conversions, v_rsq_f32, three
 * roundings each), both sides of the comparison less than 2e-6 in all, so what passes has angle(n_j, n_k) + theta_j <
 * theta_k - 6e-6 rad: room to spare against the arcs' own rounding (1e-8 rad where acos is steep).  "Larger cap" is a
 * strict order, so a dropped cap always lies inside one that stays.  Coincident centres (d = 0) give directions that
 * are not numbers: such a record neither drops nor is dropped.
 *    "Bit for bit" needs one more condition.  The arc pass works on RAW end points beta -+ alpha with beta in [0, 2 pi]
 * (lr2_sweep folds what sticks out), so an arc that lies inside another ON THE CIRCLE may lie 2 pi away from it as an
 * interval of numbers - j just above beta = 0, k just below 2 pi - and then j's end points do become a component's, and
 * the sweep adds its 2 pi to other numbers (the same area to ~1e-14, not the same bits).  beta is atan2(yd, xd) + pi: the
 * cut is the negative x axis.  A cap is therefore dropped only for a cap on the same side of the x axis (yd of equal
 * sign, neither zero): then |beta_j - beta_k| < pi as numbers, the intervals are nested as numbers, and j's end points are
 * never the least or the greatest of a component.  Which caps are in an atom's lists only decides how many records are
 * dropped, never an area; it is nevertheless a function of the atom's neighbors alone (all hits of the leading bins of
 * cos(theta) that together are no more than a list holds - not the first that arrive), so that what is dropped, and with it
 * every path the tile takes afterwards (an arc stack that proves too short sends it to another launch), does not depend on
 * the tile's shape.  The phase runs BEHIND P2's decision whether the tile fits the launch: lists as found decide that, so
 * the same atoms take the same launches with the phase on and off.  (What remains: an item with more disjoint arc pieces
 * than any launch's stack holds goes to the last launch, whose first-generation arithmetic equals this kernel's to ~1e-11,
 * not to the bit; with contained arcs gone the stack may suffice.  Inputs of > 100 neighbors per atom: tools/dev/
 * prune_fuzz.py counts them.)  tests/test_emulation.py compares the bits with and without this phase, the GPU suite runs
 * both (FREESASA_AMD_PRUNE). */
#ifndef LR2_PRUNE
#define LR2_PRUNE 1
#endif
#ifndef LR2_PRUNE_ROUNDS
#define LR2_PRUNE_ROUNDS(pairs) ((pairs) ? 3 : 2) /* rounds of 64 hits a tile may have to be pruned: 6 atoms x 20 neighbors need two, often three; the builds for more than 128 items (60 - 90 hits per tile) keep registers in scratch with more than two */
#endif
#ifndef LR2_PRUNE_LIST
#define LR2_PRUNE_LIST 4   /* caps in one of an atom's two lists.  MI355X, final form, kernel ms on coils at 20 / 100 slices (mean of three runs): 2: 2.86 / 2.71, 3: 2.78 / 2.59, 4: 2.78 / 2.56, 6: 2.77 / 2.54 */
#endif
#define LR2_DEAD_TAG 0xff  /* tag of a dropped hit (atoms of a tile: < 8) */
struct __attribute__((aligned(16))) Lr2Cap { float x, y, z, c; };
SASA_HD int lr2_prune_arg(int want, int TA, int pool) /* Lr2Args::prune: > 0 on (the chooser's figure, at most what a list holds), 0: off */
{
    /* the lists lie where P3's sort keys will be (8 B per pool record): a histogram word per (atom, side), then 20 B per cap */
    if (8 * pool < (16 + 20 * LR2_PRUNE_LIST) * 2 * TA) return 0;
    return want > LR2_PRUNE_LIST ? LR2_PRUNE_LIST : (want < 0 ? 0 : want);
}
/* on (> 0) or off by tile shape and density (host side; the figure was the number of caps wanted per list while the lists
   were filled in order of arrival, and is only a switch since they are chosen by the bins alone).
   MI355X, kernel ms without / with the phase (tools/dev/prune_ab.sh, profiles/r06_prune_ab.txt, final form): coils at 100
   slices 2.93 / 2.56, at 50 slices 1.95 / 1.81, the reference's PDB entries at 100 slices 5.12 / 4.68.  At 20 slices the
   arcs it saves are a fifth as many and the phase costs the same: random coils (20 neighbors per atom, two or three rounds
   of hits per tile of six atoms) 2.86 / 2.76, but at protein density (41 - 48 neighbors, the cover filter already takes 2 of
   3 items before their arcs) the PDB entries 2.09 / 2.04 and the lattice globules 2.13 / 2.13 under the tool, and no
   difference in the bench line's keys (profiles/r06_prune_bench_ab.txt): there it stays off (the first form - one list per
   atom, filled in order of arrival - lost 2 - 5 % there). */
static inline int lr2_prune_want(int ns, bool dense) { return ns >= 32 ? 4 : (dense ? 0 : 3); }
SASA_D int lr2_cap_bin(float c) { const int b = (int)fmaf(c, 10.0f, -2.0f); return b < 0 ? 0 : (b > 7 ? 7 : b); } /* the cover filter's bins: 0.1 wide from 0.2 */
/* Two lists per atom, one for either side of the x axis (the same-side rule above): a hit is tested against the largest caps
   of ITS side only - half the trips of one list for the same caps, and no side test inside the loop (per side the 4 largest
   caps drop 38 % of all records on coils, 35 % on 1a0q; all pairs: 43 / 44 %). */
template <int RMAX>
SASA_D void lr2_prune_contained(const Lr2Mem &m, int nh, int TA, int want, int lane)
{
    constexpr int LK = LR2_PRUNE_LIST;
    (void)want; /* (on / off since the lists are chosen by the bins alone) */
    unsigned long long *const chist = (unsigned long long *)m.keys;  /* [2 TA] a byte per bin; then the list's last bin */
    Lr2Cap *const bigc = (Lr2Cap *)(chist + 2 * TA);                 /* [2 TA][LK] the largest caps: direction, cosine ... */
    Lr2Cap *const bigs = bigc + 2 * LK * TA;                         /* [2 TA] ... and the upper bounds of their sines, a list's four in one read */
    int *const fill = m.acell;                                       /* [2 TA] over acell and lead (P0 / P1 only): caps in the list */
    if (lane < 2 * TA) { chist[lane] = 0; fill[lane] = 0; }
    LR2_SYNC();
    float nx[RMAX], ny[RMAX], nz[RMAX], cc[RMAX]; /* (the hit's bin and its sine are made again where needed: registers are short here) */
    int lis[RMAX];
#define LR2_CAP_SIN(c) LR2_SQRTF(fmaf(-(c), (c), 1.0f + 4e-6f))
/* the hit's list: 2 atom + side; -1: none (no hit, or yd = 0 / not a number: on the cut or beside it, never dropped, never listed) */
#define LR2_CAP_LIST(r, gp) ((gp) < nh && ny[r] != 0.0f && ny[r] == ny[r] ? 2 * (int)m.tag[gp] + (ny[r] > 0.0f ? 1 : 0) : -1)
    for (int r = 0; r < RMAX; ++r) {
        const int gp = lane + LR2_LANES * r;
        nx[r] = ny[r] = nz[r] = cc[r] = 0.0f;
        lis[r] = -1;
        if (gp < nh) {
            const Quad hq = m.hits[gp];
            const double ri = m.atom[(int)m.tag[gp]].w;
            const double d2 = hq.x * hq.x + hq.y * hq.y + hq.z * hq.z;
            const double K = (ri * ri - hq.w * hq.w) + d2;
            const float inv = LR2_RSQF((float)d2);
            nx[r] = (float)hq.x * inv; ny[r] = (float)hq.y * inv; nz[r] = (float)hq.z * inv;
            float c = (float)K * inv * LR2_RCPF(2.0f * (float)ri);
            c = c < -1.0f ? -1.0f : (c > 1.0f ? 1.0f : c); /* (sphere i inside sphere j: the whole sphere, theta = pi) */
            cc[r] = c;
            const int li = lis[r] = LR2_CAP_LIST(r, gp);
            if (li >= 0) LR2_ADD64_LDS(&chist[li], 1ull << (8 * lr2_cap_bin(c))); /* (a byte per bin; a carry only changes which caps are listed) */
        }
    }
    LR2_SYNC();
    if (lane < 2 * TA) {
        /* The list: ALL hits of the leading bins that together hold no more than the list does - a set that depends on the
           atom's neighbors alone, not on the order in which a tile of some shape found them: what is dropped, and with it
           every path a tile takes from here on (an arc stack that is too short sends it to another launch), is the same
           for every tile shape.  The bins' running sums byte-parallel (x 0x01010101 adds every byte to the ones above it;
           sums beyond 255 carry into the next byte and at worst list nothing - an atom with that many hits on one side has
           left this launch long before), then the first byte that exceeds the list (its bit 7 after adding 127 - LK); the
           bins before it are listed (-1: none, the first bin alone is too many). */
        const unsigned long long h = chist[lane];
        const unsigned lo = (unsigned)h * 0x01010101u, hi = (unsigned)(h >> 32) * 0x01010101u + (lo >> 24) * 0x01010101u;
        const unsigned add = (unsigned)(127 - LK) * 0x01010101u;
        const unsigned mlo = (lo + add) & 0x80808080u, mhi = (hi + add) & 0x80808080u;
        const int tb = (mlo ? __builtin_ctz(mlo) >> 3 : (mhi ? 4 + (__builtin_ctz(mhi) >> 3) : 8)) - 1;
        chist[lane] = (unsigned long long)(long long)tb;
    }
    LR2_SYNC();
    for (int r = 0; r < RMAX; ++r) {
        const int li = lis[r];
        if (li < 0) continue;
        if (lr2_cap_bin(cc[r]) <= (int)chist[li]) {
            const int slot = SASA_ATOMIC_ADD_LDS(&fill[li], 1); /* (< LK by the choice of the bins; the order of the places does not matter: the test below asks for any) */
            if (slot < LK) {
                Lr2Cap q; q.x = nx[r]; q.y = ny[r]; q.z = nz[r]; q.c = cc[r];
                bigc[LK * li + slot] = q;
                ((float *)bigs)[LK * li + slot] = LR2_CAP_SIN(cc[r]);
            }
        }
    }
    LR2_SYNC();
    for (int r = 0; r < RMAX; ++r) {
        const int gp = lane + LR2_LANES * r;
        const int li = lis[r];
        if (li < 0) continue;
        const Lr2Cap *const L = bigc + LK * li;
        const float *const S = (const float *)bigs + LK * li;
        const float clim = cc[r] - 4e-6f, sj = LR2_CAP_SIN(cc[r]);
        bool inside = false;
        int nb = fill[li];
        nb = nb < LK ? nb : LK;
        for (int k = 0; k < nb; ++k) { /* (the wave runs as many trips as its longest list) */
            const Lr2Cap q = L[k];
            const float dot = fmaf(nz[r], q.z, fmaf(ny[r], q.y, nx[r] * q.x));
            const float rhs = fmaf(sj, S[k], fmaf(cc[r], q.c, 8e-6f));
            inside = inside || (q.c <= clim && dot >= rhs); /* (a direction that is not a number compares false) */
        }
        if (inside) {
            m.tag[gp] = (unsigned char)LR2_DEAD_TAG;
            SASA_ATOMIC_ADD_LDS(&m.acnt[li >> 1], -1);
        }
    }
#undef LR2_CAP_LIST
#undef LR2_CAP_SIN
    LR2_SYNC();
}

/* ---------------------------------------------------------------- P0's global loads, one tile ahead
 * What P0 needs from global memory is a chain: the atom's sort record -> (its structure's grid ->) the first atoms of
 * the cells at both ends of each of its 9 candidate rows.  Until round 3 every tile began by waiting for those round
 * trips (P0: 9 % of a wave's life for 3 % of its instructions).  Now the chain of tile k+1 is walked while tile k
 * computes: stage A (sort records) is issued before tile k is screened, stage B (cell table) before its queue is built,
 * stages B2 (compact cell table: first atoms of the cells) and C (the tile's own atoms) before its arcs are united,
 * and P0 of tile k+1 finds the values in its registers.  nx and ny of the
 * structure's grid ride in the sort record's flag word (cell_pack_grid), so the common chain has two links, not three.
 * A tile that was not announced (the first of a wave, halves of a split tile) walks the chain on the spot, as before. */
struct Lr2Pre {
    int p0;          /* first atom of the tile the values are of (-1: none) */
    long long rcf;   /* A, row lanes: cell | (flags, nx, ny) << 32 of the row's atom */
    int s0, s1;      /* B, row lanes: first atoms of the cells at both ends of the row's run (compact cell table: the cells themselves until B2) */
    unsigned long long w0, w1; /* B1, compact cell table: the table words of the two cells */
    int rfl;         /* B: bit 0 row outside the grid, bit 1 the atom leads its cell group */
    Quad q;          /* C, lanes < TA: the atom */
    int cell, so;    /* C: its cell and its original index */
};
SASA_D void lr2_pre_none(Lr2Pre &pre)
{
    pre.p0 = -1; pre.rcf = 0; pre.s0 = pre.s1 = 0; pre.w0 = pre.w1 = 0; pre.rfl = 1; pre.cell = 0; pre.so = 0;
    pre.q.x = pre.q.y = pre.q.z = 0; pre.q.w = 1;
}
/* (Loads without branches around them: a lane that has no row or no atom of the tile loads what the tile's last atom's
 * lane loads, and P0 sorts out who is who.  A join of two paths between a load and its use makes the compiler wait for
 * every load in flight at the join - the round trip this is here to hide.) */
template <int SHAPE>
SASA_D void lr2_pre_a(const Lr2Args &a, Lr2Pre &pre, int p0, int na, int lane)
{
    const int la = lr2_div9(lane);
    pre.p0 = p0;
    pre.rcf = LR2_WARM(a, s_idx)[p0 + (la < na ? la : na - 1)].cell;
}
template <int SHAPE>
SASA_D void lr2_pre_b(const Lr2Args &a, Lr2Pre &pre, int na, int lane)
{
    const int la = lr2_div9(lane), r = lane - 9 * la;
    const int c = (int)(pre.rcf & 0xffffffffLL), hw = (int)(pre.rcf >> 32);
    const int cprev = LR2_SHFL(c, lane >= 9 ? lane - 9 : lane); /* cell of the atom before this row's atom (tile atoms are consecutive in cell order) */
    const int fl = hw & 63;
    int nx = (hw >> 6) & 8191, ny = (hw >> 19) & 8191;
    if (nx == 0) { /* (rare) a grid of 8192 cells or more along x or y: two more links */
        const GridS *const g = LR2_COLD(a, grid) + LR2_COLD(a, s_idx)[pre.p0 + (la < na ? la : na - 1)].strct;
        nx = g->nx; ny = g->ny;
        SASA_OPAQUE(nx); SASA_OPAQUE(ny); /* (waited for HERE: left in flight, these two loads make the compiler wait for every load - the ones below included - where the registers are next written, in the middle of P5) */
    }
    const int dz = lr2_div3(r < 9 ? r : 0) - 1, dy = (r < 9 ? r : 0) - 3 * (dz + 1) - 1; /* (lanes 9 TA .. 63 belong to no row) */
    const bool out = (dy < 0 && (fl & CELL_Y0)) || (dy > 0 && (fl & CELL_Y1)) ||
                     (dz < 0 && (fl & CELL_Z0)) || (dz > 0 && (fl & CELL_Z1));
    const int row = out ? c : c + nx * (dy + ny * dz);
    const int x_lo = row - ((fl & CELL_X0) ? 0 : 1), x_hi = row + ((fl & CELL_X1) ? 0 : 1);
    const unsigned long long *const tbl = LR2_WARM(a, cell_tbl);
    if (tbl) { /* (uniform) compact table: the words now, the first atoms in lr2_pre_b2 */
        pre.s0 = x_lo; pre.s1 = x_hi + 1;
        pre.w0 = tbl[x_lo >> 5]; pre.w1 = tbl[(x_hi + 1) >> 5];
    } else {
        const int *const cell_start = LR2_COLD(a, cell_start);
        pre.s0 = cell_start[x_lo]; pre.s1 = cell_start[x_hi + 1];
    }
    /* the atoms of a tile are consecutive in cell order: atoms of one cell form a group */
    pre.rfl = (out || la >= na ? 1 : 0) | (la == 0 || cprev != c ? 2 : 0);
}
/* the last link with the compact cell table: occupied-cell ranks of the two cells -> first atoms */
template <int SHAPE>
SASA_D void lr2_pre_b2(const Lr2Args &a, Lr2Pre &pre)
{
    if (!LR2_WARM(a, cell_tbl)) return; /* (uniform) */
    const int *const first = LR2_WARM(a, cell_first);
    const int r0 = cell_rank(pre.w0, pre.s0), r1 = cell_rank(pre.w1, pre.s1);
    pre.s0 = first[r0]; pre.s1 = first[r1];
}
template <int SHAPE>
SASA_D void lr2_pre_c(const Lr2Args &a, Lr2Pre &pre, int na, int lane)
{
    const int p = pre.p0 + (lane < na ? lane : na - 1);
    pre.q = a.sq[p];
    const SortIdx si = LR2_WARM(a, s_idx)[p];
    pre.cell = lane < na ? (int)(si.cell & 0xffffffffLL) : -1 - lane;
    pre.so = si.orig;
}

/* The whole tile, executed by the 64 lanes of one wave.  RMAX = rounds of 64 pair records a lane
 * keeps in registers in P3 (pool <= 64 * RMAX). */
/* returns 0: the atoms' areas are stored; 1: the tile does not fit this launch's capacities (nothing stored); 2: two
   neighbors of an atom with equal sort keys (nothing stored; see lr2_tie12): once more with tie_by_place; 3 (builds with
   WALK = false): an atom of the tile lies beyond LR2_WALK_Z (nothing stored): a tile for the walking build */
/* pre: what P0 loads, possibly fetched ahead by the previous call; (p0n, nan): the tile this wave does next (its
   own again when there is none), fetched ahead by this call */
template <int RMAX, bool COVER, bool PAIRS, int SHAPE, bool HOOKS, bool WALK>
SASA_D int lr2_tile(const Lr2Args &a, const Lr2Mem &m, int p0, int na, bool sample, bool tie_by_place, int lane_of_wave, int &wg_max_nn,
                    Lr2Pre &pre, int p0n, int nan, int &far_tiles)
{
    /* Whatever depends on the lane and the launch alone (row and atom of the lane, its LDS addresses, its item) the
       compiler computes once per wave, outside the tile loop, and holds in registers through every phase of every
       tile.  The sparse kernel has the room (126 of 128 registers, nothing in scratch) and is 1.7 % faster for it;
       the builds with the cover filter do not (8 registers in scratch): there the lane number is made something the
       compiler cannot see through, and those values are computed where they are used (117-119 registers). */
    int lane = lane_of_wave;
    if (COVER) SASA_OPAQUE(lane);
    const int TA = LR2_A_TA(a), ns = LR2_A_NS(a), mw = LR2_A_MW(a);
    const int items = na * ns;
    LR2_MARK_BEGIN;
    LR2_COUNT(0, 1);

    /* ------------------------------------------------------------ P0 load */
    if (pre.p0 != p0) { /* (uniform) not fetched ahead (a tile is announced with its own first atom and count: the halves of a split tile never find the whole tile's values): the chain, one link after the other */
        LR2_COUNT(11, 1);
        pre.p0 = p0;
        lr2_pre_c<SHAPE>(a, pre, na, lane); /* (first: it hangs on nothing, and travels with the first link of the chain) */
        lr2_pre_a<SHAPE>(a, pre, p0, na, lane);
        lr2_pre_b<SHAPE>(a, pre, na, lane);
        lr2_pre_b2<SHAPE>(a, pre);
    }
    /* (the build the neighbor hooks are launched with stops after P1, which knows nothing of slice planes: it keeps far tiles -
       round-5 advisor: handed on, they never reached the hooks, whose launch sequence ends with the main launch) */
    if (!HOOKS && (!WALK || !a.work_items) && LR2_BALLOT(lane < na && fabs(pre.q.z) > LR2_WALK_Z) != 0) { /* (uniform) an atom beyond the closed-form range of the slice planes (lr2_slice_height_at) */
        ++far_tiles; /* (counted by every build of the main launch: the share of such tiles decides which build the next batch gets) */
        if (!WALK) { /* not a tile for this build */
            lr2_pre_none(pre);
            return 3;
        }
    }
    if (lane < TA) {
        Quad q = pre.q;
        if (lane >= na) { q.x = q.y = q.z = 0; q.w = 1; }
        const double del = lane < na ? lr2_div_ns(2 * q.w, (double)ns, LR2_WARM(a, inv_ns)) : 0.0; /* = 2 Ri / ns, ref: src/sasa_lr.c:304 */
        m.atom[lane] = q; m.adel[lane] = del; m.sorig[lane] = lane < na ? pre.so : 0;
        m.acnt[lane] = 0;
        m.gsz[lane] = 0; /* (P3's list cursors) */
    }
    if (lane < 8) m.flags[lane] = 0;
    /* cell groups: the atoms of a tile are consecutive in cell order; an atom whose cell differs from its predecessor's
       leads a group.  From one ballot every lane knows the leader and the size of any atom's group (no LDS round trip) */
    unsigned lm;
    {
        const int cprev = LR2_SHFL(pre.cell, lane > 0 ? lane - 1 : 0);
        lm = (unsigned)LR2_BALLOT(lane < na && (lane == 0 || pre.cell != cprev)); /* (bits 0 .. na-1; bit 0 is set) */
    }
    int lo = 0, my_cnt = 0, info = 0; /* candidates of row `lane` (rows of atoms that do not lead a cell group, rows outside the grid and empty rows count 0) */
    {
        const int la = lr2_div9(lane);
        if (la < na && (pre.rfl & 3) == 2) { /* inside the grid, and its atom leads a group */
            const unsigned above = lm >> (la + 1);
            const int gs = (above ? la + 1 + __builtin_ctz(above) : na) - la, hc = LR2_P1_HC(gs, LR2_P1_G_OF(SHAPE));
            lo = pre.s0;
            my_cnt = LR2_MUL24(pre.s1 - pre.s0, hc); /* P1's work items of the row: (candidate, up to P1_G atoms of the group) */
            info = la | (gs << 4) | (hc << 8) | ((hc == 1 ? 0x20000 : (hc == 2 ? 0x10000 : 0xaaab)) << 10); /* (bits 10..27: 2^17 / hc, rounded up) */
        }
    }
    lr2_pre_none(pre); /* (consumed: nothing of it lives through the tile) */
    /* the rows that have work, closed up: inclusive prefix of their item counts, first candidate, group */
    const unsigned long long rm = LR2_BALLOT(my_cnt > 0);
    const int incl = lr2_scan_add(my_cnt, lane);
    const int nrows = LR2_POPC64(rm);
    const int total_c = LR2_READLANE(incl, LR2_LANES - 1);
    if (lane == 0) m.cpre[0] = 0;
    if (my_cnt > 0) {
        const int ri = LR2_RANK(rm, lane);
        m.cpre[ri + 1] = incl; m.rowlo[ri] = lo; m.rinfo[ri] = info;
    }
    LR2_SYNC();
    /* P1 decodes a work item's place in its row with a 15-bit index and a 24-bit multiplication (i / hc as
       i * (2^17 / hc) >> 17: exact for i < 2^15 only).  The rows' counts sum to total_c, so total_c < 2^15 bounds every
       one of them; a tile beyond that (a giant radius that puts tens of thousands of small atoms into one cell) does not
       fit this kernel at any tile shape and is handed on like a tile whose lists do not fit: halves, then atom by atom to
       the slab launch, which walks its candidates one by one. */
    if (total_c >= LR2_P1_ITEMS_MAX) { LR2_COUNT(9, 1); return 1; } /* (uniform) */

    LR2_STOP(0);
    LR2_MARK(0);
    /* ------------------------------------------------------------ P1 neighbors */
    int nh = 0; /* hits so far (wave-uniform) */
    {
        const int per = (total_c + LR2_LANES - 1) / LR2_LANES; /* consecutive work items per lane */
        const int f = lane * per;
        const int fend = f + per < total_c ? f + per : total_c;
        int t = 0;
        /* the row of the lane's items rides in registers (items of a lane are consecutive: a row changes once in a
           while): first item c_lo, end c_hi, first candidate rl, group */
        int c_lo = 0, c_hi = 0, rl = 0, ri_ = 0;
        if (f < total_c) { /* row of the lane's first item: largest t with cpre[t] <= f */
            for (int step = 32; step >= 1; step >>= 1)
                if (t + step <= nrows - 1 && m.cpre[t + step] <= f) t += step;
            c_lo = m.cpre[t]; c_hi = m.cpre[t + 1]; rl = m.rowlo[t]; ri_ = m.rinfo[t];
        }
        constexpr int P1_G = LR2_P1_G_OF(SHAPE);
        LR2_COUNT(3, (per + LR2_NB_UNROLL - 1) / LR2_NB_UNROLL * LR2_NB_UNROLL * P1_G);
        for (int base = 0; base < per; base += LR2_NB_UNROLL) { /* (wave-uniform trip count) */
            int q[LR2_NB_UNROLL], la0[LR2_NB_UNROLL], two[LR2_NB_UNROLL];
            double x[LR2_NB_UNROLL], y[LR2_NB_UNROLL], z[LR2_NB_UNROLL], rq[LR2_NB_UNROLL];
            for (int j = 0; j < LR2_NB_UNROLL; ++j) {
                /* no branch but the one around the row table's reads: an item that does not exist (the lane's run is
                   over) decodes to something harmless and is switched off by selects */
                const int fj = f + base + j;
                const bool live = fj < fend; /* (fend <= f + per) */
                if (live && fj >= c_hi) { /* on to the next row (every row in the table has items) */
                    ++t;
                    c_lo = c_hi; c_hi = m.cpre[t + 1]; rl = m.rowlo[t]; ri_ = m.rinfo[t];
                }
                const int lead = ri_ & 15, gs = (ri_ >> 4) & 15, hc = (ri_ >> 8) & 3;
                const unsigned i = (unsigned)(fj - c_lo) & 0x7fffu;
                const unsigned c = LR2_UMUL24(i, (unsigned)ri_ >> 10) >> 17; /* i / hc (hc <= 3: gs <= 7; i < 2^15) */
                const int h = (int)i - LR2_MUL24(c, hc);
                const int left = gs - P1_G * h; /* atoms of this item */
                q[j] = live ? rl + (int)c : -1;
                la0[j] = live ? lead + P1_G * h : 0;
                two[j] = live ? (left < P1_G ? left : P1_G) : 0;
            }
            for (int j = 0; j < LR2_NB_UNROLL; ++j) {
                const unsigned u = (unsigned)(q[j] < 0 ? 0 : q[j]);
                { const Quad v = a.sq[u]; x[j] = v.x; y[j] = v.y; z[j] = v.z; rq[j] = v.w; }
            }
            for (int g = 0; g < P1_G; ++g) {
                bool hit[LR2_NB_UNROLL];
                double dx[LR2_NB_UNROLL], dy[LR2_NB_UNROLL], dz[LR2_NB_UNROLL];
                Quad ai[LR2_NB_UNROLL];
                for (int j = 0; j < LR2_NB_UNROLL; ++j) ai[j] = m.atom[la0[j] + (g < two[j] ? g : 0)];
                for (int j = 0; j < LR2_NB_UNROLL; ++j) {
                    /* the reference's contact test, operand for operand (src/nb.c:483-492) */
                    const double cut2 = (ai[j].w + rq[j]) * (ai[j].w + rq[j]);
                    dx[j] = x[j] - ai[j].x; dy[j] = y[j] - ai[j].y; dz[j] = z[j] - ai[j].z;
                    const double d2 = dx[j] * dx[j] + dy[j] * dy[j] + dz[j] * dz[j];
                    hit[j] = (g < two[j]) & (q[j] != p0 + la0[j] + g) & (d2 < cut2); /* (two > 0 only for q >= 0; every lane computes: no branch around ten instructions) */
                }
                for (int j = 0; j < LR2_NB_UNROLL; ++j) {
                    const unsigned long long hm = LR2_BALLOT(hit[j]);
                    if (hit[j]) {
                        const int la = la0[j] + g;
                        int slot = nh + LR2_RANK(hm, lane);
                        slot = slot < a.pool ? slot : a.pool - 1; /* (a tile with more hits than the pool is redone: P2) */
                        Quad hq; hq.x = dx[j]; hq.y = dy[j]; hq.z = dz[j]; hq.w = rq[j]; /* ref: src/nb.c:445-448 */
                        m.hits[slot] = hq;
                        m.tag[slot] = (unsigned char)la;
                        /* the atom's count only: nothing here waits for the counter's old value (the hit's place in its
                           atom's list is handed out in P3, two round trips per tile instead of one per group of tests) */
                        if (!(HOOKS && (a.hooks & 2))) { /* (uniform; HOOKS: the build the neighbor hooks are launched with - the others carry none of this) */
                            LR2_INC_LDS(&m.acnt[la]);
                        } else { /* test hook: the neighbor lists themselves */
                            const int sa = SASA_ATOMIC_ADD_LDS(&m.acnt[la], 1);
                            const int cap = LR2_COLD(a, nb_cap);
                            if (sa < cap) LR2_COLD(a, nb_out)[(size_t)m.sorig[la] * cap + sa] = LR2_COLD(a, s_idx)[q[j]].orig;
                        }
                    }
                    nh += LR2_POPC64(hm);
                }
            }
        }
    }
    LR2_SYNC();

    LR2_STOP(1);
    LR2_MARK(1);
    if (HOOKS && (a.hooks & 1)) { /* (uniform) test hook: the neighbor counts are the result */
        if (lane < na) LR2_COLD(a, nn_out)[m.sorig[lane]] = m.acnt[lane];
        LR2_SYNC();
        return 0;
    }
    /* ------------------------------------------------------------ P2 offsets */
    /* offsets of the atoms' lists in the pool (lists padded to an even length): a prefix over the first lanes of
       the wave; whether the tile fits is a ballot, so every lane knows it without a flag in LDS */
    bool ovf;
    int nn_max;
    {
        const int c = lane < TA ? m.acnt[lane] : 0, pc = (c + 1) & ~1;
        const int incl = lr2_scan_add(pc, lane), cmax = lr2_scan_max16(c, lane); /* (TA <= 7: the first row) */
        if (lane < TA) m.aoff[lane] = incl - pc;
        if (lane == TA - 1) m.aoff[TA] = incl;
        const int total = LR2_READLANE(incl, TA - 1);
        nn_max = LR2_READLANE(cmax, TA - 1);
        ovf = nn_max > 32 * mw || total > a.pool || nh > a.pool || nh > LR2_LANES * RMAX;
        if (lane == 0) {
            if (nn_max > wg_max_nn) wg_max_nn = nn_max;
            if (sample) { /* demand histogram for the next batch's pool size: 1 tile in 32 of the main launch */
                const int need = total / hist_bin_width(TA);
                SASA_ATOMIC_ADD_GLB(&LR2_COLD(a, status)[ST_HIST + (need < 63 ? need : 63)], 1);
            }
        }
    }
    LR2_SYNC();
    if (ovf) { LR2_COUNT(9, 1); return 1; } /* (uniform) */
    LR2_STOP(2);
    /* ------------------------------------------------------------ P1.5 contained caps */
    /* Behind the decision whether the tile fits: a tile whose lists AS FOUND are too long for this launch is handed on
       whatever could be dropped from them, so the same atoms go through the same launches with the phase on and off (the
       last launch, for atoms with lists beyond any tile's, has the first-generation kernel's arithmetic: equal to 1e-11,
       not to the bit), and the pool's demand statistics are those of the lists as found. */
    if (LR2_PRUNE && !HOOKS) {
        const int pk = LR2_COLD(a, prune);
        /* (uniform; a tile with more rounds of hits than the build holds in registers is not pruned) */
        constexpr int PR = RMAX < LR2_PRUNE_ROUNDS(PAIRS) ? RMAX : LR2_PRUNE_ROUNDS(PAIRS);
        if (pk > 0 && nh > 0 && nh <= LR2_LANES * PR) {
            lr2_prune_contained<PR>(m, nh, TA, pk, lane);
            const int c = lane < TA ? m.acnt[lane] : 0, pc = (c + 1) & ~1; /* the offsets again, of the lists that are left */
            const int incl = lr2_scan_add(pc, lane), cmax = lr2_scan_max16(c, lane);
            if (lane < TA) m.aoff[lane] = incl - pc;
            if (lane == TA - 1) m.aoff[TA] = incl;
            nn_max = LR2_READLANE(cmax, TA - 1);
            LR2_SYNC();
        }
    }
    LR2_STOP(15);
    const int mwt = (nn_max + 31) >> 5; /* mask words this tile's longest list needs (<= mw; one on most coil tiles) */
    const bool cover = COVER && a.cover > 0 && nh >= LR2_MUL24(a.cover, na); /* (uniform) dense enough for the cover filter; COVER: the launch
                                                                                  was built with it (launches over sparse batches are not) */

    LR2_MARK(2);
    /* ------------------------------------------------------------ P3 pair records */
    {
        double r_a[RMAX], r_b[RMAX], r_beta[RMAX], r_key[RMAX];
        int r_pos[RMAX]; /* first record of the pair's atom | list length << 16; -1: no pair */
        int r_cb[RMAX];  /* cover filter: bin of the pair's cap | atom << 3 | bucket of its key << 6 | place in the bucket << 10 */
        unsigned low = 0xfffu;
        SASA_OPAQUE(low);
        unsigned long long *const chist = (unsigned long long *)m.acell; /* [TA] over acell and lead (P0 / P1 only) */
        /* Long lists (dense tiles) are ranked in two levels: 16 buckets of the key per atom (counting, one LDS atomic
           per pair), then inside the pair's own bucket - 3 keys on average where the whole list has 48.  The bucket
           is a monotone function of the KEY, so the order is the order of the keys, exactly as below.  The bucket
           counters lie where the item masks will be (free until P4; the hits over them are in registers by then). */
        const bool bk = cover && ns * mw >= 16; /* (uniform) */
        int *const bh = (int *)m.it_mask;       /* [TA][16] pairs per bucket, then first place of the bucket in the atom's list */
        if (cover) {
            if (lane < TA) chist[lane] = 0;
            LR2_SYNC();
        }
        for (int r = 0; r < RMAX; ++r) {
            const int gp = lane + LR2_LANES * r;
            r_pos[r] = -1;
            r_cb[r] = 0;
            if (gp < nh) {
                const Quad hq = m.hits[gp];
                const int la = (int)m.tag[gp];
                if (la == LR2_DEAD_TAG) continue; /* a cap inside another (P1.5) */
                const int sa = SASA_ATOMIC_ADD_LDS(&m.gsz[la], 1); /* place in the atom's list, in order of discovery (gsz: zero since P0) */
                const int o = m.aoff[la];
                const double ri = m.atom[la].w;
                double Kc, d3sq, inv_d;
                r_pos[r] = o | (m.acnt[la] << 16);
                lr2_record(hq.x, hq.y, hq.z, hq.w, ri, r_a[r], r_b[r], Kc, d3sq, inv_d);
                r_beta[r] = atan2_inv(hq.y, hq.x, inv_d) + SASA_PI; /* ref: src/sasa_lr.c:337 */
                if (r_b[r] != r_b[r]) LR2_OR_LDS((unsigned *)&m.flags[6], 1u << la); /* a coincident atom of equal radius: the area is NaN, as the reference's (0 / 0, src/sasa_lr.c:335) */
                r_key[r] = lr_rank_key(r_beta[r], tie_by_place ? (unsigned)sa : lr2_tie12(hq.x, hq.y, hq.z), low);
                if (!bk) m.keys[o + sa] = r_key[r];
                r_cb[r] = la << 3;
                if (cover) { /* (uniform) */
                    const int b = lr2_cover_bin(Kc, d3sq, ri);
                    r_cb[r] |= b;
                    LR2_ADD64_LDS(&chist[la], 1ull << (8 * b)); /* (a byte per bin: a bin of 256 neighbors or more would carry into the next one - an atom with that many does not fit this launch's lists (32 mw <= 128), and the bins only choose which neighbors the filter looks at, never an area) */
                }
            }
        }
        if (lane < TA && (m.acnt[lane] & 1)) m.keys[m.aoff[lane] + m.acnt[lane]] = INFINITY; /* never ranks below */
        if (!cover && lane < TA) m.lead[lane] = 0; /* sum of the ranks of the atom's list */
        LR2_SYNC(); /* every hit is in registers: R1 may now take the records */
        if (cover) { /* per atom: the last bin of the largest caps; counters and list bits start at zero */
            const unsigned long long h = lane < TA ? chist[lane] : 0;
            LR2_SYNC();
            if (lane < TA) {
                m.gsz[lane] = lr2_cover_last_bin(h);
                m.acell[lane] = 0;
                m.lead[lane] = 0;
                for (int w = 0; w < mw; ++w) m.cmask[LR2_MUL24(lane, mw) + w] = 0;
            }
            LR2_SYNC();
        }
        if (bk) {
            for (int q = lane; q < 16 * TA; q += LR2_LANES) bh[q] = 0;
            LR2_SYNC();
            for (int r = 0; r < RMAX; ++r) {
                if (r_pos[r] < 0) continue;
                int b = (int)(r_key[r] * (16.0 / SASA_TWOPI));
                b = b < 0 ? 0 : (b > 15 ? 15 : b);
                const int idx = SASA_ATOMIC_ADD_LDS(&bh[((r_cb[r] >> 3) & 7) * 16 + b], 1);
                r_cb[r] |= (b << 6) | (idx << 10);
            }
            LR2_SYNC();
            for (int q0 = 0; q0 < 16 * TA; q0 += LR2_LANES) { /* counts -> first places: a prefix inside every atom's row of 16 */
                const int q = q0 + lane, v = q < 16 * TA ? bh[q] : 0;
                const int incl = lr2_scan_add16(v, lane);
                LR2_SYNC();
                if (q < 16 * TA) bh[q] = incl - v;
            }
            LR2_SYNC();
            for (int r = 0; r < RMAX; ++r) {
                if (r_pos[r] < 0) continue;
                const int first = bh[((r_cb[r] >> 3) & 7) * 16 + ((r_cb[r] >> 6) & 15)];
                m.keys[(r_pos[r] & 0xffff) + first + ((r_cb[r] >> 10) & 0xff)] = r_key[r];
            }
            LR2_SYNC();
        }
        int cnt_rank_[RMAX] = {0}; (void)cnt_rank_;
        for (int r = 0; r < RMAX; ++r) {
            if (r_pos[r] < 0) continue;
            const int o = r_pos[r] & 0xffff, nn = r_pos[r] >> 16;
            const double kme = r_key[r];
            int rank = 0, t = 0;
            if (bk) { /* (uniform) the keys of the pair's own bucket only */
                const int la_ = (r_cb[r] >> 3) & 7, b = (r_cb[r] >> 6) & 15;
                const int lo = bh[la_ * 16 + b], hi = b < 15 ? bh[la_ * 16 + b + 1] : m.acnt[la_];
                rank = lo;
                for (int q = lo; q < hi; ++q) rank += m.keys[o + q] < kme ? 1 : 0;
                t = nn;
#ifdef SASA_EMU
                cnt_rank_[r] = hi - lo;
#endif
            }
            for (; t + 4 <= nn; t += 4) { /* two keys per LDS read (o is even), two reads per trip */
                const Arc2 k0 = *(const Arc2 *)(m.keys + o + t), k1 = *(const Arc2 *)(m.keys + o + t + 2);
                rank += k0.s < kme ? 1 : 0;
                rank += k0.e < kme ? 1 : 0;
                rank += k1.s < kme ? 1 : 0;
                rank += k1.e < kme ? 1 : 0;
            }
            for (; t < nn; t += 2) { /* at most two trips (lists are padded to an even length) */
                const Arc2 kk = *(const Arc2 *)(m.keys + o + t);
                rank += kk.s < kme ? 1 : 0;
                rank += kk.e < kme ? 1 : 0;
            }
            Ab16 rc; rc.a = r_a[r]; rc.b = r_b[r];
            m.ab[o + rank] = rc;
            m.beta[o + rank] = r_beta[r];
            SASA_ATOMIC_ADD_LDS(&m.lead[(r_cb[r] >> 3) & 7], rank);
            if (cover) { /* (uniform) one of the atom's largest caps: its bit in the atom's list */
                const int la = (r_cb[r] >> 3) & 7;
                if ((r_cb[r] & 7) <= m.gsz[la] && SASA_ATOMIC_ADD_LDS(&m.acell[la], 1) < LR2_COVER_MAX)
                    LR2_OR_LDS(&m.cmask[LR2_MUL24(la, mw) + (rank >> 5)], 1u << (rank & 31));
            }
        }
#ifdef SASA_EMU
        if (bk) for (int r = 0; r < RMAX; ++r) LR2_COUNT_LANES(4, 5, cnt_rank_[r]);
#endif
        if (lane < TA && (m.acnt[lane] & 1)) { /* padding record: cos(alpha) huge, never an arc */
            int zero = 0; SASA_OPAQUE(zero); /* (made here: as a hoisted 64-bit constant it was kept in scratch) */
            Ab16 rc; rc.a = (double)zero; rc.b = 1e300;
            m.ab[m.aoff[lane] + m.acnt[lane]] = rc;
            m.beta[m.aoff[lane] + m.acnt[lane]] = rc.a;
        }
    }
    LR2_SYNC();
    { /* the ranks of every list are 0 .. n-1, each once?  Equal keys (see lr2_tie12: duplicate atom records, or ~1e-16
         per atom by accident) rank alike: the tile is done once more with ties by place of discovery, as before round
         3 (identical records may stand in any order; the accident costs that tile its shape-independence) */
        const int nn_ = lane < TA ? m.acnt[lane] : 0;
        const bool twice = lane < TA && m.lead[lane] != ((LR2_MUL24(nn_, nn_) - nn_) >> 1);
        if (LR2_BALLOT(twice) != 0) return 2; /* (uniform) */
    }

    LR2_STOP(3);
    LR2_MARK(3);
    lr2_pre_a<SHAPE>(a, pre, p0n, nan, lane); /* the next tile's sort records: on their way while this tile is screened */
    /* ------------------------------------------------------------ P4 screening */
    const float inv_ns = LR2_RCPF((float)ns); /* index arithmetic only (one off either way is put right below) */
#ifndef LR2_ARC_BALANCED
#define LR2_ARC_BALANCED 1
#endif
#ifndef LR2_SCREEN5
#define LR2_SCREEN5 1 /* (0: the 3 x 100 shape screens one item per lane and round, like the generic builds) */
#endif
    /* (uniform) dense builds whose tile has an item per lane at most (3 atoms x 20 slices): the arc pass deals its lanes out
       by the items' arc counts (P6, "balanced"), which every lane still holds in a register - no queue, no P5 */
#if defined(SASA_EMU) && defined(LR2_EMU_DUMP)
    const bool direct = false;
#else
    const bool direct = LR2_ARC_BALANCED && COVER && !PAIRS && items <= LR2_LANES;
#endif
    int dcnt = 0; /* direct: arcs of the lane's item behind the cover filter */
    if (!direct) {
        m.hist[lane] = 0;
        LR2_SYNC();
    }
    /* what is left to do for an item once its arcs are known: the cover filter (dense tiles), its word for the arc pass
       or its area, its place in the queue */
    int cnt_filter_[2] = {0, 0}, cnt_filter_k_ = 0, cnt_screen_ = 0; (void)cnt_filter_; (void)cnt_filter_k_; (void)cnt_screen_;
    auto finish_item = [&](int it, int la, double t, double h2, double Ri, int o, int cnt, bool buried, bool circle) -> int {
        double area = 0;
        if (buried || !circle) cnt = 0; /* circle i inside a neighbor's: buried (ref: :327-330); no circle: ref :310-312 */
        else if (cnt == 0) area = m.adel[la] * Ri * SASA_TWOPI; /* ref: :360 with exposed_arc_length(n = 0) */
        if (cover && circle) { /* (uniform) the arcs of the largest caps, in beta order: do they cover the circle? */
            double cts = 0, cte = -INFINITY;
            for (int wi = 0; wi < mwt; ++wi) {
                unsigned wc = cnt > 0 ? (m.it_mask[LR2_MUL24(it, mw) + wi] & m.cmask[LR2_MUL24(la, mw) + wi]) : 0u;
                const int R = o + 32 * wi;
                while (wc != 0) { /* (the wave runs as many trips as its busiest lane) */
                    const int q = R + __builtin_ctz(wc);
                    wc &= wc - 1;
#ifdef SASA_EMU
                    ++cnt_filter_[cnt_filter_k_ & 1];
#endif
                    const Ab16 ab = m.ab[q];
                    const double bt = m.beta[q];
                    const double c = fma(t, ab.a, ab.b) * h2;
                    const double al = lr2_acos_lower(c);
                    const double inf = bt - al, sup = bt + al;
                    const bool fresh = inf > cte;
                    const double nts = fresh ? inf : SASA_MIN(cts, inf), nte = fresh ? sup : SASA_MAX(cte, sup);
                    if (c <= 0.9) { cts = nts; cte = nte; }
                }
            }
            if (cte - cts >= SASA_TWOPI) cnt = 0; /* covered: area 0 */
        }
#ifdef SASA_EMU
        ++cnt_filter_k_;
#endif
        m.it_tc[it] = cnt == 0 ? area : 0.5 * h2; /* (an item with arcs: 1/(4 Ri') for the arc pass - see lr2_arc_alpha -, which puts the area in its place) */
        if (direct) return cnt; /* (uniform) */
        unsigned short qt = 0xffff;
        if (cnt > 0) { /* queue: heaviest first (bin 0 = 63 arcs or more); inside a bin in order of arrival */
            const int bin = 63 - (cnt < 63 ? cnt : 63);
            const int ord = SASA_ATOMIC_ADD_LDS(&m.hist[bin], 1);
            qt = (unsigned short)((bin << 10) | ord); /* ord < items <= 512 */
        }
        m.qtmp[it] = qt;
        return cnt;
    };
    const int hp = (ns + 1) >> 1; /* pairs of slices per atom */
    if (PAIRS) { /* (the launch's tile shape: lr2_pairs_shape) */
        /* More items than lanes, but no more than two per lane (6 atoms x 20 slices): a lane screens two neighboring
           slices of one atom in ONE pass over the atom's records - every record is read once for both, the loop's
           bookkeeping and the item's own preparations are shared - where two rounds of one item per lane walk every
           list twice (round 4: the phase's wave instructions 450 -> ~360 per tile of coils). */
        int ln = lane;
        SASA_OPAQUE(ln); /* (what follows depends on the lane and the launch only: left to itself the compiler computes it once per wave and keeps - then spills - a dozen registers through every phase of every tile) */
        if (ln < LR2_MUL24(na, hp)) {
            int la = (int)(((float)ln + 0.5f) * LR2_RCPF((float)hp)), j = ln - LR2_MUL24(la, hp);
            if (j < 0) { --la; j += hp; } else if (j >= hp) { ++la; j -= hp; }
            const int sa_ = 2 * j, it0 = LR2_MUL24(la, ns) + sa_;
            const bool second = sa_ + 1 < ns; /* (ns odd: the atom's last pair is one slice) */
            const double Ri = m.atom[la].w, del = m.adel[la];
            const double zi_ = WALK ? m.atom[la].z : 0.0;
            const double t0 = lr2_slice_height_at<WALK>(sa_, del, Ri, zi_), t1 = lr2_slice_height_at<WALK>(second ? sa_ + 1 : sa_, del, Ri, zi_);
            const double A0 = Ri * Ri - t0 * t0, A1 = Ri * Ri - t1 * t1; /* Ri'^2, ref: src/sasa_lr.c:309 */
            const bool circ0 = A0 > 0, circ1 = second && A1 > 0; /* ref: :310-312 */
            double h0, h1;
            LR2_H2(circ0 ? A0 : 1.0, h0); /* 1/(2 Ri') */
            LR2_H2(circ1 ? A1 : 1.0, h1);
            const double T0 = lr2_arc_limit(circ0 ? A0 : 1.0, h0), T1 = lr2_arc_limit(circ1 ? A1 : 1.0, h1);
            const int o = m.aoff[la], nn = m.aoff[la + 1] - o;
            double vmin0 = 0.0, vmin1 = 0.0; /* least b' + a' t of the slice: at or below -T the circle lies inside a neighbor's */
            int cnt0 = 0, cnt1 = 0;
            for (int wi = 0; wi < mwt; ++wi) {
                unsigned w0 = 0, w1 = 0;
                const int k1 = nn - 32 * wi < 32 ? nn - 32 * wi : 32;
                const Ab16 *R = m.ab + (o + 32 * wi);
                int k = k1 - 2; /* from the end: neighbor k lands on bit k */
                if (k1 > 0 && (k1 & 2)) {
                    const Ab16 ra = R[k], rb = R[k + 1];
                    const double c0 = fma(t0, ra.a, ra.b), c1 = fma(t0, rb.a, rb.b);
                    const double d0 = fma(t1, ra.a, ra.b), d1 = fma(t1, rb.a, rb.b);
                    vmin0 = SASA_MIN(vmin0, SASA_MIN(c0, c1));
                    vmin1 = SASA_MIN(vmin1, SASA_MIN(d0, d1));
                    w0 = LR2_SHIFT_IN_LT(w0, c1, T0); w0 = LR2_SHIFT_IN_LT(w0, c0, T0);
                    w1 = LR2_SHIFT_IN_LT(w1, d1, T1); w1 = LR2_SHIFT_IN_LT(w1, d0, T1);
                    k -= 2;
                }
                for (; k >= 0; k -= 4) { /* four records per trip: their LDS reads are in flight together */
                    const Ab16 r2 = R[k], r3 = R[k + 1], r0 = R[k - 2], r1 = R[k - 1];
                    const double c2 = fma(t0, r2.a, r2.b), c3 = fma(t0, r3.a, r3.b);
                    const double c0 = fma(t0, r0.a, r0.b), c1 = fma(t0, r1.a, r1.b);
                    const double d2 = fma(t1, r2.a, r2.b), d3 = fma(t1, r3.a, r3.b);
                    const double d0 = fma(t1, r0.a, r0.b), d1 = fma(t1, r1.a, r1.b);
                    vmin0 = SASA_MIN(SASA_MIN(vmin0, c0), SASA_MIN(c1, SASA_MIN(c2, c3)));
                    vmin1 = SASA_MIN(SASA_MIN(vmin1, d0), SASA_MIN(d1, SASA_MIN(d2, d3)));
                    w0 = LR2_SHIFT_IN_LT(w0, c3, T0); w0 = LR2_SHIFT_IN_LT(w0, c2, T0); w0 = LR2_SHIFT_IN_LT(w0, c1, T0); w0 = LR2_SHIFT_IN_LT(w0, c0, T0);
                    w1 = LR2_SHIFT_IN_LT(w1, d3, T1); w1 = LR2_SHIFT_IN_LT(w1, d2, T1); w1 = LR2_SHIFT_IN_LT(w1, d1, T1); w1 = LR2_SHIFT_IN_LT(w1, d0, T1);
                }
                m.it_mask[LR2_MUL24(it0, mw) + wi] = w0;
                cnt0 += LR2_POPC32(w0);
                if (second) {
                    m.it_mask[LR2_MUL24(it0 + 1, mw) + wi] = w1;
                    cnt1 += LR2_POPC32(w1);
                }
            }
            finish_item(it0, la, t0, circ0 ? h0 : 0.0, Ri, o, cnt0, vmin0 <= -T0, circ0);
            if (second) finish_item(it0 + 1, la, t1, circ1 ? h1 : 0.0, Ri, o, cnt1, vmin1 <= -T1, circ1);
        }
    } else if (LR2_SCREEN5 && SHAPE == 2) {
        /* 3 atoms x 100 slices (BASELINE configs[2] as written): 300 items were 4.7 rounds of one item per lane, every round
           walking its atoms' lists again.  Here a lane screens FIVE neighboring slices of one atom in one pass over the atom's
           records (60 lanes, one round): a record is read once for the five, the loop's bookkeeping is shared - the PAIRS
           arrangement above, wider.  The same values by the same operations as the generic loop below (the shape builds give
           the generic builds' bits: tests/test_emulation.py, tests/test_gpu_parity.py).  Round 6, MI355X: the phase 739 ->
           see DESIGN.md wave instructions per tile. */
        constexpr int Q = 5, G = 100 / Q;
        int ln = lane;
        SASA_OPAQUE(ln); /* (as above: not computed once per wave and kept) */
        if (ln < LR2_MUL24(na, G)) {
            int la = (int)(((float)ln + 0.5f) * (1.0f / (float)G)), j = ln - LR2_MUL24(la, G);
            if (j < 0) { --la; j += G; } else if (j >= G) { ++la; j -= G; }
            const int s0 = Q * j, it0 = LR2_MUL24(la, ns) + s0;
            const double Ri = m.atom[la].w, del = m.adel[la];
            const double zi_ = WALK ? m.atom[la].z : 0.0;
            double t[Q], T[Q], vmin[Q]; /* (1/(2 Ri') is made again for finish_item - the same operations, the same bits: five more doubles do not fit the registers) */
            int cnt[Q];
            for (int q = 0; q < Q; ++q) {
                t[q] = lr2_slice_height_at<WALK>(s0 + q, del, Ri, zi_);
                const double A = Ri * Ri - t[q] * t[q]; /* Ri'^2, ref: src/sasa_lr.c:309 */
                double hq;
                LR2_H2(A > 0 ? A : 1.0, hq);             /* 1/(2 Ri'); no circle (ref: :310-312): finish_item is told */
                T[q] = lr2_arc_limit(A > 0 ? A : 1.0, hq);
                vmin[q] = 0.0;
                cnt[q] = 0;
            }
            const int o = m.aoff[la], nn = m.aoff[la + 1] - o;
            for (int wi = 0; wi < mwt; ++wi) {
                unsigned w[Q];
                for (int q = 0; q < Q; ++q) w[q] = 0;
                const int k1 = nn - 32 * wi < 32 ? nn - 32 * wi : 32;
                const Ab16 *R = m.ab + (o + 32 * wi);
                /* from the end: neighbor k lands on bit k; the next record is on its way while this one meets the five slices */
                Ab16 rn = R[k1 > 0 ? k1 - 1 : 0];
                for (int k = k1 - 1; k >= 0; --k) {
                    const Ab16 ra = rn;
                    rn = R[k > 0 ? k - 1 : 0];
                    for (int q = 0; q < Q; ++q) {
                        const double c = fma(t[q], ra.a, ra.b);
                        vmin[q] = SASA_MIN(vmin[q], c);
                        w[q] = LR2_SHIFT_IN_LT(w[q], c, T[q]);
                    }
                }
                for (int q = 0; q < Q; ++q) {
                    m.it_mask[LR2_MUL24(it0 + q, mw) + wi] = w[q];
                    cnt[q] += LR2_POPC32(w[q]);
                }
            }
            for (int q = 0; q < Q; ++q) {
                const double A = Ri * Ri - t[q] * t[q];
                double hq;
                LR2_H2(A > 0 ? A : 1.0, hq);
                finish_item(it0 + q, la, t[q], A > 0 ? hq : 0.0, Ri, o, cnt[q], vmin[q] <= -T[q], A > 0);
            }
        }
    } else
    for (int it = lane; it < items; it += LR2_LANES) {
        int la = (int)(((float)it + 0.5f) * inv_ns), s = it - LR2_MUL24(la, ns); /* it / ns without the integer-division sequence */
        if (s < 0) { --la; s += ns; } else if (s >= ns) { ++la; s -= ns; }
        const double Ri = m.atom[la].w, t = lr2_slice_height_at<WALK>(s, m.adel[la], Ri, WALK ? m.atom[la].z : 0.0);
        const double A = Ri * Ri - t * t; /* Ri'^2, ref: src/sasa_lr.c:309 */
        double h2 = 0;
        int cnt = 0, o = 0;
        bool buried = false;
        if (A > 0) { /* ref: :310-312 */
            LR2_H2(A, h2); /* h2 = 1/(2 Ri') */
            const double T = lr2_arc_limit(A, h2);
            o = m.aoff[la];
            const int nn = m.aoff[la + 1] - o;
            double vmin = 0.0;
            for (int wi = 0; wi < mwt; ++wi) {
                unsigned w = 0;
                const int k1 = nn - 32 * wi < 32 ? nn - 32 * wi : 32;
                const Ab16 *R = m.ab + (o + 32 * wi);
                int k = k1 - 2; /* from the end: neighbor k lands on bit k */
                if (k1 > 0 && (k1 & 2)) {
                    const double c0 = fma(t, R[k].a, R[k].b), c1 = fma(t, R[k + 1].a, R[k + 1].b);
                    vmin = SASA_MIN(vmin, c0);
                    vmin = SASA_MIN(vmin, c1);
                    w = LR2_SHIFT_IN_LT(w, c1, T);
                    w = LR2_SHIFT_IN_LT(w, c0, T);
                    k -= 2;
                }
                for (; k >= 0; k -= 4) { /* four records per trip: their LDS reads are in flight together */
                    const double c2 = fma(t, R[k].a, R[k].b), c3 = fma(t, R[k + 1].a, R[k + 1].b);
                    const double c0 = fma(t, R[k - 2].a, R[k - 2].b), c1 = fma(t, R[k - 1].a, R[k - 1].b);
                    vmin = SASA_MIN(SASA_MIN(vmin, c0), SASA_MIN(c1, SASA_MIN(c2, c3)));
                    w = LR2_SHIFT_IN_LT(w, c3, T);
                    w = LR2_SHIFT_IN_LT(w, c2, T);
                    w = LR2_SHIFT_IN_LT(w, c1, T);
                    w = LR2_SHIFT_IN_LT(w, c0, T);
                }
                m.it_mask[LR2_MUL24(it, mw) + wi] = w;
                cnt += LR2_POPC32(w);
            }
            buried = vmin <= -T;
        }
        dcnt = finish_item(it, la, t, h2, Ri, o, cnt, buried, A > 0); /* (direct: the loop has one trip) */
    }
    LR2_SYNC();
#ifdef SASA_EMU
    if (cover) { LR2_COUNT_LANES(12, 13, cnt_filter_[0]); LR2_COUNT_LANES(12, 13, cnt_filter_[1]); LR2_COUNT(10, 1); }
#endif

    LR2_STOP(4);
    LR2_MARK(4);
    lr2_pre_b<SHAPE>(a, pre, nan, lane); /* the next tile's candidate rows (compact cell table: their table words): on their way while this tile's arcs are done */
    /* ------------------------------------------------------------ P5 queue */
    int nq;
    if (direct) nq = LR2_POPC64(LR2_BALLOT(dcnt > 0));
    else {
        const int hv = m.hist[lane];
        const int incl2 = lr2_scan_add(hv, lane);
        nq = LR2_READLANE(incl2, LR2_LANES - 1);
        m.hist[lane] = incl2 - hv; /* first queue position of the bin */
        LR2_SYNC();
        if (items <= 2 * LR2_LANES) { /* (uniform; the default shape: 120) both rounds' reads in flight together: two LDS round trips, not four */
            const int it1 = lane + LR2_LANES;
            const unsigned q0 = m.qtmp[lane < items ? lane : 0], q1 = m.qtmp[it1 < items ? it1 : 0];
            const bool on0 = lane < items && q0 != 0xffffu, on1 = it1 < items && q1 != 0xffffu;
            const int h0 = m.hist[on0 ? q0 >> 10 : 0], h1 = m.hist[on1 ? q1 >> 10 : 0];
            int la0 = (int)(((float)lane + 0.5f) * inv_ns), la1 = (int)(((float)it1 + 0.5f) * inv_ns);
            { const int s = lane - LR2_MUL24(la0, ns); if (s < 0) --la0; else if (s >= ns) ++la0; }
            { const int s = it1 - LR2_MUL24(la1, ns); if (s < 0) --la1; else if (s >= ns) ++la1; }
            if (on0) m.queue[h0 + (q0 & 1023u)] = (unsigned short)(lane | (la0 << 10));
            if (on1) m.queue[h1 + (q1 & 1023u)] = (unsigned short)(it1 | (la1 << 10));
        } else
        for (int it = lane; it < items; it += LR2_LANES) {
            const unsigned qt = m.qtmp[it];
            int la = (int)(((float)it + 0.5f) * inv_ns);
            { const int s = it - LR2_MUL24(la, ns); if (s < 0) --la; else if (s >= ns) ++la; }
            if (qt != 0xffffu) m.queue[m.hist[qt >> 10] + (qt & 1023u)] = (unsigned short)(it | (la << 10));
        }
        LR2_SYNC();
    }

    LR2_STOP(5);
    LR2_MARK(5);
    lr2_pre_b2<SHAPE>(a, pre);           /* the next tile's candidate rows (compact cell table: the last link) and its atoms: on their way while this tile's arcs are united */
    lr2_pre_c<SHAPE>(a, pre, nan, lane); /* (until round 4's last session these two went out behind the arc pass and P0 waited for them: -1.5 %) */
    /* ------------------------------------------------------------ P6 arc pass */
    int maxd = 0;
    LR2_COUNT(6, nq); LR2_COUNT(8, 1);
#if defined(SASA_EMU) && defined(LR2_EMU_DUMP) /* dev only (tools/dev/arc_sched_sim.py; built with -include cstdio -include cstdlib): the arc pass's work of every tile, for scheduling studies on the CPU */
    if (lane == 0) {
        static FILE *df = fopen(getenv("LR2_EMU_DUMP") ? getenv("LR2_EMU_DUMP") : "/tmp/lr2_dump.txt", "a");
        fprintf(df, "T %d %d %d %d\n", nq, na, mwt, (int)(COVER && cover));
        for (int qi = 0; qi < nq; ++qi) {
            const int e = (int)m.queue[qi], my = e & 1023, la_ = e >> 10;
            fprintf(df, "%d %d", my, m.aoff[la_ + 1] - m.aoff[la_]);
            for (int wi = 0; wi < mw; ++wi) fprintf(df, " %x", wi < mwt ? m.it_mask[my * mw + wi] : 0u);
            fprintf(df, "\n");
        }
        fflush(df);
    }
#endif
    if (LR2_ARC_BALANCED && COVER && nq <= LR2_LANES) {
        /* Dense tiles behind the cover filter: ~22 of 60 items are left on the reference's PDB entries, with 1 to 60 arcs
           each (18 on average) - fewer items than lanes, and very unequal ones.  The 64 lanes are dealt out by ARCS: with g
           the least number for which sum ceil(arcs_i / g) <= 64, item i gets n_i = ceil(arcs_i / g) neighboring lanes, and
           lane k of them unites the arcs of ranks [arcs_i k / n_i, arcs_i (k + 1) / n_i) in the item's list - contiguous in
           beta -; then the partial unions are merged pairwise, the higher into the lower: components of the higher part
           arrive in ascending order and each contains a mid-point beyond every mid-point of the lower part, which is all
           the stack union asks of its input (see lr2_union_step).  The components, and with them every bit of the area,
           are the same however the arcs are dealt out (unions are minima and maxima of the same end points).
           Round 6 (tools/dev/arc_sched_sim.py on the emulation's dump of 1a0q, 1ubq, 3bzd): 8.5 - 9.0 trips of the arc loop
           per tile where two or four lanes per item by list POSITION (rounds 4 - 5) ran 15.5 - 16.1 and a perfect deal
           6.3 - 6.6. */
        Arc2 *stk = m.stack + lane;
        int *const tbl = m.hist; /* [64] lane -> item of the queue | place among its lanes << 8 | its lanes << 16 (P5 is done with the histogram; the arc stack takes this space over behind the second fence below) */
        int cnt = direct ? dcnt : 0; /* direct: lane = item; else lane = place in the queue */
        if (!direct && lane < nq) {
            const int my_ = (int)m.queue[lane] & 1023;
            for (int wi = 0; wi < mwt; ++wi) cnt += LR2_POPC32(m.it_mask[LR2_MUL24(my_, mw) + wi]);
        }
        const int total = LR2_READLANE(lr2_scan_add(cnt, lane), LR2_LANES - 1);
        int g = total > LR2_LANES ? (total + LR2_LANES - 1) >> 6 : 1, n = 0, incl = 0;
        for (;;) { /* (uniform; one to three trips) */
            n = (int)(((float)(cnt + g - 1) + 0.5f) * LR2_RCPF((float)g)); /* ceil(cnt / g): small integers, the half keeps the 1-ulp reciprocal on the right side */
            incl = lr2_scan_add(n, lane);
            if (LR2_READLANE(incl, LR2_LANES - 1) <= LR2_LANES) break;
            ++g;
        }
        int nmax;
        { const int v = lr2_scan_max16(n, lane); const int a0 = LR2_READLANE(v, 15), a1 = LR2_READLANE(v, 31), a2 = LR2_READLANE(v, 47), a3 = LR2_READLANE(v, 63);
          nmax = a0 > a1 ? a0 : a1; nmax = nmax > a2 ? nmax : a2; nmax = nmax > a3 ? nmax : a3; }
        tbl[lane] = -1;
        LR2_SYNC();
        for (int j = 0; j < nmax; ++j) /* (uniform) */
            if (j < n) tbl[incl - n + j] = lane | (j << 8) | (n << 16);
        LR2_SYNC();
        const int te = tbl[lane];
        LR2_SYNC();
        const bool valid = te >= 0;
        const int k = valid ? (te >> 8) & 255 : 0, nk = valid ? te >> 16 : 1;
        int my = 0, la = 0, pos0 = 0;
        const int c_src = LR2_SHFL(cnt, valid ? te & 255 : lane); /* the item's arcs */
        if (valid) { /* the first arc of this lane: the item's set bit of rank r0 = cnt k / nk */
            if (direct) {
                my = te & 255;
                la = (int)(((float)my + 0.5f) * inv_ns);
                { const int s_ = my - LR2_MUL24(la, ns); if (s_ < 0) --la; else if (s_ >= ns) ++la; }
            } else {
                const int e = (int)m.queue[te & 255];
                my = e & 1023; la = e >> 10;
            }
            const int c = c_src;
            int r = (int)(((float)LR2_MUL24(c, k) + 0.5f) * LR2_RCPF((float)nk)); /* < c: k < nk */
            unsigned w = m.it_mask[LR2_MUL24(my, mw)];
            for (int wi = 1; wi < mwt; ++wi) { /* the word it lies in */
                const int p = LR2_POPC32(w);
                if (r >= p) { r -= p; pos0 += 32; w = m.it_mask[LR2_MUL24(my, mw) + wi]; }
            }
            for (int sz = 16; sz >= 1; sz >>= 1) { /* ... and its place there: the blocks of set bits before it are stepped over */
                const int p = LR2_POPC32((w >> (pos0 & 31)) & ((1u << sz) - 1u));
                if (r >= p) { r -= p; pos0 += sz; }
            }
        }
        /* ... and the first of the lane above, where this one's part ends */
        int pos1 = LR2_SHFL(pos0, lane + 1 < LR2_LANES ? lane + 1 : lane);
        if (k + 1 >= nk) pos1 = 32 * mwt;
        Lr2Union u;
        lr2_union_reset(u);
        int cnt_shared_ = 0; (void)cnt_shared_;
        /* the lane's arcs as a window of 64 list positions from its first one (a part that spans more - an item with a
           long list, few arcs and one lane - takes a second window; LR2_ARC_WINDOW: tests build with a narrow window to
           walk that path) */
#ifndef LR2_ARC_WINDOW
#define LR2_ARC_WINDOW 64
#endif
        const int nwin = LR2_ARC_WINDOW == 64 ? (LR2_BALLOT(valid && pos1 - pos0 > 64) != 0 ? 2 : 1) : (32 * mwt + LR2_ARC_WINDOW - 1) / LR2_ARC_WINDOW; /* (32 mwt <= 128 positions) */
        for (int win = 0; win < nwin; ++win) { /* (uniform) */
            unsigned long long W = 0;
            int base = 0;
            double t = 0, hh = 0;
            if (valid) {
                const int b0 = pos0 + LR2_ARC_WINDOW * win, len = pos1 - b0 < LR2_ARC_WINDOW ? pos1 - b0 : LR2_ARC_WINDOW;
                const int i0 = b0 >> 5, sh = b0 & 31;
                const unsigned *const mk = m.it_mask + LR2_MUL24(my, mw);
                const unsigned x0 = i0 < mwt ? mk[i0 < mwt ? i0 : 0] : 0u, x1 = i0 + 1 < mwt ? mk[i0 + 1 < mwt ? i0 + 1 : 0] : 0u, x2 = i0 + 2 < mwt ? mk[i0 + 2 < mwt ? i0 + 2 : 0] : 0u;
                const unsigned lo32 = (unsigned)((((unsigned long long)x1 << 32) | x0) >> sh), hi32 = (unsigned)((((unsigned long long)x2 << 32) | x1) >> sh);
                W = ((unsigned long long)hi32 << 32) | lo32;
                if (len < 64) W &= len > 0 ? (1ull << len) - 1ull : 0ull;
                base = m.aoff[la] + b0;
                t = lr2_slice_height_at<WALK>(my - LR2_MUL24(la, ns), m.adel[la], m.atom[la].w, WALK ? m.atom[la].z : 0.0); /* as P4: bit for bit */
                hh = m.it_tc[my];
            }
            const Ab16 *const Rab = m.ab + base;
            const double *const Rbt = m.beta + base;
            while (W != 0) { /* (the wave runs as many trips as its busiest lane) */
                const int q = __builtin_ctzll(W);
                W &= W - 1;
#ifdef SASA_EMU
                ++cnt_shared_;
#endif
                const Ab16 ab = Rab[q];
                const double bt = Rbt[q];
                const double alpha = lr2_arc_alpha(t, ab, hh);
                lr2_union_step(bt - alpha, bt + alpha, u, stk, LR2_A_DS(a), maxd); /* ref: :338-339 */
            }
        }
        LR2_COUNT_LANES(14, 15, cnt_shared_);
        for (int st = 1; st < nmax; st <<= 1) { /* merge: lane k + st of an item into its lane k, for k a multiple of 2 st */
            LR2_SYNC();
            const int src = lane + st < LR2_LANES ? lane + st : lane;
            const int d_s = LR2_SHFL(u.depth, src);
            const double bs_s = lr2_shfl_f64(u.bs, src), be_s = lr2_shfl_f64(u.be, src);
            const double ts_s = lr2_shfl_f64(u.ts, src), te_s = lr2_shfl_f64(u.te, src);
            if (valid && (k & (2 * st - 1)) == 0 && k + st < nk) {
                const Arc2 *col = m.stack + src;
                for (int c = 0; c < d_s - 2; ++c) { /* (rare: the components below the two in registers) */
                    const Arc2 kk = col[(c < LR2_A_DS(a) ? c : 0) * LR2_LANES];
                    lr2_union_step(kk.s, kk.e, u, stk, LR2_A_DS(a), maxd);
                }
                if (d_s >= 2) lr2_union_step(bs_s, be_s, u, stk, LR2_A_DS(a), maxd);
                if (d_s >= 1) lr2_union_step(ts_s, te_s, u, stk, LR2_A_DS(a), maxd);
            }
        }
        if (valid && k == 0) m.it_tc[my] = m.adel[la] * m.atom[la].w * lr2_sweep(u, stk, LR2_A_DS(a)); /* ref: :360 */
    } else if (COVER && nq * 2 <= LR2_LANES) {
        const int shb = nq * 4 <= LR2_LANES ? 2 : 1, share = 1 << shb; /* (uniform) 4 lanes per item, or 2 */
        /* Few items are left (dense tiles behind the cover filter: ~9 of 60, two of them with ~27 arcs): one item
           per lane would leave 55 lanes idle for as long as the longest item takes.  Four lanes (two, from 17 items)
           share an item instead: each unites the arcs of its part of the atom's neighbor list (contiguous in beta), then the
           partial unions are merged pairwise, the higher into the lower — components of the higher part arrive in
           ascending order and each contains a mid-point beyond every mid-point of the lower part, which is all the
           stack union asks of its input (see lr2_union_step). */
        Arc2 *stk = m.stack + lane;
        const int qi = lane >> shb, j = lane & (share - 1);
        const bool valid = qi < nq;
        int my = 0, la = 0;
        Lr2Union u;
        lr2_union_reset(u);
        int cnt_shared_ = 0; (void)cnt_shared_;
        if (valid) {
            const int e = (int)m.queue[qi];
            my = e & 1023; la = e >> 10;
            const int o = m.aoff[la], nn = m.aoff[la + 1] - o;
            const double t = lr2_slice_height_at<WALK>(my - LR2_MUL24(la, ns), m.adel[la], m.atom[la].w, WALK ? m.atom[la].z : 0.0); /* as P4: bit for bit */
            const double hh = m.it_tc[my];
            const int lo = LR2_MUL24(nn, j) >> shb, hi = LR2_MUL24(nn, j + 1) >> shb; /* list positions of this lane */
            for (int wi = 0; wi < mwt; ++wi) {
                int a0 = lo - 32 * wi, a1 = hi - 32 * wi;
                a0 = a0 < 0 ? 0 : (a0 > 32 ? 32 : a0);
                a1 = a1 < 0 ? 0 : (a1 > 32 ? 32 : a1);
                const unsigned below1 = a1 >= 32 ? 0xffffffffu : ((1u << a1) - 1u), below0 = a0 >= 32 ? 0xffffffffu : ((1u << a0) - 1u);
                unsigned w = m.it_mask[LR2_MUL24(my, mw) + wi] & below1 & ~below0;
                const int R = o + 32 * wi;
                while (w != 0) { /* (the wave runs as many trips as its busiest lane) */
                    const int q = R + __builtin_ctz(w);
                    w &= w - 1;
#ifdef SASA_EMU
                    ++cnt_shared_;
#endif
                    const Ab16 ab = m.ab[q];
                    const double bt = m.beta[q];
                    const double alpha = lr2_arc_alpha(t, ab, hh);
                    lr2_union_step(bt - alpha, bt + alpha, u, stk, LR2_A_DS(a), maxd); /* ref: :338-339 */
                }
            }
        }
        LR2_COUNT_LANES(14, 15, cnt_shared_);
        for (int st = 1; st < share; st <<= 1) { /* merge: lane j + st into lane j, for j a multiple of 2 st */
            LR2_SYNC();
            const int src = lane + st < LR2_LANES ? lane + st : lane;
            const int d_s = LR2_SHFL(u.depth, src);
            const double bs_s = lr2_shfl_f64(u.bs, src), be_s = lr2_shfl_f64(u.be, src);
            const double ts_s = lr2_shfl_f64(u.ts, src), te_s = lr2_shfl_f64(u.te, src);
            if (valid && (j & (2 * st - 1)) == 0) {
                const Arc2 *col = m.stack + src;
                for (int c = 0; c < d_s - 2; ++c) { /* (rare: the components below the two in registers) */
                    const Arc2 k = col[(c < LR2_A_DS(a) ? c : 0) * LR2_LANES];
                    lr2_union_step(k.s, k.e, u, stk, LR2_A_DS(a), maxd);
                }
                if (d_s >= 2) lr2_union_step(bs_s, be_s, u, stk, LR2_A_DS(a), maxd);
                if (d_s >= 1) lr2_union_step(ts_s, te_s, u, stk, LR2_A_DS(a), maxd);
            }
        }
        if (valid && j == 0) m.it_tc[my] = m.adel[la] * m.atom[la].w * lr2_sweep(u, stk, LR2_A_DS(a)); /* ref: :360 */
    } else {
        Arc2 *stk = m.stack + lane;
        int next = LR2_LANES;
        int my = LR2_NONE, la = 0, wleft = 0;
        unsigned w = 0;
        const Ab16 *Rab = m.ab;       /* record of bit 0 of the current mask word: its coefficients ... */
        const double *Rbt = m.beta;   /* ... and its direction (two running addresses: one instruction each per arc, no index to add first) */
        const unsigned *mk = m.it_mask; /* current mask word */
        double t = 0, hh = 0;
        Lr2Union u;
        lr2_union_reset(u);
/* on to the item's next mask word with a bit set: mw - 1 straight steps (mw is 2 unless lists are long) */
#define LR2_NEXT_WORD()                                                                            \
    for (int k_ = 1; k_ < mwt; ++k_)                                                               \
        if (w == 0 && wleft > 0) { ++mk; Rab += 32; Rbt += 32; --wleft; w = *mk; }
#define LR2_FETCH(idx)                                                                             \
    do {                                                                                           \
        const int e_ = (idx) < nq ? (int)m.queue[(idx)] : LR2_NONE;                                \
        my = e_ == LR2_NONE ? LR2_NONE : (e_ & 1023); w = 0; wleft = 0;                            \
        if (my != LR2_NONE) {                                                                      \
            la = e_ >> 10;                                                                         \
            { const int o_ = m.aoff[la]; Rab = m.ab + o_; Rbt = m.beta + o_; } hh = m.it_tc[my];     \
            t = lr2_slice_height_at<WALK>(my - LR2_MUL24(la, ns), m.adel[la], m.atom[la].w, WALK ? m.atom[la].z : 0.0); /* as P4: bit for bit */ \
            mk = m.it_mask + LR2_MUL24(my, mw); w = *mk; wleft = mwt - 1;                          \
            LR2_NEXT_WORD();                                                                       \
        }                                                                                          \
    } while (0)
        { int ln = lane; SASA_OPAQUE(ln); LR2_FETCH(ln); } /* (as in P4: not hoisted out of the tile loop) */
        for (;;) {
            /* a refill is due when `due` lanes wait: the launch's threshold while the queue has items, all 64 after */
            const int due = next < nq && a.refill < LR2_LANES ? a.refill : LR2_LANES;
            for (;;) { /* arc steps until a refill is due */
                const bool act = w != 0;
                const unsigned long long am = LR2_BALLOT(act);
                if (LR2_LANES - LR2_POPC64(am) >= due) break;
                LR2_COUNT(1, 1);
                if (act) {
                    const int q = __builtin_ctz(w);
                    w &= w - 1;
                    const Ab16 ab = Rab[q];
                    const double bt = Rbt[q];
                    const double alpha = lr2_arc_alpha(t, ab, hh);
                    lr2_union_step(bt - alpha, bt + alpha, u, stk, LR2_A_DS(a), maxd); /* ref: :338-339 */
                }
                for (int more = 1; more < LR2_ARC_STEPS; ++more) /* further steps before the waiting lanes are counted again */
                    if (w != 0) {
                        const int q = __builtin_ctz(w);
                        w &= w - 1;
                        const Ab16 ab = Rab[q];
                        const double bt = Rbt[q];
                        const double alpha = lr2_arc_alpha(t, ab, hh);
                        lr2_union_step(bt - alpha, bt + alpha, u, stk, LR2_A_DS(a), maxd);
                    }
            }
            /* refill: a lane that has used up its mask word moves on to the item's next word or, when the item
               is finished, stores its area and takes the next item of the queue */
            LR2_NEXT_WORD();
            const unsigned long long im = LR2_BALLOT(w == 0); /* finished (or without an item) */
            LR2_COUNT(2, 1);
            if (w == 0) {
                if (my != LR2_NONE) m.it_tc[my] = m.adel[la] * m.atom[la].w * lr2_sweep(u, stk, LR2_A_DS(a)); /* ref: :360 */
                lr2_union_reset(u);
                LR2_FETCH(next + LR2_RANK(im, lane));
            }
            next += LR2_POPC64(im);
            if (next - LR2_POPC64(im) >= nq && LR2_BALLOT(w != 0) == 0) break;
        }
#undef LR2_FETCH
#undef LR2_NEXT_WORD
    }
    LR2_SYNC();
    LR2_MARK(6);

    /* ------------------------------------------------------------ P7 store */
    const bool deep = LR2_BALLOT(maxd - 2 > LR2_A_DS(a)) != 0; /* an arc stack column was too short: the tile is redone */
    if (!deep && lane < na) {
        double s = 0;
        const double *const tc = m.it_tc + LR2_MUL24(lane, ns);
        int k = 0;
        for (; k + 10 <= ns; k += 10) { /* slice order, ref: :305-361; ten reads in flight (20 slices: two LDS round trips) */
            const double v0 = tc[k], v1 = tc[k + 1], v2 = tc[k + 2], v3 = tc[k + 3], v4 = tc[k + 4];
            const double v5 = tc[k + 5], v6 = tc[k + 6], v7 = tc[k + 7], v8 = tc[k + 8], v9 = tc[k + 9];
            s += v0; s += v1; s += v2; s += v3; s += v4; s += v5; s += v6; s += v7; s += v8; s += v9;
        }
        for (; k + 4 <= ns; k += 4) { /* four reads in flight */
            const double v0 = tc[k], v1 = tc[k + 1], v2 = tc[k + 2], v3 = tc[k + 3];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < ns; ++k) s += tc[k];
        if ((m.flags[6] >> lane) & 1) s = NAN; /* duplicate atom record (see P3) */
        LR2_WARM(a, sasa)[m.sorig[lane]] = s; /* (generic build: the pointer is read here, once per tile) */
    }
    LR2_SYNC();
    LR2_MARK(7);
    if (deep) LR2_COUNT(7, 1);
    return deep ? 1 : 0;
}

/* The work items of one wave (items first, first + stride, ... of the launch).  A tile that does not fit is redone
 * at once as two halves (1.5 % of the 6-atom tiles of random coils at a pool of 224 records); what still does not fit
 * goes to the next launch's list. */
template <int RMAX, bool COVER, bool PAIRS, int SHAPE = 0, bool HOOKS = false, bool WALK = false>
SASA_D void lr2_wave(const Lr2Args &a, const Lr2Mem &m, int first, int stride, int lane, int &wg_max_nn)
{
    /* all tiles (main launch, rounded up to whole XCD groups) or the items of a work list */
    int n_work = a.work_items ? *a.work_count : ((a.n_tiles + 7) >> 3) << 3;
    int tile0 = 0, n_seg = a.n_tiles; /* the part of the launch this workgroup belongs to: its first tile, its tiles */
    if (!a.work_items && a.seg_grid > 0) {
        if (first < a.seg_grid) { n_seg = a.seg_tiles; stride = a.seg_grid; }
        else { tile0 = a.seg_tiles; n_seg = a.n_tiles - a.seg_tiles; first -= a.seg_grid; stride -= a.seg_grid; }
        n_work = ((n_seg + 7) >> 3) << 3;
    }
    int splits = 0, far_tiles = 0;
    Lr2Pre pre;
    lr2_pre_none(pre);
    for (int w = first; w < n_work; w += stride) {
        int p0, na, tile = 0;
        int p0n = -1, nan = 0; /* what this wave does after this item (main launch), for P0's loads to run ahead */
        if (a.work_items) {
            const long long e = a.work_items[w];
            p0 = (int)(e & 0xffffffffLL); na = (int)(e >> 32);
        } else {
            tile = xcd_tile(w, n_seg);
            if (tile >= n_seg) continue; /* uniform per wave */
            tile += tile0;
            p0 = tile * LR2_A_TA(a);
            na = a.n_atoms - p0 < LR2_A_TA(a) ? a.n_atoms - p0 : LR2_A_TA(a);
            const int tn = w + stride < n_work ? xcd_tile(w + stride, n_seg) : n_seg;
            if (tn < n_seg) { p0n = (tile0 + tn) * LR2_A_TA(a); nan = a.n_atoms - p0n < LR2_A_TA(a) ? a.n_atoms - p0n : LR2_A_TA(a); }
        }
        if (na <= 0) continue;
        /* one call site (the tile is ~9000 instructions): a tile that does not fit is redone as two halves, a
           half that does not fit either goes to the next launch's list */
        int rest0 = 0, rest_n = 0;
        bool whole = true, sample = !a.work_items && (tile & 31) == 0, by_place = false;
        for (;;) {
            /* (what comes next: the second half of a split tile, else the wave's next tile) */
            const bool nxt = rest_n > 0 || nan > 0;
            int fail = lr2_tile<RMAX, COVER, PAIRS, SHAPE, HOOKS, WALK>(a, m, p0, na, sample && whole, by_place, lane, wg_max_nn, pre,
                                             rest_n > 0 ? rest0 : (nxt ? p0n : p0), rest_n > 0 ? rest_n : (nxt ? nan : na), far_tiles);
            sample = false;
            if (fail == 3) { /* an atom beyond the closed-form range of the slice planes: the whole tile to the next launch, whose build walks to them (no halves: they would say the same; not a split tile either - ST_FAR counts these, ST_SPLIT does not) */
                if (lane == 0) lr2_overflow(a, p0, na, ERR_NEIGHBOR_CAP);
                break;
            }
            if (fail == 2) { /* equal sort keys: once more, ties by place of discovery */
                if (!by_place) { by_place = true; continue; }
                fail = 1;
            }
            by_place = false;
            if (fail && whole && na > 1) {
                ++splits;
                whole = false;
                const int h = (na + 1) >> 1;
                rest0 = p0 + h; rest_n = na - h; na = h;
                continue;
            }
            if (fail) {
                if (whole) ++splits;
                if (lane == 0) lr2_overflow(a, p0, na, ERR_NEIGHBOR_CAP);
            }
            whole = false;
            if (rest_n == 0) break;
            p0 = rest0; na = rest_n; rest_n = 0;
        }
    }
    if (lane == 0 && splits > 0) {
        int *const split_count = LR2_COLD(a, split_count);
        if (split_count) SASA_ATOMIC_ADD_GLB(&split_count[first & 63], splits);
    }
    if (lane == 0 && far_tiles > 0 && !a.work_items) SASA_ATOMIC_ADD_GLB(&LR2_COLD(a, status)[ST_FAR], far_tiles);
}

/* launch configuration (host side; shared by gpu_engine.hip and the test emulation) */
struct Lr2Cfg {
    int TA, ns, pool, mw, ds, refill, rmax;
    int lds;
};
#define LR2_ITEMS_CAP 512   /* TA * ns of a tile */
#define LR2_NS_MAX 256      /* finer resolutions use the first-generation kernel */
#define LR2_RMAX_MAIN 4
#define LR2_RMAX_MID 6
#define LR2_MID_BLOCKS 32768 /* workgroups of the second launch (they share its work list) */

static inline bool lr2_supported(int ns) { return ns >= 1 && ns <= LR2_NS_MAX; }
/* tile shapes whose (atom, slice) items are more than the wave's lanes but at most two per lane (6 atoms x 20 slices:
   120): the screening gives every lane two neighboring slices of one atom (P4) */
static inline bool lr2_pairs_shape(int TA, int ns) { return TA * ns > LR2_LANES && TA * ((ns + 1) / 2) <= LR2_LANES; }
static inline bool lr2_default_shape(int TA, int ns, int mw, int ds) { return TA == LR2_SHAPE_TA && ns == LR2_SHAPE_NS && mw == LR2_SHAPE_MW && ds == LR2_SHAPE_DS; }
/* the build (template parameter SHAPE) a tile shape has, 0: none */
static inline int lr2_shape_id(int TA, int ns, int mw, int ds)
{
    for (int s = 1; s <= 4; ++s)
        if (TA == lr2_shape_ta(s) && ns == lr2_shape_ns(s) && mw == lr2_shape_mw(s) && ds == lr2_shape_ds(s)) return s;
    return 0;
}

/* nn_hint: neighbor records one atom needs (with its safety margin), 0 = unknown; nn_max_hint: the longest
 * neighbor list expected (0 = unknown): the masks of an item get ceil(nn_max / 32) words, two at least */
/* LDS decides how many one-wave tiles a CU holds (160 KB; at most 16 with the registers of the 4-waves-per-SIMD
 * build).  Measured on coils at 20 / 40 / 60 / 100 slices and on globules: a launch at 16 tiles per CU beats the
 * same tile shape one LDS step down by 6-9 % (two steps: 20 %).  A tile whose neighbor records exceed the pool is
 * redone as two halves: each % of such tiles costs ~0.5 % while a half still has an item for every lane (>= 64
 * items), ~1.2 % when it does not (6 atoms x 20 slices -> halves of 60 items).  More atoms per tile beat fewer at
 * equal occupancy (the queue of the arc pass has more to balance).
 * lr2_step_eff: throughput relative to 16 tiles per CU; lr2_split_cost: cost of a split tile, in tiles. */
static inline double lr2_step_eff(int tiles_per_cu)
{
    return tiles_per_cu >= 16 ? 1.0 : (tiles_per_cu >= 14 ? 0.92 : (tiles_per_cu >= 12 ? 0.80 : 0.80 * tiles_per_cu / 12.0));
}
static inline bool lr2_halves_fill_the_wave(int TA, int ns) { return ((TA + 1) / 2) * ns >= LR2_LANES; }
static inline double lr2_split_cost(int TA, int ns) { return lr2_halves_fill_the_wave(TA, ns) ? 0.5 : 1.2; }
/* share of split tiles a shape may have before one atom less per tile is the better shape */
static inline double lr2_split_limit(int TA, int ns) { return lr2_halves_fill_the_wave(TA, ns) ? 0.12 : 0.06; }
/* largest pool (even, <= pool_max) with which tiles_per_cu tiles fit a CU; 0: not even the smallest */
static inline int lr2_pool_for_step(int TA, int ns, int mw, int ds, int tiles_per_cu, int pool_max)
{
    const int cu_lds = 160 * 1024;
    int pool = 16;
    if (lr2_layout(TA, ns, pool, mw, ds).total * tiles_per_cu > cu_lds) return 0;
    while (pool + 2 <= pool_max && lr2_layout(TA, ns, pool + 2, mw, ds).total * tiles_per_cu <= cu_lds) pool += 2;
    return pool;
}

/* nn_hint: neighbor records per atom of the tile at the 96th percentile of demand (0: unknown); nn_max_hint: the
 * longest list expected; last_ta / last_split: atoms per tile of the previous batch of this kind on the context
 * and the share of its tiles that exceeded the 16-per-CU pool (last_ta 0: no history) */
static inline Lr2Cfg lr2_choose_cfg(int ns, double nn_hint = 0, int ta_override = 0, int nn_max_hint = 0, int last_ta = 0,
                                    double last_split = 0)
{
    Lr2Cfg c;
    c.ns = ns;
    const int pool_max = LR2_LANES * LR2_RMAX_MAIN;
    /* mask words per item: two (64 neighbors) unless lists that long are common - the rare atom above the
       capacity sends its tile to the next launch, which costs less than LDS for every tile (occupancy) */
    c.mw = 2;
    if (nn_hint > 0 && 1.45 * nn_hint > 64 && nn_max_hint > 64) c.mw = (nn_max_hint + 31) / 32;
    if (c.mw > 4) c.mw = 4;
    c.ds = 2;
    c.refill = 24; /* waiting lanes that trigger an item switch.  Emulation, coils: 19.3 / 19.5 / 20.0 / 20.6 / 21.1 / 22.6 arc iterations and 4.7 / 3.9 / 3.0 / 2.6 / 2.4 / 2.0 switches per tile at 4 / 8 / 16 / 24 / 32 / 48; with ~50 instructions per arc step and ~105 per switch the work is least at 16-32, and the MI355X agrees (round 4, 1e7 atoms: 9.57 / 9.56 / 9.58 ms at 16 / 24 / 32; round 3, with a switch of ~165: flat from 16 to 48) */
    /* atoms per tile: as many as give at most ~320 items and run 16 tiles per CU with few enough split tiles */
    int ta_cap = 320 / ns;
    if (ta_cap < 1) ta_cap = 1;
    if (ta_cap > 6) ta_cap = 6;
    while (ta_cap > 1 && ta_cap * ns > LR2_ITEMS_CAP) --ta_cap;
    const double per_atom = nn_hint > 0 ? nn_hint : 32.0;
    int ta;
    if (last_ta > 0) {
        /* history: stay, unless too many tiles were split (one atom less) or one atom more clearly fits */
        ta = last_ta < ta_cap ? last_ta : ta_cap;
        /* Protein density at the default resolution (lists beyond 64 common, <= 32 slices: tiles of 3 or 4 atoms) - round
           5, measured on the final kernel: with 3 atoms per tile the reference's PDB entries run 2.3 % faster than with the
           4 that just fit (6.08 against 6.24 ms), the bench's lattice globules 3 % (5.66 / 5.83) - an item per lane in the
           screening, fewer tiles redone as halves - so a tile of such a batch gets a fourth atom only where the pool has a
           tenth to spare.  (Another lattice, tools/gpu_shapes.py g100, prefers 4 by 9 %: its four atoms share a cell more
           often, and the neighbor search tests a candidate against all four in one work item.  A rule on the atoms per
           cell that told the two apart sent the bench's globules the wrong way, and was taken out again.) */
        const bool dense20 = c.mw >= 3 && ns <= 32;
        const double fit = dense20 ? 0.90 : 1.0;
        if (ta > 1 && (last_split > lr2_split_limit(ta, ns) || (dense20 && per_atom * ta + 8 > fit * lr2_pool_for_step(ta, ns, c.mw, c.ds, 16, pool_max)))) --ta;
        else if (ta < ta_cap && per_atom * (ta + 1) + 8 <= fit * lr2_pool_for_step(ta + 1, ns, c.mw, c.ds, 16, pool_max)) ++ta;
    } else {
        /* first batch: from the density estimate; 0.8 (0.95) of the 96th-percentile demand in the pool leaves
           ~10 % (~5 %) of the tiles to be split */
        ta = ta_cap;
        while (ta > 1 && (lr2_halves_fill_the_wave(ta, ns) ? 0.8 : 0.95) * (per_atom * ta + 8) >
                             lr2_pool_for_step(ta, ns, c.mw, c.ds, 16, pool_max))
            --ta;
    }
    if (ta_override > 0 && ta_override <= 7 && ta_override * ns <= LR2_ITEMS_CAP) ta = ta_override;
    c.TA = ta;
    /* the pool: everything the 16-tile step has room for; a shape that needs more than that even for 80 % of its
       demand (very dense input at one atom per tile, or a forced shape) runs at the step that holds it */
    c.pool = lr2_pool_for_step(ta, ns, c.mw, c.ds, 16, pool_max);
    int want = (int)(0.8 * (per_atom * ta + 8));
    if (want > pool_max) want = pool_max;
    for (int nblk = 15; c.pool < want && nblk >= 1; --nblk) c.pool = lr2_pool_for_step(ta, ns, c.mw, c.ds, nblk, pool_max);
    if (c.pool < 16) c.pool = 16;
    c.lds = lr2_layout(c.TA, c.ns, c.pool, c.mw, c.ds).total;
    c.rmax = (c.pool + LR2_LANES - 1) / LR2_LANES;
    return c;
}

/* Rounds of 64 pair records (P3) per ATOM that tiles of `ta` atoms would take, from the sampled demand histogram of a
 * batch run with tiles of TA atoms (records per tile scale with the atoms per tile), and the share of such tiles that
 * would not fit `pool` records.  P3 costs a round whatever it holds: 3 atoms x 48 neighbors are three rounds for three
 * atoms, 4 x 48 three rounds for four. */
static inline double lr2_rounds_per_atom(const int *hist, int TA, int ta, int pool, double *above = nullptr)
{
    long long total = 0;
    for (int k = 0; k < 64; ++k) total += hist[k];
    if (above) *above = 0;
    if (total <= 0 || TA <= 0 || ta <= 0) return 0;
    const int w = hist_bin_width(TA);
    double rounds = 0, ab = 0;
    for (int k = 0; k < 64; ++k) {
        if (!hist[k]) continue;
        const double d = (k + 0.5) * w * (double)ta / (double)TA;
        rounds += hist[k] * (double)(((int)d + LR2_LANES - 1) / LR2_LANES);
        if (d > pool) ab += hist[k];
    }
    if (above) *above = ab / (double)total;
    return rounds / (double)total / (double)ta;
}

/* neighbor records per tile that all but ~4 % of the tiles of the last batch needed (sampled demand
 * histogram of P2): what decides how many atoms a tile of the next batch of this kind gets */
static inline int lr2_need_from_hist(const int *hist, int TA)
{
    long long total = 0;
    for (int k = 0; k < 64; ++k) total += hist[k];
    if (total <= 0) return 0;
    long long allowed = total / 25, acc = 0;
    int k = 63;
    for (; k > 0; --k) {
        acc += hist[k];
        if (acc > allowed) break;
    }
    if (k >= 63) return 0;
    return ((k + 1) * hist_bin_width(TA) + 1) & ~1;
}

/* The pool (neighbor records per tile) the next batch of this kind should run with, from the sampled demand
 * histogram of P2: for every occupancy step, the largest pool that fits it and the share f of tiles above it; the
 * step with the least (1 + lr2_split_cost f) / lr2_step_eff(tiles per CU).  split16: f at 16 tiles per CU (what
 * lr2_choose_cfg wants to know about this tile shape).  0: no histogram. */
static inline int lr2_pool_from_hist(const int *hist, int TA, int ns, int mw, int ds, double *split16 = nullptr)
{
    long long total = 0;
    for (int k = 0; k < 64; ++k) total += hist[k];
    if (split16) *split16 = 0;
    if (total <= 0) return 0;
    const int w = hist_bin_width(TA), pool_max = LR2_LANES * LR2_RMAX_MAIN;
    double best_cost = 0;
    int best_pool = 0, last_pool = 0;
    for (int nblk = 16; nblk >= 6; --nblk) {
        const int pool = lr2_pool_for_step(TA, ns, mw, ds, nblk, pool_max);
        if (pool == 0 || pool == last_pool) continue;
        last_pool = pool;
        double above = 0; /* tiles needing more than pool records (uniform within a bin; the last bin is open) */
        for (int k = 0; k < 64; ++k) {
            const int lo = k * w, hi = lo + w - 1;
            if (k == 63 || lo > pool) above += hist[k];
            else if (hi > pool) above += hist[k] * (double)(hi - pool) / w;
        }
        const double f = above / (double)total;
        if (nblk == 16 && split16) *split16 = f;
        const double cost = (1.0 + lr2_split_cost(TA, ns) * f) / lr2_step_eff(nblk);
        if (best_pool == 0 || cost < best_cost) { best_cost = cost; best_pool = pool; }
        if (f == 0) break; /* larger pools only cost occupancy */
    }
    return best_pool;
}
static inline Lr2Cfg lr2_mid_cfg(const Lr2Cfg &main_cfg)
{
    Lr2Cfg c = main_cfg;
    c.rmax = LR2_RMAX_MID;
    c.pool = LR2_LANES * LR2_RMAX_MID;
    c.mw = 4;
    c.ds = 6;
    c.lds = lr2_layout(c.TA, c.ns, c.pool, c.mw, c.ds).total;
    return c;
}

/* the third launch runs the first-generation kernel (slab-backed lists, any neighbor count) over the
 * SAME tiling: one wave, TA atoms, slice areas in an LDS table */
static inline TileCfg lr_slab_cfg(int TA, int ns)
{
    TileCfg c;
    c.B = 64; c.TA = TA; c.tab = 1; c.items = TA * ns; c.cap_idx = 128; c.pool = 64 * TA; c.lr = 1; c.ds = 3;
    c.lds = tile_fixed_bytes(c.TA, c.items) + tile_list_bytes(c.TA, c.cap_idx, c.pool, c.lr, c.ds, c.B);
    return c;
}

} /* namespace sasa */
#endif
