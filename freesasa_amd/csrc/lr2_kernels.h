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
 *   P2 offsets
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
 * Tiles that do not fit the LDS capacities of a launch go to the next launch's work list exactly
 * as in sasa_kernels.h (larger LDS lists, then the slab-backed first-generation kernel).
 *
 * Arithmetic: fp64 only; -ffp-contract=off; fma only where written.
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
#else
#define LR2_BALLOT(p) __builtin_amdgcn_ballot_w64(p)
#define LR2_SYNC() __syncthreads() /* one wave per workgroup: no s_barrier is emitted, only the LDS fence */
#define LR2_SHFL(v, src) __shfl((v), (src), 64)
#define LR2_POPC64(m) __popcll(m)
#define LR2_POPC32(m) __popc(m)
#define LR2_RANK(m, lane) ((int)__builtin_amdgcn_mbcnt_hi((unsigned)((m) >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)(m), 0u)))
#define LR2_SHIFT_IN_LT1(w, c) sasa_shift_in_lt1((w), (c))
#endif

namespace sasa {

#define LR2_LANES 64
#define LR2_NONE 0xffff

struct Lr2Args {
    const double *sx, *sy, *sz, *sr;
    const int *s_orig, *s_struct;
    const long long *s_cell;
    const GridS *grid;
    const int *cell_start;
    int n_atoms, n_tiles;
    int TA;     /* atoms per tile, 9*TA <= 64 */
    int ns;     /* slices per atom */
    int pool;   /* neighbor records per tile (even) */
    int mw;     /* 32-bit mask words per item: at most 32*mw neighbors per atom */
    int ds;     /* spilled levels of the arc stack */
    int refill; /* waiting lanes that trigger a refill of the arc pass */
    double *sasa;
    int *ovf_count; /* tiles that do not fit are appended to the next launch's work list */
    int *ovf_tiles;
    const int *work_tiles; /* tile ids to (re)do; null in the main launch (all tiles) */
    const int *work_count;
    int *status;
};

struct __attribute__((aligned(16))) Pair16 { double a, b; };

/* LDS layout of one tile (byte offsets), shared by the host (launch size) and the device */
struct Lr2Layout {
    int o_atoms, o_ints, o_t, o_ch, o_mask, o_queue, o_qtmp, o_r1, o_r2, total;
};
SASA_HD int lr2_a16(int v) { return (v + 15) & ~15; }
SASA_HD int lr2_n_ints(int TA) { return 6 * TA + 1 + 8 + 18 * TA + (9 * TA + 2) + 64; }
SASA_HD Lr2Layout lr2_layout(int TA, int ns, int pool, int mw, int ds)
{
    Lr2Layout L;
    const int items = TA * ns;
    int p = 0;
    L.o_atoms = p; p += lr2_a16(8 * 5 * TA);
    L.o_ints = p;  p += lr2_a16(4 * lr2_n_ints(TA));
    L.o_t = p;     p += lr2_a16(8 * items);
    L.o_ch = p;    p += lr2_a16(8 * items);
    L.o_mask = p;  p += lr2_a16(4 * items * mw);
    L.o_queue = p; p += lr2_a16(2 * items);
    L.o_qtmp = p;  p += lr2_a16(2 * items);
    /* R1: the hits of P1 (32 B each), overwritten in P3 by the records (a', b': 16 B, beta: 8 B) */
    L.o_r1 = p;    p += 32 * pool;
    /* R2: sort keys and hit tags of P1..P3, then the arc stack of P6 */
    const int r2a = lr2_a16(8 * pool) + lr2_a16(4 * pool), r2b = 16 * LR2_LANES * ds;
    L.o_r2 = p;    p += r2a > r2b ? r2a : r2b;
    L.total = p;
    return L;
}

struct Lr2Mem {
    double *ax, *ay, *az, *aR, *adel; /* [TA] tile atoms; adel = 2 Ri / ns (ref: src/sasa_lr.c:304) */
    int *acell, *lead, *gsz, *acnt, *aoff, *sorig, *flags, *rowlo, *rowcnt, *cpre, *hist;
    double *it_t;   /* [items] slice height relative to the atom centre, z - zi */
    double *it_ch;  /* [items] 1/(2 Ri') of a queued item; its slice area once it is done */
    unsigned *it_mask; /* [items*mw] neighbors that cut an arc */
    unsigned short *queue; /* [items] items with arcs, heaviest first */
    unsigned short *qtmp;  /* [items] bin and arrival order of an item before the bins are laid out */
    Quad *hits;     /* [pool] (xd, yd, zd, Rj) of the neighbors found, in order of discovery */
    Pair16 *ab;     /* [pool] records, sorted by beta inside each atom's list */
    double *beta;   /* [pool] */
    double *keys;   /* [pool] beta with the list position in its low mantissa bits */
    unsigned *tag;  /* [pool] atom (low 8 bits) and list position of hit */
    Arc *stack;     /* [ds][64] */
};
/* flags: 0 tile overflow, 1 stack overflow, 2 max neighbor count, 3 hits found, 4 queue length */

SASA_D Lr2Mem lr2_carve(const Lr2Args &a, char *smem)
{
    const Lr2Layout L = lr2_layout(a.TA, a.ns, a.pool, a.mw, a.ds);
    const int TA = a.TA;
    Lr2Mem m;
    m.ax = (double *)(smem + L.o_atoms); m.ay = m.ax + TA; m.az = m.ay + TA; m.aR = m.az + TA; m.adel = m.aR + TA;
    int *q = (int *)(smem + L.o_ints);
    m.acell = q; q += TA; m.lead = q; q += TA; m.gsz = q; q += TA; m.acnt = q; q += TA; m.aoff = q; q += TA + 1;
    m.sorig = q; q += TA; m.flags = q; q += 8; m.rowlo = q; q += 9 * TA; m.rowcnt = q; q += 9 * TA; m.cpre = q; q += 9 * TA + 2;
    m.hist = q;
    m.it_t = (double *)(smem + L.o_t);
    m.it_ch = (double *)(smem + L.o_ch);
    m.it_mask = (unsigned *)(smem + L.o_mask);
    m.queue = (unsigned short *)(smem + L.o_queue);
    m.qtmp = (unsigned short *)(smem + L.o_qtmp);
    m.hits = (Quad *)(smem + L.o_r1);
    m.ab = (Pair16 *)(smem + L.o_r1);
    m.beta = (double *)(smem + L.o_r1 + 16 * a.pool);
    m.keys = (double *)(smem + L.o_r2);
    m.tag = (unsigned *)(smem + L.o_r2 + lr2_a16(8 * a.pool));
    m.stack = (Arc *)(smem + L.o_r2);
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
 * divides 0 by 0; two coincident atoms of equal radius get b' = NaN: no arc, as sasa_kernels.h). */
SASA_D void lr2_record(double xd, double yd, double zd, double rj, double ri, double &ap, double &bp)
{
    const double D = xd * xd + yd * yd; /* ref: src/nb.c:438 */
    double g = 0, h = 0;
    if (D > 0) sqrt_rh(D, g, h);
    const double inv = D > 0 ? 2.0 * h : 0x1p500; /* 1/dij */
    const double K = (ri * ri - rj * rj) + (D + zd * zd);
    ap = -2.0 * zd * inv;
    bp = K * inv;
    if (!(D > 0) && zd == 0 && K == 0) bp = NAN;
}

/* one step of the arc union on raw end points (inf may be negative, sup may exceed 2 pi: every arc
 * contains its beta in [0, 2 pi], so only the lowest component can start below 0 and only the
 * highest can end above 2 pi; lr2_sweep folds them back) */
SASA_D void lr2_union_step(double inf, double sup, double &ts, double &te, int &depth, Arc *stk, int ds, int &err)
{
    const bool fresh = inf > te; /* te = -inf while there is no component */
    if (fresh && depth > 0) {
        if (depth - 1 < ds) {
            Arc t; t.s = ts; t.e = te;
            stk[(depth - 1) * LR2_LANES] = t;
        } else {
            err = 1;
        }
    }
    ts = fresh ? inf : SASA_MIN(ts, inf);
    te = fresh ? sup : SASA_MAX(te, sup);
    depth += fresh ? 1 : 0;
    if (!fresh)
        while (depth > 1) { /* the merged component may now reach the ones below it */
            const int lv = depth - 2 < ds ? depth - 2 : 0; /* (beyond ds: err is set, the tile is redone) */
            const Arc lo = stk[lv * LR2_LANES];
            if (lo.e < ts) break;
            ts = SASA_MIN(ts, lo.s);
            --depth;
        }
}

/* exposed arc length from the final components (ascending; the top one in ts/te).
 * ref: src/sasa_lr.c:340-351 (arcs through the origin) and :389-408 (sweep) */
SASA_D double lr2_sweep(double ts, double te, int depth, const Arc *stk, int ds)
{
    if (depth == 0) return SASA_TWOPI; /* ref: :392 */
    const double b_s = depth > 1 ? stk[0].s : ts; /* lowest component */
    const bool wrap = b_s < 0 || te > SASA_TWOPI;
    const double Vlo = b_s < 0 ? b_s + SASA_TWOPI : SASA_TWOPI; /* ref: :340 */
    const double Vhi = te > SASA_TWOPI ? ts : SASA_TWOPI;        /* the piece [inf, 2pi] of an arc whose sup wraps */
    const double V = SASA_MIN(Vlo, Vhi);
    const double W = te > SASA_TWOPI ? te - SASA_TWOPI : 0.0;   /* ref: :341 */
    double sum = 0, sup = W;
    for (int c = 0; c < depth; ++c) {
        const bool top = c == depth - 1;
        const int lv = c < ds ? c : 0;
        const Arc k = top ? Arc{ts, te} : stk[lv * LR2_LANES];
        const double ce = top && te > SASA_TWOPI ? SASA_TWOPI : k.e;
        if (wrap && k.s >= V) break; /* sorted behind the [V, 2pi] piece: covered */
        if (sup < k.s) sum += k.s - sup;
        if (ce > sup) sup = ce;
    }
    if (wrap) {
        if (sup < V) sum += V - sup;
        sup = SASA_TWOPI;
    }
    return sum + SASA_TWOPI - sup; /* ref: :407 */
}

/* The whole tile, executed by the 64 lanes of one wave.  RMAX = rounds of 64 pair records a lane
 * keeps in registers in P3 (pool <= 64 * RMAX). */
template <int RMAX>
SASA_D void lr2_tile(const Lr2Args &a, const Lr2Mem &m, int tile, int lane, int &wg_max_nn)
{
    const int TA = a.TA, ns = a.ns, mw = a.mw;
    const int p0 = tile * TA;
    const int na = a.n_atoms - p0 < TA ? a.n_atoms - p0 : TA;
    const int items = na * ns;

    /* ------------------------------------------------------------ P0 load */
    if (lane < TA) {
        if (lane < na) {
            const int p = p0 + lane;
            const double R = a.sr[p];
            m.ax[lane] = a.sx[p]; m.ay[lane] = a.sy[p]; m.az[lane] = a.sz[p]; m.aR[lane] = R;
            m.adel[lane] = 2 * R / ns; /* ref: src/sasa_lr.c:304 */
            m.acell[lane] = (int)(a.s_cell[p] & 0xffffffffLL);
            m.sorig[lane] = a.s_orig[p];
        } else {
            m.ax[lane] = m.ay[lane] = m.az[lane] = 0; m.aR[lane] = 1; m.adel[lane] = 0;
            m.acell[lane] = -1 - lane;
            m.sorig[lane] = 0;
        }
        m.acnt[lane] = 0;
    }
    if (lane < 8) m.flags[lane] = 0;
    m.hist[lane] = 0;
    int my_cnt = 0; /* candidates of row `lane` (rows of atoms that do not lead a cell group count 0) */
    if (lane < 9 * TA) {
        const int la = lane / 9, r = lane - 9 * la;
        int lo = 0, cnt = 0;
        if (la < na) { /* as tile_phase_load of sasa_kernels.h: three dependent round trips */
            const int p = p0 + la;
            const int sid = a.s_struct[p];
            const long long cf = a.s_cell[p];
            const int nx = a.grid[sid].nx, ny = a.grid[sid].ny;
            const int c = (int)(cf & 0xffffffffLL), fl = (int)(cf >> 32);
            const int dy = (r % 3) - 1, dz = (r / 3) - 1;
            const bool out = (dy < 0 && (fl & CELL_Y0)) || (dy > 0 && (fl & CELL_Y1)) ||
                             (dz < 0 && (fl & CELL_Z0)) || (dz > 0 && (fl & CELL_Z1));
            const int row = out ? c : c + nx * (dy + ny * dz);
            const int x_lo = row - ((fl & CELL_X0) ? 0 : 1), x_hi = row + ((fl & CELL_X1) ? 0 : 1);
            const int s0 = a.cell_start[x_lo], s1 = a.cell_start[x_hi + 1];
            lo = out ? 0 : s0;
            cnt = out ? 0 : s1 - s0;
            /* the atoms of a tile are consecutive in cell order: atoms of one cell form a group */
            const bool leads = la == 0 || (int)(a.s_cell[p - 1] & 0xffffffffLL) != c;
            my_cnt = leads ? cnt : 0;
        }
        m.rowlo[lane] = lo;
        m.rowcnt[lane] = cnt;
    }
    LR2_SYNC();
    if (lane < TA) {
        const bool lead = lane < na && (lane == 0 || m.acell[lane] != m.acell[lane - 1]);
        int gs = 0;
        if (lead) { gs = 1; while (lane + gs < na && m.acell[lane + gs] == m.acell[lane]) ++gs; }
        m.lead[lane] = lead ? 1 : 0;
        m.gsz[lane] = gs;
        if (gs > 0) SASA_ATOMIC_MAX_LDS(&m.flags[5], gs);
        if (lane < na) { /* slice heights, accumulated like the reference (src/sasa_lr.c:304-307) */
            const double zi = m.az[lane], Ri = m.aR[lane], delta = m.adel[lane];
            double z = zi - Ri - 0.5 * delta;
            for (int s = 0; s < ns; ++s) {
                z += delta;
                m.it_t[lane * ns + s] = z - zi; /* exact; |z - zi| is the reference's di (:308) */
            }
        }
    }
    /* inclusive prefix of the candidate counts over the rows */
    int incl = my_cnt;
    for (int d = 1; d < LR2_LANES; d <<= 1) {
        const int v = LR2_SHFL(incl, lane >= d ? lane - d : lane);
        if (lane >= d) incl += v;
    }
    if (lane == 0) m.cpre[0] = 0;
    if (lane < 9 * TA) m.cpre[lane + 1] = incl;
    LR2_SYNC();

    /* ------------------------------------------------------------ P1 neighbors */
    const int nrows = 9 * TA;
    const int total_c = m.cpre[nrows];
    int nh = 0; /* hits so far (wave-uniform) */
    {
        const int per = (total_c + LR2_LANES - 1) / LR2_LANES; /* consecutive candidates per lane */
        const int gmax = m.flags[5];
        int f = lane * per;
        const int fend = f + per < total_c ? f + per : total_c;
        int t = 0;
        if (f < total_c) { /* row of the lane's first candidate: largest t with cpre[t] <= f */
            for (int step = 32; step >= 1; step >>= 1)
                if (t + step <= nrows && m.cpre[t + step] <= f) t += step;
        }
        for (int base = 0; base < per; base += 2) { /* (wave-uniform trip count) */
            int q[2], la0[2], gs[2];
            double x[2], y[2], z[2], rq[2];
            for (int j = 0; j < 2; ++j) {
                const int fj = f + base + j;
                if (base + j < per && fj < fend) {
                    while (fj >= m.cpre[t + 1]) ++t;
                    q[j] = m.rowlo[t] + (fj - m.cpre[t]);
                    la0[j] = t / 9;
                    gs[j] = m.gsz[la0[j]];
                } else {
                    q[j] = -1; la0[j] = 0; gs[j] = 0;
                }
            }
            for (int j = 0; j < 2; ++j) {
                const unsigned u = (unsigned)(q[j] < 0 ? 0 : q[j]);
                x[j] = a.sx[u]; y[j] = a.sy[u]; z[j] = a.sz[u]; rq[j] = a.sr[u];
            }
            for (int g = 0; g < gmax; ++g)
                for (int j = 0; j < 2; ++j) {
                    const int la = la0[j] + g;
                    bool hit = false;
                    double dx = 0, dy = 0, dz = 0;
                    if (g < gs[j] && q[j] != p0 + la) {
                        /* the reference's contact test, operand for operand (src/nb.c:483-492) */
                        const double ri = m.aR[la];
                        const double cut2 = (ri + rq[j]) * (ri + rq[j]);
                        dx = x[j] - m.ax[la]; dy = y[j] - m.ay[la]; dz = z[j] - m.az[la];
                        hit = dx * dx + dy * dy + dz * dz < cut2;
                    }
                    const unsigned long long hm = LR2_BALLOT(hit);
                    if (hm) {
                        if (hit) {
                            const int slot = nh + LR2_RANK(hm, lane);
                            const int sa = SASA_ATOMIC_ADD_LDS(&m.acnt[la], 1);
                            if (slot < a.pool) {
                                Quad hq; hq.x = dx; hq.y = dy; hq.z = dz; hq.w = rq[j]; /* ref: src/nb.c:445-448 */
                                m.hits[slot] = hq;
                                m.tag[slot] = (unsigned)la | ((unsigned)sa << 8);
                            }
                        }
                        nh += LR2_POPC64(hm);
                    }
                }
        }
    }
    LR2_SYNC();

    /* ------------------------------------------------------------ P2 offsets */
    if (lane < TA) {
        int off = 0;
        for (int k = 0; k < lane; ++k) off += (m.acnt[k] + 1) & ~1; /* lists padded to an even length */
        const int c = m.acnt[lane];
        m.aoff[lane] = off;
        if (c > 32 * mw) m.flags[0] = 1;
        if (lane == TA - 1) {
            m.aoff[TA] = off + ((c + 1) & ~1);
            if (off + ((c + 1) & ~1) > a.pool) m.flags[0] = 1;
        }
        SASA_ATOMIC_MAX_LDS(&m.flags[2], c);
    }
    if (lane == 0 && (nh > a.pool || nh > LR2_LANES * RMAX)) m.flags[0] = 1;
    LR2_SYNC();
    if (lane == 0) {
        if (m.flags[2] > wg_max_nn) wg_max_nn = m.flags[2];
        if (!a.work_tiles && (tile & 31) == 0) { /* demand histogram for the next batch's pool size: 1 tile in 32 */
            const int need = m.aoff[TA] / hist_bin_width(TA);
            SASA_ATOMIC_ADD_GLB(&a.status[ST_HIST + (need < 63 ? need : 63)], 1);
        }
    }
    if (m.flags[0]) { /* uniform: the tile goes to the next launch */
        if (lane == 0) {
            if (!a.ovf_tiles) {
                SASA_ATOMIC_MAX_GLB(&a.status[ST_ERROR], (int)ERR_NEIGHBOR_CAP);
            } else {
                const int w = SASA_ATOMIC_ADD_GLB(a.ovf_count, 1);
                a.ovf_tiles[w] = tile;
            }
        }
        LR2_SYNC();
        return;
    }

    /* ------------------------------------------------------------ P3 pair records */
    {
        double r_a[RMAX], r_b[RMAX], r_beta[RMAX];
        int r_la[RMAX], r_sa[RMAX];
        unsigned low = 0xfffu;
        SASA_OPAQUE(low);
        for (int r = 0; r < RMAX; ++r) {
            const int gp = lane + LR2_LANES * r;
            r_la[r] = -1;
            if (gp < nh) {
                const Quad hq = m.hits[gp];
                const unsigned tg = m.tag[gp];
                const int la = (int)(tg & 0xffu), sa = (int)(tg >> 8);
                r_la[r] = la; r_sa[r] = sa;
                lr2_record(hq.x, hq.y, hq.z, hq.w, m.aR[la], r_a[r], r_b[r]);
                r_beta[r] = atan2_fast(hq.y, hq.x) + SASA_PI; /* ref: src/sasa_lr.c:337 */
                m.keys[m.aoff[la] + sa] = lr_rank_key(r_beta[r], (unsigned)sa, low);
            }
        }
        if (lane < TA && (m.acnt[lane] & 1)) m.keys[m.aoff[lane] + m.acnt[lane]] = INFINITY; /* never ranks below */
        LR2_SYNC(); /* every hit is in registers: R1 may now take the records */
        for (int r = 0; r < RMAX; ++r) {
            if (r_la[r] < 0) continue;
            const int la = r_la[r], o = m.aoff[la], nn = m.acnt[la];
            const double kme = lr_rank_key(r_beta[r], (unsigned)r_sa[r], low);
            int rank = 0;
            for (int t = 0; t < nn; t += 2) { /* two keys per LDS read (o is even) */
                const Arc kk = *(const Arc *)(m.keys + o + t);
                rank += kk.s < kme ? 1 : 0;
                rank += kk.e < kme ? 1 : 0;
            }
            Pair16 pr; pr.a = r_a[r]; pr.b = r_b[r];
            m.ab[o + rank] = pr;
            m.beta[o + rank] = r_beta[r];
        }
        if (lane < TA && (m.acnt[lane] & 1)) { /* padding record: cos(alpha) huge, never an arc */
            const int pp = m.aoff[lane] + m.acnt[lane];
            Pair16 pr; pr.a = 0; pr.b = 1e300;
            m.ab[pp] = pr;
            m.beta[pp] = 0;
        }
    }
    LR2_SYNC();

    /* ------------------------------------------------------------ P4 screening */
    const float inv_ns = 1.0f / (float)ns; /* index arithmetic only */
    for (int it = lane; it < items; it += LR2_LANES) {
        int la = (int)(((float)it + 0.5f) * inv_ns), s = it - la * ns; /* it / ns without the integer-division sequence */
        if (s < 0) { --la; s += ns; } else if (s >= ns) { ++la; s -= ns; }
        const double Ri = m.aR[la], t = m.it_t[it];
        const double A = Ri * Ri - t * t; /* Ri'^2, ref: src/sasa_lr.c:309 */
        double area = 0;
        int cnt = 0;
        if (A > 0) { /* ref: :310-312 */
            double Rip, h2;
            sqrt_rh(A, Rip, h2); /* h2 = 1/(2 Ri') */
            const int o = m.aoff[la], nn = m.aoff[la + 1] - o;
            double cmin = 1.0;
            for (int wi = 0; wi < mw; ++wi) {
                unsigned w = 0;
                const int k1 = nn - 32 * wi < 32 ? nn - 32 * wi : 32;
                for (int k = k1 - 2; k >= 0; k -= 2) { /* from the end: neighbor k lands on bit k */
                    const Pair16 q0 = m.ab[o + 32 * wi + k], q1 = m.ab[o + 32 * wi + k + 1];
                    const double c0 = fma(t, q0.a, q0.b) * h2, c1 = fma(t, q1.a, q1.b) * h2;
                    cmin = SASA_MIN(cmin, c0);
                    cmin = SASA_MIN(cmin, c1);
                    w = LR2_SHIFT_IN_LT1(w, c1);
                    w = LR2_SHIFT_IN_LT1(w, c0);
                }
                m.it_mask[it * mw + wi] = w;
                cnt += LR2_POPC32(w);
            }
            if (cmin <= -1.0) cnt = 0; /* circle i inside a neighbor's: buried (ref: :327-330) */
            else if (cnt == 0) area = m.adel[la] * Ri * SASA_TWOPI; /* ref: :360 with exposed_arc_length(n = 0) */
            else area = h2; /* parked for the arc pass */
        }
        m.it_ch[it] = area;
        unsigned short qt = 0xffff;
        if (cnt > 0) { /* queue: heaviest first (bin 0 = 63 arcs or more); inside a bin in order of arrival */
            const int bin = 63 - (cnt < 63 ? cnt : 63);
            const int ord = SASA_ATOMIC_ADD_LDS(&m.hist[bin], 1);
            qt = (unsigned short)((bin << 10) | ord); /* ord < items <= 512 */
        }
        m.qtmp[it] = qt;
    }
    LR2_SYNC();

    /* ------------------------------------------------------------ P5 queue */
    int nq;
    {
        const int hv = m.hist[lane];
        int incl = hv;
        for (int d = 1; d < LR2_LANES; d <<= 1) {
            const int v = LR2_SHFL(incl, lane >= d ? lane - d : lane);
            if (lane >= d) incl += v;
        }
        nq = LR2_SHFL(incl, LR2_LANES - 1);
        m.hist[lane] = incl - hv; /* first queue position of the bin */
        LR2_SYNC();
        for (int it = lane; it < items; it += LR2_LANES) {
            const unsigned qt = m.qtmp[it];
            if (qt != 0xffffu) m.queue[m.hist[qt >> 10] + (qt & 1023u)] = (unsigned short)it;
        }
    }
    LR2_SYNC();

    /* ------------------------------------------------------------ P6 arc pass */
    int err = 0;
    {
        Arc *stk = m.stack + lane;
        int next = LR2_LANES;
        int my = lane < nq ? (int)m.queue[lane] : LR2_NONE;
        int wi = 0, o = 0, la = 0;
        unsigned w = 0;
        double t = 0, h2 = 0, ts = 0, te = -INFINITY;
        int depth = 0;
        if (my != LR2_NONE) {
            la = (int)(((float)my + 0.5f) * inv_ns);
            { int s = my - la * ns; if (s < 0) --la; else if (s >= ns) ++la; }
            o = m.aoff[la]; t = m.it_t[my]; h2 = m.it_ch[my]; w = m.it_mask[my * mw];
        }
        for (;;) {
            while (w == 0 && my != LR2_NONE && wi + 1 < mw) { ++wi; w = m.it_mask[my * mw + wi]; }
            const bool act = w != 0;
            const unsigned long long am = LR2_BALLOT(act);
            const int idle = LR2_LANES - LR2_POPC64(am);
            if (am == 0 || (next < nq && idle >= a.refill)) {
                /* refill: lanes whose item is finished store its area and take the next items of the queue */
                if (!act) {
                    if (my != LR2_NONE) m.it_ch[my] = m.adel[la] * m.aR[la] * lr2_sweep(ts, te, depth, stk, a.ds); /* ref: :360 */
                    const int idx = next + LR2_RANK(~am, lane);
                    my = idx < nq ? (int)m.queue[idx] : LR2_NONE;
                    wi = 0; w = 0; ts = 0; te = -INFINITY; depth = 0;
                    if (my != LR2_NONE) {
                        la = (int)(((float)my + 0.5f) * inv_ns);
                        { int s = my - la * ns; if (s < 0) --la; else if (s >= ns) ++la; }
                        o = m.aoff[la]; t = m.it_t[my]; h2 = m.it_ch[my]; w = m.it_mask[my * mw];
                    }
                }
                if (am == 0 && next >= nq) break;
                next += idle;
                continue;
            }
            if (act) {
                const int k = __builtin_ctz(w) + 32 * wi;
                w &= w - 1;
                const Pair16 q = m.ab[o + k];
                const double bt = m.beta[o + k];
                const double alpha = acos_fast(fma(t, q.a, q.b) * h2); /* the screening's value, bit for bit */
                lr2_union_step(bt - alpha, bt + alpha, ts, te, depth, stk, a.ds, err); /* ref: :338-339 */
            }
        }
    }
    if (err) m.flags[1] = 1;
    LR2_SYNC();

    /* ------------------------------------------------------------ P7 store */
    if (m.flags[1]) { /* an arc stack overflowed: the tile is redone by the next launch */
        if (lane == 0) {
            if (!a.ovf_tiles) {
                SASA_ATOMIC_MAX_GLB(&a.status[ST_ERROR], (int)ERR_STACK_CAP);
            } else {
                const int w = SASA_ATOMIC_ADD_GLB(a.ovf_count, 1);
                a.ovf_tiles[w] = tile;
            }
        }
    } else if (lane < na) {
        double s = 0;
        for (int k = 0; k < ns; ++k) s += m.it_ch[lane * ns + k]; /* slice order, ref: :305-361 */
        a.sasa[m.sorig[lane]] = s;
    }
    LR2_SYNC();
}

/* launch configuration (host side; shared by gpu_engine.hip and the test emulation) */
struct Lr2Cfg {
    int TA, ns, pool, mw, ds, refill, rmax;
    int lds;
};
#define LR2_ITEMS_CAP 512   /* TA * ns of a tile */
#define LR2_NS_MAX 256      /* finer resolutions use the first-generation kernel */
#define LR2_RMAX_MAIN 3
#define LR2_RMAX_MID 8

static inline bool lr2_supported(int ns) { return ns >= 1 && ns <= LR2_NS_MAX; }

/* nn_hint: neighbor records one atom needs (with its safety margin), 0 = unknown */
static inline Lr2Cfg lr2_choose_cfg(int ns, double nn_hint = 0, int ta_override = 0)
{
    Lr2Cfg c;
    c.ns = ns;
    /* about two items per lane let the queue balance the arc pass; more atoms cost LDS (occupancy) */
    int ta = (2 * LR2_LANES + ns / 2) / ns;
    if (ta < 1) ta = 1;
    if (ta > 6) ta = 6;
    while (ta > 1 && ta * ns > LR2_ITEMS_CAP) --ta;
    const int pool_max = LR2_LANES * LR2_RMAX_MAIN;
    if (nn_hint > 0) /* dense inputs: fewer atoms per tile, so that a tile's records fit the registers of P3 */
        while (ta > 1 && nn_hint * ta + 8 > pool_max) --ta;
    if (ta_override > 0 && ta_override <= 7 && ta_override * ns <= LR2_ITEMS_CAP) ta = ta_override;
    c.TA = ta;
    c.rmax = LR2_RMAX_MAIN;
    c.pool = nn_hint > 0 ? (int)(nn_hint * ta + 8) : 32 * ta;
    c.pool = (c.pool + 1) & ~1;
    if (c.pool > pool_max) c.pool = pool_max;
    if (c.pool < 16) c.pool = 16;
    c.mw = 2;
    c.ds = 3;
    c.refill = 16;
    c.lds = lr2_layout(c.TA, c.ns, c.pool, c.mw, c.ds).total;
    return c;
}
static inline Lr2Cfg lr2_mid_cfg(const Lr2Cfg &main_cfg)
{
    Lr2Cfg c = main_cfg;
    c.rmax = LR2_RMAX_MID;
    c.pool = LR2_LANES * LR2_RMAX_MID;
    c.mw = 8;
    c.ds = 8;
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
