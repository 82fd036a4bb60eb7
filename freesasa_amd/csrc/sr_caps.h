/* Shrake & Rupley, third arrangement (round 6): the test points a neighbor covers, LOOKED UP instead of tested.
 *
 * The reference tests every point of an atom against its neighbors until one covers it (src/sasa_sr.c:311-330).  The
 * points are the SAME unit directions u_k for every atom (src/sasa_sr.c:56-90), and "neighbor j covers point k of atom
 * i" is, in exact arithmetic, a cap on atom i's sphere:
 *
 *     |c_i + R_i u_k - c_j|^2 <= R_j^2   <=>   u_k . v/|v|  >=  g,     v = c_j - c_i,  g = (R_i^2 + |v|^2 - R_j^2) / (2 R_i |v|)
 *
 * so which points a neighbor covers depends on a direction and one number only.  A table (built on the host from the
 * caller's unit points, once per point set: sr_captab_build) holds, for every cell of a cube map of directions (N x N
 * cells per face) and every one of L intervals of g, two masks of the points:
 *     DEF   points covered by EVERY cap whose direction lies in the cell and whose g lies in the interval
 *           (with a margin: covered whatever the rounding of the reference's own test and of the lookup's fp32 does),
 *     BAND  points some such cap may cover and not in DEF: only these need the reference's test, operand for operand.
 * One lane per neighbor record looks its entry up (32 bytes from a table that lives in the L2 cache), ORs DEF into the
 * atom's mask, and the few (neighbor, point) pairs of BAND whose point no neighbor has covered for sure - 8 per atom on
 * the reference's PDB entries, 17 on random coils, where the reference runs ~800 point tests per atom - are tested the
 * reference's way.  Counts and areas are bit-exact by construction: a point counts as covered if the reference's own
 * test says so, or if it lies so far inside a cap that no rounding in the reference's test can say otherwise.
 *
 * Margins (in units of the cosine u.v/|v|):
 *   the reference computes the point in two rounded steps and the distance in six more; its verdict can differ from
 *   the exact-arithmetic one only if |u.v/|v| - g| < delta, delta <= [3.9e-16 R_j (|c_i|_inf + R_i) + 3.4e-16 R_j^2 +
 *   R_i^2 ||u|^2 - 1| / 2] / (R_i |v|); the lookup bounds it per pair (sr_cap_lookup: 1e-14 (...)/(R_i |v|), with the
 *   rounding of its own fp64 numerator inside) and sends the pair to the test for ALL points when that exceeds
 *   SR_CAP_DEV_MARGIN (huge coordinates, nearly coincident centres) - so does everything that is not a number;
 *   g and the direction are rounded to fp32, reciprocal and root are the hardware's 1-ulp ones (1e-6 in g; a direction within 3e-7 of a cell border may land in the cell
 *   next door: the cells' angular radius is padded by SR_CAP_RHO_PAD); the table's margin SR_CAP_MARGIN covers both. */
#ifndef SR_CAPS_H
#define SR_CAPS_H

#define SR_CAP_WORDS 4           /* mask words: up to 128 test points (more: the second arrangement, sr_phase_points) */
#define SR_CAP_POINTS_MAX (32 * SR_CAP_WORDS)
#define SR_CAP_N_DEFAULT 16      /* cells per cube-map face edge */
#define SR_CAP_L_DEFAULT 32      /* intervals of g in [-1, 1] */
#define SR_CAP_MARGIN 2e-4       /* table margin in g */
#define SR_CAP_DEV_MARGIN 1e-4f  /* what the per-pair bound of the reference's rounding may use of it */
#define SR_CAP_RHO_PAD 2e-5      /* radians */
#define SR_CAP_UNIT_TOL 4e-15    /* | |u|^2 - 1 | of the caller's points, or no table */

struct alignas(16) SrCapEntry { unsigned def[SR_CAP_WORDS], band[SR_CAP_WORDS]; };

#ifdef SASA_EMU
#define SASA_ATOMIC_OR_LDS(p, v) (*(p) |= (v))
extern long long sr_caps_count_emu[4]; /* tests only: records looked up, points asked through the list, asked by a lane whose items found the list full, records sent to the test for all points */
#define SR_CAPS_COUNT(k, n) (sr_caps_count_emu[(k)] += (n))
#define SASA_ATOMIC_OR64_LDS(p, v) (*(p) |= (v))
#define SASA_RCPF(x) (1.0f / (x))
#define SASA_RSQF(x) (1.0f / sqrtf(x))
#define SASA_WAVE_ANY(p) (p) /* (a thread of the emulation decides for itself) */
#else
#define SR_CAPS_COUNT(k, n) ((void)0)
#define SASA_ATOMIC_OR64_LDS(p, v) atomicOr((p), (v))
#define SASA_RCPF(x) __builtin_amdgcn_rcpf(x) /* 1 ulp: inside the margins (see above) */
#define SASA_RSQF(x) __builtin_amdgcn_rsqf(x)
#define SASA_WAVE_ANY(p) (__builtin_amdgcn_ballot_w64(p) != 0) /* (uniform: true if the predicate holds in any lane of the wave) */
#define SASA_ATOMIC_OR_LDS(p, v) atomicOr((p), (v))
#endif

/* cube-map cell of a direction (not normalised; not the zero vector) */
SASA_HD int sr_cap_cell(float vx, float vy, float vz, int N)
{
    /* selects, no branches (the three cases are equally likely: a wave would run them all) */
    const float ax = fabsf(vx), ay = fabsf(vy), az = fabsf(vz);
    const bool fx = ax >= ay && ax >= az, fy = !fx && ay >= az;
    const float major = fx ? vx : (fy ? vy : vz), s = fx ? vy : (fy ? vz : vx), t = fx ? vz : (fy ? vx : vy);
    const int face = (fx ? 0 : (fy ? 2 : 4)) + (major < 0 ? 1 : 0);
    const float inv = SASA_RCPF(fabsf(major)), h = 0.5f * (float)N;
    int ix = (int)((s * inv + 1.0f) * h), iy = (int)((t * inv + 1.0f) * h);
    ix = ix < 0 ? 0 : (ix > N - 1 ? N - 1 : ix);
    iy = iy < 0 ? 0 : (iy > N - 1 ? N - 1 : iy);
    return (face * N + iy) * N + ix;
}
SASA_HD int sr_cap_level(float g, int L)
{
    int l = (int)((g + 1.0f) * (0.5f * (float)L));
    return l < 0 ? 0 : (l > L - 1 ? L - 1 : l);
}
SASA_HD unsigned sr_cap_full_word(int np, int w) /* the points that exist, word w */
{
    const int left = np - 32 * w;
    return left >= 32 ? 0xffffffffu : (left <= 0 ? 0u : ((1u << left) - 1u));
}

/* the masks of one neighbor record: which of atom i's points it covers for sure (def), which have to be asked (band).
   Without branches: the table is read in every case (at a valid place), the special cases are selects. */
SASA_D void sr_cap_lookup(const TileArgs &a, double xi, double yi, double zi, double ri, const Quad rec,
                          unsigned def[SR_CAP_WORDS], unsigned band[SR_CAP_WORDS])
{
    const double dx = rec.x - xi, dy = rec.y - yi, dz = rec.z - zi;
    const double d2 = dx * dx + dy * dy + dz * dz;
    const double num = ri * ri + d2 - rec.w * rec.w;
    const float fri = (float)ri, frq = (float)rec.w, fd2 = (float)d2, fd = fd2 * SASA_RSQF(fd2);
    const float cabs = fabsf((float)xi) + fabsf((float)yi) + fabsf((float)zi);
    const float ird = SASA_RCPF(fri * fd); /* (coincident centres: 0 * inf, not a number - and so is everything below) */
    const float bound = 1e-14f * (frq * (cabs + fri + frq) + fri * fri) * ird; /* of the reference's rounding, in g */
    const float g = 0.5f * (float)num * ird;
    const bool ok = bound <= SR_CAP_DEV_MARGIN && fri * fd <= 1e30f && g == g; /* (false for anything that is not a number, and for R_i |v| so large that its reciprocal is zero; true: g is a number, or +-inf) */
    const bool all = ok && g < -1.0f - (float)SR_CAP_MARGIN;  /* sphere i inside sphere j */
    const bool none = ok && g > 1.0f + (float)SR_CAP_MARGIN;  /* sphere j inside sphere i: it covers no point of i's surface */
    const bool tab = ok && !all && !none;
    const int cell = sr_cap_cell(tab ? (float)dx : 1.0f, tab ? (float)dy : 0.0f, tab ? (float)dz : 0.0f, a.cap_n);
    const SrCapEntry e = ((const SrCapEntry *)a.captab)[cell * a.cap_l + sr_cap_level(tab ? g : 0.0f, a.cap_l)];
    for (int w = 0; w < SR_CAP_WORDS; ++w) { def[w] = e.def[w]; band[w] = e.band[w]; }
    if (SASA_WAVE_ANY(!tab)) /* (rare, so a branch for the whole wave: nested spheres, and what the bound refuses) */
        for (int w = 0; w < SR_CAP_WORDS; ++w) {
            const unsigned full = sr_cap_full_word(a.n_res, w);
            def[w] = tab ? def[w] : (all ? full : 0u);
            band[w] = tab ? band[w] : (ok ? 0u : full); /* not ok: every point has to be asked */
        }
}

/* LDS of the arrangement: the tile's survivor table of the second arrangement (TileMem::contrib, 2 * items dwords) holds
   SR_CAP_COPIES copies of every atom's DEF words (a record ORs into copy lane % SR_CAP_COPIES: the records of an atom
   are neighboring lanes, and 41 lanes' atomics on one address are served one after the other - measured, round 6: with one
   copy the lookup pass took 2.1 ms of a 3.9 ms kernel), the atoms' COV words (points the reference's test found covered),
   then the list of (atom, record, point) triples to test.  Sized by sr_tile_items. */
#ifndef SR_CAP_COPIES
#define SR_CAP_COPIES 4
#endif
#ifndef SR_CAP_REG_TRIPS
#define SR_CAP_REG_TRIPS 3 /* records per thread whose BAND words stay in registers between the two passes (more: looked up again; six atoms of a protein are ~260 records = three per thread of 128: measured 2 / 3 trips in registers 2.79 / 2.74 ms on the PDB entries) */
#endif
SASA_D unsigned *sr_caps_def(const TileMem &m, int la, int copy) { return (unsigned *)m.contrib + SR_CAP_WORDS * (SR_CAP_COPIES * la + copy); }
SASA_D unsigned *sr_caps_cov(const TileArgs &a, const TileMem &m, int la) { return (unsigned *)m.contrib + SR_CAP_WORDS * (SR_CAP_COPIES * a.TA + la); }
SASA_D unsigned *sr_caps_list(const TileArgs &a, const TileMem &m) { return (unsigned *)m.contrib + SR_CAP_WORDS * (SR_CAP_COPIES + 1) * a.TA; }
SASA_D int sr_caps_list_cap(const TileArgs &a, int items) { return 2 * items - SR_CAP_WORDS * (SR_CAP_COPIES + 1) * a.TA; }
#define SR_CAP_PACK(la, k, pt) (((unsigned)(la) << 14) | ((unsigned)(k) << 7) | (unsigned)(pt)) /* k, pt < 128 */
#define SR_CAP_NONE 0xffffffffu

/* (the unit points stay in global memory: a copy in the LDS - 2.4 KB per workgroup - was measured, round 6, and cost a tenth
   of the resident tiles and 8 % of the time) */
/* beside the load phase: the words cleared, and where the atoms' results go (their places in the caller's order: read here,
   so that the store at the tile's end does not wait for them) */
SASA_D void sr_caps_clear(const TileArgs &a, TileMem &m, int tile, int tid, int B)
{
    unsigned *w = (unsigned *)m.contrib;
    for (int t = tid; t < SR_CAP_WORDS * (SR_CAP_COPIES + 1) * a.TA; t += B) w[t] = 0;
    if (tid < tile_atoms(a, tile)) m.aexp[tid] = a.s_idx[tile_first_atom(a, tile) + tid].orig;
}

/* behind the barrier that follows the neighbor phase (as sr_phase_lists): does every list fit its segment, the longest
   list - and where each atom's records start in the tile's run of records (the passes below take records, not atoms) */
SASA_D void sr_caps_lists(const TileArgs &a, TileMem &m, int tid)
{
    if (tid >= a.TA) return;
    const int c = m.acnt[tid];
    if (c > a.cap_idx) m.flags[0] = 1;
    SASA_ATOMIC_MAX_LDS(&m.flags[2], c);
    int before = 0;
    for (int t = 0; t < tid; ++t) before += m.acnt[t];
    m.aoff[tid] = before;
    if (tid == a.TA - 1) m.aoff[a.TA] = before + c;
}
/* record number s of the tile -> (atom, place in its list) */
SASA_D void sr_caps_record(const TileArgs &a, const TileMem &m, int s, int &la, int &k)
{
    la = 0;
    for (int t = 1; t < a.TA; ++t) la += s >= m.aoff[t] ? 1 : 0;
    k = s - m.aoff[la];
}

/* the reference's test of one point against one neighbor record (R_j not yet squared in the record) */
SASA_D bool sr_caps_test(const TileMem &m, const double *upts, int la, const Quad rec, int pt)
{
    const double ri = m.aR[la];
    /* test point = unit * ri, then + centre: two rounded steps (ref: src/coord.c:331-342, 314-329; as sr_point) */
    double tx = upts[3 * pt] * ri, ty = upts[3 * pt + 1] * ri, tz = upts[3 * pt + 2] * ri;
    tx += m.ax[la]; ty += m.ay[la]; tz += m.az[la];
    Quad q = rec;
    q.w = rec.w * rec.w; /* ref: src/sasa_sr.c:146 */
    return sr_inside(q, tx, ty, tz);
}

struct SrCapRegs { unsigned band[SR_CAP_REG_TRIPS][SR_CAP_WORDS]; int lak[SR_CAP_REG_TRIPS]; /* (atom << 8) | place in its list */ };
struct alignas(16) SrCapWords { unsigned w[SR_CAP_WORDS]; };

/* pass 1: a thread per neighbor record of the tile (records tid, tid + B, ...): what the neighbor covers for sure is ORed
   into its atom's words */
SASA_D int sr_caps_lookup_one(const TileArgs &a, TileMem &m, int s, int tid, unsigned band[SR_CAP_WORDS])
{
    int la, k;
    sr_caps_record(a, m, s, la, k);
    unsigned def[SR_CAP_WORDS];
    SR_CAPS_COUNT(0, 1);
    sr_cap_lookup(a, m.ax[la], m.ay[la], m.az[la], m.aR[la], m.pq[la * a.cap_idx + k], def, band);
    unsigned long long *D = (unsigned long long *)sr_caps_def(m, la, tid & (SR_CAP_COPIES - 1));
    for (int w = 0; w < SR_CAP_WORDS; w += 2)
        SASA_ATOMIC_OR64_LDS(&D[w >> 1], (unsigned long long)def[w] | ((unsigned long long)def[w + 1] << 32));
    return (la << 8) | k;
}
SASA_D void sr_caps_lookup_pass(const TileArgs &a, TileMem &m, int tid, int B, SrCapRegs &r)
{
    const int total = m.flags[0] ? 0 : m.aoff[a.TA];
#pragma unroll
    for (int trip = 0; trip < SR_CAP_REG_TRIPS; ++trip) { /* (unrolled: the words stay in registers, and the trips' table reads are in flight together) */
        const int s = tid + trip * B;
        for (int w = 0; w < SR_CAP_WORDS; ++w) r.band[trip][w] = 0;
        r.lak[trip] = 0;
        if (s < total) r.lak[trip] = sr_caps_lookup_one(a, m, s, tid, r.band[trip]);
    }
    for (int s = tid + SR_CAP_REG_TRIPS * B; s < total; s += B) { /* (tiles of more records: the second launch's, or small workgroups) */
        unsigned band[SR_CAP_WORDS];
        (void)sr_caps_lookup_one(a, m, s, tid, band);
    }
}

/* pass 2, behind a barrier (every atom's words are complete): the points that must be asked, to the tile's list */
SASA_D void sr_caps_todo_one(const TileArgs &a, TileMem &m, int la, int k, unsigned td[SR_CAP_WORDS], int items)
{
    SrCapWords d = *(const SrCapWords *)sr_caps_def(m, la, 0);
    for (int c = 1; c < SR_CAP_COPIES; ++c) {
        const SrCapWords e = *(const SrCapWords *)sr_caps_def(m, la, c);
        for (int w = 0; w < SR_CAP_WORDS; ++w) d.w[w] |= e.w[w];
    }
    int cnt = 0;
    for (int w = 0; w < SR_CAP_WORDS; ++w) { td[w] &= ~d.w[w]; cnt += __builtin_popcount(td[w]); }
    if (cnt == 0) return;
    unsigned *list = sr_caps_list(a, m);
    const int cap = sr_caps_list_cap(a, items);
    const double *upts = a.unit_pts;
    int slot = SASA_ATOMIC_ADD_LDS(&m.flags[3], cnt);
    const unsigned long long b0 = (unsigned long long)td[0] | ((unsigned long long)td[1] << 32), b1 = (unsigned long long)td[2] | ((unsigned long long)td[3] << 32);
    if (slot + cnt <= cap) {
        /* (64 points per loop: a wave runs as many trips as its busiest lane has bits) */
        const unsigned head = SR_CAP_PACK(la, k, 0);
        for (unsigned long long b = b0; b; b &= b - 1) list[slot++] = head | (unsigned)__builtin_ctzll(b);
        for (unsigned long long b = b1; b; b &= b - 1) list[slot++] = head | (unsigned)(64 + __builtin_ctzll(b));
    } else {
        /* the list is full (a pair whose every point has to be asked, times many): this thread asks for itself, and
           fills what is left of the list with blanks (the threads after it find it full too) */
        SR_CAPS_COUNT(2, cnt);
        for (; slot < cap; ++slot) list[slot] = SR_CAP_NONE;
        const Quad rec = m.pq[la * a.cap_idx + k];
        unsigned *V = sr_caps_cov(a, m, la);
#pragma unroll 1
        for (int h = 0; h < 2; ++h)
            for (unsigned long long b = h ? b1 : b0; b; b &= b - 1) {
                const int pt = 64 * h + __builtin_ctzll(b);
                if (sr_caps_test(m, upts, la, rec, pt)) SASA_ATOMIC_OR_LDS(&V[pt >> 5], 1u << (pt & 31));
            }
    }
}
SASA_D void sr_caps_todo_pass(const TileArgs &a, TileMem &m, int tid, int B, int items, SrCapRegs &r)
{
    const int total = m.flags[0] ? 0 : m.aoff[a.TA];
#pragma unroll
    for (int trip = 0; trip < SR_CAP_REG_TRIPS; ++trip) {
        const int s = tid + trip * B;
        if (s < total) sr_caps_todo_one(a, m, r.lak[trip] >> 8, r.lak[trip] & 255, r.band[trip], items);
    }
    for (int s = tid + SR_CAP_REG_TRIPS * B; s < total; s += B) { /* (looked up again: see SR_CAP_REG_TRIPS) */
        int la, k;
        sr_caps_record(a, m, s, la, k);
        unsigned def[SR_CAP_WORDS], td[SR_CAP_WORDS];
        sr_cap_lookup(a, m.ax[la], m.ay[la], m.az[la], m.aR[la], m.pq[la * a.cap_idx + k], def, td);
        sr_caps_todo_one(a, m, la, k, td, items);
    }
}

/* behind a barrier: the list, one (atom, record, point) per thread */
SASA_D void sr_caps_exact(const TileArgs &a, TileMem &m, int tid, int B, int items)
{
    if (m.flags[0]) return;
    const unsigned *list = sr_caps_list(a, m);
    const int cap = sr_caps_list_cap(a, items);
    const double *upts = a.unit_pts;
    const int n = m.flags[3] < cap ? m.flags[3] : cap;
    for (int it = tid; it < n; it += B) {
        const unsigned e = list[it];
        if (e == SR_CAP_NONE) continue;
        SR_CAPS_COUNT(1, 1);
        const int la = (int)(e >> 14), k = (int)((e >> 7) & 127u), pt = (int)(e & 127u);
        if (sr_caps_test(m, upts, la, m.pq[la * a.cap_idx + k], pt)) SASA_ATOMIC_OR_LDS(&sr_caps_cov(a, m, la)[pt >> 5], 1u << (pt & 31));
    }
}

SASA_D void sr_caps_store(const TileArgs &a, TileMem &m, int tile, int tid)
{
    if (m.flags[0]) return;
    const int na = tile_atoms(a, tile);
    if (tid < na) {
        const unsigned *V = sr_caps_cov(a, m, tid);
        int covered = 0;
        for (int w = 0; w < SR_CAP_WORDS; ++w) {
            unsigned d = V[w];
            for (int c = 0; c < SR_CAP_COPIES; ++c) d |= sr_caps_def(m, tid, c)[w];
            covered += __builtin_popcount(d);
        }
        const double ri = m.aR[tid];
        const int n_surface = a.n_res - covered;
        const int i = m.aexp[tid]; /* (sr_caps_clear) */
        a.sasa[i] = (4.0 * SASA_PI * ri * ri * n_surface) / a.n_res; /* ref: src/sasa_sr.c:337 */
        if (a.counts) a.counts[i] = n_surface;
    }
}

/* The table for a set of unit points: 6 N^2 cells x L intervals, an SrCapEntry each.  false: these points are not unit
 * vectors to SR_CAP_UNIT_TOL or there are more than SR_CAP_POINTS_MAX of them (the caller keeps the second arrangement). */
static inline bool sr_captab_build(const double *unit, int np, int N, int L, std::vector<SrCapEntry> &out)
{
    if (np < 1 || np > SR_CAP_POINTS_MAX || N < 1 || L < 1) return false;
    for (int k = 0; k < np; ++k) {
        const double n2 = unit[3 * k] * unit[3 * k] + unit[3 * k + 1] * unit[3 * k + 1] + unit[3 * k + 2] * unit[3 * k + 2];
        if (!(fabs(n2 - 1.0) <= SR_CAP_UNIT_TOL)) return false;
    }
    out.assign((size_t)6 * N * N * L, SrCapEntry());
    std::vector<double> lo(np), hi(np);
    for (int face = 0; face < 6; ++face)
        for (int iy = 0; iy < N; ++iy)
            for (int ix = 0; ix < N; ++ix) {
                /* sr_cap_cell's map: axis = face / 2 carries +-1, the next axis s, the one after t */
                const int axis = face >> 1;
                const double sg = (face & 1) ? -1.0 : 1.0;
                auto dir = [&](double s, double t, double *v) {
                    v[axis] = sg; v[(axis + 1) % 3] = s; v[(axis + 2) % 3] = t;
                    const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                    v[0] /= n; v[1] /= n; v[2] /= n;
                };
                const double s0 = -1.0 + 2.0 * ix / N, s1 = -1.0 + 2.0 * (ix + 1) / N, t0 = -1.0 + 2.0 * iy / N, t1 = -1.0 + 2.0 * (iy + 1) / N;
                double c[3], q[3];
                dir(0.5 * (s0 + s1), 0.5 * (t0 + t1), c);
                double rho = 0; /* angular radius of the cell about c: its farthest corner */
                for (int cs = 0; cs < 4; ++cs) {
                    dir((cs & 1) ? s1 : s0, (cs & 2) ? t1 : t0, q);
                    double d = c[0] * q[0] + c[1] * q[1] + c[2] * q[2];
                    d = d > 1 ? 1 : (d < -1 ? -1 : d);
                    const double ang = acos(d);
                    if (ang > rho) rho = ang;
                }
                rho += SR_CAP_RHO_PAD;
                const double cr = cos(rho), sr = sin(rho);
                for (int k = 0; k < np; ++k) { /* u_k . w over the cell's directions w lies in [lo, hi] = [cos(phi + rho), cos(phi - rho)], phi the angle between u_k and c */
                    double d = c[0] * unit[3 * k] + c[1] * unit[3 * k + 1] + c[2] * unit[3 * k + 2];
                    d = d > 1 ? 1 : (d < -1 ? -1 : d);
                    const double sn = sqrt(1.0 - d * d);                  /* sin(phi) */
                    lo[k] = d > -cr ? d * cr - sn * sr - 2e-8 : -1.0;     /* (phi + rho < pi; the 2e-8: this arithmetic's own rounding - sqrt(1 - d^2) is off by up to 1e-8 where u_k is the cell's centre or its antipode) */
                    hi[k] = d < cr ? d * cr + sn * sr + 2e-8 : 1.0;       /* (phi > rho) */
                }
                SrCapEntry *E = out.data() + (size_t)((face * N + iy) * N + ix) * L;
                for (int l = 0; l < L; ++l) {
                    const double g0 = -1.0 + 2.0 * l / L, g1 = -1.0 + 2.0 * (l + 1) / L;
                    for (int k = 0; k < np; ++k) {
                        const bool def = lo[k] >= g1 + SR_CAP_MARGIN, poss = hi[k] >= g0 - SR_CAP_MARGIN;
                        if (def) E[l].def[k >> 5] |= 1u << (k & 31);
                        else if (poss) E[l].band[k >> 5] |= 1u << (k & 31);
                    }
                }
            }
    return true;
}

#endif /* SR_CAPS_H */
