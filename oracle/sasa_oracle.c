/*
 * sasa_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see sasa_oracle.h).
 *
 * Plain C99 restatement of the reference hot path.  Every floating-point expression
 * that reaches the result keeps the reference's operand order and is compiled with
 * -ffp-contract=off, so on x86-64/glibc the per-atom values are bit-identical to the
 * reference's (checked in tests/test_oracle.py against oracle/_ref).
 *
 * Deliberate structural differences (none changes a result bit):
 *   - neighbor sets are UNIQUE and stored CSR; the reference's ragged lists contain
 *     ~10% duplicates (src/nb.c:106 double-visits three cell-pair directions), which its
 *     algorithms are insensitive to (OR-predicate in S&R, idempotent arc union in L&R);
 *   - every atom scans its 27 surrounding cells itself instead of the reference's
 *     "forward cells + symmetric insert" (src/nb.c:86-130, 409-451);
 *   - S&R with zero neighbors is defined (all points exposed); the reference reads an
 *     uninitialised buffer there (src/sasa_sr.c:313 with nn == 0).
 */
#include "sasa_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static const double ORACLE_TWOPI = 2 * M_PI; /* ref: src/sasa_lr.c:25 */

/* ---------------------------------------------------------------- test points */

void oracle_test_points(int n_points, double *tp)
{
    /* ref: src/sasa_sr.c:60-76 — longitude and z are ACCUMULATED, not k*step */
    const double dlong = M_PI * (3 - sqrt(5)), dz = 2.0 / n_points;
    double longitude = 0, z = 1 - dz / 2;
    int k;
    for (k = 0; k < n_points; ++k) {
        double r = sqrt(1 - z * z);
        tp[3 * k] = cos(longitude) * r;
        tp[3 * k + 1] = sin(longitude) * r;
        tp[3 * k + 2] = z;
        z -= dz;
        longitude += dlong;
    }
}

/* ---------------------------------------------------------------- cell grid */

typedef struct {
    double x0, y0, z0, d;
    int nx, ny, nz;
    int *cell_start; /* ncells+1 */
    int *order;      /* atoms sorted by cell */
} grid_t;

static int cell_coord(double v, double v0, double d)
{
    return (int)((v - v0) / d); /* ref: src/nb.c:137-140 */
}

static int grid_build(grid_t *g, const double *xyz, const double *r_ext, int n)
{
    double lo[3], hi[3], rmax = 0;
    int i, k, ncells, *cell_of = NULL, *fill = NULL;

    for (i = 0; i < n; ++i) rmax = fmax(r_ext[i], rmax); /* ref: src/nb.c:242-254 */
    g->d = 2 * rmax;                                      /* ref: src/nb.c:543 */
    if (!(g->d > 0)) return ORACLE_FAIL;

    for (k = 0; k < 3; ++k) lo[k] = hi[k] = xyz[k];
    for (i = 1; i < n; ++i)
        for (k = 0; k < 3; ++k) {
            lo[k] = fmin(xyz[3 * i + k], lo[k]);
            hi[k] = fmax(xyz[3 * i + k], hi[k]);
        }
    /* ref: src/nb.c:61-70 — box padded by half a cell on each side */
    g->x0 = lo[0] - g->d / 2.;
    g->y0 = lo[1] - g->d / 2.;
    g->z0 = lo[2] - g->d / 2.;
    g->nx = (int)ceil((hi[0] + g->d / 2. - g->x0) / g->d);
    g->ny = (int)ceil((hi[1] + g->d / 2. - g->y0) / g->d);
    g->nz = (int)ceil((hi[2] + g->d / 2. - g->z0) / g->d);
    ncells = g->nx * g->ny * g->nz;

    g->cell_start = calloc((size_t)ncells + 1, sizeof(int));
    g->order = malloc(sizeof(int) * (size_t)n);
    cell_of = malloc(sizeof(int) * (size_t)n);
    fill = calloc((size_t)ncells, sizeof(int));
    if (!g->cell_start || !g->order || !cell_of || !fill) {
        free(cell_of);
        free(fill);
        return ORACLE_FAIL;
    }
    for (i = 0; i < n; ++i) {
        int ix = cell_coord(xyz[3 * i], g->x0, g->d);
        int iy = cell_coord(xyz[3 * i + 1], g->y0, g->d);
        int iz = cell_coord(xyz[3 * i + 2], g->z0, g->d);
        cell_of[i] = ix + g->nx * (iy + g->ny * iz); /* ref: src/nb.c:74-83 */
        ++g->cell_start[cell_of[i] + 1];
    }
    for (k = 0; k < ncells; ++k) g->cell_start[k + 1] += g->cell_start[k];
    for (i = 0; i < n; ++i) g->order[g->cell_start[cell_of[i]] + fill[cell_of[i]]++] = i;
    free(cell_of);
    free(fill);
    return ORACLE_OK;
}

static void grid_free(grid_t *g)
{
    free(g->cell_start);
    free(g->order);
}

/* Visit every j != i with |ci-cj|^2 < (ri+rj)^2.  pass 0 counts, pass 1 stores. */
static int neighbors_of(const grid_t *g, const double *xyz, const double *r, int i, int *out)
{
    const double xi = xyz[3 * i], yi = xyz[3 * i + 1], zi = xyz[3 * i + 2], ri = r[i];
    const int ix = cell_coord(xi, g->x0, g->d);
    const int iy = cell_coord(yi, g->y0, g->d);
    const int iz = cell_coord(zi, g->z0, g->d);
    int cx, cy, cz, k, nn = 0;

    for (cz = iz - 1; cz <= iz + 1; ++cz) {
        if (cz < 0 || cz >= g->nz) continue;
        for (cy = iy - 1; cy <= iy + 1; ++cy) {
            if (cy < 0 || cy >= g->ny) continue;
            for (cx = ix - 1; cx <= ix + 1; ++cx) {
                int c;
                if (cx < 0 || cx >= g->nx) continue;
                c = cx + g->nx * (cy + g->ny * cz);
                for (k = g->cell_start[c]; k < g->cell_start[c + 1]; ++k) {
                    const int j = g->order[k];
                    double rj, cut2, dx, dy, dz;
                    if (j == i) continue;
                    /* ref: src/nb.c:483-492 */
                    rj = r[j];
                    cut2 = (ri + rj) * (ri + rj);
                    dx = xyz[3 * j] - xi;
                    dy = xyz[3 * j + 1] - yi;
                    dz = xyz[3 * j + 2] - zi;
                    if (dx * dx + dy * dy + dz * dz < cut2) {
                        if (out) out[nn] = j;
                        ++nn;
                    }
                }
            }
        }
    }
    return nn;
}

int oracle_neighbors(const double *xyz, const double *r_ext, int n,
                     int **start_out, int **idx_out)
{
    grid_t g = {0};
    int i, *start = NULL, *idx = NULL;

    *start_out = *idx_out = NULL;
    if (n <= 0 || grid_build(&g, xyz, r_ext, n)) {
        grid_free(&g);
        return ORACLE_FAIL;
    }
    start = malloc(sizeof(int) * ((size_t)n + 1));
    if (!start) goto fail;
    start[0] = 0;
    for (i = 0; i < n; ++i) start[i + 1] = start[i] + neighbors_of(&g, xyz, r_ext, i, NULL);
    idx = malloc(sizeof(int) * (size_t)(start[n] > 0 ? start[n] : 1));
    if (!idx) goto fail;
    for (i = 0; i < n; ++i) neighbors_of(&g, xyz, r_ext, i, idx + start[i]);
    grid_free(&g);
    *start_out = start;
    *idx_out = idx;
    return ORACLE_OK;
fail:
    grid_free(&g);
    free(start);
    free(idx);
    return ORACLE_FAIL;
}

/* ---------------------------------------------------------------- Shrake-Rupley */

int oracle_shrake_rupley(const double *xyz, const double *radii, int n,
                         double probe, int n_points, double *sasa, int *counts)
{
    double *r = NULL, *r2 = NULL, *unit = NULL;
    int *start = NULL, *idx = NULL, i, ret = ORACLE_FAIL;

    if (n <= 0 || n_points <= 0) return ORACLE_FAIL;
    r = malloc(sizeof(double) * (size_t)n);
    r2 = malloc(sizeof(double) * (size_t)n);
    unit = malloc(sizeof(double) * 3 * (size_t)n_points);
    if (!r || !r2 || !unit) goto done;

    for (i = 0; i < n; ++i) { /* ref: src/sasa_sr.c:143-147 */
        double ri = radii[i] + probe;
        r[i] = ri;
        r2[i] = ri * ri;
    }
    oracle_test_points(n_points, unit);
    if (oracle_neighbors(xyz, r, n, &start, &idx)) goto done;

    for (i = 0; i < n; ++i) {
        const double ri = r[i];
        const double *vi = xyz + 3 * i;
        const int *nbi = idx + start[i];
        const int nni = start[i + 1] - start[i];
        int n_surface = 0, p, k;

        for (p = 0; p < n_points; ++p) {
            /* ref: src/sasa_sr.c:296-298 via coord.c:331-342 then :306-329 —
               scale, then translate: two separately rounded operations */
            double tx = unit[3 * p] * ri, ty = unit[3 * p + 1] * ri, tz = unit[3 * p + 2] * ri;
            int covered = 0;
            tx += vi[0];
            ty += vi[1];
            tz += vi[2];
            for (k = 0; k < nni && !covered; ++k) {
                /* ref: src/sasa_sr.c:320-324 — covered iff d^2 <= r2[a].  The
                   reference's "last hit first" shortcut (:313-317) only reorders
                   the same tests; the per-point outcome is an OR over neighbors. */
                const int a = nbi[k];
                double dx = tx - xyz[3 * a], dy = ty - xyz[3 * a + 1], dz = tz - xyz[3 * a + 2];
                if (dx * dx + dy * dy + dz * dz <= r2[a]) covered = 1;
            }
            if (!covered) ++n_surface;
        }
        if (counts) counts[i] = n_surface;
        sasa[i] = (4.0 * M_PI * ri * ri * n_surface) / n_points; /* ref: src/sasa_sr.c:337 */
    }
    ret = ORACLE_OK;
done:
    free(r);
    free(r2);
    free(unit);
    free(start);
    free(idx);
    return ret;
}

/* ---------------------------------------------------------------- Lee-Richards */

double oracle_exposed_arc_length(double *arc, int n)
{
    int i, j;
    double sum, sup;

    if (n == 0) return ORACLE_TWOPI; /* ref: src/sasa_lr.c:396 */

    /* order the (start,end) pairs by start (ref: src/sasa_lr.c:367-385);
       any correct sort gives the same sweep result, see DESIGN.md */
    for (i = 1; i < n; ++i) {
        const double s = arc[2 * i], e = arc[2 * i + 1];
        for (j = i; j > 0 && arc[2 * (j - 1)] > s; --j) {
            arc[2 * j] = arc[2 * (j - 1)];
            arc[2 * j + 1] = arc[2 * (j - 1) + 1];
        }
        arc[2 * j] = s;
        arc[2 * j + 1] = e;
    }
    /* ref: src/sasa_lr.c:399-407 */
    sum = arc[0];
    sup = arc[1];
    for (i = 1; i < n; ++i) {
        if (sup < arc[2 * i]) sum += arc[2 * i] - sup;
        if (arc[2 * i + 1] > sup) sup = arc[2 * i + 1];
    }
    return sum + ORACLE_TWOPI - sup;
}

typedef struct {
    double tests, zpass, arcs, buried, max_arcs, nn_sum;
} lr_counters;

static double lr_atom(const double *xyz, const double *R, int i, const int *nbi, int nni,
                      int ns, double *arc, lr_counters *cnt)
{
    const double xi = xyz[3 * i], yi = xyz[3 * i + 1], zi = xyz[3 * i + 2], Ri = R[i];
    const double delta = 2 * Ri / ns; /* ref: src/sasa_lr.c:304 */
    double z = zi - Ri - 0.5 * delta, sasa = 0;
    int s, k;

    for (s = 0; s < ns; ++s) {
        double di, Ri_p2, Ri_p;
        int n_arcs = 0, buried = 0;

        z += delta; /* ref: src/sasa_lr.c:307 — accumulated */
        di = fabs(zi - z);
        Ri_p2 = Ri * Ri - di * di;
        if (Ri_p2 < 0) continue;
        Ri_p = sqrt(Ri_p2);
        if (Ri_p <= 0) continue;

        for (k = 0; k < nni; ++k) {
            const int j = nbi[k];
            const double zj = xyz[3 * j + 2], Rj = R[j];
            const double dj = fabs(zj - z);
            double Rj_p2, Rj_p, xd, yd, dij, alpha, beta, inf, sup;

            if (cnt) cnt->tests += 1;
            if (!(dj < Rj)) continue; /* ref: src/sasa_lr.c:320 */
            if (cnt) cnt->zpass += 1;
            Rj_p2 = Rj * Rj - dj * dj;
            Rj_p = sqrt(Rj_p2);
            /* ref: src/nb.c:426-440 — xd = x_nb - x_i, xyd = sqrt(xd^2+yd^2) */
            xd = xyz[3 * j] - xi;
            yd = xyz[3 * j + 1] - yi;
            dij = sqrt(xd * xd + yd * yd);
            if (dij >= Ri_p + Rj_p) continue; /* ref: :324 */
            if (dij + Ri_p < Rj_p) {          /* ref: :327-330 */
                buried = 1;
                break;
            }
            if (dij + Rj_p < Ri_p) continue; /* ref: :331 */
            alpha = acos((Ri_p2 + dij * dij - Rj_p2) / (2.0 * Ri_p * dij)); /* ref: :335 */
            beta = atan2(yd, xd) + M_PI;                                      /* ref: :337 */
            inf = beta - alpha;
            sup = beta + alpha;
            if (inf < 0) inf += ORACLE_TWOPI;
            if (sup > 2 * M_PI) sup -= ORACLE_TWOPI;
            if (sup < inf) { /* ref: :344-351 — split at the origin */
                arc[2 * n_arcs] = 0;
                arc[2 * n_arcs + 1] = sup;
                arc[2 * n_arcs + 2] = inf;
                arc[2 * n_arcs + 3] = ORACLE_TWOPI;
                n_arcs += 2;
            } else {
                arc[2 * n_arcs] = inf;
                arc[2 * n_arcs + 1] = sup;
                n_arcs += 1;
            }
        }
        if (cnt) {
            if (buried) cnt->buried += 1;
            else {
                cnt->arcs += n_arcs;
                if (n_arcs > cnt->max_arcs) cnt->max_arcs = n_arcs;
            }
        }
        if (!buried) sasa += delta * Ri * oracle_exposed_arc_length(arc, n_arcs); /* ref: :360 */
    }
    return sasa;
}

static int lr_run(const double *xyz, const double *radii, int n, double probe, int n_slices,
                  double *sasa, lr_counters *cnt)
{
    double *R = NULL, *arc = NULL;
    int *start = NULL, *idx = NULL, i, max_nn = 0, ret = ORACLE_FAIL;

    if (n <= 0 || n_slices <= 0) return ORACLE_FAIL;
    R = malloc(sizeof(double) * (size_t)n);
    if (!R) goto done;
    for (i = 0; i < n; ++i) R[i] = radii[i] + probe; /* ref: src/sasa_lr.c:135-138 */
    if (oracle_neighbors(xyz, R, n, &start, &idx)) goto done;
    for (i = 0; i < n; ++i)
        if (start[i + 1] - start[i] > max_nn) max_nn = start[i + 1] - start[i];
    arc = malloc(sizeof(double) * 4 * (size_t)(max_nn > 0 ? max_nn : 1)); /* ref: :92 */
    if (!arc) goto done;
    for (i = 0; i < n; ++i) {
        double a = lr_atom(xyz, R, i, idx + start[i], start[i + 1] - start[i], n_slices, arc, cnt);
        if (sasa) sasa[i] = a;
    }
    if (cnt) cnt->nn_sum = start[n];
    ret = ORACLE_OK;
done:
    free(R);
    free(arc);
    free(start);
    free(idx);
    return ret;
}

int oracle_lee_richards(const double *xyz, const double *radii, int n,
                        double probe, int n_slices, double *sasa)
{
    return lr_run(xyz, radii, n, probe, n_slices, sasa, NULL);
}

int oracle_lr_work_stats(const double *xyz, const double *radii, int n,
                         double probe, int n_slices, double *stats)
{
    lr_counters c = {0, 0, 0, 0, 0, 0};
    int ret = lr_run(xyz, radii, n, probe, n_slices, NULL, &c);
    stats[0] = c.tests;
    stats[1] = c.zpass;
    stats[2] = c.arcs;
    stats[3] = c.buried;
    stats[4] = c.max_arcs;
    stats[5] = c.nn_sum;
    return ret;
}

double oracle_total(const double *sasa, int n)
{
    double t = 0; /* ref: src/freesasa.c:113-116 */
    int i;
    for (i = 0; i < n; ++i) t += sasa[i];
    return t;
}
