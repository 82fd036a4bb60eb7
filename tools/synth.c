/*
 * synth.c — seeded synthetic structure generators for bench.py and the tests
 * (workload tooling; not part of the SASA product path and not part of the oracle).
 *
 * Generators follow SURVEY.md §8(d):
 *   coil(n, seed)    self-avoiding random walk, bond 1.5 A, no atom closer than 2.8 A
 *                    to any atom more than 3 bonds back (hash-grid check, back-track 5
 *                    atoms after 50 failed tries)      -> ~17-23 neighbors/atom
 *   globule(n, seed) simple-cubic lattice, spacing `a` (2.6 A = protein-like packing),
 *                    every coordinate jittered uniformly by +-0.45 A -> ~47 neighbors/atom
 * Radii are drawn uniformly from the ProtOr radius set {1.42,1.46,1.61,1.64,1.76,1.88}.
 * RNG: splitmix64-seeded xoshiro256**, so a (generator, n, seed) triple is the same
 * bytes on every machine.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct {
    uint64_t s[4];
} rng_t;

static uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static void rng_seed(rng_t *r, uint64_t seed)
{
    int i;
    for (i = 0; i < 4; ++i) r->s[i] = splitmix64(&seed);
}

static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static uint64_t rng_next(rng_t *r)
{
    uint64_t *s = r->s, result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return result;
}

static double rng_uniform(rng_t *r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }

static const double PROTOR_R[6] = {1.42, 1.46, 1.61, 1.64, 1.76, 1.88};

static void fill_radii(rng_t *rng, double *radii, int n)
{
    int i;
    for (i = 0; i < n; ++i) radii[i] = PROTOR_R[rng_next(rng) % 6];
}

/* ------------------------------------------------------------------ coil */

#define COIL_BOND 1.5
#define COIL_EXCL 2.8
#define COIL_SKIP 3
#define HASH_BITS 18

typedef struct {
    int *head; /* 1<<HASH_BITS buckets -> atom index, -1 = empty */
    int *next; /* per atom chain */
} hashgrid_t;

static uint32_t cell_hash(int ix, int iy, int iz)
{
    uint32_t h = (uint32_t)ix * 73856093u ^ (uint32_t)iy * 19349663u ^ (uint32_t)iz * 83492791u;
    return h & ((1u << HASH_BITS) - 1);
}

static int cell_of(double v) { return (int)floor(v / COIL_EXCL); }

static int coil_clash(const hashgrid_t *g, const double *xyz, int n_placed, const double *p)
{
    int cx = cell_of(p[0]), cy = cell_of(p[1]), cz = cell_of(p[2]), dx, dy, dz, j;
    for (dx = -1; dx <= 1; ++dx)
        for (dy = -1; dy <= 1; ++dy)
            for (dz = -1; dz <= 1; ++dz)
                for (j = g->head[cell_hash(cx + dx, cy + dy, cz + dz)]; j >= 0; j = g->next[j]) {
                    double ex, ey, ez;
                    if (j >= n_placed - COIL_SKIP) continue; /* bonded neighborhood */
                    ex = xyz[3 * j] - p[0];
                    ey = xyz[3 * j + 1] - p[1];
                    ez = xyz[3 * j + 2] - p[2];
                    if (ex * ex + ey * ey + ez * ez < COIL_EXCL * COIL_EXCL) return 1;
                }
    return 0;
}

static void grid_insert(hashgrid_t *g, const double *xyz, int i)
{
    uint32_t h = cell_hash(cell_of(xyz[3 * i]), cell_of(xyz[3 * i + 1]), cell_of(xyz[3 * i + 2]));
    g->next[i] = g->head[h];
    g->head[h] = i;
}

static void grid_remove(hashgrid_t *g, const double *xyz, int i)
{
    /* atoms are removed in LIFO order, so atom i is the head of its bucket */
    uint32_t h = cell_hash(cell_of(xyz[3 * i]), cell_of(xyz[3 * i + 1]), cell_of(xyz[3 * i + 2]));
    g->head[h] = g->next[i];
}

int synth_coil(int n, uint64_t seed, double *xyz, double *radii)
{
    rng_t rng;
    hashgrid_t g;
    int placed = 1, fails = 0, i;

    if (n <= 0) return -1;
    rng_seed(&rng, seed);
    g.head = malloc(sizeof(int) << HASH_BITS);
    g.next = malloc(sizeof(int) * (size_t)n);
    if (!g.head || !g.next) {
        free(g.head);
        free(g.next);
        return -1;
    }
    for (i = 0; i < (1 << HASH_BITS); ++i) g.head[i] = -1;
    xyz[0] = xyz[1] = xyz[2] = 0;
    grid_insert(&g, xyz, 0);

    while (placed < n) {
        double u = 2 * rng_uniform(&rng) - 1, phi = 2 * M_PI * rng_uniform(&rng);
        double s = sqrt(1 - u * u), p[3];
        p[0] = xyz[3 * (placed - 1)] + COIL_BOND * s * cos(phi);
        p[1] = xyz[3 * (placed - 1) + 1] + COIL_BOND * s * sin(phi);
        p[2] = xyz[3 * (placed - 1) + 2] + COIL_BOND * u;
        if (coil_clash(&g, xyz, placed, p)) {
            if (++fails >= 50) { /* dead end: back-track 5 atoms */
                int back = placed > 5 ? 5 : placed - 1;
                for (i = 0; i < back; ++i) grid_remove(&g, xyz, --placed);
                fails = 0;
            }
            continue;
        }
        memcpy(xyz + 3 * placed, p, sizeof p);
        grid_insert(&g, xyz, placed);
        ++placed;
        fails = 0;
    }
    fill_radii(&rng, radii, n);
    free(g.head);
    free(g.next);
    return 0;
}

/* ------------------------------------------------------------------ globule */

int synth_globule(int n, uint64_t seed, double spacing, double *xyz, double *radii)
{
    rng_t rng;
    int m = 1, i;

    if (n <= 0 || !(spacing > 0)) return -1;
    rng_seed(&rng, seed);
    while ((long)m * m * m < n) ++m;
    for (i = 0; i < n; ++i) {
        int ix = i % m, iy = (i / m) % m, iz = i / (m * m);
        xyz[3 * i] = ix * spacing + 0.9 * (rng_uniform(&rng) - 0.5);
        xyz[3 * i + 1] = iy * spacing + 0.9 * (rng_uniform(&rng) - 0.5);
        xyz[3 * i + 2] = iz * spacing + 0.9 * (rng_uniform(&rng) - 0.5);
    }
    fill_radii(&rng, radii, n);
    return 0;
}

/* A batch of n_structs coils of n atoms each, seeds seed0 + k, written back to back. */
int synth_coil_batch(int n_structs, int n, uint64_t seed0, double *xyz, double *radii)
{
    int k;
    for (k = 0; k < n_structs; ++k)
        if (synth_coil(n, seed0 + (uint64_t)k, xyz + 3 * (size_t)k * n, radii + (size_t)k * n)) return -1;
    return 0;
}

/* Trajectory frame: base coordinates + uniform jitter in [-amp, amp] (seeded per frame). */
int synth_jitter(const double *base, int n, uint64_t seed, double amp, double *out)
{
    rng_t rng;
    int i;
    rng_seed(&rng, seed);
    for (i = 0; i < 3 * n; ++i) out[i] = base[i] + amp * (2 * rng_uniform(&rng) - 1);
    return 0;
}
