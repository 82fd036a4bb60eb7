/*
 * seam.c — the reference's internal algorithm seam, backed by the MI355X engine.
 *
 * Implements, with the reference's exact signatures and error behaviour,
 *     int freesasa_lee_richards (double *sasa, const coord_t*, const double *radii, const freesasa_parameters*)
 *     int freesasa_shrake_rupley(double *sasa, const coord_t*, const double *radii, const freesasa_parameters*)
 * (src/freesasa_internal.h:74-103; bodies in src/sasa_lr.c:156-216 and src/sasa_sr.c:168-224).
 * These two are all that the reference's freesasa_calc() (src/freesasa.c:97-107) needs from
 * sasa_lr.o, sasa_sr.o and nb.o; this object + gpu_engine.o replace those three objects in
 * a drop-in build.  Error hooks (freesasa_fail_wloc, freesasa_warn) are
 * resolved from the reference's util.o there, and from api.c in the stand-alone library.
 *
 * n_threads only sizes the reference's pthread pool; results do not depend on it (the
 * reference's own test, tests/test_freesasa.c:404-429).  It is validated exactly as the
 * reference validates it and otherwise ignored.
 */
#include <assert.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/freesasa_amd.h"
#include "../../include/freesasa_gpu.h"

#define SEAM_MAX_THREADS 16 /* src/sasa_lr.c:17, src/sasa_sr.c:16 */

/* src/freesasa_internal.h:29-32, src/util.c:62-113 */
int freesasa_fail_wloc(const char *file, int line, const char *format, ...);
int freesasa_warn(const char *format, ...);
#define fail_msg(...) freesasa_fail_wloc(__FILE__, __LINE__, __VA_ARGS__)

static int run_one(int alg, double *sasa, const coord_t *xyz, const double *radii,
                   double probe, int resolution)
{
    char msg[512];
    const int64_t offsets[2] = {0, xyz->n};
    if (freesasa_gpu_calc_batch(xyz->xyz, radii, offsets, 1, alg, probe, resolution,
                                sasa, NULL, NULL, -1, msg, (int)sizeof msg))
        return fail_msg("%s", msg);
    return FREESASA_SUCCESS;
}

int freesasa_lee_richards(double *sasa, const coord_t *xyz, const double *radii,
                          const freesasa_parameters *param)
{
    int n_atoms, n_threads, resolution;

    assert(sasa);
    assert(xyz);
    assert(radii);

    if (param == NULL) param = &freesasa_default_parameters; /* src/sasa_lr.c:169 */
    n_atoms = xyz->n;
    n_threads = param->n_threads;
    resolution = param->lee_richards_n_slices;

    if (n_threads > SEAM_MAX_THREADS) /* src/sasa_lr.c:177-179 */
        return fail_msg("L&R does not support more than %d threads", SEAM_MAX_THREADS);
    if (resolution <= 0) /* src/sasa_lr.c:181-183 */
        return fail_msg("%d slices per atom invalid resolution in L&R, must be > 0\n", resolution);
    if (n_atoms == 0) /* src/sasa_lr.c:185-187: sasa is left untouched */
        return freesasa_warn("in %s(): empty coordinates", __func__);
    if (n_threads > n_atoms) /* src/sasa_lr.c:189-193 */
        freesasa_warn("no sense in having more threads than atoms, only using %d threads", n_atoms);

    return run_one(FREESASA_LEE_RICHARDS, sasa, xyz, radii, param->probe_radius, resolution);
}

int freesasa_shrake_rupley(double *sasa, const coord_t *xyz, const double *radii,
                           const freesasa_parameters *param)
{
    int n_atoms, n_threads, resolution;

    assert(sasa);
    assert(xyz);
    assert(radii);

    /* the reference dereferences param before this check (src/sasa_sr.c:173-181); here NULL
       simply means defaults, as documented in src/freesasa_internal.h:66-67 */
    if (param == NULL) param = &freesasa_default_parameters;
    n_atoms = xyz->n;
    n_threads = param->n_threads;
    resolution = param->shrake_rupley_n_points;

    if (n_threads > SEAM_MAX_THREADS) /* src/sasa_sr.c:188-190 */
        return fail_msg("S&R does not support more than %d threads", SEAM_MAX_THREADS);
    if (resolution <= 0) /* src/sasa_sr.c:191-193 */
        return fail_msg("%d test points invalid resolution in S&R, must be > 0\n", resolution);
    if (n_atoms == 0) /* src/sasa_sr.c:194 */
        return freesasa_warn("in %s(): empty coordinates", __func__);
    if (n_threads > n_atoms) /* src/sasa_sr.c:195-199 */
        freesasa_warn("no sense in having more threads than atoms, only using %d threads", n_atoms);

    return run_one(FREESASA_SHRAKE_RUPLEY, sasa, xyz, radii, param->probe_radius, resolution);
}
