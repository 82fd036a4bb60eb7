/*
 * api.c — the stand-alone part of the drop-in boundary: what the reference keeps in
 * src/freesasa.c:31-153 (defaults, result objects, freesasa_calc and its public wrappers),
 * the two coord_t helpers the boundary needs (src/coord.c:12-32, 72-88) and the
 * error-reporting hooks (src/util.c:37-141, src/log.c:12-32).  In a drop-in build of the
 * reference these all stay the reference's own objects and only seam.c + gpu_engine.hip
 * are linked in.
 */
#include <assert.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/freesasa_amd.h"

/* ------------------------------------------------------------- error hooks */

static const char *lib_name = "freesasa"; /* src/util.c:12-16 */
static FILE *errlog = NULL;               /* src/util.c:18 */
static freesasa_verbosity verbosity = FREESASA_V_NORMAL; /* src/log.c:12 */

int freesasa_set_verbosity(freesasa_verbosity v)
{
    if (v == FREESASA_V_NORMAL || v == FREESASA_V_NOWARNINGS ||
        v == FREESASA_V_SILENT || v == FREESASA_V_DEBUG) {
        verbosity = v;
        return FREESASA_SUCCESS;
    }
    return FREESASA_WARN;
}

freesasa_verbosity freesasa_get_verbosity(void) { return verbosity; }

void freesasa_set_err_out(FILE *fp)
{
    assert(fp);
    errlog = fp;
}

FILE *freesasa_get_err_out(void) { return errlog; }

int freesasa_fail_wloc(const char *file, int line, const char *format, ...)
{
    FILE *fp = errlog ? errlog : stderr;
    va_list arg;

    if (verbosity == FREESASA_V_SILENT) return FREESASA_FAIL;
    fprintf(fp, "%s:%s:%d: error: ", lib_name, file, line);
    va_start(arg, format);
    vfprintf(fp, format, arg);
    va_end(arg);
    fputc('\n', fp);
    fflush(fp);
    return FREESASA_FAIL;
}

int freesasa_warn(const char *format, ...)
{
    FILE *fp = errlog ? errlog : stderr;
    va_list arg;

    if (verbosity == FREESASA_V_NOWARNINGS || verbosity == FREESASA_V_SILENT) return FREESASA_WARN;
    fprintf(fp, "%s: warning: ", lib_name);
    va_start(arg, format);
    vfprintf(fp, format, arg);
    va_end(arg);
    fputc('\n', fp);
    fflush(fp);
    return FREESASA_WARN;
}

#define fail_msg(...) freesasa_fail_wloc(__FILE__, __LINE__, __VA_ARGS__)
#define mem_fail() freesasa_fail_wloc(__FILE__, __LINE__, "Out of memory")

/* ------------------------------------------------------------- defaults, results */

const int FREESASA_DEF_NUMBER_THREADS = 2; /* src/freesasa.c:31-36 with threads enabled */

const freesasa_parameters freesasa_default_parameters = {
    FREESASA_DEF_ALGORITHM,
    FREESASA_DEF_PROBE_RADIUS,
    FREESASA_DEF_SR_N,
    FREESASA_DEF_LR_N,
    2};

void freesasa_result_free(freesasa_result *r)
{
    if (r) {
        free(r->sasa);
        free(r);
    }
}

static freesasa_result *result_new(int n)
{
    freesasa_result *r = malloc(sizeof *r);
    if (r == NULL) {
        mem_fail();
        return NULL;
    }
    /* libc malloc: the caller releases it with freesasa_result_free (src/freesasa.c:45-74) */
    r->sasa = malloc(sizeof(double) * (size_t)n);
    if (r->sasa == NULL) {
        mem_fail();
        free(r);
        return NULL;
    }
    r->n_atoms = n;
    return r;
}

/* ------------------------------------------------------------- calc */

freesasa_result *freesasa_calc(const coord_t *c, const double *radii,
                               const freesasa_parameters *parameters)
{
    freesasa_result *result;
    int ret = FREESASA_SUCCESS, i;

    assert(c);
    assert(radii);

    result = result_new(c->n);
    if (result == NULL) {
        fail_msg("");
        return NULL;
    }
    if (parameters == NULL) parameters = &freesasa_default_parameters;

    switch (parameters->alg) { /* src/freesasa.c:97-107 */
    case FREESASA_SHRAKE_RUPLEY:
        ret = freesasa_shrake_rupley(result->sasa, c, radii, parameters);
        break;
    case FREESASA_LEE_RICHARDS:
        ret = freesasa_lee_richards(result->sasa, c, radii, parameters);
        break;
    default:
        assert(0);
        break;
    }
    if (ret == FREESASA_FAIL) {
        freesasa_result_free(result);
        return NULL;
    }
    /* sequential host sum in atom order, so equal per-atom values give a bit-equal total
       (src/freesasa.c:113-116) */
    result->total = 0;
    for (i = 0; i < c->n; ++i) result->total += result->sasa[i];
    result->parameters = *parameters;
    return result;
}

freesasa_result *freesasa_calc_coord(const double *xyz, const double *radii, int n,
                                     const freesasa_parameters *parameters)
{
    coord_t linked; /* zero-copy view of the caller's array (src/coord.c:72-88) */
    freesasa_result *result;

    assert(xyz);
    assert(radii);
    assert(n > 0);

    linked.n = n;
    linked.is_linked = 1;
    linked.xyz = (double *)xyz;
    result = freesasa_calc(&linked, radii, parameters);
    if (result == NULL) fail_msg("");
    return result;
}

/* The structure accessors live in the reference's structure.c.  Weak references: resolved
   when this library is combined with the reference, NULL otherwise. */
extern const coord_t *freesasa_structure_xyz(const freesasa_structure *) __attribute__((weak));
extern const double *freesasa_structure_radius(const freesasa_structure *) __attribute__((weak));

freesasa_result *freesasa_calc_structure(const freesasa_structure *structure,
                                         const freesasa_parameters *parameters)
{
    assert(structure);
    if (!freesasa_structure_xyz || !freesasa_structure_radius) {
        fail_msg("freesasa_calc_structure() needs the reference's structure module "
                 "(freesasa_structure_xyz/_radius); use freesasa_calc_coord() or link it in");
        return NULL;
    }
    return freesasa_calc(freesasa_structure_xyz(structure),
                         freesasa_structure_radius(structure), parameters);
}
