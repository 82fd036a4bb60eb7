/*
 * api.c — the stand-alone half of the drop-in boundary (include/freesasa_amd.h): result objects, the three public
 * calc entry points over the seam (seam.c), the default parameters and the message hooks.  What the caller can
 * observe is fixed by the reference (src/freesasa.c:31-153, src/util.c:89-141, src/log.c:12-32): NULL on failure with
 * a message on the error stream, `sasa` and the result released separately by freesasa_result_free, the total as a
 * sequential sum in atom order; how it is done here is this project's own - one message writer behind both hooks,
 * one function that owns a result from allocation to hand-over with a single way out on failure.
 * A drop-in build of the reference does not contain this file: there the reference's own freesasa.o / util.o / log.o
 * stay and only seam.c + the engine are linked in.
 */
#include <assert.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/freesasa_amd.h"
#include "hostfault.h"

/* ------------------------------------------------------------- messages */

static struct {
    FILE *stream;               /* NULL: stderr (ref: src/util.c:18) */
    freesasa_verbosity level;   /* ref: src/log.c:12 */
} g_log = {NULL, FREESASA_V_NORMAL};

/* one line "freesasa:<where> <kind>: <text>" on the error stream; the return code is the caller's to pass on */
static int emit(int code, const char *kind, const char *file, int line, const char *format, va_list ap)
{
    const int quiet = code == FREESASA_FAIL ? g_log.level == FREESASA_V_SILENT
                                            : (g_log.level == FREESASA_V_SILENT || g_log.level == FREESASA_V_NOWARNINGS);
    if (!quiet) {
        FILE *fp = g_log.stream ? g_log.stream : stderr;
        if (file) fprintf(fp, "freesasa:%s:%d: %s: ", file, line, kind); /* ref: src/util.c:89-113 */
        else fprintf(fp, "freesasa: %s: ", kind);                        /* ref: src/util.c:115-129 */
        vfprintf(fp, format, ap);
        fputc('\n', fp);
        fflush(fp);
    }
    return code;
}

int freesasa_fail_wloc(const char *file, int line, const char *format, ...)
{
    va_list ap;
    va_start(ap, format);
    const int rc = emit(FREESASA_FAIL, "error", file, line, format, ap);
    va_end(ap);
    return rc;
}

int freesasa_warn(const char *format, ...)
{
    va_list ap;
    va_start(ap, format);
    const int rc = emit(FREESASA_WARN, "warning", NULL, 0, format, ap);
    va_end(ap);
    return rc;
}

int freesasa_set_verbosity(freesasa_verbosity v)
{
    switch (v) {
    case FREESASA_V_NORMAL: case FREESASA_V_NOWARNINGS: case FREESASA_V_SILENT: case FREESASA_V_DEBUG:
        g_log.level = v;
        return FREESASA_SUCCESS;
    }
    return FREESASA_WARN; /* ref: src/log.c:14-26 */
}
freesasa_verbosity freesasa_get_verbosity(void) { return g_log.level; }
void freesasa_set_err_out(FILE *fp) { assert(fp); g_log.stream = fp; }
FILE *freesasa_get_err_out(void) { return g_log.stream; }

#define FAIL_HERE(...) freesasa_fail_wloc(__FILE__, __LINE__, __VA_ARGS__)

/* ------------------------------------------------------------- defaults */

const int FREESASA_DEF_NUMBER_THREADS = 2; /* a thread-enabled reference build (src/freesasa.c:31-36) */
const freesasa_parameters freesasa_default_parameters = {
    FREESASA_DEF_ALGORITHM, FREESASA_DEF_PROBE_RADIUS, FREESASA_DEF_SR_N, FREESASA_DEF_LR_N, 2};

/* ------------------------------------------------------------- results */

void freesasa_result_free(freesasa_result *r)
{
    if (!r) return;
    free(r->sasa); /* both libc malloc: a caller may also release them itself, as with the reference (src/freesasa.c:68-74) */
    free(r);
}

/* A result for `c` computed with `p` (never NULL here), or NULL with the reason on the error stream: the one place
   that allocates, dispatches on the algorithm (src/freesasa.c:97-107), sums and hands over. */
static freesasa_result *compute(const coord_t *c, const double *radii, const freesasa_parameters *p)
{
    freesasa_result *r = hf_calloc(1, sizeof *r);
    const char *why = "Out of memory";
    if (r && (r->sasa = hf_malloc(sizeof(double) * (size_t)(c->n > 0 ? c->n : 1))) != NULL) {
        int rc;
        assert(p->alg == FREESASA_SHRAKE_RUPLEY || p->alg == FREESASA_LEE_RICHARDS);
        rc = p->alg == FREESASA_SHRAKE_RUPLEY ? freesasa_shrake_rupley(r->sasa, c, radii, p)
                                              : freesasa_lee_richards(r->sasa, c, radii, p);
        if (rc != FREESASA_FAIL) { /* (FREESASA_WARN, n == 0: an empty result, as the reference returns) */
            double total = 0;
            int i;
            for (i = 0; i < c->n; ++i) total += r->sasa[i]; /* in atom order: equal areas, equal total (src/freesasa.c:113-116) */
            r->total = total;
            r->n_atoms = c->n;
            r->parameters = *p;
            return r;
        }
        why = ""; /* the seam has said what went wrong; the reference adds an empty line per level (src/freesasa.c:91, :109) */
    }
    FAIL_HERE("%s", why);
    freesasa_result_free(r);
    return NULL;
}

freesasa_result *freesasa_calc(const coord_t *c, const double *radii, const freesasa_parameters *parameters)
{
    assert(c);
    assert(radii);
    return compute(c, radii, parameters ? parameters : &freesasa_default_parameters);
}

freesasa_result *freesasa_calc_coord(const double *xyz, const double *radii, int n, const freesasa_parameters *parameters)
{
    coord_t view; /* the caller's array, borrowed for the call (what src/coord.c:72-88 builds on the heap) */
    freesasa_result *r;
    assert(xyz);
    assert(radii);
    assert(n > 0);
    view.n = n; view.is_linked = 1; view.xyz = (double *)xyz;
    r = freesasa_calc(&view, radii, parameters);
    if (!r) FAIL_HERE("");
    return r;
}

/* The structure accessors live in the reference's structure.c.  Weak references: resolved when this library is
   combined with the reference, NULL otherwise. */
extern const coord_t *freesasa_structure_xyz(const freesasa_structure *) __attribute__((weak));
extern const double *freesasa_structure_radius(const freesasa_structure *) __attribute__((weak));

freesasa_result *freesasa_calc_structure(const freesasa_structure *structure, const freesasa_parameters *parameters)
{
    assert(structure);
    if (!freesasa_structure_xyz || !freesasa_structure_radius) {
        FAIL_HERE("freesasa_calc_structure() needs the reference's structure module "
                  "(freesasa_structure_xyz/_radius); use freesasa_calc_coord() or link it in");
        return NULL;
    }
    return freesasa_calc(freesasa_structure_xyz(structure), freesasa_structure_radius(structure), parameters);
}
