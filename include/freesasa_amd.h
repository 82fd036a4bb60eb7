/*
 * freesasa_amd.h — the drop-in C boundary of the MI355X SASA engine.
 *
 * This header declares, with the reference's names, layouts and semantics, exactly the
 * part of FreeSASA's public API that IS the per-atom SASA hot path.  A program compiled
 * against the reference's freesasa.h and using only these entry points can be linked
 * against libfreesasa_amd.so unchanged; each declaration cites the reference interface
 * it replaces (paths relative to the reference tree).
 *
 * Not declared here (they stay with the reference; see INTEGRATION.md for how the two
 * are combined): structure/PDB/mmCIF input, classifiers, selections, result trees and
 * output writers.
 */
#ifndef FREESASA_AMD_H
#define FREESASA_AMD_H

#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* replaces src/freesasa.h:89-92 — same enumerators, same values */
enum freesasa_algorithm {
    FREESASA_LEE_RICHARDS, /* z-slice arc integration */
    FREESASA_SHRAKE_RUPLEY /* test-point occlusion */
};
#ifndef __cplusplus
typedef enum freesasa_algorithm freesasa_algorithm;
#endif

/* replaces src/freesasa.h:103-108 */
enum freesasa_verbosity {
    FREESASA_V_NORMAL,
    FREESASA_V_NOWARNINGS,
    FREESASA_V_SILENT,
    FREESASA_V_DEBUG
};
#ifndef __cplusplus
typedef enum freesasa_verbosity freesasa_verbosity;
#endif

/* replaces src/freesasa.h:115-118 */
#define FREESASA_DEF_ALGORITHM FREESASA_LEE_RICHARDS
#define FREESASA_DEF_PROBE_RADIUS 1.4
#define FREESASA_DEF_SR_N 100
#define FREESASA_DEF_LR_N 20

/* replaces src/freesasa.h:143 (value 2, as in a thread-enabled reference build; the GPU
   path accepts and ignores 1..16 and rejects > 16 exactly like src/sasa_lr.c:177) */
extern const int FREESASA_DEF_NUMBER_THREADS;

/* replaces src/freesasa.h:151-155 */
enum freesasa_error_codes {
    FREESASA_SUCCESS = 0,
    FREESASA_FAIL = -1,
    FREESASA_WARN = -2
};

/* replaces src/freesasa.h:232-238 — 32 bytes, identical field order */
struct freesasa_parameters {
    freesasa_algorithm alg;
    double probe_radius;
    int shrake_rupley_n_points;
    int lee_richards_n_slices;
    int n_threads;
};
#ifndef __cplusplus
typedef struct freesasa_parameters freesasa_parameters;
#endif

/* replaces src/freesasa.h:248, src/freesasa.c:38-43 */
extern const freesasa_parameters freesasa_default_parameters;

/* opaque; owned by the reference's structure.c (src/freesasa.h:259) */
typedef struct freesasa_structure freesasa_structure;

/* replaces src/freesasa.h:267-272 — 56 bytes; sasa is malloc()ed, n_atoms doubles */
struct freesasa_result {
    double total;
    double *sasa;
    int n_atoms;
    freesasa_parameters parameters;
};
#ifndef __cplusplus
typedef struct freesasa_result freesasa_result;
#endif

/* replaces src/freesasa.h:475-479, src/freesasa.c:122-142.
   xyz = x1,y1,z1,...; radii WITHOUT probe; n > 0 (asserted, as in the reference);
   parameters == NULL means defaults.  Returns NULL on failure (message through the
   error hooks below).  Inputs are borrowed for the duration of the call. */
freesasa_result *
freesasa_calc_coord(const double *xyz, const double *radii, int n,
                    const freesasa_parameters *parameters);

/* replaces src/freesasa.h:455-457, src/freesasa.c:144-153.  Needs the reference's
   structure accessors (freesasa_structure_xyz / freesasa_structure_radius, src/structure.c:
   1106-1111, 1390-1395); when those are not linked in it fails with a message. */
freesasa_result *
freesasa_calc_structure(const freesasa_structure *structure,
                        const freesasa_parameters *parameters);

/* replaces src/freesasa.h:527, src/freesasa.c:68-74 */
void freesasa_result_free(freesasa_result *result);

/* error-reporting conventions of the hot path, replacing src/freesasa.h:703-727 and
   src/util.c:131-141, src/log.c:12-32 in the stand-alone library */
int freesasa_set_verbosity(freesasa_verbosity v);
freesasa_verbosity freesasa_get_verbosity(void);
void freesasa_set_err_out(FILE *err);
FILE *freesasa_get_err_out(void);

/* ---- internal seam (src/freesasa_internal.h:74-103, src/coord.h:26-38) -------------
   The narrowest replacement point: the reference's freesasa_calc() (src/freesasa.c:97-107)
   calls exactly these two.  Replacing sasa_lr.o, sasa_sr.o and nb.o of libfreesasa.a by
   libfreesasa_amd_seam leaves the CLI, Python bindings and every output path untouched. */
typedef struct coord_t {
    int n;         /* number of 3-vectors */
    int is_linked; /* 1: xyz borrowed from the caller */
    double *xyz;   /* x1,y1,z1,...,xn,yn,zn */
} coord_t;

/* sasa[n] caller-allocated, fully overwritten.  Returns FREESASA_SUCCESS, FREESASA_FAIL
   (n_threads > 16, resolution <= 0, device failure) or FREESASA_WARN (n == 0, sasa
   untouched) — src/sasa_lr.c:156-216, src/sasa_sr.c:168-224. */
int freesasa_lee_richards(double *sasa, const coord_t *xyz, const double *radii,
                          const freesasa_parameters *param);
int freesasa_shrake_rupley(double *sasa, const coord_t *xyz, const double *radii,
                           const freesasa_parameters *param);

/* src/freesasa_internal.h:120-123, src/freesasa.c:76-120 */
freesasa_result *freesasa_calc(const coord_t *c, const double *radii,
                               const freesasa_parameters *parameters);

#ifdef __cplusplus
}
#endif
#endif /* FREESASA_AMD_H */
