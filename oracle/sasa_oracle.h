/*
 * sasa_oracle.h — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's per-atom SASA hot path
 * (/root/reference/src/nb.c, sasa_sr.c, sasa_lr.c, the calc part of freesasa.c).
 * It is the checker for the HIP path.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  Nothing under freesasa_amd/ links,
 * includes or calls it; the product library fails loudly when no GPU is present.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks it
 *   - bit-for-bit against the real reference built by oracle/Makefile
 *     (oracle/_ref/libfreesasa_ref.so) when that library is present, and
 *   - against the reference's own golden numbers (tests/test_freesasa.c:155-200,
 *     305-327, 441-451; tests/data/1ubq.B.pdb; src/sasa_lr.c:455-475) through the
 *     committed fixtures under tests/golden/.
 */
#ifndef SASA_ORACLE_H
#define SASA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_OK 0
#define ORACLE_FAIL (-1)

/* Golden-section spiral unit test points, tp[3*N] (ref: src/sasa_sr.c:56-90). */
void oracle_test_points(int n_points, double *tp);

/* Union sweep over buried arcs; arc = n (start,end) pairs, sorted in place
 * (ref: src/sasa_lr.c:367-408). */
double oracle_exposed_arc_length(double *arc, int n);

/* Unique neighbor sets in CSR form.  r_ext[i] already includes the probe.
 * Contact iff dx*dx+dy*dy+dz*dz < (ri+rj)^2, strict (ref: src/nb.c:481-495).
 * start[n+1] and *idx_out are malloc'd by the callee; free with free(). */
int oracle_neighbors(const double *xyz, const double *r_ext, int n,
                     int **start_out, int **idx_out);

/* Shrake-Rupley.  sasa[n] and counts[n] (exposed test points; may be NULL).
 * (ref: src/sasa_sr.c:108-166, 276-338). */
int oracle_shrake_rupley(const double *xyz, const double *radii, int n,
                         double probe, int n_points, double *sasa, int *counts);

/* Lee-Richards.  (ref: src/sasa_lr.c:104-154, 270-364). */
int oracle_lee_richards(const double *xyz, const double *radii, int n,
                        double probe, int n_slices, double *sasa);

/* Sequential total in atom order (ref: src/freesasa.c:113-116). */
double oracle_total(const double *sasa, int n);

/* Instrumentation for design decisions (not part of any parity claim):
 * per-atom work counters of the L&R loop, and the depth a "sorted-by-beta
 * component stack" would reach.  stats[0]=pair-slice tests, [1]=z-test passes,
 * [2]=arcs, [3]=buried slices, [4]=max arcs in a slice, [5]=sum of neighbors. */
int oracle_lr_work_stats(const double *xyz, const double *radii, int n,
                         double probe, int n_slices, double *stats);

#ifdef __cplusplus
}
#endif
#endif
