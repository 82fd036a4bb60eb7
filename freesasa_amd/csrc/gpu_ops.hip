/*
 * gpu_ops.hip — device-side aggregates over per-atom areas (segments, atom classes, residues: what the reference's
 * result tree adds up on the host, src/node.c:717-764, src/classifier.c:830-866, src/rsa.c:14-25) and the test hooks
 * that run the integer / exact parts of the Lee-Richards kernel on their own.  Host code; kernels in gpu_kernels.hip.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdint.h>
#include <string.h>
#include <utility>
#include <vector>

#include "engine_internal.h"

extern "C" int freesasa_gpu_segment_sums_dev(freesasa_gpu_ctx *c, const double *d_sasa, const int64_t *seg,
                                             int n_segs, double *d_out)
{
    if (!c) return -1;
    return guarded_ctx(c, [&]() -> int {
    c->err[0] = 0;
    if (!d_sasa || !seg || !d_out || n_segs <= 0) return ctx_fail(c, "bad argument");
    for (int k = 0; k < n_segs; ++k)
        if (seg[k + 1] < seg[k]) return ctx_fail(c, "segment offsets must be non-decreasing");
    HIP_TRY(c, hipSetDevice(c->device));
    if (ensure(c, c->seg, sizeof(int64_t) * ((size_t)n_segs + 1))) return -1;
    HIP_TRY(c, hipMemcpyAsync(c->seg.p, seg, sizeof(int64_t) * ((size_t)n_segs + 1), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, kl_segment_sums(d_sasa, (const int64_t *)c->seg.p, n_segs, seg[n_segs] - seg[0] < (int64_t)64 * n_segs, d_out, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
    });
}

extern "C" int freesasa_gpu_class_sums_dev(freesasa_gpu_ctx *c, const double *d_sasa, const unsigned char *d_class,
                                           const int64_t *offsets, int n_structs, double *d_out)
{
    if (!c) return -1;
    return guarded_ctx(c, [&]() -> int {
    c->err[0] = 0;
    if (!d_sasa || !d_class || !offsets || !d_out || n_structs <= 0) return ctx_fail(c, "bad argument");
    for (int k = 0; k < n_structs; ++k)
        if (offsets[k + 1] < offsets[k]) return ctx_fail(c, "structure offsets must be non-decreasing");
    HIP_TRY(c, hipSetDevice(c->device));
    if (ensure(c, c->seg, sizeof(int64_t) * ((size_t)n_structs + 1))) return -1;
    HIP_TRY(c, hipMemcpyAsync(c->seg.p, offsets, sizeof(int64_t) * ((size_t)n_structs + 1), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, kl_class_sums(d_sasa, d_class, (const int64_t *)c->seg.p, n_structs, d_out, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
    });
}

extern "C" int freesasa_gpu_residue_areas_dev(freesasa_gpu_ctx *c, const double *d_sasa, const unsigned char *d_class,
                                              const unsigned char *d_backbone, const int64_t *res_first, int n_res,
                                              const short *ref_row, const double *ref_table, int ref_rows,
                                              double *d_abs, double *d_rel)
{
    if (!c) return -1;
    return guarded_ctx(c, [&]() -> int {
    c->err[0] = 0;
    if (!d_sasa || !d_class || !d_backbone || !res_first || !d_abs || n_res <= 0) return ctx_fail(c, "bad argument");
    if (d_rel && (!ref_row || !ref_table || ref_rows <= 0)) return ctx_fail(c, "relative areas need the reference rows and table");
    for (int k = 0; k < n_res; ++k) {
        if (res_first[k + 1] < res_first[k]) return ctx_fail(c, "residue offsets must be non-decreasing");
        if (d_rel && ref_row[k] >= ref_rows) return ctx_fail(c, "reference row out of range");
    }
    HIP_TRY(c, hipSetDevice(c->device));
    /* one staging buffer: offsets, reference table, reference rows */
    const size_t b_first = sizeof(int64_t) * ((size_t)n_res + 1);
    const size_t b_table = d_rel ? sizeof(double) * 5 * (size_t)ref_rows : 0;
    const size_t b_rows = d_rel ? sizeof(short) * (size_t)n_res : 0;
    if (ensure(c, c->seg, b_first + b_table + b_rows)) return -1;
    char *base = (char *)c->seg.p;
    HIP_TRY(c, hipMemcpyAsync(base, res_first, b_first, hipMemcpyHostToDevice, c->stream));
    if (d_rel) {
        HIP_TRY(c, hipMemcpyAsync(base + b_first, ref_table, b_table, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(base + b_first + b_table, ref_row, b_rows, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, kl_residue_areas(d_sasa, d_class, d_backbone, (const int64_t *)base, d_rel ? (const short *)(base + b_first + b_table) : nullptr,
                                d_rel ? (const double *)(base + b_first) : nullptr, d_abs, d_rel, n_res, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
    });
}

/* ------------------------------------------------------------------ test hooks of the L&R kernel's integer parts */

/* The neighbor sets the Lee-Richards kernel finds (ref: freesasa_nb_new with radii + probe, src/nb.c:524-557, what
 * tests/test_nb.c checks): per atom, in original order, the number of neighbors and, if d_nb is given, the first
 * nb_cap of them (original atom indices, in order of discovery).  Device pointers; d_nb may be NULL. */
extern "C" int freesasa_gpu_lr_neighbors_dev(freesasa_gpu_ctx *c, const double *d_xyz, const double *d_radii, const int64_t *offsets,
                                             int n_structs, double probe, int *d_nn, int *d_nb, int nb_cap)
{
    if (!c) return -1;
    return guarded_ctx(c, [&]() -> int {
    if (!d_nn || (d_nb && nb_cap <= 0)) return ctx_fail(c, "bad argument");
    const int64_t n = offsets && n_structs > 0 ? offsets[n_structs] : 0;
    if (n <= 0) return ctx_fail(c, "empty batch");
    if (freesasa_gpu_wait(c)) return -1; /* (batches submitted asynchronously come first, as for every synchronous entry) */
    if (hipSetDevice(c->device) != hipSuccess || ensure(c, c->h_sasa, 8 * (size_t)n)) return -1;
    struct Hooks { /* (taken off the context on every way out) */
        freesasa_gpu_ctx *c;
        ~Hooks() { c->dbg_nn = c->dbg_nb = nullptr; c->dbg_cap = 0; }
    } hooks{c};
    c->dbg_nn = d_nn; c->dbg_nb = d_nb; c->dbg_cap = nb_cap;
    return run_batch(c, true, d_xyz, d_radii, offsets, n_structs, probe, 20, nullptr, (double *)c->h_sasa.p, nullptr, nullptr);
    });
}

/* The exposed arc length of n_sets sets of arcs (start, end pairs in [0, 2 pi], set k = arcs first[k] .. first[k+1]),
 * computed on the device by the arc union and sweep of the Lee-Richards kernel (ref: exposed_arc_length,
 * src/sasa_lr.c:389-408, and its KATs :455-475).  Host arrays; at most 64 sets. */
extern "C" int freesasa_gpu_arc_union_dev(freesasa_gpu_ctx *c, const double *arcs, const int *first, int n_sets, double *out)
{
    if (!c) return -1;
    return guarded_ctx(c, [&]() -> int {
    if (!arcs || !first || !out || n_sets <= 0 || n_sets > 64) return ctx_fail(c, "bad argument");
    const int total = first[n_sets];
    /* the arc pass feeds the union in the order of the arcs' mid-points (the neighbors' directions) */
    std::vector<double> sorted(2 * (size_t)(total > 0 ? total : 1));
    for (int k = 0; k < n_sets; ++k) {
        std::vector<std::pair<double, double>> v;
        for (int i = first[k]; i < first[k + 1]; ++i) v.emplace_back(arcs[2 * i], arcs[2 * i + 1]);
        std::stable_sort(v.begin(), v.end(), [](const std::pair<double, double> &x, const std::pair<double, double> &y) {
            return x.first + x.second < y.first + y.second; });
        for (size_t i = 0; i < v.size(); ++i) { sorted[2 * (first[k] + i)] = v[i].first; sorted[2 * (first[k] + i) + 1] = v[i].second; }
    }
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t b_arcs = sizeof(double) * sorted.size(), b_first = sizeof(int) * ((size_t)n_sets + 1);
    if (ensure(c, c->seg, b_arcs + b_first + 8 * 64 + 64)) return -1;
    char *base = (char *)c->seg.p;
    HIP_TRY(c, hipMemcpyAsync(base, sorted.data(), b_arcs, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(base + b_arcs, first, b_first, hipMemcpyHostToDevice, c->stream));
    double *d_out = (double *)(base + ((b_arcs + b_first + 15) & ~(size_t)15));
    HIP_TRY(c, kl_arc_kat((const double *)base, (const int *)(base + b_arcs), n_sets, d_out, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out, d_out, sizeof(double) * (size_t)n_sets, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
    });
}

