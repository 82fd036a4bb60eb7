/*
 * gpu_engine.hip — the per-device context (workspace, status words, events), the launch sequence of one batch,
 * asynchronous batches, and the device-pointer entry points of include/freesasa_gpu.h.  Host code only: the kernels
 * and their launchers live in gpu_kernels.hip (engine_internal.h).  No CPU path: without a HIP device every entry
 * point fails with a message.
 */
#include <hip/hip_runtime.h>

#include <math.h>
#include <atomic>
#include <new>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "engine_internal.h"

using namespace sasa;

/* ------------------------------------------------------------------ context */

int ctx_fail(freesasa_gpu_ctx *c, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof c->err, fmt, ap);
    va_end(ap);
    return -1;
}

static inline int *ctx_status_h(freesasa_gpu_ctx *c) { return c->pinned + c->slot * (ST_WORDS + 4); }
static inline hipEvent_t *ctx_ev(freesasa_gpu_ctx *c) { return c->evs[c->slot]; }
static int drain_pending(freesasa_gpu_ctx *c);

/* Test hook (include/freesasa_gpu.h, freesasa_gpu_test_fail_after): the n-th device / page-locked allocation from
 * now on fails, like the reference's interposed malloc (tests/tools.c:10-48) makes its n-th malloc fail. */
static std::atomic<int> g_fail_after(0);
extern "C" void freesasa_gpu_test_fail_after(int n) { g_fail_after.store(n > 0 ? n : 0); }
static bool alloc_fails()
{
    int v = g_fail_after.load();
    while (v > 0) {
        if (g_fail_after.compare_exchange_weak(v, v - 1)) return v == 1;
    }
    return false;
}
hipError_t dev_malloc(void **p, size_t bytes)
{
    if (alloc_fails()) { *p = nullptr; return hipErrorOutOfMemory; }
    return hipMalloc(p, bytes);
}
hipError_t host_malloc(void **p, size_t bytes)
{
    if (alloc_fails()) { *p = nullptr; return hipErrorOutOfMemory; }
    return hipHostMalloc(p, bytes, hipHostMallocDefault);
}

int ensure(freesasa_gpu_ctx *c, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return 0;
    /* Batches in flight may be using the buffer about to be freed: wait for the stream, nothing more.  Their verdicts
       are already on their way into their own sets of page-locked words (enqueue_tail) and are read when they are
       collected; collecting them HERE - in the middle of another batch's set-up - could redo one of them through
       run_batch and leave this batch with status words, arguments and buffers that are no longer its own. */
    if (b.p && (c->pend[0].active || c->pend[1].active)) HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (b.p) HIP_TRY(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 4 + 256; /* slack: trajectories grow/shrink a little */
    HIP_TRY(c, dev_malloc(&b.p, want));
    b.cap = want;
    return 0;
}

extern "C" int freesasa_gpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" void freesasa_gpu_ctx_destroy(freesasa_gpu_ctx *c);
extern "C" freesasa_gpu_ctx *freesasa_gpu_ctx_create(int device, void *stream)
{
    int n = freesasa_gpu_device_count();
    if (n <= 0 || device >= n) return nullptr;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    freesasa_gpu_ctx *c = new (std::nothrow) freesasa_gpu_ctx(); /* (no exception across the C boundary: engine_internal.h) */
    if (!c) return nullptr;
    c->device = device;
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            freesasa_gpu_ctx_destroy(c);
            return nullptr;
        }
        c->own_stream = true;
    }
    for (int s_ = 0; s_ < 2; ++s_) {
        for (int k = 0; k < 4; ++k)
            if (hipEventCreate(&c->evs[s_][k]) != hipSuccess) {
                freesasa_gpu_ctx_destroy(c);
                return nullptr;
            }
        if (hipEventCreateWithFlags(&c->done[s_], hipEventDisableTiming) != hipSuccess) {
            freesasa_gpu_ctx_destroy(c);
            return nullptr;
        }
    }
    if (host_malloc((void **)&c->pinned, sizeof(int) * (2 * (ST_WORDS + 4) + 4)) != hipSuccess) { /* (two sets of status words, four words for the parser's read-backs) */
        freesasa_gpu_ctx_destroy(c);
        return nullptr;
    }
    return c;
}

extern "C" void freesasa_gpu_ctx_destroy(freesasa_gpu_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf *all[] = {&c->chunk_struct, &c->chunk_begin, &c->chunk_len, &c->struct_chunk0, &c->bpart, &c->offsets, &c->grid, &c->ncells, &c->sid, &c->cell_of, &c->rank, &c->cell_start,
                     &c->blk_sums, &c->cell_tbl, &c->cell_first, &c->sq, &c->s_idx,
                     &c->status, &c->ovf_tiles, &c->ovf_tiles2, &c->ovf_atoms, &c->unit_pts, &c->captab, &c->slab, &c->seg,
                     &c->h_xyz, &c->h_radii, &c->h_sasa, &c->h_counts, &c->h_totals};
    for (DevBuf *b : all)
        if (b->p) (void)hipFree(b->p);
    for (DevBuf &b : c->parse)
        if (b.p) (void)hipFree(b.p);
    for (int s_ = 0; s_ < 2; ++s_) {
        for (int k = 0; k < 4; ++k)
            if (c->evs[s_][k]) (void)hipEventDestroy(c->evs[s_][k]);
        if (c->done[s_]) (void)hipEventDestroy(c->done[s_]);
    }
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->stage_in) (void)hipHostFree(c->stage_in);
    if (c->stage_out) (void)hipHostFree(c->stage_out);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" void freesasa_gpu_ctx_set_timing(freesasa_gpu_ctx *c, int enable) { c->timing = enable != 0; }
extern "C" void freesasa_gpu_ctx_get_stats(const freesasa_gpu_ctx *c, freesasa_gpu_stats *out) { *out = c->stats; }
extern "C" const char *freesasa_gpu_ctx_last_error(const freesasa_gpu_ctx *c) { return c ? c->err : "no context"; }

static const char *err_text(int code)
{
    switch (code) {
    case ERR_BAD_RADIUS: return "cell size 2*max(radius+probe) is not positive and finite";
    case ERR_GRID_TOO_BIG: return "cell list too large for the coordinate extent (out of memory in the reference)";
    case ERR_BAD_COORD: return "non-finite coordinate";
    case ERR_NEIGHBOR_CAP: return "an atom has more neighbors than the GPU fallback path supports (4096 per atom, 16384 per tile)";
    case ERR_STACK_CAP: return "more disjoint arcs in one slice than the GPU fallback path supports";
    default: return "unknown device-side error";
    }
}

/* ------------------------------------------------------------------ L&R, second generation */

/* run_batch_once's third outcome: the cell table was too small, redo the batch (c->cells_hint has the size) */
#define RC_RETRY 2

static int judge_status(freesasa_gpu_ctx *c, const int *status_h, long long *total_cells);
/* Status words [0, words) and K2's cell total, read back behind everything enqueued so far.  Returns 0, RC_RETRY,
 * or -1 with the batch's error text set. */
static int collect_status(freesasa_gpu_ctx *c, int n_structs, int words, long long *total_cells)
{
    hipStream_t st = c->stream;
    int *status_h = ctx_status_h(c);
    (void)words; (void)n_structs; /* (all of them: the cell total rides behind the others, sasa_kernels.h ST_CELLS) */
    HIP_TRY(c, hipMemcpyAsync(status_h, c->status.p, sizeof(int) * (size_t)ST_WORDS, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    return judge_status(c, status_h, total_cells);
}

/* the verdict of a batch's status words (already in host memory) */
static int judge_status(freesasa_gpu_ctx *c, const int *status_h, long long *total_cells)
{
    const long long *total_p = (const long long *)(status_h + ST_CELLS);
    *total_cells = *total_p;
    if (status_h[ST_ERROR]) return ctx_fail(c, "%s", err_text(status_h[ST_ERROR]));
    if (*total_p <= 0 || *total_p > c->max_cells) return ctx_fail(c, "%s", err_text(ERR_GRID_TOO_BIG));
    if (status_h[ST_RETRY]) { /* bit flags (workgroups of one launch may raise both) */
        if (status_h[ST_RETRY] & 2) c->sort_fused = false; /* a structure k_sort_struct does not hold (atoms or cells): this context sorts the general way from now on */
        /* bit 0: the cell table was too small.  The total is every structure's demand (k_sort_struct: all but those it
           does not hold, which the general pipeline's own count covers on the next pass) */
        if ((status_h[ST_RETRY] & 1) && *total_p + *total_p / 16 + 1024 > c->cells_hint) c->cells_hint = *total_p + *total_p / 16 + 1024;
        return RC_RETRY;
    }
    return 0;
}

/* The tail of a batch on the stream: per-structure totals, then the status words and the cell total into the
 * current set of page-locked host words, then the set's end-of-batch event.  Nothing is waited for. */
static int enqueue_tail(freesasa_gpu_ctx *c, const PipeArgs &pa, int n_structs, double *d_sasa, double *d_totals)
{
    hipStream_t st = c->stream;
    if (d_totals) {
        /* the chunk partials reuse the bounds kernels' scratch (56 bytes per chunk, free by now) */
        HIP_TRY(c, kl_totals(pa, c->n_chunks, n_structs, (const double *)d_sasa, (double *)c->bpart.p, d_totals, st));
    }
    if (c->timing) HIP_TRY(c, hipEventRecord(ctx_ev(c)[3], st));
    int *status_h = ctx_status_h(c);
    HIP_TRY(c, hipMemcpyAsync(status_h, c->status.p, sizeof(int) * (size_t)ST_WORDS, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipEventRecord(c->done[c->slot], st));
    return 0;
}

/* Wait for the batch whose tail went into the current set, and read its verdict and statistics there. */
static int complete_batch(freesasa_gpu_ctx *c, int n, int n_structs, int tile_atoms, int block_threads, int lds)
{
    HIP_TRY(c, hipEventSynchronize(c->done[c->slot]));
    const int *status_h = ctx_status_h(c);
    long long total_cells = 0;
    const int rcs = judge_status(c, status_h, &total_cells);
    freesasa_gpu_stats &S = c->stats;
    S.n_atoms = n; S.n_cells = total_cells; S.n_structs = n_structs;
    S.max_neighbors = status_h[ST_MAX_NN]; S.fallback_tiles = status_h[ST_OVF_TILES];
    for (int k = 0; k < 64; ++k) S.fallback_tiles += status_h[ST_SPLIT + k]; /* (L&R: tiles redone as halves) */
    if (!c->pend[c->slot].walk) S.fallback_tiles += status_h[ST_FAR]; /* (... and tiles with an atom beyond LR2_WALK_Z, which an ordinary build of the main launch hands to the second - counted on their own since round 6, not as split tiles; the walking build keeps them) */
    S.tile_atoms = tile_atoms; S.block_threads = block_threads; S.lds_bytes = lds;
    S.ms_prep = S.ms_kernel = S.ms_total = 0;
    if (c->timing) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx_ev(c)[0], ctx_ev(c)[1]) == hipSuccess) S.ms_prep = ms;
        if (hipEventElapsedTime(&ms, ctx_ev(c)[1], ctx_ev(c)[2]) == hipSuccess) S.ms_kernel = ms;
        if (hipEventElapsedTime(&ms, ctx_ev(c)[0], ctx_ev(c)[3]) == hipSuccess) S.ms_total = ms;
    }
    if (rcs) return rcs;
    if (total_cells + total_cells / 32 > c->cells_hint) c->cells_hint = total_cells + total_cells / 32;
    kl_dump_phase_clocks();
    return 0;
}

static int finish_batch(freesasa_gpu_ctx *c, const PipeArgs &pa, int n, int n_structs, double *d_sasa,
                        double *d_totals, int tile_atoms, int block_threads, int lds, int *)
{
    if (enqueue_tail(c, pa, n_structs, d_sasa, d_totals)) return -1;
    return complete_batch(c, n, n_structs, tile_atoms, block_threads, lds);
}

/* what a completed Lee-Richards batch teaches the context about the next one of its kind (trajectory frames, sweeps) */
static void lr2_learn(freesasa_gpu_ctx *c, const int *status_h, int TA, int ns, int mw, int ds, int n_tiles)
{
    const int learnt = lr2_need_from_hist(status_h + ST_HIST, TA);
    if (learnt > 0) c->hint_nn = (double)(learnt - 8) / TA;
    c->hint_pool2 = lr2_pool_from_hist(status_h + ST_HIST, TA, ns, mw, ds, &c->hint_split2); /* (see there: pool vs occupancy) */
    c->hint_ta2 = TA; c->hint_mw2 = mw;
    c->hint_nn_max = status_h[ST_MAX_NN] + 4; /* the longest list of this batch, a little room */
    if (getenv("FREESASA_AMD_SHOW_SHAPE")) { /* (dev aid) */
        fprintf(stderr, "lr2 learnt: TA %d need96 %d nn %.1f pool2 %d split16 %.3f |", TA, learnt, c->hint_nn, c->hint_pool2, c->hint_split2);
        for (int ta = TA - 1; ta <= TA + 1; ++ta) {
            if (ta < 1 || ta > 7) continue;
            double ab = 0;
            const double r = lr2_rounds_per_atom(status_h + ST_HIST, TA, ta, lr2_pool_for_step(ta, ns, mw, ds, 16, LR2_LANES * LR2_RMAX_MAIN), &ab);
            fprintf(stderr, " ta %d: %.3f rounds/atom, %.3f above the pool;", ta, r, ab);
        }
        fprintf(stderr, "\n");
    }
    c->hint_far = 4LL * status_h[ST_FAR] >= (long long)n_tiles && n_tiles > 0; /* (see lr2_slice_height_at: such tiles go through the second launch unless the main launch walks itself) */
}

/* defer: enqueue only (freesasa_gpu_lr_batch_dev_async); the caller completes the batch later (complete_pending) */
static int run_lr2(freesasa_gpu_ctx *c, const PipeArgs &pa, int n, int n_structs, int resolution, double *d_sasa,
                   double *d_totals, bool defer)
{
    hipStream_t st = c->stream;
    int *status_h = ctx_status_h(c);
    int ta_env = 0, pool_env = 0, ds_env = -1, refill_env = 0;
    if (const char *e = getenv("FREESASA_AMD_LR2")) (void)sscanf(e, "%d,%d,%d,%d", &ta_env, &pool_env, &ds_env, &refill_env); /* tuning aid: "TA,pool,ds,refill" */
    if (c->hint_res[0] != resolution) {
        /* no demand history for this resolution on this context: size the neighbor pool from the local
           density (atoms in an atom's own cell; ~3.1 neighbors per such atom on coils, globules and
           proteins alike).  One 8-byte readback, first call only. */
        long long total_cells = 0;
        const int rcs = collect_status(c, n_structs, 8, &total_cells);
        if (rcs) return rcs;
        c->hint_nn = status_h[ST_OCC_N] > 0 ? 1.25 * 3.1 * (double)status_h[ST_OCC_SUM] / (double)status_h[ST_OCC_N] + 2.0 : 0.0;
        c->hint_nn_max = (int)(1.45 * c->hint_nn); /* longest list ~ 1.6 x the mean on coils, globules and proteins alike */
        c->hint_res[0] = resolution;
        c->hint_pool2 = 0;
    }
    Lr2Cfg cfg = lr2_choose_cfg(resolution, c->hint_nn, ta_env, c->hint_nn_max, c->hint_pool2 > 0 ? c->hint_ta2 : 0, c->hint_split2);
    if (c->hint_pool2 > 0 && c->hint_ta2 == cfg.TA && c->hint_mw2 == cfg.mw) { /* same tile shape as the last batch: its demand histogram decides */
        cfg.pool = c->hint_pool2;
        cfg.rmax = (cfg.pool + LR2_LANES - 1) / LR2_LANES;
    }
    if (pool_env > 0) { cfg.pool = (pool_env + 1) & ~1; cfg.rmax = (cfg.pool + LR2_LANES - 1) / LR2_LANES; }
    if (ds_env >= 0) cfg.ds = ds_env;
    if (refill_env > 0) cfg.refill = refill_env;
    cfg.lds = lr2_layout(cfg.TA, cfg.ns, cfg.pool, cfg.mw, cfg.ds).total;
    if (getenv("FREESASA_AMD_SHOW_SHAPE")) fprintf(stderr, "lr2 shape: TA %d ns %d pool %d mw %d ds %d refill %d rmax %d lds %d cover %d\n", cfg.TA, cfg.ns, cfg.pool, cfg.mw, cfg.ds, cfg.refill, cfg.rmax, cfg.lds, c->hint_nn >= 1.35 * LR2_COVER_DENSITY ? 1 : 0); /* (dev aid) */
    const int n_tiles = (n + cfg.TA - 1) / cfg.TA;
    if (ensure(c, c->ovf_tiles, sizeof(long long) * (2 * (size_t)n_tiles + 2)) || ensure(c, c->ovf_atoms, sizeof(int) * ((size_t)n + 8)))
        return -1;

    Lr2Args la;
    memset(&la, 0, sizeof la);
    la.sq = pa.sq;
    la.s_idx = pa.s_idx;
    la.grid = pa.grid; la.cell_start = pa.cell_start; la.cell_tbl = pa.cell_tbl; la.cell_first = pa.cell_first;
    la.n_atoms = n; la.n_tiles = n_tiles; la.TA = cfg.TA; la.ns = resolution;
    la.pool = cfg.pool; la.mw = cfg.mw; la.ds = cfg.ds; la.refill = cfg.refill;
    /* (hint_nn: what all but 4 % of the last batch's tiles needed per atom, or 1.25 x the density sample: coils ~30, proteins ~58) */
    la.cover = c->hint_nn >= 1.35 * LR2_COVER_DENSITY ? LR2_COVER_DENSITY : 0;
    if (const char *e = getenv("FREESASA_AMD_COVER")) la.cover = atoi(e); /* tuning aid: neighbor records per atom from which a tile runs the cover filter; 0: never */
    /* contained caps (lr2_prune_contained): the largest caps an atom's hits are tested against */
    {
        int want = lr2_prune_want(resolution, la.cover > 0);
        if (const char *e = getenv("FREESASA_AMD_PRUNE")) want = atoi(e); /* tuning / test aid: caps wanted per list (at most LR2_PRUNE_LIST); 0: off */
        la.prune = lr2_prune_arg(want, cfg.TA, cfg.pool);
    }
    la.sasa = d_sasa; la.status = (int *)c->status.p;
    la.inv_ns = 1.0 / (double)resolution;
    la.ovf_items = (long long *)c->ovf_tiles.p;
    la.ovf_count = (int *)c->status.p + ST_OVF2_TILES;
    la.split_count = (int *)c->status.p + ST_SPLIT;

    const int grid_all = ((n_tiles + 7) / 8) * 8;
    int grid_main = grid_all > 147456 ? 147456 : grid_all;
    if (const char *e = getenv("FREESASA_AMD_GRID")) { /* tuning aid */
        const int g = atoi(e);
        if (g >= 8) grid_main = g < grid_all ? (g / 8) * 8 : grid_all;
    }
    /* the last quarter of a full-size launch's workgroups takes two tiles each, the others share the rest (Lr2Args::seg_grid) */
    la.seg_grid = 0; la.seg_tiles = 0;
    if (grid_main == 147456 && !getenv("FREESASA_AMD_ONE_PART")) {
        const int g2 = grid_main / 4, t2 = 2 * g2;
        if (n_tiles >= 8 * t2) { la.seg_grid = grid_main - g2; la.seg_tiles = ((n_tiles - t2) / 8) * 8; }
    }
    la.nn_out = c->dbg_nn; la.nb_out = c->dbg_nb; la.nb_cap = c->dbg_cap;
    la.hooks = (c->dbg_nn ? 1 : 0) | (c->dbg_nb ? 2 : 0);
    la.walk = c->hint_far && c->hint_res[0] == resolution ? 1 : 0;
    if (const char *e = getenv("FREESASA_AMD_WALK")) la.walk = atoi(e); /* tuning / test aid: force (1) or forbid (0) the walking build of the main launch */
    hipError_t le = kl_lr2_main(cfg.rmax, grid_main, (size_t)cfg.lds, st, la);
    if (le != hipSuccess) return ctx_fail(c, "tile kernel launch failed: %s", hipGetErrorString(le));
    if (c->timing) HIP_TRY(c, hipEventRecord(ctx_ev(c)[2], st));
    if (c->dbg_nn) return finish_batch(c, pa, n, n_structs, d_sasa, nullptr, cfg.TA, 64, cfg.lds, status_h);
    /* second launch: halves that did not fit either: larger LDS lists, more registers */
    const Lr2Cfg cm = lr2_mid_cfg(cfg);
    {
        Lr2Args lm = la;
        lm.pool = cm.pool; lm.mw = cm.mw; lm.ds = cm.ds;
        if (!getenv("FREESASA_AMD_COVER")) lm.cover = LR2_COVER_DENSITY;
        lm.work_items = (const long long *)c->ovf_tiles.p;
        lm.work_count = (const int *)c->status.p + ST_OVF2_TILES;
        lm.ovf_items = nullptr;
        lm.ovf_atoms = (int *)c->ovf_atoms.p;
        lm.ovf_count = (int *)c->status.p + ST_OVF3_ATOMS;
        lm.split_count = nullptr;
        /* (usually a handful of halves; but a batch far from the origin sends EVERY tile here - lr2_slice_height_at - so the
           grid is sized to fill the chip then: workgroups that find the list short leave at once) */
        const int grid_mid = n_tiles < LR2_MID_BLOCKS ? n_tiles : LR2_MID_BLOCKS;
        le = kl_lr2_mid(grid_mid, (size_t)cm.lds, st, lm);
        if (le != hipSuccess) return ctx_fail(c, "second tile launch failed: %s", hipGetErrorString(le));
    }
    /* last launch: whatever is left (pathological densities), atom by atom: the first-generation kernel with its
       lists in a global slab */
    {
        const TileCfg fb = fallback_cfg(lr_slab_cfg(1, resolution), true);
        const size_t stride = tile_slab_bytes(fb.TA, fb.cap_idx, fb.pool, fb.lr, fb.ds, fb.B);
        if (ensure(c, c->slab, stride * SASA_FB_BLOCKS)) return -1;
        TileArgs tf;
        memset(&tf, 0, sizeof tf);
        tf.sq = pa.sq;
        tf.s_idx = pa.s_idx;
        tf.grid = pa.grid; tf.cell_start = pa.cell_start; tf.cell_tbl = pa.cell_tbl; tf.cell_first = pa.cell_first;
        tf.n_atoms = n; tf.n_tiles = n; tf.TA = 1; tf.n_res = resolution; tf.tab = fb.tab;
        tf.sasa = d_sasa; tf.lr = 1; tf.status = (int *)c->status.p;
        tf.cap_idx = fb.cap_idx; tf.pool = fb.pool; tf.ds = fb.ds;
        tf.work_tiles = (const int *)c->ovf_atoms.p;
        tf.work_count = (const int *)c->status.p + ST_OVF3_ATOMS;
        tf.slab = (char *)c->slab.p;
        tf.slab_stride = (long long)stride;
        le = kl_lr_tile(2, fb, tf, SASA_FB_BLOCKS, fb.lds, st, false);
        if (le != hipSuccess) return ctx_fail(c, "fallback kernel launch failed: %s", hipGetErrorString(le));
    }
    if (enqueue_tail(c, pa, n_structs, d_sasa, d_totals)) return -1;
    freesasa_gpu_ctx::Pend &P = c->pend[c->slot];
    P.n = n; P.TA = cfg.TA; P.mw = cfg.mw; P.ds = cfg.ds; P.lds = cfg.lds; P.n_structs = n_structs; P.resolution = resolution; P.walk = la.walk != 0;
    if (defer) return 0;
    const int rc = complete_batch(c, n, n_structs, cfg.TA, 64, cfg.lds);
    if (rc) return rc;
    lr2_learn(c, status_h, cfg.TA, cfg.ns, cfg.mw, cfg.ds, n_tiles);
    return 0;
}

/* ------------------------------------------------------------------ one batch */

static int run_batch_once(freesasa_gpu_ctx *c, bool lr, const double *d_xyz, const double *d_radii,
                     const int64_t *offsets, int n_structs, double probe, int resolution,
                     const double *unit_points, double *d_sasa, int *d_counts, double *d_totals, bool defer = false, bool *deferred = nullptr)
{
    c->err[0] = 0;
    if (deferred) *deferred = false;
    if (!d_xyz || !d_radii || !offsets || !d_sasa) return ctx_fail(c, "null argument");
    if (n_structs <= 0) return ctx_fail(c, "n_structs must be > 0");
    if (resolution <= 0) return ctx_fail(c, "resolution must be > 0");
    if (offsets[0] != 0) return ctx_fail(c, "offsets[0] must be 0");
    for (int s = 0; s < n_structs; ++s)
        if (offsets[s + 1] < offsets[s]) return ctx_fail(c, "offsets must be non-decreasing");
    const int64_t n64 = offsets[n_structs];
    if (n64 <= 0) return ctx_fail(c, "empty batch");
    if (n64 > (int64_t)1 << 30) return ctx_fail(c, "batch too large (max 2^30 atoms per call)");
    const int n = (int)n64;

    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t nb = (size_t)n;

    /* workspace */
    if (ensure(c, c->offsets, sizeof(int64_t) * ((size_t)n_structs + 1)) || ensure(c, c->grid, sizeof(GridS) * (size_t)n_structs) ||
        ensure(c, c->ncells, sizeof(long long) * ((size_t)n_structs + 1)) || ensure(c, c->sid, 4 * nb) ||
        ensure(c, c->cell_of, 8 * nb) || ensure(c, c->rank, 4 * nb) || ensure(c, c->sq, 32 * nb) ||
        ensure(c, c->s_idx, 16 * nb) ||
        ensure(c, c->status, sizeof(int) * ST_WORDS))
        return -1;

    /* offsets and the chunk table derived from them: upload only when they changed
       (trajectory frames and repeated batches reuse them) */
    if ((int)c->offsets_host.size() != n_structs + 1 ||
        memcmp(c->offsets_host.data(), offsets, sizeof(int64_t) * ((size_t)n_structs + 1)) != 0) {
        if (c->pend[0].active || c->pend[1].active) HIP_TRY(c, hipStreamSynchronize(st)); /* (batches in flight read the tables about to be overwritten; they are collected later, from their own sets of host words: see ensure) */
        c->offsets_host.clear(); /* (set again below, once the tables derived from it are on the device) */
        std::vector<int> cs, cl, sc0((size_t)n_structs + 1);
        std::vector<int64_t> cb;
        for (int s = 0; s < n_structs; ++s) {
            sc0[s] = (int)cs.size();
            for (int64_t b = offsets[s]; b < offsets[s + 1]; b += SASA_BOUNDS_CHUNK) {
                const int64_t e = b + SASA_BOUNDS_CHUNK < offsets[s + 1] ? b + SASA_BOUNDS_CHUNK : offsets[s + 1];
                cs.push_back(s); cb.push_back(b); cl.push_back((int)(e - b));
            }
        }
        sc0[n_structs] = (int)cs.size();
        c->n_chunks = (int)cs.size();
        const size_t nc = cs.size();
        if (ensure(c, c->chunk_struct, 4 * nc) || ensure(c, c->chunk_begin, 8 * nc) || ensure(c, c->chunk_len, 4 * nc) ||
            ensure(c, c->struct_chunk0, 4 * ((size_t)n_structs + 1)) || ensure(c, c->bpart, 56 * nc))
            return -1;
        HIP_TRY(c, hipMemcpy(c->offsets.p, offsets, sizeof(int64_t) * ((size_t)n_structs + 1), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->chunk_struct.p, cs.data(), 4 * nc, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->chunk_begin.p, cb.data(), 8 * nc, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->chunk_len.p, cl.data(), 4 * nc, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->struct_chunk0.p, sc0.data(), 4 * ((size_t)n_structs + 1), hipMemcpyHostToDevice));
        c->offsets_host.assign(offsets, offsets + n_structs + 1);
    }
    /* (a whole number of 64-byte lines - the buffer has the room, ensure() adds slack: the runtime clears a length that is
       not a multiple of 16 with TWO fill kernels, seen in the round-6 kernel trace of the 200 000-atom case, 7 us apart) */
    HIP_TRY(c, hipMemsetAsync(c->status.p, 0, (sizeof(int) * ST_WORDS + 63) & ~(size_t)63, st));
    if (c->timing) HIP_TRY(c, hipEventRecord(ctx_ev(c)[0], st));

    PipeArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.xyz = d_xyz; pa.radii = d_radii; pa.offsets = (const int64_t *)c->offsets.p;
    pa.n_structs = n_structs; pa.n_atoms = n; pa.probe = probe; pa.max_cells = c->max_cells;
    pa.shared_radii = c->shared_radii ? 1 : 0;
    pa.n_chunks = c->n_chunks; pa.chunk_struct = (const int *)c->chunk_struct.p; pa.chunk_begin = (const int64_t *)c->chunk_begin.p;
    pa.chunk_len = (const int *)c->chunk_len.p; pa.struct_chunk0 = (const int *)c->struct_chunk0.p; pa.bpart = (double *)c->bpart.p;
    pa.grid = (GridS *)c->grid.p; pa.ncells = (long long *)c->ncells.p;
    pa.sid = (int *)c->sid.p; pa.cell_of = (long long *)c->cell_of.p; pa.rank = (int *)c->rank.p;
    pa.sq = (Quad *)c->sq.p;
    pa.s_idx = (SortIdx *)c->s_idx.p;
    pa.status = (int *)c->status.p;
    pa.cells_total = (long long *)((int *)c->status.p + ST_CELLS);
    pa.cells_total_at = pa.cells_total - pa.ncells; /* (both 8-byte aligned device addresses) */
    if (probe != c->hint_probe) { /* launch-shape history is per (resolution, probe radius) */
        c->hint_res[0] = c->hint_res[1] = 0;
        c->hint_pool2 = 0;
        c->hint_probe = probe;
    }
    pa.occ_stride = c->hint_res[lr ? 0 : 1] == resolution ? 0 : (n / 256 > 0 ? n / 256 : 1);

    /* The cell table is sized WITHOUT waiting for K2's total: for what the context has seen so far, and for a
       first batch 10 cells per atom (sparse random coils need 9, proteins 1-2) plus 256 per structure.  K2 checks
       the real total against it on the device (ST_RETRY, see PIPE_GATE); the total itself reaches the host with
       the status words at the end of the batch. */
    int *status_h = ctx_status_h(c);
    long long cells_cap = 10LL * n + 256LL * n_structs; /* (0.4 GB for 1e7 atoms; a sparser batch is redone once with K2's size) */
    if (cells_cap < c->cells_hint) cells_cap = c->cells_hint;
    if (cells_cap > c->max_cells) cells_cap = c->max_cells;
    const int nblk_scan = (int)((cells_cap + 1 + (long long)SASA_PIPE_B * SASA_SCAN_ITEMS - 1) / ((long long)SASA_PIPE_B * SASA_SCAN_ITEMS));
    /* batches of small structures: bounds, grid and cell sort of a structure in one workgroup (k_sort_struct) */
    long long biggest = 0;
    for (int s_ = 0; s_ < n_structs; ++s_) biggest = offsets[s_ + 1] - offsets[s_] > biggest ? offsets[s_ + 1] - offsets[s_] : biggest;
    /* (... and a call with a few SMALL structures - freesasa_calc_coord on one protein - where one launch instead of nine
       is worth more than the general pipeline's parallelism: round 5, MI355X, median of freesasa_calc_coord on 1UBQ / 1A0Q /
       a 4000-atom coil, L&R 142 -> 134 / 174 -> 165 / 185 -> 177 us, S&R 131 -> 116 / 124 -> 115 / 140 -> 130 us; from
       8000 atoms on the general pipeline wins; FREESASA_AMD_SMALL_FUSED=n moves the limit) */
    static const long long small_fused = getenv("FREESASA_AMD_SMALL_FUSED") ? atoll(getenv("FREESASA_AMD_SMALL_FUSED")) : 4096;
    const bool fused = c->sort_fused && biggest <= SORT_ATOMS && (n_structs >= 8 || biggest <= small_fused) && !getenv("FREESASA_AMD_NO_FUSED_SORT");
    /* ... which writes the cell table in its compact form for the Lee-Richards tile kernel (PipeArgs::cell_tbl) */
    /* (... and, since round 5, for the Shrake-Rupley kernel too: tile_phase_load looks cells up through cell_first_atom) */
    const bool compact = fused && (!lr || (lr2_supported(resolution) && !getenv("FREESASA_AMD_LR1"))) && !getenv("FREESASA_AMD_DENSE_CELLS");
    if (compact) {
        if (ensure(c, c->cell_tbl, sizeof(unsigned long long) * ((size_t)(cells_cap >> 5) + 2)) || ensure(c, c->cell_first, sizeof(int) * (nb + (size_t)n_structs + 2)))
            return -1;
        pa.cell_tbl = (unsigned long long *)c->cell_tbl.p;
        pa.cell_first = (int *)c->cell_first.p;
    } else {
        const size_t desc_cap_was = c->blk_sums.cap;
        if (ensure(c, c->cell_start, sizeof(int) * ((size_t)cells_cap + 4)) || ensure(c, c->blk_sums, sizeof(unsigned long long) * ((size_t)nblk_scan + 1)))
            return -1;
        /* the chained scan's block descriptors carry the batch's epoch (sasa_kernels.h, K4): fresh memory must not look
           like a descriptor of this epoch, so a new array is cleared once (epoch 0 is never used).  "New" is a new
           CAPACITY, not a new address: the allocator may hand the grown array the address of the one just freed, and the
           part beyond the old length then holds whatever was there before - an earlier context's descriptors with small
           epochs like this one's (round 6, tools/dev/prune_fuzz.py: a batch that is redone with a larger cell table -
           ST_RETRY - scanned with such descriptors as its predecessors', scattered atoms out of bounds, and the GPU
           faulted; one batch in a few thousand random ones, and only behind certain other batches) */
        if (c->blk_sums.cap != desc_cap_was) HIP_TRY(c, hipMemsetAsync(c->blk_sums.p, 0, c->blk_sums.cap, st));
    }
    pa.cell_start = (int *)c->cell_start.p;
    pa.scan_desc = (unsigned long long *)c->blk_sums.p;
    c->scan_epoch = c->scan_epoch >= (1 << 30) - 1 ? 1 : c->scan_epoch + 1;
    pa.scan_epoch = c->scan_epoch;
    pa.zero_n = cells_cap + 2;
    pa.cells_cap = cells_cap;

    if (fused) {
        HIP_TRY(c, kl_prep_fused(pa, st));
    } else {
        HIP_TRY(c, kl_prep_general(pa, cells_cap, st));
    }
    if (c->timing) HIP_TRY(c, hipEventRecord(ctx_ev(c)[1], st));

    /* Lee & Richards at ordinary resolutions: the second-generation kernel (lr2_kernels.h) */
    if (lr && lr2_supported(resolution) && !getenv("FREESASA_AMD_LR1")) {
        const int rc2 = run_lr2(c, pa, n, n_structs, resolution, d_sasa, d_totals, defer && !c->dbg_nn);
        if (rc2 == 0 && defer && !c->dbg_nn && deferred) *deferred = true;
        return rc2;
    }

    /* fused tile kernel */
    const int hi = lr ? 0 : 1;
    if (c->hint_res[hi] != resolution) {
        /* no demand history for this resolution on this context: estimate the neighbor count from the
           local density (atoms in an atom's own cell; ~3.1 neighbors per such atom on coils, globules
           and proteins alike) so that the very first launch already has a fitting neighbor pool and,
           for dense inputs, the bucket-sort variant.  One 8-byte readback, first call only. */
        long long total_cells = 0;
        const int rcs = collect_status(c, n_structs, 8, &total_cells);
        if (rcs) return rcs;
        if (status_h[ST_OCC_N] > 0) {
            const double nn_est = 3.1 * (double)status_h[ST_OCC_SUM] / (double)status_h[ST_OCC_N];
            const TileCfg probe_cfg = choose_cfg(resolution, lr, 0, !lr && resolution <= SR_CAP_POINTS_MAX);
            int pool = (int)(1.35 * nn_est * probe_cfg.TA + 16.0);
            pool = (pool + 1) & ~1;
            if (pool < 32) pool = 32;
            if (pool > 4096) pool = 4096;
            if (!lr) { /* S&R: records per atom's segment: the longest list of a tile is ~1.5 x the mean */
                pool = (((int)(1.5 * nn_est) + 8 + 7) / 8) * 8;
                pool = pool < SR_CAP_MIN ? SR_CAP_MIN : (pool > SR_CAP_MAX ? SR_CAP_MAX : pool);
            }
            c->hint_res[hi] = resolution;
            c->hint_ta[hi] = probe_cfg.TA;
            c->hint_pool[hi] = pool;
            if (lr) c->hint_bucket = nn_est > 30.0;
        }
    }
    if (!lr) {
        if (!unit_points) return ctx_fail(c, "unit_points is null");
        if (ensure(c, c->unit_pts, sizeof(double) * 3 * (size_t)resolution)) return -1;
        /* pageable host-to-device copies stall the stream: upload the points only when they change */
        if (c->unit_host.size() != 3 * (size_t)resolution ||
            memcmp(c->unit_host.data(), unit_points, sizeof(double) * 3 * (size_t)resolution) != 0) {
            c->unit_host.assign(unit_points, unit_points + 3 * (size_t)resolution);
            if (hipMemcpyAsync(c->unit_pts.p, c->unit_host.data(), sizeof(double) * 3 * (size_t)resolution, hipMemcpyHostToDevice, st) != hipSuccess) {
                c->unit_host.clear();
                return ctx_fail(c, "upload of the test points failed");
            }
            /* the third arrangement's table of cap masks for these points (sr_caps.h; FREESASA_AMD_SR_CAPS=0 keeps the
               second arrangement, "N,L" sets the table's resolution: tuning aids) */
            int tn = SR_CAP_N_DEFAULT, tl = SR_CAP_L_DEFAULT;
            if (const char *e = getenv("FREESASA_AMD_SR_CAPS")) {
                int n_ = 0, l_ = 0;
                if (sscanf(e, "%d,%d", &n_, &l_) == 2 && n_ >= 1 && n_ <= 64 && l_ >= 1 && l_ <= 256) { tn = n_; tl = l_; }
                else if (atoi(e) == 0) tn = 0;
            }
            c->captab_n = c->captab_l = 0;
            if (tn > 0 && sr_captab_build(c->unit_host.data(), resolution, tn, tl, c->captab_host)) {
                const size_t bytes = sizeof(SrCapEntry) * c->captab_host.size();
                if (ensure(c, c->captab, bytes)) { c->unit_host.clear(); return -1; }
                if (hipMemcpyAsync(c->captab.p, c->captab_host.data(), bytes, hipMemcpyHostToDevice, st) != hipSuccess) {
                    c->unit_host.clear();
                    return ctx_fail(c, "upload of the cap table failed");
                }
                c->captab_n = tn; c->captab_l = tl;
            }
        }
    }

    const bool sr_caps = !lr && c->captab_n > 0;
    TileCfg cfg = choose_cfg(resolution, lr, c->hint_res[hi] == resolution ? c->hint_pool[hi] : 0, sr_caps, c->hint_res[hi] == resolution ? c->hint_ta[hi] : 0);
    if (const char *e = getenv("FREESASA_AMD_CFG")) { /* tuning aid: "B,TA,pool,ds" */
        int b = 0, t = 0, pl = 0, d = 0;
        if (sscanf(e, "%d,%d,%d,%d", &b, &t, &pl, &d) == 4 && (b == 64 || b == 128 || b == 256) && t >= 1 && t <= b &&
            (!lr || !cfg.tab || t * resolution <= 4096)) {
            cfg.B = b; cfg.TA = t; cfg.pool = pl; cfg.ds = lr ? d : 0;
            if (!lr) { cfg.cap_idx = pl > 0 ? pl : cfg.cap_idx; cfg.pool = cfg.TA * cfg.cap_idx; } /* (S&R: "pool" = records per atom) */
            if (!lr) cfg.tab = sr_survivors_fit(cfg.TA, resolution) ? 1 : 0;
            cfg.items = lr ? (cfg.tab ? cfg.TA * resolution : cfg.B) : (cfg.tab ? sr_tile_items(cfg.TA, resolution, sr_caps) : 1);
            cfg.lds = tile_fixed_bytes(cfg.TA, cfg.items) + tile_list_bytes(cfg.TA, cfg.cap_idx, cfg.pool, cfg.lr, cfg.ds, cfg.B);
        }
    }
    /* whatever the estimate or the tuning aid asked for: the main launch's lists must fit the CU's LDS (dense tiles
       then go to the later launches instead of failing the launch) */
    while (cfg.lds > 160 * 1024 && (lr ? cfg.pool > 32 : cfg.cap_idx > 8)) {
        if (lr) {
            cfg.pool = (cfg.pool * 3 / 4) & ~1;
        } else { /* S&R: the pool IS TA segments of cap_idx records (sr_nb_test writes pq[la * cap_idx + s]): they shrink together */
            cfg.cap_idx = ((cfg.cap_idx * 3 / 4) + 7) & ~7;
            cfg.pool = cfg.TA * cfg.cap_idx;
        }
        cfg.lds = tile_fixed_bytes(cfg.TA, cfg.items) + tile_list_bytes(cfg.TA, cfg.cap_idx, cfg.pool, cfg.lr, cfg.ds, cfg.B);
    }
    if (cfg.lds > 160 * 1024) return ctx_fail(c, "tile shape does not fit the LDS of a compute unit (%zu bytes)", cfg.lds);
    if (getenv("FREESASA_AMD_SHOW_SHAPE")) fprintf(stderr, "%s tile shape: B %d TA %d cap_idx %d pool %d ds %d lds %zu (hint %d)\n", lr ? "lr1" : "sr", cfg.B, cfg.TA, cfg.cap_idx, cfg.pool, cfg.ds, cfg.lds, c->hint_res[hi] == resolution ? c->hint_pool[hi] : 0); /* (dev aid) */
    const int n_tiles = (n + cfg.TA - 1) / cfg.TA;
    if (ensure(c, c->ovf_tiles, sizeof(int) * ((size_t)n_tiles + 1)) || ensure(c, c->ovf_tiles2, sizeof(int) * ((size_t)n_tiles + 1))) return -1;

    TileArgs ta;
    memset(&ta, 0, sizeof ta);
    ta.sq = pa.sq;
    ta.s_idx = pa.s_idx;
    ta.grid = pa.grid; ta.cell_start = pa.cell_start; ta.cell_tbl = pa.cell_tbl; ta.cell_first = pa.cell_first;
    ta.n_atoms = n; ta.n_tiles = n_tiles; ta.TA = cfg.TA; ta.n_res = resolution; ta.tab = cfg.tab;
    ta.sasa = d_sasa; ta.counts = d_counts;
    ta.cap_idx = cfg.cap_idx; ta.pool = cfg.pool; ta.lr = cfg.lr; ta.ds = cfg.ds;
    ta.ovf_count = (int *)c->status.p + ST_OVF_TILES;
    ta.ovf_tiles = (int *)c->ovf_tiles.p;
    ta.work_tiles = nullptr;
    ta.work_count = nullptr;
    ta.status = (int *)c->status.p;
    if (!lr) {
        ta.unit_pts = (const double *)c->unit_pts.p;
        if (sr_caps) { ta.captab = c->captab.p; ta.cap_n = c->captab_n; ta.cap_l = c->captab_l; }
    }

    /* workgroups loop over tiles (w, w + grid, ...): ~150k workgroups measured ~2% better than
       one workgroup per tile, a grid of exactly the resident workgroups 25% worse (tiles vary) */
    int grid_main = ((n_tiles + 7) / 8) * 8;
    if (grid_main > 147456) grid_main = 147456;
    if (const char *e = getenv("FREESASA_AMD_GRID")) { /* tuning aid */
        const int g = atoi(e);
        if (g >= 8 && g < grid_main) grid_main = (g / 8) * 8;
    }
    hipError_t le;
    const bool bucket = lr && c->hint_bucket && c->hint_res[0] == resolution;
    le = lr ? kl_lr_tile(0, cfg, ta, grid_main, cfg.lds, st, bucket) : kl_sr_tile(0, cfg, ta, grid_main, cfg.lds, st);
    if (le != hipSuccess) return ctx_fail(c, "tile kernel launch failed: %s", hipGetErrorString(le));
    if (c->timing) HIP_TRY(c, hipEventRecord(ctx_ev(c)[2], st));

    /* second launch: the tiles whose lists did not fit the small LDS capacities (the blocks read
       the count on the device; normally a fraction of a percent of the tiles) */
    {
        const TileCfg mc = mid_cfg(cfg, lr);
        TileArgs tm = ta;
        tm.cap_idx = mc.cap_idx; tm.pool = mc.pool; tm.ds = mc.ds;
        tm.work_tiles = (const int *)c->ovf_tiles.p;
        tm.work_count = (const int *)c->status.p + ST_OVF_TILES;
        tm.ovf_tiles = (int *)c->ovf_tiles2.p;
        tm.ovf_count = (int *)c->status.p + ST_OVF2_TILES;
        const int grid_mid = n_tiles < SASA_MID_BLOCKS ? n_tiles : SASA_MID_BLOCKS;
        le = lr ? kl_lr_tile(1, mc, tm, grid_mid, mc.lds, st, bucket) : kl_sr_tile(1, mc, tm, grid_mid, mc.lds, st);
        if (le != hipSuccess) return ctx_fail(c, "second tile launch failed: %s", hipGetErrorString(le));
    }
    /* third launch: whatever is left (pathological densities), lists in a global slab */
    {
        const TileCfg fb = fallback_cfg(cfg, lr);
        const size_t stride = tile_slab_bytes(fb.TA, fb.cap_idx, fb.pool, fb.lr, fb.ds, fb.B);
        /* (Shrake-Rupley: a workgroup's slice of the slab is TA segments of SASA_FB_CAP records, 1.4 MB for 8-atom tiles - 92 MB
           for 64 workgroups in EVERY context, and the multi-device drivers hold a context per lane (round-5 advisor).  The launch
           only ever sees what overflowed two launches: a quarter of the workgroups serves it) */
        const int fb_blocks = lr ? SASA_FB_BLOCKS : SASA_FB_BLOCKS / 4;
        if (ensure(c, c->slab, stride * (size_t)fb_blocks)) return -1;
        TileArgs tf = ta;
        tf.cap_idx = fb.cap_idx; tf.pool = fb.pool; tf.ds = fb.ds;
        tf.work_tiles = (const int *)c->ovf_tiles2.p;
        tf.work_count = (const int *)c->status.p + ST_OVF2_TILES;
        tf.ovf_tiles = nullptr;
        tf.ovf_count = nullptr;
        tf.slab = (char *)c->slab.p;
        tf.slab_stride = (long long)stride;
        tf.captab = nullptr; /* (segments of 4096 records in the slab: the second arrangement ... */
        if (sr_caps) tf.tab = 0; /* ... without its survivor table: the tile's LDS is sized for the third's masks and list) */
        le = lr ? kl_lr_tile(2, fb, tf, fb_blocks, fb.lds, st, false) : kl_sr_tile(2, fb, tf, fb_blocks, fb.lds, st);
        if (le != hipSuccess) return ctx_fail(c, "fallback kernel launch failed: %s", hipGetErrorString(le));
    }

    if (d_totals) {
        /* the chunk partials reuse the bounds kernels' scratch (56 bytes per chunk, free by now) */
        HIP_TRY(c, kl_totals(pa, c->n_chunks, n_structs, (const double *)d_sasa, (double *)c->bpart.p, d_totals, st));
    }
    if (c->timing) HIP_TRY(c, hipEventRecord(ctx_ev(c)[3], st));

    long long total_cells = 0;
    const int rcs = collect_status(c, n_structs, ST_WORDS, &total_cells);

    freesasa_gpu_stats &S = c->stats;
    S.n_atoms = n; S.n_cells = total_cells; S.n_structs = n_structs;
    S.max_neighbors = status_h[ST_MAX_NN]; S.fallback_tiles = status_h[ST_OVF_TILES];
    S.tile_atoms = cfg.TA; S.block_threads = cfg.B; S.lds_bytes = (int)cfg.lds;
    S.ms_prep = S.ms_kernel = S.ms_total = 0;
    if (c->timing) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx_ev(c)[0], ctx_ev(c)[1]) == hipSuccess) S.ms_prep = ms;
        if (hipEventElapsedTime(&ms, ctx_ev(c)[1], ctx_ev(c)[2]) == hipSuccess) S.ms_kernel = ms;
        if (hipEventElapsedTime(&ms, ctx_ev(c)[0], ctx_ev(c)[3]) == hipSuccess) S.ms_total = ms;
    }
    if (rcs) return rcs;
    if (total_cells + total_cells / 32 > c->cells_hint) c->cells_hint = total_cells + total_cells / 32;
    kl_dump_phase_clocks();
    /* learn the pool size for the next batch of this kind (trajectory frames, sweeps) */
    c->hint_res[hi] = resolution;
    c->hint_ta[hi] = cfg.TA;
    c->hint_pool[hi] = lr ? pool_from_hist(status_h + ST_HIST, cfg.TA) : sr_cap_from_hist(status_h + ST_HIST);
    if (!lr && getenv("FREESASA_AMD_SHOW_HIST")) { fprintf(stderr, "sr hist:"); for (int k = 0; k < 64; ++k) fprintf(stderr, " %d", status_h[ST_HIST + k]); fprintf(stderr, " -> cap %d, ovf %d\n", c->hint_pool[hi], status_h[ST_OVF_TILES]); } /* (dev aid) */
    if (lr) c->hint_bucket = mean_from_hist(status_h + ST_HIST, cfg.TA) > 30.0 * cfg.TA;
    return 0;
}

static int run_batch_impl(freesasa_gpu_ctx *c, bool lr, const double *d_xyz, const double *d_radii,
                     const int64_t *offsets, int n_structs, double probe, int resolution,
                     const double *unit_points, double *d_sasa, int *d_counts, double *d_totals)
{
    int rc = run_batch_once(c, lr, d_xyz, d_radii, offsets, n_structs, probe, resolution, unit_points, d_sasa, d_counts, d_totals);
    /* once more when the cell table was too small (first batch of a very sparse kind; now with K2's size), and once more
       when a structure did not fit the per-structure cell sort (the general pipeline from then on) */
    for (int again = 0; rc == RC_RETRY && again < 3; ++again)
        rc = run_batch_once(c, lr, d_xyz, d_radii, offsets, n_structs, probe, resolution, unit_points, d_sasa, d_counts, d_totals);
    if (rc == RC_RETRY) return ctx_fail(c, "cell table sizing did not converge");
    return rc;
}

/* On failure nothing may still be running on the stream when the caller gets control back (it
 * is entitled to free its buffers right away). */
int run_batch(freesasa_gpu_ctx *c, bool lr, const double *d_xyz, const double *d_radii,
              const int64_t *offsets, int n_structs, double probe, int resolution,
              const double *unit_points, double *d_sasa, int *d_counts, double *d_totals)
{
    const int rc = run_batch_impl(c, lr, d_xyz, d_radii, offsets, n_structs, probe, resolution, unit_points,
                                  d_sasa, d_counts, d_totals);
    if (rc) (void)hipStreamSynchronize(c->stream);
    return rc;
}

/* ------------------------------------------------------------------ asynchronous batches
 * freesasa_gpu_lr_batch_dev_async enqueues a batch and returns; up to two may be in flight on a context.  Each leaves
 * its host-side set (status words, events) behind; complete_pending waits for its end-of-batch event and reads the
 * verdict there.  The rare batch whose cell table was too small (RC_RETRY) is redone synchronously, with everything
 * else on the stream drained first. */
static int complete_pending(freesasa_gpu_ctx *c, int slot)
{
    freesasa_gpu_ctx::Pend &P = c->pend[slot];
    if (!P.active) return 0;
    const int keep = c->slot;
    c->slot = slot;
    int rc = complete_batch(c, P.n, P.n_structs, P.TA, 64, P.lds);
    /* (what it teaches is keyed to the resolution and probe the hints stand for NOW: a later batch of another kind may
       have reset them since this one was enqueued) */
    if (rc == 0 && P.resolution == c->hint_res[0] && P.probe == c->hint_probe) lr2_learn(c, ctx_status_h(c), P.TA, P.resolution, P.mw, P.ds, (P.n + P.TA - 1) / P.TA);
    c->slot = keep;
    P.active = false;
    if (rc == RC_RETRY) {
        /* redo it the synchronous way (which sizes the table from this pass's count); the other batch in flight, if
           any, ran into the same table and is redone when it is collected */
        (void)hipStreamSynchronize(c->stream);
        std::vector<int64_t> offs;
        offs.swap(P.offsets);
        rc = run_batch(c, true, P.d_xyz, P.d_radii, offs.data(), P.n_structs, P.probe, P.resolution, nullptr, P.d_sasa, nullptr, P.d_totals);
    }
    return rc;
}
/* collect every batch in flight, oldest first; -1 if any failed (the context's error text is the last failure's) */
static int drain_pending(freesasa_gpu_ctx *c)
{
    int rc = 0;
    for (int k = 0; k < 2; ++k) {
        const int slot = (c->slot + k) & 1; /* c->slot is the set the next batch takes: the older one of two in flight */
        if (c->pend[slot].active && complete_pending(c, slot)) rc = -1;
    }
    return rc;
}

extern "C" int freesasa_gpu_wait(freesasa_gpu_ctx *c)
{
    if (!c) return -1;
    if (hipSetDevice(c->device) != hipSuccess) return ctx_fail(c, "hipSetDevice failed");
    return guarded_ctx(c, [&]() -> int { return drain_pending(c); });
}

extern "C" int freesasa_gpu_lr_batch_dev_async(freesasa_gpu_ctx *c, const double *d_xyz, const double *d_radii,
                                               const int64_t *offsets, int n_structs, double probe, int n_slices,
                                               double *d_sasa, double *d_totals)
{
    if (!c) return -1;
    if (hipSetDevice(c->device) != hipSuccess) return ctx_fail(c, "hipSetDevice failed");
    return guarded_ctx(c, [&]() -> int {
    /* the set this batch takes may still belong to the batch submitted two calls ago: collect that one first */
    if (c->pend[c->slot].active && complete_pending(c, c->slot)) return -1;
    bool deferred = false;
    int rc = run_batch_once(c, true, d_xyz, d_radii, offsets, n_structs, probe, n_slices, nullptr, d_sasa, nullptr, d_totals, true, &deferred);
    if (rc == RC_RETRY || (rc == 0 && !deferred)) {
        /* a first batch of its kind (its density is read back before the tile kernel is shaped) that has to be
           redone, or a path without a deferred tail: finish it the synchronous way */
        if (rc == RC_RETRY) rc = run_batch(c, true, d_xyz, d_radii, offsets, n_structs, probe, n_slices, nullptr, d_sasa, nullptr, d_totals);
        return rc;
    }
    if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
    freesasa_gpu_ctx::Pend &P = c->pend[c->slot];
    P.offsets.assign(offsets, offsets + n_structs + 1); /* (may throw: the set is not marked in flight before it has what a redo needs) */
    P.active = true;
    P.d_xyz = d_xyz; P.d_radii = d_radii;
    P.probe = probe; P.d_sasa = d_sasa; P.d_totals = d_totals;
    c->slot ^= 1;
    return 0;
    });
}

extern "C" int freesasa_gpu_lr_batch_dev(freesasa_gpu_ctx *c, const double *d_xyz, const double *d_radii,
                                         const int64_t *offsets, int n_structs, double probe, int n_slices,
                                         double *d_sasa, double *d_totals)
{
    if (!c) return -1;
    if (freesasa_gpu_wait(c)) return -1; /* (batches submitted asynchronously come first) */
    return guarded_ctx(c, [&]() -> int { return run_batch(c, true, d_xyz, d_radii, offsets, n_structs, probe, n_slices, nullptr, d_sasa, nullptr, d_totals); });
}

extern "C" int freesasa_gpu_sr_batch_dev(freesasa_gpu_ctx *c, const double *d_xyz, const double *d_radii,
                                         const int64_t *offsets, int n_structs, double probe, int n_points,
                                         const double *unit_points, double *d_sasa, int *d_counts, double *d_totals)
{
    if (!c) return -1;
    if (freesasa_gpu_wait(c)) return -1;
    return guarded_ctx(c, [&]() -> int { return run_batch(c, false, d_xyz, d_radii, offsets, n_structs, probe, n_points, unit_points, d_sasa, d_counts, d_totals); });
}

