/*
 * emu.cpp — TESTS ONLY.  Drives the phase functions of freesasa_amd/csrc/sasa_kernels.h on the
 * CPU, one "thread" at a time, in the same launch sequence as gpu_engine.hip::run_batch, so the
 * kernel logic (indexing, capacities, overflow hand-off, arithmetic order) can be checked against
 * the oracle in the GPU-less build container.  Built as tests/emu/libsasa_emu.so by `make emu`;
 * never linked into libfreesasa_amd.so.  With host libm the emulated L&R must equal the oracle
 * bit for bit, which pins everything except the device's acos/atan2.
 */
#define SASA_EMU 1
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include <ucontext.h>

#include "../../freesasa_amd/csrc/lr2_kernels.h"

using namespace sasa;

/* ---------------------------------------------------------------------------------------------
 * A wave in lock step: the second-generation L&R kernel (lr2_kernels.h) talks across its 64 lanes
 * (ballot, shuffle, one-wave barriers).  Here every lane is a fiber (ucontext); a cross-lane
 * operation deposits the lane's operand and yields to the scheduler, which resumes the lanes once
 * all of them have arrived — the device's semantics for wave-uniform control flow. */
namespace sasa_emu {
long long lr2_count[16];
static const int W = 64;
static ucontext_t g_main, g_fiber[W];
static bool g_done[W];
static int g_lane = -1;
static long long g_dep[W];
static unsigned long long g_ballot;
static void (*g_body)(int lane, void *ctx);
static void *g_ctx;
static std::vector<char> g_stacks;

static void yield_to_scheduler() { swapcontext(&g_fiber[g_lane], &g_main); }
unsigned long long wave_ballot(bool p)
{
    g_dep[g_lane] = p ? 1 : 0;
    yield_to_scheduler(); /* the scheduler folds the deposits into g_ballot before anyone resumes */
    return g_ballot;
}
void wave_sync()
{
    g_dep[g_lane] = 0;
    yield_to_scheduler();
}
static long long g_snap[W]; /* the deposits of the last completed round */
long long wave_exchange(long long v, int src)
{
    g_dep[g_lane] = v;
    yield_to_scheduler();
    return g_snap[src];
}
static void trampoline()
{
    g_body(g_lane, g_ctx);
    g_done[g_lane] = true;
    swapcontext(&g_fiber[g_lane], &g_main);
}
/* run body(lane, ctx) for the 64 lanes of one wave */
static void run_wave(void (*body)(int, void *), void *ctx)
{
    const size_t STK = 256 * 1024;
    if (g_stacks.empty()) g_stacks.resize(STK * W);
    g_body = body; g_ctx = ctx;
    for (int l = 0; l < W; ++l) {
        getcontext(&g_fiber[l]);
        g_fiber[l].uc_stack.ss_sp = g_stacks.data() + STK * l;
        g_fiber[l].uc_stack.ss_size = STK;
        g_fiber[l].uc_link = &g_main;
        makecontext(&g_fiber[l], trampoline, 0);
        g_done[l] = false;
    }
    for (;;) {
        bool any = false;
        for (int l = 0; l < W; ++l) {
            if (g_done[l]) continue;
            any = true;
            g_lane = l;
            swapcontext(&g_main, &g_fiber[l]);
        }
        if (!any) break;
        unsigned long long b = 0;
        for (int l = 0; l < W; ++l) {
            if (!g_done[l] && g_dep[l]) b |= 1ull << l;
            g_snap[l] = g_dep[l];
        }
        g_ballot = b;
    }
    g_lane = -1;
}
} /* namespace sasa_emu */

namespace sasa { long long sr_caps_count_emu[4]; }
static int emu_sr_caps_n = SR_CAP_N_DEFAULT, emu_sr_caps_l = SR_CAP_L_DEFAULT; /* S&R: the third arrangement's table (0: the second arrangement) */
extern "C" void emu_sr_caps_counts(long long *out, int reset) { for (int k = 0; k < 4; ++k) { out[k] = sasa::sr_caps_count_emu[k]; if (reset) sasa::sr_caps_count_emu[k] = 0; } }
/* the table itself, for tests/test_sr_caps.py: 6 N^2 L entries of 8 words (DEF then BAND); 0 if these points get no table */
extern "C" int emu_sr_captab(const double *unit, int np, int n, int l, unsigned *out)
{
    std::vector<SrCapEntry> t;
    if (!sr_captab_build(unit, np, n, l, t)) return 0;
    memcpy(out, t.data(), sizeof(SrCapEntry) * t.size());
    return (int)t.size();
}
extern "C" int emu_sr_cap_cell(float vx, float vy, float vz, int n) { return sr_cap_cell(vx, vy, vz, n); }
extern "C" int emu_sr_cap_level(float g, int l) { return sr_cap_level(g, l); }
extern "C" void emu_set_sr_caps(int n, int l) { emu_sr_caps_n = n; emu_sr_caps_l = l; }
static bool emu_bucket = true; /* the BUCKET kernel variant; emu_set_bucket(0) emulates the plain one */
extern "C" void emu_set_bucket(int on) { emu_bucket = on != 0; }

template <bool GLOBAL>
static void emu_tile_kernel(bool lr, const TileCfg &cfg, TileArgs a, int grid)
{
    const int B = cfg.B;
    std::vector<char> smem(cfg.lds + 64);
    for (int blk = 0; blk < grid; ++blk) {
        TileMem m = tile_carve<GLOBAL>(a, smem.data(), cfg.items, B, blk);
        const int n_work = a.work_tiles ? *a.work_count : ((a.n_tiles + 7) >> 3) << 3;
        std::vector<int> wg_max_nn(B, 0);
        for (int w = blk; w < n_work; w += grid) {
            const int tile = a.work_tiles ? a.work_tiles[w] : xcd_tile(w, a.n_tiles);
            if (tile >= a.n_tiles) continue;
#define PHASE(call) for (int tid = 0; tid < B; ++tid) { call; }
            if (!lr) { /* Shrake-Rupley: records written by the neighbor phase, no offsets / pairs phases (gpu_kernels.hip, k_sr_tile) */
                PHASE(sr_phase_load(a, m, tile, tid, B));
                PHASE(sr_phase_neighbors(a, m, tile, tid, B));
                const bool caps = !GLOBAL && a.captab && a.tab && sr_order_in_wave(a.cap_idx) && B >= 64; /* third arrangement (sr_caps.h), as launch_sr picks it */
                if (caps) { PHASE(sr_caps_lists(a, m, tid)); } else { PHASE(sr_phase_lists(a, m, tid)); }
                PHASE(sr_report<GLOBAL>(a, m, tile, tid, wg_max_nn[tid]));
                if (caps) {
                    PHASE(sr_caps_clear(a, m, tile, tid, B)); /* (the device clears beside the load phase: the words are not read before) */
                    std::vector<SrCapRegs> regs(B);
                    PHASE(sr_caps_lookup_pass(a, m, tid, B, regs[tid]));
                    PHASE(sr_caps_todo_pass(a, m, tid, B, cfg.items, regs[tid]));
                    PHASE(sr_caps_exact(a, m, tid, B, cfg.items));
                    PHASE(sr_caps_store(a, m, tile, tid));
                    continue;
                }
                PHASE(sr_order_serial(a, m, tid)); /* (the device orders a list with one wave's ballots: the same partition up to the order inside each part, which no count depends on) */
                PHASE(sr_phase_points(a, m, tile, tid, B));
                PHASE(sr_phase_points2(a, m, tid, B));
                PHASE(sr_phase_store(a, m, tile, tid));
                continue;
            }
            PHASE(tile_phase_load(a, m, tile, tid, B, lr && emu_bucket));
            PHASE(tile_phase_neighbors(a, m, tile, tid, B));
            PHASE(tile_phase_offsets(a, m, tid));
            if (lr) {
                PHASE(tile_report<GLOBAL>(a, m, tile, tid, wg_max_nn[tid]); lr_phase_beta(a, m, tid, B, emu_bucket));
                if (emu_bucket && lr_bucket_path(a, m, B)) {
                    std::vector<RankRegs> rrs(B);
                    PHASE(lr_phase_prefix(a, m, tid));
                    PHASE(lr_phase_scatter(a, m, tid, B));
                    PHASE(lr_phase_rank2(a, m, tid, B, rrs[tid]));
                    PHASE(lr_phase_write(a, m, tid, B, rrs[tid]));
                } else {
                    PHASE(lr_phase_rank(a, m, tid, B));
                }
                PHASE(lr_phase_slices(a, m, tile, tid, B));
                PHASE(lr_phase_store<GLOBAL>(a, m, tile, tid, B));
            }
        }
        PHASE(tile_report_flush(a, tid, wg_max_nn[tid]));
#undef PHASE
    }
}

/* second-generation L&R kernel: one wave (64 fibers) per tile */
static int emu_lr2 = 1, emu_lr2_ta = 0, emu_lr2_refill = 0;
extern "C" void emu_lr2_counts(long long *out, int reset) { for (int k = 0; k < 16; ++k) { out[k] = sasa_emu::lr2_count[k]; if (reset) sasa_emu::lr2_count[k] = 0; } }
extern "C" void emu_set_lr2(int on, int ta, int refill) { emu_lr2 = on; emu_lr2_ta = ta; emu_lr2_refill = refill; }

/* emu_set_lr2_opts: bit 0 = run the shape-specialised builds where the device would (launch_lr2_main: template parameter
   SHAPE 1..4, with the COVER / PAIRS / RMAX combination each is launched with); bit 1 = the COMPACT cell table
   (k_sort_struct's form: a bit per cell, occupied cells before each 32-cell word, first atoms of the occupied cells)
   instead of the dense one, for single-structure batches (whose cell numbering starts at a multiple of 32 either way) */
static int emu_lr2_opts = 0, emu_last_shape = 0, emu_last_compact = 0;
extern "C" void emu_set_lr2_opts(int opts) { emu_lr2_opts = opts; }
extern "C" int emu_last_lr2_variant(void) { return emu_last_shape | (emu_last_compact << 4); } /* what the last main launch ran: SHAPE | compact table << 4 */
struct Lr2Run { const Lr2Args *a; Lr2Mem *m; int first, stride; int rmax; int *wg_max; int shape; };
static void lr2_lane_body(int lane, void *ctx)
{
    Lr2Run *r = (Lr2Run *)ctx;
    const bool pairs = r->rmax == LR2_RMAX_MAIN && lr2_pairs_shape(r->a->TA, r->a->ns); /* (as launch_lr2_main) */
    if (r->shape == 1) lr2_wave<LR2_RMAX_MAIN, false, true, 1>(*r->a, *r->m, r->first, r->stride, lane, r->wg_max[lane]);
    else if (r->shape == 2) lr2_wave<2, false, false, 2>(*r->a, *r->m, r->first, r->stride, lane, r->wg_max[lane]);
    else if (r->shape == 3) lr2_wave<LR2_RMAX_MAIN, true, false, 3>(*r->a, *r->m, r->first, r->stride, lane, r->wg_max[lane]);
    else if (r->shape == 4) lr2_wave<LR2_RMAX_MAIN, true, true, 4>(*r->a, *r->m, r->first, r->stride, lane, r->wg_max[lane]);
    else if (r->rmax == LR2_RMAX_MAIN && pairs) lr2_wave<LR2_RMAX_MAIN, true, true>(*r->a, *r->m, r->first, r->stride, lane, r->wg_max[lane]);
    else if (r->rmax == LR2_RMAX_MAIN) lr2_wave<LR2_RMAX_MAIN, true, false>(*r->a, *r->m, r->first, r->stride, lane, r->wg_max[lane]);
    else lr2_wave<LR2_RMAX_MID, true, false, 0, false, true>(*r->a, *r->m, r->first, r->stride, lane, r->wg_max[lane]); /* (the second launch: WALK) */
}
static void emu_lr2_kernel(const Lr2Cfg &cfg, Lr2Args a, int grid, bool main_launch = false)
{
    std::vector<char> smem(cfg.lds + 64);
    int shape = 0;
    if (main_launch && (emu_lr2_opts & 1)) { /* the build launch_lr2_main would pick */
        const int sid = lr2_shape_id(a.TA, a.ns, a.mw, a.ds);
        if (sid == 1 && cfg.rmax == LR2_RMAX_MAIN && lr2_pairs_shape(a.TA, a.ns)) shape = 1;
        if (sid == 2 && cfg.rmax <= 2) shape = 2;
        if ((sid == 3 || sid == 4) && cfg.rmax == LR2_RMAX_MAIN) shape = sid;
    }
    if (main_launch) { emu_last_shape = shape; emu_last_compact = a.cell_tbl != nullptr; }
    for (int blk = 0; blk < grid; ++blk) {
        Lr2Mem m = shape == 1 ? lr2_carve<1>(a, smem.data()) : shape == 2 ? lr2_carve<2>(a, smem.data()) : shape == 3 ? lr2_carve<3>(a, smem.data())
                 : shape == 4 ? lr2_carve<4>(a, smem.data()) : lr2_carve<0>(a, smem.data());
        std::vector<int> wg_max(64, 0);
        Lr2Run run = {&a, &m, blk, grid, cfg.rmax, wg_max.data(), shape};
        sasa_emu::run_wave(lr2_lane_body, &run);
        if (wg_max[0] > a.status[ST_MAX_NN]) a.status[ST_MAX_NN] = wg_max[0];
    }
}

/* force_*: <= 0 keeps the production launch configuration; positive values shrink the
 * capacities so that small inputs exercise the overflow -> fallback path.
 * stats_out[10]: error, fallback tiles, max nn, TA, B, lds, total cells, items, slices that
 * needed the exact path, filtered-vs-exact cross-check failures. */
extern "C" int emu_run_batch(int lr, const double *xyz, const double *radii, const int64_t *offsets,
                             int n_structs, double probe, int resolution, const double *unit_pts,
                             double *sasa, int *counts, double *totals, long long *stats_out,
                             int force_cap_idx, int force_pool, int force_ds, int fb_cap_idx,
                             int fb_pool, int fb_ds, int mid_cap_idx, int mid_pool, int mid_ds)
{
    const int n = (int)offsets[n_structs];
    const int PB = SASA_PIPE_B;
    std::vector<GridS> grid(n_structs);
    std::vector<long long> ncells(n_structs + 1);
    std::vector<int> sid(n), rank(n), status(ST_WORDS, 0);
    std::vector<long long> cell_of(n);
    std::vector<SortIdx> s_idx(n);
    std::vector<Quad> sq(n);

    PipeArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.xyz = xyz; pa.radii = radii; pa.offsets = offsets; pa.n_structs = n_structs; pa.n_atoms = n;
    pa.probe = probe; pa.max_cells = 1LL << 28;
    pa.grid = grid.data(); pa.ncells = ncells.data(); pa.sid = sid.data(); pa.cell_of = cell_of.data();
    pa.rank = rank.data(); pa.sq = sq.data();
    pa.s_idx = s_idx.data(); pa.status = status.data(); pa.cells_total = (long long *)(status.data() + ST_CELLS);

    /* chunk table, as gpu_engine.hip builds it */
    std::vector<int> cs, cl, sc0(n_structs + 1);
    std::vector<int64_t> cb;
    for (int s = 0; s < n_structs; ++s) {
        sc0[s] = (int)cs.size();
        for (int64_t b = offsets[s]; b < offsets[s + 1]; b += SASA_BOUNDS_CHUNK) {
            const int64_t e = b + SASA_BOUNDS_CHUNK < offsets[s + 1] ? b + SASA_BOUNDS_CHUNK : offsets[s + 1];
            cs.push_back(s); cb.push_back(b); cl.push_back((int)(e - b));
        }
    }
    sc0[n_structs] = (int)cs.size();
    std::vector<double> bpart(7 * cs.size() + 7);
    pa.n_chunks = (int)cs.size(); pa.chunk_struct = cs.data(); pa.chunk_begin = cb.data(); pa.chunk_len = cl.data();
    pa.struct_chunk0 = sc0.data(); pa.bpart = bpart.data();
    { /* k_bounds + k_grid */
        std::vector<double> red(7 * PB), red2(7 * 16);
        for (int ch = 0; ch < pa.n_chunks; ++ch) {
            for (int t = 0; t < PB; ++t) bounds_phase0(pa, red.data(), ch, t, PB);
            for (int t = 0; t < PB; ++t) bounds_phase1(red.data(), red2.data(), t, PB);
            for (int t = 0; t < PB; ++t) bounds_phase2(pa, red2.data(), ch, t);
        }
        std::vector<double> gpart(7 * PB);
        for (int s0 = 0; s0 < n_structs; s0 += PB / SASA_GRID_GROUP) {
            for (int t = 0; t < PB; ++t) grid_phase0(pa, gpart.data(), s0, t);
            for (int t = 0; t < PB; ++t) grid_phase1(pa, gpart.data(), s0, t);
        }
    }
    { /* k_cell_base */
        std::vector<long long> part(PB), part2(16);
        for (int t = 0; t < PB; ++t) cellbase_phase0(pa, part.data(), t, PB);
        for (int t = 0; t < PB; ++t) cellbase_phase1(part.data(), part2.data(), t, PB);
        for (int t = 0; t < PB; ++t) cellbase_phase1b(pa, part2.data(), t);
        for (int t = 0; t < PB; ++t) cellbase_phase2(pa, part.data(), part2.data(), t, PB);
    }
    stats_out[0] = status[ST_ERROR];
    if (status[ST_ERROR]) return -1;
    const long long total_cells = *cell_total(pa);
    const int nblk_scan = (int)((total_cells + 1 + (long long)PB * SASA_SCAN_ITEMS - 1) / ((long long)PB * SASA_SCAN_ITEMS));
    std::vector<int> cell_start(total_cells + 2, 0);
    std::vector<unsigned long long> scan_desc(nblk_scan + 1, 0ULL); /* (fresh descriptors are cleared once: gpu_engine.hip) */
    pa.cell_start = cell_start.data(); pa.scan_desc = scan_desc.data(); pa.scan_epoch = 1;

    const int nblk_atoms = (n + PB - 1) / PB;
    for (int b = 0; b < nblk_atoms; ++b)
    {
        std::vector<int> cells(PB), base(PB);
        for (int t = 0; t < PB; ++t) count_phase0(pa, cells.data(), b * PB + t, t);
        for (int t = 0; t < PB; ++t) count_phase1(pa, cells.data(), base.data(), t, PB);
        for (int t = 0; t < PB; ++t) count_phase2(pa, cells.data(), base.data(), b * PB + t, t);
    }
    { /* k_scan: the chained scan, its workgroups one after the other (every look-back finds its predecessors done) */
        std::vector<int> part(PB), part2(SASA_SCAN_GROUP);
        std::vector<ScanRegs> regs(PB);
        for (int w = 0; w < nblk_scan; ++w) {
            const int b = scan_take_block(pa);
            int before = 0;
            for (int t = 0; t < PB; ++t) scan_phase0(pa, total_cells, part.data(), b, t, PB, regs[t]);
            for (int t = 0; t < PB; ++t) scan_phase1(part.data(), part2.data(), t);
            for (int t = 0; t < PB; ++t) scan_phase2(pa, part2.data(), &before, b, t);
            for (int t = 0; t < PB; ++t) scan_phase3(pa, total_cells, part.data(), part2.data(), &before, b, t, PB, regs[t]);
        }
    }
    for (int b = 0; b < nblk_atoms; ++b)
        for (int t = 0; t < PB; ++t) scatter_atom(pa, b * PB + t);

    if (lr && emu_lr2 && lr2_supported(resolution)) {
        /* main and second launch: lr2_kernels.h; third launch: the slab-backed first-generation kernel
           over the same tiling */
        Lr2Cfg c2 = lr2_choose_cfg(resolution, 0, emu_lr2_ta);
        if (force_pool > 0) c2.pool = force_pool < LR2_LANES * LR2_RMAX_MAIN ? (force_pool + 1) & ~1 : LR2_LANES * LR2_RMAX_MAIN;
        if (force_cap_idx > 0) c2.mw = (force_cap_idx + 31) / 32;
        if (force_ds >= 0) c2.ds = force_ds;
        if (emu_lr2_refill > 0) c2.refill = emu_lr2_refill;
        c2.lds = lr2_layout(c2.TA, c2.ns, c2.pool, c2.mw, c2.ds).total;
        const int n_tiles2 = (n + c2.TA - 1) / c2.TA;
        std::vector<long long> ovf2x(2 * n_tiles2 + 2);
        std::vector<int> ovf3(n + 8);
        Lr2Args la;
        memset(&la, 0, sizeof la);
        la.sq = pa.sq; la.s_idx = pa.s_idx;
        la.grid = pa.grid; la.cell_start = pa.cell_start; la.n_atoms = n; la.n_tiles = n_tiles2;
        la.TA = c2.TA; la.ns = resolution; la.pool = c2.pool; la.mw = c2.mw; la.ds = c2.ds; la.refill = c2.refill; la.inv_ns = 1.0 / (double)resolution;
        la.cover = getenv("EMU_LR2_COVER") ? atoi(getenv("EMU_LR2_COVER")) : LR2_COVER_DENSITY;
        la.prune = lr2_prune_arg(getenv("EMU_LR2_PRUNE") ? atoi(getenv("EMU_LR2_PRUNE")) : 4, c2.TA, c2.pool); /* (contained caps, P1.5: caps wanted per atom; 0: off) */
        la.sasa = sasa; la.status = status.data();
        /* main launch: a tile that does not fit is split in place; halves that still do not fit go to the list */
        la.ovf_items = ovf2x.data(); la.ovf_count = status.data() + ST_OVF2_TILES; la.split_count = status.data() + ST_SPLIT;
        /* the compact cell table, built from the dense one the way k_sort_struct's stage F leaves it (single structure) */
        std::vector<unsigned long long> ctbl;
        std::vector<int> cfirst;
        if ((emu_lr2_opts & 2) && n_structs == 1) {
            const long long C = ncells[0];
            ctbl.assign((size_t)((C + 1 + 31) >> 5) + 1, 0ull);
            int occ = 0;
            for (long long w = 0; w * 32 < C; ++w) {
                unsigned bits = 0;
                const int before = occ;
                for (int b = 0; b < 32 && w * 32 + b < C; ++b)
                    if (cell_start[w * 32 + b + 1] > cell_start[w * 32 + b]) { bits |= 1u << b; cfirst.push_back(cell_start[w * 32 + b]); ++occ; }
                ctbl[(size_t)w] = (unsigned long long)bits | ((unsigned long long)(unsigned)before << 32);
            }
            cfirst.push_back(n); /* the entry behind the last cell: the structure's end */
            if ((C & 31) == 0) ctbl[(size_t)(C >> 5)] = (unsigned long long)(unsigned)occ << 32; /* (cell C opens a word of its own) */
            la.cell_tbl = ctbl.data(); la.cell_first = cfirst.data();
        }
        emu_lr2_kernel(c2, la, ((n_tiles2 + 7) / 8) * 8, true);
        Lr2Cfg cm = lr2_mid_cfg(c2);
        if (mid_cap_idx > 0) cm.mw = (mid_cap_idx + 31) / 32;
        if (mid_pool > 0) cm.pool = mid_pool;
        if (mid_ds >= 0) cm.ds = mid_ds;
        cm.lds = lr2_layout(cm.TA, cm.ns, cm.pool, cm.mw, cm.ds).total;
        {
            Lr2Args lm = la;
            lm.pool = cm.pool; lm.mw = cm.mw; lm.ds = cm.ds;
            lm.work_items = ovf2x.data(); lm.work_count = status.data() + ST_OVF2_TILES;
            lm.ovf_items = nullptr; lm.ovf_atoms = ovf3.data(); lm.ovf_count = status.data() + ST_OVF3_ATOMS;
            lm.split_count = nullptr;
            emu_lr2_kernel(cm, lm, 5);
        }
        {
            TileCfg fb = fallback_cfg(lr_slab_cfg(1, resolution), true);
            if (fb_cap_idx > 0) fb.cap_idx = fb_cap_idx;
            if (fb_pool > 0) fb.pool = fb_pool;
            if (fb_ds > 0) fb.ds = fb_ds;
            const size_t stride = tile_slab_bytes(fb.TA, fb.cap_idx, fb.pool, fb.lr, fb.ds, fb.B);
            const int fb_blocks = 3;
            std::vector<char> slab(stride * fb_blocks + 64);
            TileArgs tf;
            memset(&tf, 0, sizeof tf);
            tf.sq = pa.sq; tf.s_idx = pa.s_idx;
            tf.grid = pa.grid; tf.cell_start = pa.cell_start; tf.cell_tbl = la.cell_tbl; tf.cell_first = la.cell_first;
            tf.n_atoms = n; tf.n_tiles = n; tf.TA = 1; tf.n_res = resolution; tf.tab = fb.tab;
            tf.sasa = sasa; tf.lr = 1; tf.status = status.data();
            tf.cap_idx = fb.cap_idx; tf.pool = fb.pool; tf.ds = fb.ds;
            tf.work_tiles = ovf3.data(); tf.work_count = status.data() + ST_OVF3_ATOMS;
            tf.slab = slab.data(); tf.slab_stride = (long long)stride;
            emu_tile_kernel<true>(true, fb, tf, fb_blocks);
        }
        if (totals) {
            double part[SASA_TOT_B];
            std::vector<double> chunk_tot(pa.n_chunks > 0 ? pa.n_chunks : 1);
            for (int ch = 0; ch < pa.n_chunks; ++ch) {
                for (int l = 0; l < SASA_TOT_B; ++l) totals_chunk_phase0(pa, sasa, part, ch, l);
                for (int l = 0; l < SASA_TOT_B; ++l) totals_chunk_phase1(part, chunk_tot.data(), ch, l);
            }
            for (int s = 0; s < ((n_structs + 255) / 256) * 256; ++s) totals_struct(pa, chunk_tot.data(), totals, s);
        }
        long long splits = 0;
        for (int k = 0; k < 64; ++k) splits += status[ST_SPLIT + k];
        stats_out[0] = status[ST_ERROR]; stats_out[1] = splits; stats_out[2] = status[ST_MAX_NN];
        stats_out[3] = c2.TA; stats_out[4] = 64; stats_out[5] = (long long)c2.lds; stats_out[6] = total_cells;
        stats_out[7] = c2.TA * resolution;
        stats_out[8] = status[ST_OVF3_ATOMS]; stats_out[9] = status[ST_OVF2_TILES];
        return status[ST_ERROR] ? -1 : 0;
    }
    TileCfg cfg = choose_cfg(resolution, lr != 0, 0, !lr && emu_sr_caps_n > 0 && resolution <= SR_CAP_POINTS_MAX);
    if (force_cap_idx > 0) cfg.cap_idx = force_cap_idx;
    if (force_pool > 0) cfg.pool = force_pool;
    if (force_ds >= 0 && lr) cfg.ds = force_ds;
    if (!lr) cfg.pool = cfg.TA * cfg.cap_idx; /* (S&R: a segment of cap_idx records per atom) */
    cfg.lds = tile_fixed_bytes(cfg.TA, cfg.items) + tile_list_bytes(cfg.TA, cfg.cap_idx, cfg.pool, cfg.lr, cfg.ds, cfg.B);
    const int n_tiles = (n + cfg.TA - 1) / cfg.TA;
    std::vector<int> ovf_tiles(n_tiles + 1);

    TileArgs ta;
    memset(&ta, 0, sizeof ta);
    ta.sq = pa.sq;
    ta.s_idx = pa.s_idx;
    ta.grid = pa.grid; ta.cell_start = pa.cell_start;
    ta.n_atoms = n; ta.n_tiles = n_tiles; ta.TA = cfg.TA; ta.n_res = resolution; ta.tab = cfg.tab;
    ta.unit_pts = unit_pts; ta.sasa = sasa; ta.counts = counts;
    ta.cap_idx = cfg.cap_idx; ta.pool = cfg.pool; ta.lr = cfg.lr; ta.ds = cfg.ds;
    ta.ovf_count = status.data() + ST_OVF_TILES; ta.ovf_tiles = ovf_tiles.data(); ta.status = status.data();
    std::vector<SrCapEntry> captab;
    if (!lr && emu_sr_caps_n > 0 && sr_captab_build(unit_pts, resolution, emu_sr_caps_n, emu_sr_caps_l, captab)) {
        ta.captab = captab.data(); ta.cap_n = emu_sr_caps_n; ta.cap_l = emu_sr_caps_l;
    }

    ta.work_tiles = nullptr; ta.work_count = nullptr;
    emu_tile_kernel<false>(lr != 0, cfg, ta, ((n_tiles + 7) / 8) * 8);

    std::vector<int> ovf_tiles2(n_tiles + 1);
    {
        TileCfg mc = mid_cfg(cfg, lr != 0);
        if (mid_cap_idx > 0) mc.cap_idx = mid_cap_idx;
        if (mid_pool > 0) mc.pool = mid_pool;
        if (mid_ds >= 0 && lr) mc.ds = mid_ds;
        if (!lr) mc.pool = mc.TA * mc.cap_idx;
        mc.lds = tile_fixed_bytes(mc.TA, mc.items) + tile_list_bytes(mc.TA, mc.cap_idx, mc.pool, mc.lr, mc.ds, mc.B);
        TileArgs tm = ta;
        tm.cap_idx = mc.cap_idx; tm.pool = mc.pool; tm.ds = mc.ds;
        tm.work_tiles = ovf_tiles.data(); tm.work_count = status.data() + ST_OVF_TILES;
        tm.ovf_tiles = ovf_tiles2.data(); tm.ovf_count = status.data() + ST_OVF2_TILES;
        emu_tile_kernel<false>(lr != 0, mc, tm, 5); /* fewer blocks than work items: the loop wraps */
    }
    {
        TileCfg fb = fallback_cfg(cfg, lr != 0);
        if (fb_cap_idx > 0) fb.cap_idx = fb_cap_idx;
        if (fb_pool > 0) fb.pool = fb_pool;
        if (fb_ds > 0 && lr) fb.ds = fb_ds;
        if (!lr) fb.pool = fb.TA * fb.cap_idx;
        const size_t stride = tile_slab_bytes(fb.TA, fb.cap_idx, fb.pool, fb.lr, fb.ds, fb.B);
        const int fb_blocks = 3; /* fewer than SASA_FB_BLOCKS so that the work loop wraps */
        std::vector<char> slab(stride * fb_blocks + 64);
        TileArgs tf = ta;
        tf.cap_idx = fb.cap_idx; tf.pool = fb.pool; tf.ds = fb.ds;
        tf.work_tiles = ovf_tiles2.data(); tf.work_count = status.data() + ST_OVF2_TILES;
        tf.ovf_tiles = nullptr; tf.ovf_count = nullptr;
        tf.slab = slab.data(); tf.slab_stride = (long long)stride;
        if (tf.captab) { tf.captab = nullptr; tf.tab = 0; } /* (as the engine: the slab launch runs the second arrangement without its survivor table) */
        emu_tile_kernel<true>(lr != 0, fb, tf, fb_blocks);
    }
    if (totals) {
        double part[SASA_TOT_B];
        std::vector<double> chunk_tot(pa.n_chunks > 0 ? pa.n_chunks : 1);
        for (int ch = 0; ch < pa.n_chunks; ++ch) {
            for (int l = 0; l < SASA_TOT_B; ++l) totals_chunk_phase0(pa, sasa, part, ch, l);
            for (int l = 0; l < SASA_TOT_B; ++l) totals_chunk_phase1(part, chunk_tot.data(), ch, l);
        }
        for (int s = 0; s < ((n_structs + 255) / 256) * 256; ++s) totals_struct(pa, chunk_tot.data(), totals, s);
    }

    stats_out[0] = status[ST_ERROR]; stats_out[1] = status[ST_OVF_TILES]; stats_out[2] = status[ST_MAX_NN];
    stats_out[3] = cfg.TA; stats_out[4] = cfg.B; stats_out[5] = (long long)cfg.lds; stats_out[6] = total_cells;
    stats_out[7] = cfg.items;
    stats_out[8] = status[ST_OVF2_TILES]; stats_out[9] = 0;
    return status[ST_ERROR] ? -1 : 0;
}

/* scalar device math helpers, exposed for unit tests */
/* the aggregation kernels' phase functions (per-residue and per-class sums) */
/* host-side launch shaping of the L&R kernel (lr2_kernels.h), for tests/test_emulation.py */
extern "C" void emu_lr2_shape(int ns, double nn_hint, int nn_max_hint, int last_ta, double last_split, int *out /* TA, pool, mw, ds, lds, rmax */)
{
    const Lr2Cfg c = lr2_choose_cfg(ns, nn_hint, 0, nn_max_hint, last_ta, last_split);
    out[0] = c.TA; out[1] = c.pool; out[2] = c.mw; out[3] = c.ds; out[4] = c.lds; out[5] = c.rmax;
}
extern "C" int emu_lr2_pool_from_hist(const int *hist, int TA, int ns, int mw, int ds) { return lr2_pool_from_hist(hist, TA, ns, mw, ds); }
extern "C" int emu_lr2_lds(int TA, int ns, int pool, int mw, int ds) { return lr2_layout(TA, ns, pool, mw, ds).total; }
extern "C" int emu_lr2_prune(int ns, int dense, int TA, int pool) { return lr2_prune_arg(lr2_prune_want(ns, dense != 0), TA, pool); } /* what gpu_engine.hip puts into Lr2Args::prune */

extern "C" void emu_segsum_small(const double *sasa, const int64_t *seg, int n_segs, double *out)
{
    for (int k = 0; k < ((n_segs + 255) / 256) * 256; ++k) segsum_small(sasa, seg, out, k, n_segs);
}
extern "C" void emu_class_sums(const double *sasa, const unsigned char *cls, const int64_t *offsets, int n_structs, double *out)
{
    std::vector<double> part(3 * SASA_TOT_B);
    for (int s = 0; s < n_structs; ++s) {
        for (int l = 0; l < SASA_TOT_B; ++l) class_phase0(sasa, cls, offsets, part.data(), s, l);
        for (int l = 0; l < SASA_TOT_B; ++l) class_phase1(part.data(), out, s, l);
    }
}

extern "C" void emu_residue_areas(const double *sasa, const unsigned char *cls, const unsigned char *bb, const int64_t *res_first,
                                  const short *ref_row, const double *ref_table, int n_res, double *abs_out, double *rel_out)
{
    for (int r = 0; r < ((n_res + 255) / 256) * 256; ++r) residue_areas(sasa, cls, bb, res_first, ref_row, ref_table, abs_out, rel_out, r, n_res);
}

extern "C" void emu_atan2_inv(const double *y, const double *x, double *out, int n)
{
    for (int i = 0; i < n; ++i) {
        double g, h;
        const double D = x[i] * x[i] + y[i] * y[i];
        if (D > 0) { sqrt_rh(D, g, h); out[i] = atan2_inv(y[i], x[i], 2.0 * h); } else out[i] = 0;
    }
}
extern "C" void emu_acos_fast2(const double *x, double *out, int n)
{
    for (int i = 0; i < n; ++i) out[i] = acos_fast2(x[i]);
}
extern "C" void emu_acos_fast(const double *x, double *out, int n)
{
    for (int i = 0; i < n; ++i) out[i] = acos_fast(x[i]);
}
extern "C" void emu_sqrt_rh(const double *x, double *g, double *h, int n)
{
    for (int i = 0; i < n; ++i) sqrt_rh(x[i], g[i], h[i]);
}
extern "C" void emu_atan2_fast(const double *y, const double *x, double *out, int n)
{
    for (int i = 0; i < n; ++i) out[i] = atan2_fast(y[i], x[i]);
}

/* the arc union and sweep of the Lee-Richards kernel on sets of arcs given in the order of their mid-points
   (what freesasa_gpu_arc_union_dev runs on the device); raw end points allowed (start < 0, end > 2 pi) */
extern "C" void emu_arc_union(const double *arcs, const int *first, int n_sets, int ds, double *out)
{
    std::vector<Arc2> stack((size_t)(ds > 0 ? ds : 1) * LR2_LANES);
    for (int k = 0; k < n_sets; ++k) out[k] = lr2_arc_kat(arcs, first, k, stack.data(), ds);
}
