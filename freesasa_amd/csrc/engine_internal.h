/*
 * engine_internal.h — what the translation units of the engine share (never installed; the public surface is
 * include/freesasa_gpu.h):
 *
 *   gpu_kernels.hip    every __global__ wrapper around the phase functions of sasa_kernels.h / lr2_kernels.h and the
 *                      kl_* launchers below (the only file that holds device code)
 *   gpu_engine.hip     the per-device context (workspace, status words, events), the launch sequence of one batch,
 *                      asynchronous batches, the device-pointer entry points
 *   gpu_ops.hip        device-side aggregates (segments, classes, residues) and the kernel test hooks
 *   gpu_hostbatch.hip  host-pointer batches: the context pool, one device, several devices, the pipelined form
 *   gpu_drivers.hip    file sweep and trajectory drivers (one device or a list of devices), done-lists
 */
#ifndef FREESASA_AMD_ENGINE_INTERNAL_H
#define FREESASA_AMD_ENGINE_INTERNAL_H

#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stddef.h>
#include <thread>
#include <vector>

#include "../../include/freesasa_gpu.h"
#include "../../include/freesasa_ingest.h"
#include "sasa_kernels.h"
#include "lr2_kernels.h"

/* ------------------------------------------------------------------ kernel launchers (gpu_kernels.hip) */

/* k_sort_struct: one workgroup sorts one structure in LDS */
#define SORT_B 1024
#define SORT_APT 16
#define SORT_ATOMS (SORT_B * SORT_APT - 256) /* atoms of a structure (the 256 short of 16 threads' worth: two workgroups' LDS per CU) */
#define SORT_CELLS (1 << 18) /* cells in LDS at a time (a bit each); a structure with more is done in that many passes */
#define SORT_WORDS (SORT_CELLS / 32)
#define SORT_CELL_BITS 26    /* cells of one structure this kernel can number (6 more bits hold the border flags) */

/* cell sort of a batch: per structure in one workgroup (batches of small structures), or the general pipeline
   (bounds, grid, cell numbering, histogram, scan, scatter: any structure size) */
hipError_t kl_prep_fused(const sasa::PipeArgs &pa, hipStream_t st);
hipError_t kl_prep_general(const sasa::PipeArgs &pa, long long cells_cap, hipStream_t st);
/* Lee-Richards, second generation: main launch (the build is picked by tile shape and pair-record rounds), second
   launch (larger LDS lists, more registers) */
hipError_t kl_lr2_main(int rmax, int grid, size_t lds, hipStream_t st, const sasa::Lr2Args &la);
hipError_t kl_lr2_mid(int grid, size_t lds, hipStream_t st, const sasa::Lr2Args &la);
/* first-generation tile kernels: tier 0 main launch, 1 second launch, 2 last launch (lists in a global slab) */
hipError_t kl_lr_tile(int tier, const sasa::TileCfg &c, const sasa::TileArgs &t, int grid, size_t lds, hipStream_t st, bool bucket);
hipError_t kl_sr_tile(int tier, const sasa::TileCfg &c, const sasa::TileArgs &t, int grid, size_t lds, hipStream_t st);
/* per-structure totals in two levels (chunk partials in `bpart`) */
hipError_t kl_totals(const sasa::PipeArgs &pa, int n_chunks, int n_structs, const double *d_sasa, double *bpart, double *d_totals, hipStream_t st);
hipError_t kl_segment_sums(const double *d_sasa, const int64_t *d_seg, int n_segs, bool short_segments, double *d_out, hipStream_t st);
hipError_t kl_class_sums(const double *d_sasa, const unsigned char *d_class, const int64_t *d_offsets, int n_structs, double *d_out, hipStream_t st);
hipError_t kl_residue_areas(const double *d_sasa, const unsigned char *d_class, const unsigned char *d_backbone, const int64_t *d_res_first,
                            const short *d_ref_row, const double *d_ref_table, double *d_abs, double *d_rel, int n_res, hipStream_t st);
hipError_t kl_arc_kat(const double *d_arcs, const int *d_first, int n_sets, double *d_out, hipStream_t st);
hipError_t kl_widen_f32(const float *d_in, double *d_out, long long n, hipStream_t st);
hipError_t kl_narrow_f64(const double *d_in, float *d_out, long long n, hipStream_t st);
void kl_dump_phase_clocks(void); /* (dev builds with -DSASA_PHASE_TIMING; else nothing) */

/* ------------------------------------------------------------------ context (gpu_engine.hip) */

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct freesasa_gpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool timing = false;
    bool shared_radii = false; /* d_radii holds ONE structure's radii (trajectory frames) */
    char err[512] = {0};
    freesasa_gpu_stats stats = {};
    /* Two sets of what the HOST reads of a batch (page-locked status words, stage events, end-of-batch event): a batch
       submitted with freesasa_gpu_lr_batch_dev_async leaves its set behind until it is collected, while the next one
       is enqueued with the other set.  The device side needs no second copy: the copies into a set are enqueued at
       the end of their batch, in stream order before the next batch resets the device words. */
    int slot = 0;
    hipEvent_t evs[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
    hipEvent_t done[2] = {nullptr, nullptr};
    struct Pend {
        bool active = false;
        /* the call, for the rare batch that has to be redone (cell table sizing, see RC_RETRY) */
        const double *d_xyz = nullptr, *d_radii = nullptr;
        std::vector<int64_t> offsets;
        int n_structs = 0, resolution = 0;
        double probe = 0;
        double *d_sasa = nullptr, *d_totals = nullptr;
        /* what completing it needs */
        int n = 0, TA = 0, mw = 0, ds = 0, lds = 0;
        bool walk = false; /* the main launch was the walking build (far tiles stay in it: statistics) */
    } pend[2];
    /* workspace */
    DevBuf offsets, grid, ncells, sid, cell_of, rank, cell_start, blk_sums, cell_tbl, cell_first;
    DevBuf chunk_struct, chunk_begin, chunk_len, struct_chunk0, bpart;
    int n_chunks = 0;
    DevBuf sq, s_idx;
    DevBuf status, ovf_tiles, ovf_tiles2, ovf_atoms, unit_pts, slab, seg;
    std::vector<int64_t> offsets_host; /* last uploaded offsets */
    std::vector<double> unit_host;     /* last uploaded S&R unit points */
    /* S&R, third arrangement (sr_caps.h): the table of cap masks of unit_host, rebuilt when the points change */
    DevBuf captab;
    std::vector<sasa::SrCapEntry> captab_host;
    int captab_n = 0, captab_l = 0;    /* its resolution; 0: no table for these points (more than 128, not unit vectors, switched off) */
    /* host staging for freesasa_gpu_calc_batch */
    DevBuf h_xyz, h_radii, h_sasa, h_counts, h_totals;
    void *stage_in = nullptr, *stage_out = nullptr; /* page-locked host staging of freesasa_gpu_calc_batch_pipelined */
    size_t stage_in_cap = 0, stage_out_cap = 0;
    int *pinned = nullptr; /* page-locked host words for the small device->host readbacks: two sets of ST_WORDS + 4 */
    long long max_cells = 1LL << 30;
    long long cells_hint = 0; /* cells the last batch needed, with a margin: the table is never sized below it */
    int scan_epoch = 0;       /* batches that went through the general cell sort's chained scan (PipeArgs::scan_epoch) */
    /* adaptive neighbor-pool size, per algorithm: (resolution, TA) it was learnt for and the value */
    int hint_res[2] = {0, 0}, hint_ta[2] = {0, 0}, hint_pool[2] = {0, 0};
    double hint_probe = -1.0; /* the probe radius the hints were learnt with (another probe: other neighbor counts, so they start over) */
    bool hint_bucket = false; /* L&R: the last batch had long neighbor lists */
    bool sort_fused = true;   /* the per-structure cell sort (k_sort_struct) until a batch turns out not to fit it */
    double hint_nn = 0;       /* L&R (lr2): neighbor records per atom the main launch should hold */
    int hint_nn_max = 0;      /* ... and the longest neighbor list expected (mask words per item) */
    double hint_split2 = 0;   /* ... the share of its tiles above the 16-tiles-per-CU pool */
    bool hint_far = false;    /* ... a quarter or more of its tiles had an atom beyond LR2_WALK_Z: the next batch gets the walking build of the main launch */
    int hint_pool2 = 0, hint_ta2 = 0, hint_mw2 = 0; /* ... and the pool the last batch's demand histogram asks for, for tiles of that shape */
    /* the device-side parser's workspace and what its two phases hand each other (gpu_parse.hip) */
    DevBuf parse[11];
    std::vector<long long> parse_off;
    long long parse_atoms = 0;
    int parse_lines = 0, parse_files = 0, parse_options = 0;
    unsigned parse_T = 0;
    int *dbg_nn = nullptr, *dbg_nb = nullptr; /* test hook: freesasa_gpu_lr_neighbors_dev */
    int dbg_cap = 0;
};

int ctx_fail(freesasa_gpu_ctx *c, const char *fmt, ...) __attribute__((format(printf, 2, 3))); /* sets the context's error text, returns -1 */

#define HIP_TRY(c, call)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return ctx_fail((c), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

/* device / page-locked allocations (both honour the fault-injection hook freesasa_gpu_test_fail_after) */
hipError_t dev_malloc(void **p, size_t bytes);
hipError_t host_malloc(void **p, size_t bytes);
/* grow a workspace buffer to at least `bytes` (waits for the stream first when batches are in flight) */
int ensure(freesasa_gpu_ctx *c, DevBuf &b, size_t bytes);

/* One batch on device pointers, synchronous: redone when the cell table was too small; on failure nothing is still
   running on the stream when the caller gets control back.  lr: Lee-Richards (resolution = slices), else Shrake-Rupley
   (resolution = test points, unit_points on the host). */
int run_batch(freesasa_gpu_ctx *c, bool lr, const double *d_xyz, const double *d_radii, const int64_t *offsets, int n_structs,
              double probe, int resolution, const double *unit_points, double *d_sasa, int *d_counts, double *d_totals);

/* ------------------------------------------------------------------ host-side helpers (gpu_hostbatch.hip) */

/* A small pool of contexts so that concurrent host threads (the reference library is re-entrant,
   doc/doxy-main.md:741-756) each get their own stream and workspace. */
freesasa_gpu_ctx *pool_get(int device);
void pool_put(freesasa_gpu_ctx *c);
int set_err(char *out, int len, const char *msg); /* returns -1 */

/* ------------------------------------------------------------------ the C boundary and C++ exceptions
 * The engine is C++ behind a C ABI whose contract is the reference's: NULL / FREESASA_FAIL / -1 with a message, never
 * exit() (ref: src/util.c:89-113) - and therefore never an exception: a std::bad_alloc from a std::vector, or a
 * std::system_error from a thread that cannot start under a cgroup's pid limit, must not reach a C caller (it would end
 * in std::terminate).  Every extern "C" entry runs its body through guarded() / guarded_ctx(); every worker-thread
 * body catches for itself (an exception that leaves a std::thread's function terminates the process) and reports
 * through its FirstError / error slot; what a scope owns - threads, pooled contexts, loader batches, descriptors - is
 * held by the small RAII types below so that unwinding releases it.  tests/test_hostfault.py walks the n-th host
 * allocation / thread creation failing (hostfault.h) through the drivers. */
extern "C" int freesasa_hostfault_hit(void); /* hostfault.c: 1 = this thread creation has to fail (tests) */
/* text for the exception being handled (call inside a catch block) */
const char *exception_text(char *buf, size_t len) noexcept;

template <class F> int guarded(char *err_out, int err_len, F &&body) noexcept
{
    try { return body(); }
    catch (...) { char msg[200]; return set_err(err_out, err_len, exception_text(msg, sizeof msg)); }
}
/* ... for the entries that report through a context: nothing may still run on its stream when the caller is back */
template <class F> int guarded_ctx(freesasa_gpu_ctx *c, F &&body) noexcept
{
    try { return body(); }
    catch (...) {
        char msg[200];
        exception_text(msg, sizeof msg);
        if (!c) return -1;
        (void)hipStreamSynchronize(c->stream);
        return ctx_fail(c, "%s", msg);
    }
}

/* ------------------------------------------------------------------ a device's host side: its NUMA node
 * On an 8-GPU node every GPU hangs off one socket's PCIe root; a lane that reads files into page-locked staging and
 * feeds that GPU should run - and have its staging allocated - on that socket (the reference has no counterpart: its
 * threads share one structure in one address space, src/sasa_lr.c:219-253).  hwloc-free: the device's PCI address
 * (hipDeviceGetPCIBusId) -> <sysfs>/bus/pci/devices/<address>/numa_node -> <sysfs>/devices/system/node/node<k>/cpulist.
 * DeviceNodeScope binds the CALLING thread to those CPUs (intersected with the CPUs it is allowed now) for the length
 * of a scope and puts its old mask back; nothing happens when the node is unknown (-1: single-socket boxes, VMs), the
 * intersection is empty, or FREESASA_AMD_NO_AFFINITY is set.  node_cpus_for_pci is the mapping itself, testable on
 * a made-up sysfs tree without a GPU (freesasa_gpu_test_node_cpus, tests/test_multidevice.py). */
int node_cpus_for_pci(const char *sysfs_root, const char *pci_address, int *cpus_out, int cap); /* CPUs found (<= cap stored), 0: no node known, -1: unreadable */
struct DeviceNodeScope {
    bool bound = false;
    unsigned long old_mask[16] = {0}; /* cpu_set_t of 1024 CPUs */
    explicit DeviceNodeScope(int device);
    DeviceNodeScope(const DeviceNodeScope &) = delete;
    DeviceNodeScope &operator=(const DeviceNodeScope &) = delete;
    ~DeviceNodeScope();
};

/* threads of one scope: joined on every way out of it */
struct ThreadGroup {
    std::vector<std::thread> th;
    ThreadGroup() = default;
    ThreadGroup(const ThreadGroup &) = delete;
    ThreadGroup &operator=(const ThreadGroup &) = delete;
    ~ThreadGroup() { join(); }
    void join() noexcept
    {
        for (auto &t : th)
            if (t.joinable()) t.join();
        th.clear();
    }
    /* false: the thread did not start (no memory for its state, EAGAIN from the system, or the test hook) */
    template <class F, class... A> bool spawn(F &&f, A &&...a) noexcept
    {
        try {
            if (freesasa_hostfault_hit()) return false;
            th.emplace_back(std::forward<F>(f), std::forward<A>(a)...);
            return true;
        } catch (...) { return false; }
    }
};

/* a pooled context for the length of a scope; on the way out nothing is left running on its stream */
struct PoolLease {
    freesasa_gpu_ctx *c;
    explicit PoolLease(int device) : c(pool_get(device)) {}
    PoolLease(const PoolLease &) = delete;
    PoolLease &operator=(const PoolLease &) = delete;
    ~PoolLease()
    {
        if (!c) return;
        c->shared_radii = false;
        (void)hipStreamSynchronize(c->stream);
        pool_put(c);
    }
};
bool host_pinned(const void *p); /* page-locked already (hipHostMalloc / hipHostRegister, e.g. a pinned tensor)? */
int ensure_pinned(freesasa_gpu_ctx *c, void **p, size_t *cap, size_t bytes); /* grow a context's page-locked staging buffer */

#endif
