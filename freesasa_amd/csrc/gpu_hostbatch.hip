/*
 * gpu_hostbatch.hip — host-pointer batches (include/freesasa_gpu.h): the pool of contexts behind the re-entrant
 * entry points, one batch on one device, a batch cut over a list of devices, and the pipelined form whose PCIe copies
 * run under the kernels of other chunks.  Host code; kernels in gpu_kernels.hip.
 */
#include <hip/hip_runtime.h>

#include <atomic>
#include <exception>
#include <mutex>
#include <new>
#include <sched.h>
#include <stdlib.h>
#include <system_error>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <thread>
#include <vector>

#include "engine_internal.h"

/* ------------------------------------------------------------------ host-pointer batch */

/* A small pool of contexts so that concurrent host threads (the reference library is
 * re-entrant, doc/doxy-main.md:741-756) each get their own stream and workspace. */
static std::mutex g_pool_mu;
static std::vector<freesasa_gpu_ctx *> g_pool;

freesasa_gpu_ctx *pool_get(int device)
{
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t k = 0; k < g_pool.size(); ++k)
            if (device < 0 || g_pool[k]->device == device) {
                freesasa_gpu_ctx *c = g_pool[k];
                g_pool.erase(g_pool.begin() + k);
                return c;
            }
    }
    return freesasa_gpu_ctx_create(device, nullptr);
}
void pool_put(freesasa_gpu_ctx *c) /* (called from destructors: must not throw - a context the pool cannot list is destroyed) */
{
    try {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_pool.push_back(c);
    } catch (...) {
        freesasa_gpu_ctx_destroy(c);
    }
}

/* Destroy the idle contexts of the pool (their streams, workspaces and staging buffers): device memory goes back
 * to the runtime; the next host-pointer call builds what it needs again. */
extern "C" void freesasa_gpu_release_pool(void)
{
    std::vector<freesasa_gpu_ctx *> idle;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        idle.swap(g_pool);
    }
    for (freesasa_gpu_ctx *c : idle) freesasa_gpu_ctx_destroy(c);
}

int set_err(char *out, int len, const char *msg)
{
    if (out && len > 0) snprintf(out, (size_t)len, "%s", msg);
    return -1;
}

const char *exception_text(char *buf, size_t len) noexcept
{
    try { throw; }
    catch (const std::bad_alloc &) { snprintf(buf, len, "out of host memory"); }
    catch (const std::system_error &e) { snprintf(buf, len, "system error: %s", e.what()); }
    catch (const std::exception &e) { snprintf(buf, len, "internal error: %s", e.what()); }
    catch (...) { snprintf(buf, len, "internal error (unknown C++ exception)"); }
    return buf;
}

/* ------------------------------------------------------------------ the NUMA node of a device (engine_internal.h) */

static bool read_small_file(const char *path, char *buf, size_t len)
{
    FILE *f = fopen(path, "r");
    if (!f) return false;
    const size_t n = fread(buf, 1, len - 1, f);
    fclose(f);
    buf[n] = 0;
    return n > 0;
}
/* "0-63,128-191\n" -> CPU numbers; returns how many the list names (the first `cap` are stored), -1 on a malformed list */
static int parse_cpulist(const char *text, int *cpus_out, int cap)
{
    int n = 0;
    const char *p = text;
    while (*p && *p != '\n') {
        char *e;
        const long a = strtol(p, &e, 10);
        if (e == p || a < 0) return -1;
        long b = a;
        p = e;
        if (*p == '-') { b = strtol(p + 1, &e, 10); if (e == p + 1 || b < a) return -1; p = e; }
        for (long c = a; c <= b; ++c) { if (n < cap && cpus_out) cpus_out[n] = (int)c; ++n; }
        if (*p == ',') ++p;
        else if (*p && *p != '\n') return -1;
    }
    return n;
}
int node_cpus_for_pci(const char *sysfs_root, const char *pci_address, int *cpus_out, int cap)
{
    if (!sysfs_root || !pci_address) return -1;
    char path[512], buf[4096], addr[64];
    size_t k = 0;
    for (; pci_address[k] && k + 1 < sizeof addr; ++k) addr[k] = (char)(pci_address[k] >= 'A' && pci_address[k] <= 'F' ? pci_address[k] + 32 : pci_address[k]); /* sysfs names are lower case */
    addr[k] = 0;
    snprintf(path, sizeof path, "%s/bus/pci/devices/%s/numa_node", sysfs_root, addr);
    if (!read_small_file(path, buf, sizeof buf)) return -1;
    const int node = atoi(buf);
    if (node < 0) return 0; /* the platform names no node for the device */
    snprintf(path, sizeof path, "%s/devices/system/node/node%d/cpulist", sysfs_root, node);
    if (!read_small_file(path, buf, sizeof buf)) return -1;
    return parse_cpulist(buf, cpus_out, cap);
}
extern "C" int freesasa_gpu_test_node_cpus(const char *sysfs_root, const char *pci_address, int *cpus_out, int cap)
{
    return node_cpus_for_pci(sysfs_root, pci_address, cpus_out, cap);
}

DeviceNodeScope::DeviceNodeScope(int device)
{
    static_assert(sizeof(cpu_set_t) <= sizeof old_mask, "cpu_set_t");
    if (getenv("FREESASA_AMD_NO_AFFINITY")) return;
    char addr[64] = {0};
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return;
    if (hipDeviceGetPCIBusId(addr, (int)sizeof addr, device) != hipSuccess) { (void)hipGetLastError(); return; }
    int cpus[1024];
    const char *root = getenv("FREESASA_AMD_SYSFS_ROOT"); /* (tests) */
    const int n = node_cpus_for_pci(root ? root : "/sys", addr, cpus, 1024);
    if (n <= 0) return;
    cpu_set_t now, want;
    if (sched_getaffinity(0, sizeof now, &now) != 0) return;
    CPU_ZERO(&want);
    int common = 0;
    for (int k = 0; k < n && k < 1024; ++k)
        if (cpus[k] < CPU_SETSIZE && CPU_ISSET(cpus[k], &now)) { CPU_SET(cpus[k], &want); ++common; }
    if (common == 0 || common == CPU_COUNT(&now)) return; /* (nothing to narrow) */
    if (sched_setaffinity(0, sizeof want, &want) != 0) return;
    memcpy(old_mask, &now, sizeof now);
    bound = true;
}
DeviceNodeScope::~DeviceNodeScope()
{
    if (!bound) return;
    cpu_set_t old;
    memcpy(&old, old_mask, sizeof old);
    (void)sched_setaffinity(0, sizeof old, &old);
}

extern "C" int freesasa_gpu_calc_batch(const double *xyz, const double *radii, const int64_t *offsets, int n_structs,
                                       int alg, double probe, int resolution, double *sasa_out, int *counts_out,
                                       double *totals_out, int device, char *err_out, int err_len)
{
    if (err_out && err_len > 0) err_out[0] = 0;
    if (!xyz || !radii || !offsets || !sasa_out) return set_err(err_out, err_len, "null argument");
    if (freesasa_gpu_device_count() <= 0)
        return set_err(err_out, err_len, "no HIP device available: libfreesasa_amd has no CPU path");
    return guarded(err_out, err_len, [&]() -> int {
    PoolLease lease(device); /* (returned to the pool, its stream idle, on every way out - an exception included) */
    freesasa_gpu_ctx *c = lease.c;
    if (!c) return set_err(err_out, err_len, "could not create a GPU context");
    int ret = -1;
    do {
        if (n_structs <= 0 || offsets[n_structs] <= 0) { ctx_fail(c, "empty batch"); break; }
        const size_t n = (size_t)offsets[n_structs];
        if (hipSetDevice(c->device) != hipSuccess) { ctx_fail(c, "hipSetDevice failed"); break; }
        if (ensure(c, c->h_xyz, 24 * n) || ensure(c, c->h_radii, 8 * n) || ensure(c, c->h_sasa, 8 * n) ||
            ensure(c, c->h_counts, 4 * n) || ensure(c, c->h_totals, 8 * (size_t)n_structs))
            break;
        if (hipMemcpyAsync(c->h_xyz.p, xyz, 24 * n, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(c->h_radii.p, radii, 8 * n, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
            ctx_fail(c, "host-to-device copy failed");
            break;
        }
        if (alg == 0) {
            ret = run_batch(c, true, (double *)c->h_xyz.p, (double *)c->h_radii.p, offsets, n_structs, probe, resolution,
                            nullptr, (double *)c->h_sasa.p, nullptr, totals_out ? (double *)c->h_totals.p : nullptr);
        } else if (alg == 1) {
            std::vector<double> tp(3 * (size_t)(resolution > 0 ? resolution : 1));
            if (resolution > 0) freesasa_gpu_test_points(resolution, tp.data());
            ret = run_batch(c, false, (double *)c->h_xyz.p, (double *)c->h_radii.p, offsets, n_structs, probe, resolution,
                            tp.data(), (double *)c->h_sasa.p, counts_out ? (int *)c->h_counts.p : nullptr,
                            totals_out ? (double *)c->h_totals.p : nullptr);
        } else {
            ctx_fail(c, "unknown algorithm %d", alg);
        }
        if (ret) break;
        ret = -1;
        if (hipMemcpyAsync(sasa_out, c->h_sasa.p, 8 * n, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { ctx_fail(c, "device-to-host copy failed"); break; }
        if (counts_out && alg == 1 &&
            hipMemcpyAsync(counts_out, c->h_counts.p, 4 * n, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { ctx_fail(c, "device-to-host copy failed"); break; }
        if (totals_out &&
            hipMemcpyAsync(totals_out, c->h_totals.p, 8 * (size_t)n_structs, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { ctx_fail(c, "device-to-host copy failed"); break; }
        if (hipStreamSynchronize(c->stream) != hipSuccess) { ctx_fail(c, "stream synchronize failed"); break; }
        ret = 0;
    } while (0);
    if (ret) {
        (void)hipStreamSynchronize(c->stream); /* the caller's arrays must not be read after we return */
        set_err(err_out, err_len, c->err[0] ? c->err : "GPU batch failed");
    }
    return ret;
    });
}

/* ------------------------------------------------------------------ several GPUs, one process */

/* Independent structures shard with no exchange: the batch is cut into contiguous runs of
 * structures with about equal atom counts, one run per device of the mask, each run handled by its
 * own host thread through freesasa_gpu_calc_batch (its own pooled context, stream and workspace).
 * Contiguous runs need no gather: every device reads and writes its slice of the caller's arrays. */
/* cuts[k] = first structure of shard k (cuts[n_parts] = n_structs): where the running atom count passes
 * k/n_parts of the total; shards may be empty when there are fewer structures than parts */
extern "C" void freesasa_gpu_shard_cuts(const int64_t *offsets, int n_structs, int n_parts, int *cuts)
{
    cuts[0] = 0;
    const int64_t base = offsets[0], total = offsets[n_structs] - base;
    for (int k = 1, s = 0; k < n_parts; ++k) {
        const int64_t want = base + total * k / n_parts;
        while (s < n_structs && offsets[s] < want) ++s;
        cuts[k] = s < cuts[k - 1] ? cuts[k - 1] : s;
    }
    cuts[n_parts] = n_structs;
}

extern "C" int freesasa_gpu_calc_batch_devices(const double *xyz, const double *radii, const int64_t *offsets, int n_structs,
                                               int alg, double probe, int resolution, double *sasa_out, int *counts_out,
                                               double *totals_out, const int *devices, int n_devices, char *err_out, int err_len)
{
    if (err_out && err_len > 0) err_out[0] = 0;
    if (!xyz || !radii || !offsets || !sasa_out || n_structs <= 0 || !devices || n_devices <= 0)
        return set_err(err_out, err_len, "bad argument");
    const int n_dev = freesasa_gpu_device_count();
    if (n_dev <= 0) return set_err(err_out, err_len, "no HIP device available: libfreesasa_amd has no CPU path");
    for (int k = 0; k < n_devices; ++k)
        if (devices[k] < 0 || devices[k] >= n_dev) return set_err(err_out, err_len, "device index out of range");
    return guarded(err_out, err_len, [&]() -> int {
    const int nd = n_devices;
    std::vector<int> cut(nd + 1);
    freesasa_gpu_shard_cuts(offsets, n_structs, nd, cut.data());
    std::vector<int> rc(nd, 0);
    std::vector<std::vector<char>> errs(nd, std::vector<char>(256, 0));
    auto run = [&](int k) noexcept {
        try {
            const int s0 = cut[k], ns = cut[k + 1] - cut[k];
            if (ns <= 0 || offsets[s0 + ns] == offsets[s0]) return;
            std::vector<int64_t> off(ns + 1); /* the shard's own CSR offsets start at 0 */
            for (int i = 0; i <= ns; ++i) off[i] = offsets[s0 + i] - offsets[s0];
            const int64_t a0 = offsets[s0];
            rc[k] = freesasa_gpu_calc_batch(xyz + 3 * a0, radii + a0, off.data(), ns, alg, probe, resolution, sasa_out + a0,
                                            counts_out ? counts_out + a0 : nullptr, totals_out ? totals_out + s0 : nullptr,
                                            devices[k], errs[k].data(), (int)errs[k].size());
        } catch (...) { /* (an exception that leaves a thread's function ends the process) */
            exception_text(errs[k].data(), errs[k].size());
            rc[k] = -1;
        }
    };
    bool started = true;
    {
        ThreadGroup tg; /* joined on every way out */
        for (int k = 1; k < nd && started; ++k) started = tg.spawn(run, k);
        if (started) run(0);
    }
    if (!started) return set_err(err_out, err_len, "could not start a worker thread");
    for (int k = 0; k < nd; ++k)
        if (rc[k]) return set_err(err_out, err_len, errs[k].data()[0] ? errs[k].data() : "a device shard failed");
    return 0;
    });
}

extern "C" int freesasa_gpu_calc_batch_multi(const double *xyz, const double *radii, const int64_t *offsets, int n_structs,
                                             int alg, double probe, int resolution, double *sasa_out, int *counts_out,
                                             double *totals_out, unsigned device_mask, char *err_out, int err_len)
{
    if (err_out && err_len > 0) err_out[0] = 0;
    const int n_dev = freesasa_gpu_device_count();
    if (n_dev <= 0) return set_err(err_out, err_len, "no HIP device available: libfreesasa_amd has no CPU path");
    int devs[32], nd = 0;
    for (int d = 0; d < 32 && d < n_dev; ++d)
        if (device_mask & (1u << d)) devs[nd++] = d;
    if (nd == 0) return set_err(err_out, err_len, "device mask selects no available device");
    return freesasa_gpu_calc_batch_devices(xyz, radii, offsets, n_structs, alg, probe, resolution, sasa_out, counts_out,
                                           totals_out, devs, nd, err_out, err_len);
}

/* ------------------------------------------------------------------ host arrays in, host arrays out, pipelined */

/* One host pointer: page-locked already (hipHostMalloc / hipHostRegister, e.g. a pinned tensor)? */
bool host_pinned(const void *p)
{
    hipPointerAttribute_t at;
    if (!p || hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

/* Grow a context's page-locked staging buffer (for callers whose arrays are pageable). */
int ensure_pinned(freesasa_gpu_ctx *c, void **p, size_t *cap, size_t bytes)
{
    if (bytes <= *cap) return 0;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr; *cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (host_malloc(p, want) != hipSuccess) return ctx_fail(c, "out of page-locked host memory (%zu bytes)", want);
    *cap = want;
    return 0;
}

/* The batch is cut into chunks of whole structures (about chunk_atoms atoms each) that n_lanes host threads take
 * from a shared counter; every lane owns a pooled context (stream, workspace, staging) and runs
 *     host -> device copy,  cell sort + tile kernels,  device -> host copy
 * for its chunk while the other lanes are in a different stage: PCIe in, kernels and PCIe out of different
 * chunks overlap.  Page-locked caller arrays are copied by DMA in place; pageable ones go through the lane's
 * page-locked staging buffers (the memcpy of one lane overlaps the DMA of another). */
extern "C" int freesasa_gpu_calc_batch_pipelined(const double *xyz, const double *radii, const int64_t *offsets, int n_structs,
                                                 int alg, double probe, int resolution, double *sasa_out, int *counts_out,
                                                 double *totals_out, int device, int n_lanes, long long chunk_atoms,
                                                 char *err_out, int err_len)
{
    if (err_out && err_len > 0) err_out[0] = 0;
    if (!xyz || !radii || !offsets || !sasa_out || n_structs <= 0) return set_err(err_out, err_len, "bad argument");
    if (alg != 0 && alg != 1) return set_err(err_out, err_len, "unknown algorithm");
    if (freesasa_gpu_device_count() <= 0)
        return set_err(err_out, err_len, "no HIP device available: libfreesasa_amd has no CPU path");
    const bool pin_in = host_pinned(xyz) && host_pinned(radii);
    const bool pin_out = host_pinned(sasa_out) && (!counts_out || host_pinned(counts_out)) && (!totals_out || host_pinned(totals_out));
    if (n_lanes <= 0) n_lanes = pin_in && pin_out ? 3 : 4; /* (measured, 1e7 atoms: 1 / 2 / 3 / 4 / 6 lanes with DMA in place 24.0 / 22.4 /
                                                              15.9 / 17.1 / 16.6 ms - one lane each in PCIe in, kernels, PCIe out;
                                                              staging through page-locked buffers also spends host memcpy time) */
    if (n_lanes > 8) n_lanes = 8;
    if (chunk_atoms <= 0) chunk_atoms = 1250000;
    return guarded(err_out, err_len, [&]() -> int {
    std::vector<int> cut(1, 0);
    for (int s = 0; s < n_structs; ++s)
        if (offsets[s + 1] - offsets[cut.back()] >= chunk_atoms && s + 1 < n_structs) cut.push_back(s + 1);
    cut.push_back(n_structs);
    const int n_chunks = (int)cut.size() - 1;
    if (n_lanes > n_chunks) n_lanes = n_chunks;
    std::vector<double> tp;
    if (alg == 1) { tp.resize(3 * (size_t)(resolution > 0 ? resolution : 1)); if (resolution > 0) freesasa_gpu_test_points(resolution, tp.data()); }
    std::atomic<int> next(0), failed(0);
    std::vector<std::vector<char>> errs(n_lanes, std::vector<char>(256, 0));
    auto lane = [&](int id) noexcept {
      try {
        PoolLease lease(device);
        freesasa_gpu_ctx *c = lease.c;
        if (!c) { snprintf(errs[id].data(), 256, "could not create a GPU context"); failed = 1; return; }
        std::vector<int64_t> off;
        for (;;) {
            const int k = next.fetch_add(1);
            if (k >= n_chunks || failed.load()) break;
            const int s0 = cut[k], ns = cut[k + 1] - cut[k];
            const int64_t a0 = offsets[s0];
            const size_t n = (size_t)(offsets[s0 + ns] - a0);
            if (n == 0) { if (totals_out) for (int i = 0; i < ns; ++i) totals_out[s0 + i] = 0; continue; }
            off.resize((size_t)ns + 1);
            for (int i = 0; i <= ns; ++i) off[i] = offsets[s0 + i] - a0;
            int rc = -1;
            do {
                if (hipSetDevice(c->device) != hipSuccess) { ctx_fail(c, "hipSetDevice failed"); break; }
                if (ensure(c, c->h_xyz, 24 * n) || ensure(c, c->h_radii, 8 * n) || ensure(c, c->h_sasa, 8 * n) ||
                    ensure(c, c->h_counts, 4 * n) || ensure(c, c->h_totals, 8 * (size_t)ns))
                    break;
                const double *src_xyz = xyz + 3 * a0, *src_r = radii + a0;
                if (!pin_in) {
                    if (ensure_pinned(c, &c->stage_in, &c->stage_in_cap, 32 * n)) break;
                    memcpy(c->stage_in, src_xyz, 24 * n);
                    memcpy((char *)c->stage_in + 24 * n, src_r, 8 * n);
                    src_xyz = (const double *)c->stage_in;
                    src_r = (const double *)((char *)c->stage_in + 24 * n);
                }
                if (hipMemcpyAsync(c->h_xyz.p, src_xyz, 24 * n, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                    hipMemcpyAsync(c->h_radii.p, src_r, 8 * n, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
                    ctx_fail(c, "host-to-device copy failed");
                    break;
                }
                if (run_batch(c, alg == 0, (double *)c->h_xyz.p, (double *)c->h_radii.p, off.data(), ns, probe, resolution,
                              alg == 1 ? tp.data() : nullptr, (double *)c->h_sasa.p, counts_out && alg == 1 ? (int *)c->h_counts.p : nullptr,
                              totals_out ? (double *)c->h_totals.p : nullptr))
                    break;
                const bool want_counts = counts_out && alg == 1;
                double *dst_sasa = sasa_out + a0, *dst_tot = totals_out ? totals_out + s0 : nullptr;
                int *dst_cnt = want_counts ? counts_out + a0 : nullptr;
                const size_t out_bytes = 8 * n + (want_counts ? 4 * n : 0) + (dst_tot ? 8 * (size_t)ns : 0);
                if (!pin_out) {
                    if (ensure_pinned(c, &c->stage_out, &c->stage_out_cap, out_bytes)) break;
                    dst_sasa = (double *)c->stage_out;
                    dst_cnt = want_counts ? (int *)((char *)c->stage_out + 8 * n) : nullptr;
                    dst_tot = totals_out ? (double *)((char *)c->stage_out + 8 * n + (want_counts ? 4 * n : 0)) : nullptr;
                }
                bool ok = hipMemcpyAsync(dst_sasa, c->h_sasa.p, 8 * n, hipMemcpyDeviceToHost, c->stream) == hipSuccess;
                if (ok && want_counts) ok = hipMemcpyAsync(dst_cnt, c->h_counts.p, 4 * n, hipMemcpyDeviceToHost, c->stream) == hipSuccess;
                if (ok && dst_tot) ok = hipMemcpyAsync(dst_tot, c->h_totals.p, 8 * (size_t)ns, hipMemcpyDeviceToHost, c->stream) == hipSuccess;
                if (!ok) { ctx_fail(c, "device-to-host copy failed"); break; }
                if (hipStreamSynchronize(c->stream) != hipSuccess) { ctx_fail(c, "stream synchronize failed"); break; }
                if (!pin_out) {
                    memcpy(sasa_out + a0, dst_sasa, 8 * n);
                    if (want_counts) memcpy(counts_out + a0, dst_cnt, 4 * n);
                    if (totals_out) memcpy(totals_out + s0, dst_tot, 8 * (size_t)ns);
                }
                rc = 0;
            } while (0);
            if (rc) {
                (void)hipStreamSynchronize(c->stream); /* nothing may still read the caller's arrays when we return */
                snprintf(errs[id].data(), 256, "%s", c->err[0] ? c->err : "GPU batch failed");
                failed = 1;
                break;
            }
        }
      } catch (...) { /* (the lease has left the stream idle and returned the context) */
        exception_text(errs[id].data(), errs[id].size());
        failed = 1;
      }
    };
    {
        ThreadGroup tg;
        for (int k = 1; k < n_lanes; ++k)
            if (!tg.spawn(lane, k)) { snprintf(errs[0].data(), 256, "could not start a worker thread"); failed = 1; break; }
        if (!failed.load()) lane(0);
    }
    if (failed.load())
        for (int k = 0; k < n_lanes; ++k)
            if (errs[k][0]) return set_err(err_out, err_len, errs[k].data());
    return failed.load() ? set_err(err_out, err_len, "GPU batch failed") : 0;
    });
}

