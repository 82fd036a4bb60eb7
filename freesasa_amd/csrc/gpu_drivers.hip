/*
 * gpu_drivers.hip — the drivers for BASELINE configs[3] and configs[4] (include/freesasa_gpu.h): the structure sweep
 * over PDB / mmCIF files, the same sweep from a binary cache, and the trajectory drivers — each over ONE device or a
 * LIST of devices of the node.  Host code; kernels in gpu_kernels.hip.
 *
 * What replaces what: the reference reads one file per run of its CLI (src/main.cc:763-779) and spreads ONE structure
 * over <= 16 pthreads (src/sasa_lr.c:219-253).  Here the unit of parallel work is a batch of whole structures (a
 * shard of whole frames), and the units are independent: there is no exchange between devices, only a shared list of
 * work.  One worker (sweep) or a few lanes (trajectory, cache sweep) per entry of devices[] take the next unit from a
 * shared counter — largest first for the file sweep (LPT on the file sizes: atoms are proportional to bytes) — so
 * that a device that finishes early takes more; every result lands at its own place of the caller's arrays / the
 * result files (pwrite at the unit's offset), and ONE done-list, appended to under a mutex after a unit's results
 * are on disk, serves all devices.  The host CPUs THE CGROUP GRANTS (freesasa_ingest_usable_cpus: a GPU box shows 256
 * and grants 16) are divided among the devices' loaders.  A device may appear in the list more than once (its units
 * then overlap their copies and kernels; the tests run device lists [0, 0, 0] and [0] * 8 on a one-GPU box).
 * Results are bit-identical to the single-device drivers': a unit's numbers do not depend on who computed it.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <fcntl.h>
#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <time.h>
#include <unistd.h>
#include <vector>

#include "engine_internal.h"
#include "gpu_parse.h"

namespace {

bool pread_all(int fd, void *buf, size_t bytes, long long off)
{
    char *p = (char *)buf;
    while (bytes) {
        const ssize_t r = pread(fd, p, bytes, (off_t)off);
        if (r <= 0) return false;
        p += r; off += r; bytes -= (size_t)r;
    }
    return true;
}
bool pwrite_all(int fd, const void *buf, size_t bytes, long long off)
{
    const char *p = (const char *)buf;
    while (bytes) {
        const ssize_t r = pwrite(fd, p, bytes, (off_t)off);
        if (r <= 0) return false;
        p += r; off += r; bytes -= (size_t)r;
    }
    return true;
}

/* the device list of a call: every entry an existing device (entries may repeat) */
int check_devices(const int *devices, int n_devices, char *err_out, int err_len)
{
    const int n_dev = freesasa_gpu_device_count();
    if (n_dev <= 0) return set_err(err_out, err_len, "no HIP device available: libfreesasa_amd has no CPU path");
    if (!devices || n_devices <= 0 || n_devices > 64) return set_err(err_out, err_len, "bad device list (1 .. 64 entries)");
    for (int k = 0; k < n_devices; ++k)
        if (devices[k] < -1 || devices[k] >= n_dev) return set_err(err_out, err_len, "device index out of range");
    return 0;
}

/* CPUs THIS PROCESS may count on: what the cgroup grants (freesasa_ingest_usable_cpus), divided among the ranks of the node
   when a launcher says there are several (one process per GPU: torchrun exports LOCAL_WORLD_SIZE).  What every default
   below - loader threads, lanes per device - starts from: eight ranks of a box that grants 16 CPUs get two each, not
   sixteen each (the host budget). */
int process_cpus()
{
    int total = freesasa_ingest_usable_cpus();
    if (const char *lws = getenv("LOCAL_WORLD_SIZE")) {
        const int ranks = atoi(lws);
        if (ranks > 1) total /= ranks;
    }
    return total < 1 ? 1 : total;
}
/* host threads of one of n_workers loaders: the caller's total (or, <= 0, this process's CPUs) divided among them */
int threads_per_worker(int n_threads, int n_workers)
{
    const int total = n_threads > 0 ? n_threads : process_cpus();
    const int per = total / (n_workers > 0 ? n_workers : 1);
    return per < 1 ? 1 : per;
}

/* what a driver's scope owns besides threads and contexts (engine_internal.h): released on every way out of it */
struct Fd {
    int fd = -1;
    Fd() = default;
    Fd(const Fd &) = delete;
    Fd &operator=(const Fd &) = delete;
    ~Fd() { if (fd >= 0) close(fd); }
};
struct Batch { /* a loader batch (freesasa_ingest.h) */
    freesasa_ingest_batch b;
    Batch() { memset(&b, 0, sizeof b); }
    Batch(const Batch &) = delete;
    Batch &operator=(const Batch &) = delete;
    ~Batch() { freesasa_ingest_free(&b); }
    void take(Batch &o) { freesasa_ingest_free(&b); b = o.b; memset(&o.b, 0, sizeof o.b); }
};
struct Cache {
    freesasa_ingest_cache *c = nullptr;
    Cache() = default;
    Cache(const Cache &) = delete;
    Cache &operator=(const Cache &) = delete;
    ~Cache() { if (c) freesasa_ingest_cache_close(c); }
};

/* ---- the device-side parser's input (gpu_parse.hip): a batch's files read - not parsed - into page-locked memory */
/* The text lives in one of the worker's CONTEXT's two page-locked staging buffers (stage_in / stage_out: they stay with the
   pooled context from call to call).  Until round 6's last session every sweep allocated and freed its own: hipHostMalloc
   and hipHostFree of 50 MB take 5 - 7 ms each and hold the runtime's lock while they do - in the kernel trace of a 70 ms
   sweep (tools/dev/sweep_trace.sh) the first 24 ms saw five batches where the steady state does twenty-two, and the last
   batch's tile kernel waited 11 ms for the OTHER worker to free its buffers. */
struct Staged {
    unsigned char *text = nullptr; /* page-locked: the files one after the other, each in a slot of its size + 1 and ending with '\n' */
    size_t T = 0;
    void **slot = nullptr;         /* the context's buffer and its capacity */
    size_t *slot_cap = nullptr;
    std::vector<ParseFile> files;  /* [n + 1] */
    int rc = 0;                    /* -1: no page-locked memory */
    Staged(void **slot_, size_t *cap_) : slot(slot_), slot_cap(cap_) {}
    Staged(const Staged &) = delete;
    Staged &operator=(const Staged &) = delete;
    void swap(Staged &o) { std::swap(text, o.text); std::swap(T, o.T); std::swap(slot, o.slot); std::swap(slot_cap, o.slot_cap); files.swap(o.files); std::swap(rc, o.rc); }
};
/* n files -> out, with `threads` readers (each file: one pread loop, then the one-line-at-a-time look at an mmCIF file's text
   before its _atom_site loop: freesasa_ingest_cif_locate); a file that cannot be read is left to the host parser, which
   reports it */
void stage_files(const char *const *paths, int n, int options, int threads, Staged *out)
{
    out->rc = 0;
    out->files.assign((size_t)n + 1, ParseFile());
    std::vector<long long> size((size_t)n, 0);
    size_t T = 0;
    for (int f = 0; f < n; ++f) {
        struct stat st;
        size[f] = (paths[f] && stat(paths[f], &st) == 0 && st.st_size > 0) ? (long long)st.st_size : 0;
        out->files[f].beg = (unsigned)T;
        T += (size_t)size[f] + 1;
    }
    if (T >= (1ULL << 31)) { out->rc = -2; return; }
    out->files[n].beg = (unsigned)T;
    out->T = T;
    if (T + 64 > *out->slot_cap) {
        if (*out->slot) (void)hipHostFree(*out->slot);
        *out->slot = nullptr; *out->slot_cap = 0;
        const size_t want = T + T / 8 + 4096;
        if (host_malloc(out->slot, want) != hipSuccess) { out->rc = -1; out->text = nullptr; return; }
        *out->slot_cap = want;
    }
    out->text = (unsigned char *)*out->slot;
    std::atomic<int> next(0);
    auto reader = [&]() noexcept {
        for (;;) {
            const int f = next.fetch_add(1);
            if (f >= n) break;
            ParseFile &pf = out->files[f];
            unsigned char *dst = out->text + pf.beg;
            const size_t slot = (size_t)size[f] + 1;
            size_t got = 0;
            bool ok = false;
            if (paths[f]) {
                const int fd = open(paths[f], O_RDONLY);
                if (fd >= 0) {
                    ok = true;
                    while (got < (size_t)size[f]) {
                        const ssize_t r = pread(fd, dst + got, (size_t)size[f] - got, (off_t)got);
                        if (r < 0) { ok = false; break; }
                        if (r == 0) break;
                        got += (size_t)r;
                    }
                    close(fd);
                }
            }
            pf.no_final_nl = (got > 0 && dst[got - 1] != '\n') ? 1 : 0;
            memset(dst + got, '\n', slot - got); /* (the slot's spare byte, and whatever a file that shrank left) */
            pf.kind = PARSE_HOST; pf.ncol = 0; pf.row0 = 0;
            if (!ok || (long long)got != size[f] || (options & FREESASA_INGEST_RADIUS_FROM_OCCUPANCY)) continue;
            int ncol = 0;
            size_t row0 = 0;
            const int kind = freesasa_ingest_cif_locate((const char *)dst, got, &ncol, pf.slot, &row0);
            if (kind == 0) pf.kind = PARSE_PDB;
            else if (kind == 1) { pf.kind = PARSE_CIF; pf.ncol = (short)ncol; pf.row0 = pf.beg + (unsigned)row0; }
        }
    };
    ThreadGroup tg;
    for (int t = 1; t < threads && t < n; ++t)
        if (!tg.spawn(reader)) break; /* (fewer readers then) */
    reader();
}
std::atomic<long long> g_parse_dev_files(0), g_parse_host_files(0);

/* first failure of a set of workers wins; the others stop taking work */
struct FirstError {
    std::mutex mu;
    std::atomic<int> failed{0};
    char text[256] = {0};
    void set(const char *msg)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed.load()) snprintf(text, sizeof text, "%s", msg && msg[0] ? msg : "GPU driver failed");
        failed = 1;
    }
    void set_exception() noexcept /* inside a catch block */
    {
        char msg[200];
        try { set(exception_text(msg, sizeof msg)); } catch (...) { failed = 1; }
    }
};

/* ------------------------------------------------------------------ structure sweep: files */

struct SweepRec { double total, cls[3]; long long atoms; int status, pad; };
static_assert(sizeof(SweepRec) == 48, "result record");

/* Files -> per-structure totals.  Every worker owns a pooled context of its device and a loader: while batch k is on
 * the GPU the loader's threads (include/freesasa_ingest.h) read the batch the worker took next.  Inputs that fail to
 * load get total 0 and their loader status; the call only fails for GPU errors.
 * done_path (may be NULL): the sweep's done-list — a first line with the run's parameters, then "shard <batch> <first
 * file> <files>" per finished batch — next to a result file <done_path>.bin that holds, per file, total | class sums
 * (3) | atoms | status (fixed 48-byte records), written before the batch is listed.  A call that finds the done-list of
 * the same run takes the listed batches' results from the result file and only computes the others.
 * Returns 0 done, 1 stopped after max_new_batches, -1 error. */
int sweep_impl(const char *const *paths, int n_paths, int ingest_options, int n_threads,
               int alg, double probe, int resolution, long long batch_atoms,
               double *totals_out, double *class_sums_out, long long *atoms_out, int *status_out,
               const char *done_path, long long max_new_batches, const int *devices, int n_devices, char *err_out, int err_len)
{
    if (err_out && err_len > 0) err_out[0] = 0;
    if (!paths || n_paths < 0 || !totals_out || !status_out) return set_err(err_out, err_len, "null argument");
    if (alg != 0 && alg != 1) return set_err(err_out, err_len, "unknown algorithm");
    if (check_devices(devices, n_devices, err_out, err_len)) return -1;
    if (n_paths == 0) return 0;
    /* (round 6, MI355X box, 1.2e7 protein atoms in 7172 files, 16 CPUs, two workers on the device, parser on the device,
       batches of 5e5 / 1e6 / 1.5e6 / 2e6 atoms: 2.2 / 2.5 / 2.4 / 2.3e8 atoms/s once the workers' page-locked staging stays
       with their contexts - tools/dev/sweep_profile.py; with a staging buffer allocated and freed per call, as until the
       round's last session: 1.7 / 1.5 / - / 1.1e8; host parser 1.25 / 1.34 / - / 1.31e8) */
    if (batch_atoms <= 0) batch_atoms = 1000000;
    /* ONE device in the list: two workers on it, so that the upload of one batch runs under the kernels of the other (a
       worker's batch is a chain on one stream: text or arrays over PCIe, parse, cell sort, tile kernels) */
    const int two[2] = {devices[0], devices[0]};
    if (n_devices == 1 && !getenv("FREESASA_AMD_SWEEP_ONE_WORKER")) { devices = two; n_devices = 2; }
    return guarded(err_out, err_len, [&]() -> int {
    /* batches of roughly batch_atoms atoms, estimated from the file sizes (~81 bytes per ATOM line) */
    std::vector<int> cut(1, 0);
    std::vector<long long> batch_bytes;
    {
        long long bytes = 0;
        for (int k = 0; k < n_paths; ++k) {
            struct stat st;
            bytes += (paths[k] && stat(paths[k], &st) == 0) ? (long long)st.st_size : 0;
            if (bytes >= batch_atoms * 81 && k + 1 < n_paths) { cut.push_back(k + 1); batch_bytes.push_back(bytes); bytes = 0; }
        }
        cut.push_back(n_paths);
        batch_bytes.push_back(bytes);
    }
    const int n_batches = (int)cut.size() - 1;
    /* done-list and result file */
    std::vector<char> done((size_t)n_batches, 0);
    Fd f_done, f_res; /* (closed on every way out) */
    int &fd_done = f_done.fd, &fd_res = f_res.fd;
    if (done_path) {
        unsigned long long h = 1469598103934665603ULL; /* FNV-1a over the files' names, sizes and modification times: the done-list belongs to THESE files as they are now */
        for (int k = 0; k < n_paths; ++k) {
            for (const char *q = paths[k] ? paths[k] : ""; ; ++q) { h = (h ^ (unsigned char)*q) * 1099511628211ULL; if (!*q) break; }
            struct stat st;
            long long id[3] = {-1, -1, -1};
            if (paths[k] && stat(paths[k], &st) == 0) { id[0] = (long long)st.st_size; id[1] = (long long)st.st_mtim.tv_sec; id[2] = (long long)st.st_mtim.tv_nsec; }
            for (size_t q = 0; q < sizeof id; ++q) h = (h ^ ((const unsigned char *)id)[q]) * 1099511628211ULL;
        }
        char head[256];
        snprintf(head, sizeof head, "freesasa_amd sweep done-list v2 n_files=%d batches=%d files=%016llx options=%d alg=%d resolution=%d probe=%.17g\n",
                 n_paths, n_batches, h, ingest_options & ~FREESASA_INGEST_PARSE_ON_DEVICE, alg, resolution, probe); /* (who parses does not change a result: not part of the run's name) */
        const std::string res_path = std::string(done_path) + ".bin";
        bool resume = false;
        if (FILE *fp = fopen(done_path, "r")) {
            char line[256];
            if (fgets(line, sizeof line, fp)) {
                if (strcmp(line, head) != 0) { fclose(fp); return set_err(err_out, err_len, "the done-list belongs to a sweep with other parameters or other (changed) input files"); }
                resume = true;
                int b, first, count;
                while (fgets(line, sizeof line, fp))
                    if (sscanf(line, "shard %d %d %d", &b, &first, &count) == 3 && b >= 0 && b < n_batches && first == cut[b] &&
                        count == cut[b + 1] - cut[b] && line[strlen(line) - 1] == '\n')
                        done[(size_t)b] = 1;
            }
            fclose(fp);
        }
        fd_res = open(res_path.c_str(), resume ? O_RDWR | O_CREAT : O_RDWR | O_CREAT | O_TRUNC, 0644);
        fd_done = open(done_path, resume ? O_WRONLY | O_APPEND : O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd_res < 0 || fd_done < 0 || (!resume && write(fd_done, head, strlen(head)) != (ssize_t)strlen(head)))
            return set_err(err_out, err_len, "cannot open the done-list or its result file");
        for (int b = 0; b < n_batches; ++b) { /* results of the batches already done */
            if (!done[(size_t)b]) continue;
            std::vector<SweepRec> recs((size_t)(cut[b + 1] - cut[b]));
            if (!pread_all(fd_res, recs.data(), sizeof(SweepRec) * recs.size(), (long long)sizeof(SweepRec) * cut[b])) { done[(size_t)b] = 0; continue; }
            for (size_t k = 0; k < recs.size(); ++k) {
                const int f = cut[b] + (int)k;
                totals_out[f] = recs[k].total; status_out[f] = recs[k].status;
                if (atoms_out) atoms_out[f] = recs[k].atoms;
                if (class_sums_out) for (int q = 0; q < 3; ++q) class_sums_out[3 * f + q] = recs[k].cls[q];
            }
        }
    }
    std::vector<int> todo;
    for (int b = 0; b < n_batches; ++b)
        if (!done[(size_t)b]) todo.push_back(b);
    bool stopped = false;
    if (max_new_batches > 0 && (long long)todo.size() > max_new_batches) { todo.resize((size_t)max_new_batches); stopped = true; }
    if (todo.empty()) return stopped ? 1 : 0;
    /* largest first (LPT): whoever is free takes the largest batch left */
    std::stable_sort(todo.begin(), todo.end(), [&](int x, int y) { return batch_bytes[(size_t)x] > batch_bytes[(size_t)y]; });
    int n_workers = n_devices < (int)todo.size() ? n_devices : (int)todo.size();
    const int loader_threads = threads_per_worker(n_threads, n_workers);
    const bool want_cls = class_sums_out != nullptr || done_path != nullptr;
    std::vector<double> tp;
    if (alg == 1) { tp.resize(3 * (size_t)(resolution > 0 ? resolution : 1)); if (resolution > 0) freesasa_gpu_test_points(resolution, tp.data()); }
    std::atomic<size_t> next(0);
    std::mutex done_mu;
    FirstError fe;

    const bool dev_parse = (ingest_options & FREESASA_INGEST_PARSE_ON_DEVICE) != 0;
    ingest_options &= ~FREESASA_INGEST_PARSE_ON_DEVICE;
    /* the results of a finished batch to the result file, then its line in the done-list (both modes) */
    auto record = [&](freesasa_gpu_ctx *c, int b, int first, int ns, const long long *atoms, const double *cls_src) -> int {
        std::vector<SweepRec> recs((size_t)ns);
        for (int k = 0; k < ns; ++k) {
            SweepRec &r = recs[(size_t)k];
            memset(&r, 0, sizeof r);
            r.total = totals_out[first + k]; r.status = status_out[first + k];
            r.atoms = atoms ? atoms[k] : 0;
            if (cls_src) for (int q = 0; q < 3; ++q) r.cls[q] = cls_src[3 * k + q];
        }
        char line[96];
        const int len = snprintf(line, sizeof line, "shard %d %d %d\n", b, first, ns);
        std::lock_guard<std::mutex> lk(done_mu); /* (the records of a batch lie at their own offset; the list is appended to by one worker at a time) */
        if (!pwrite_all(fd_res, recs.data(), sizeof(SweepRec) * recs.size(), (long long)sizeof(SweepRec) * first) || fdatasync(fd_res) != 0 ||
            write(fd_done, line, (size_t)len) != len || fdatasync(fd_done) != 0)
            return ctx_fail(c, "could not record the finished batch in the done-list");
        return 0;
    };

    /* The sweep with the parser ON THE DEVICE (FREESASA_INGEST_PARSE_ON_DEVICE; gpu_parse.hip): the loader threads only READ the
       next batch's files into page-locked memory while this batch's text is uploaded, parsed, classified and swept on the GPU.
       Files the device refuses are read by the host parser and appended to the batch as further structures. */
    /* dev aid (FREESASA_AMD_SWEEP_PROFILE): where a worker's wall clock goes - staging of its first batch, parse (upload, line
       and atom counts back), the files left to the host parser, the tile kernels up to the totals, the done-list, waiting for
       the loader of the next batch */
    const bool sprof = getenv("FREESASA_AMD_SWEEP_PROFILE") != nullptr;
    std::atomic<long long> tp_first(0), tp_parse(0), tp_host(0), tp_run(0), tp_rec(0), tp_join(0), tp_stage(0);
    auto now_ns = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (long long)ts.tv_sec * 1000000000LL + ts.tv_nsec; };
    auto worker_dev = [&](int w) noexcept {
      try {
        DeviceNodeScope node(devices[w]);
        PoolLease lease(devices[w]);
        freesasa_gpu_ctx *c = lease.c;
        if (!c) { fe.set("could not create a GPU context"); return; }
        Staged cur_s(&c->stage_in, &c->stage_in_cap), nxt_s(&c->stage_out, &c->stage_out_cap);
        auto stage = [&](int b, Staged *out) noexcept {
            const long long t0 = sprof ? now_ns() : 0;
            try { stage_files(paths + cut[b], cut[b + 1] - cut[b], ingest_options, loader_threads, out); } catch (...) { out->rc = -3; }
            if (sprof) tp_stage += now_ns() - t0;
        };
        size_t ti = next.fetch_add(1);
        { const long long t0 = sprof ? now_ns() : 0;
          if (ti < todo.size()) stage(todo[ti], &cur_s);
          if (sprof) tp_first += now_ns() - t0; }
        while (ti < todo.size() && !fe.failed.load()) {
            const int b = todo[ti];
            const size_t tn = next.fetch_add(1);
            ThreadGroup loader;
            if (tn < todo.size() && !loader.spawn(stage, todo[tn], &nxt_s)) { fe.set("could not start a loader thread"); break; }
            const int first = cut[b], ns = cut[b + 1] - cut[b];
            std::vector<int> atoms((size_t)ns), status((size_t)ns), host((size_t)ns);
            std::vector<long long> atoms64((size_t)ns, 0);
            std::vector<double> tot, cls;
            std::vector<int> fb;  /* the files the device left to the host parser ... */
            Batch hb;             /* ... as the host read them (alive until the stream is idle: its arrays are copied from) */
            int ret = -1;
            do {
                if (cur_s.rc) { ctx_fail(c, cur_s.rc == -1 ? "out of page-locked host memory (file staging)" : (cur_s.rc == -2 ? "a batch of files larger than 2 GB: use a smaller batch_atoms" : "out of host memory (file staging)")); break; }
                if (hipSetDevice(c->device) != hipSuccess) { ctx_fail(c, "hipSetDevice failed"); break; }
                long long total = 0;
                long long tq = sprof ? now_ns() : 0;
                if (parse_batch_dev_begin(c, cur_s.text, cur_s.T, cur_s.files.data(), ns, ingest_options, atoms.data(), status.data(), host.data(), &total)) break;
                if (sprof) { const long long t1 = now_ns(); tp_parse += t1 - tq; tq = t1; }
                /* the files the device left to the host parser: read now, appended behind the device's atoms */
                for (int k = 0; k < ns; ++k) if (host[(size_t)k]) fb.push_back(k);
                g_parse_dev_files += ns - (long long)fb.size(); g_parse_host_files += (long long)fb.size();
                if (!fb.empty()) {
                    std::vector<const char *> fp;
                    for (int k : fb) fp.push_back(paths[first + k]);
                    const int lrc = freesasa_ingest_pdb_files(fp.data(), (int)fp.size(), ingest_options, loader_threads, &hb.b);
                    if (lrc) { ctx_fail(c, "loader failed with code %d", lrc); break; }
                }
                const long long extra = hb.b.n_atoms, n_all = total + extra;
                if (sprof) { const long long t1 = now_ns(); tp_host += t1 - tq; tq = t1; }
                if (parse_batch_dev_finish(c, extra)) break;
                const int nst = ns + (int)fb.size();
                std::vector<int64_t> off((size_t)nst + 1);
                off[0] = 0;
                for (int k = 0; k < ns; ++k) off[(size_t)k + 1] = off[(size_t)k] + atoms[(size_t)k];
                for (size_t j = 0; j < fb.size(); ++j) off[(size_t)ns + j + 1] = total + hb.b.offsets[j + 1];
                for (int k = 0; k < ns; ++k) {
                    status_out[first + k] = status[(size_t)k];
                    totals_out[first + k] = 0;
                    atoms64[(size_t)k] = atoms[(size_t)k];
                    if (class_sums_out) class_sums_out[3 * (first + k)] = class_sums_out[3 * (first + k) + 1] = class_sums_out[3 * (first + k) + 2] = 0;
                }
                for (size_t j = 0; j < fb.size(); ++j) { status_out[first + fb[j]] = hb.b.status[j]; atoms64[(size_t)fb[j]] = hb.b.offsets[j + 1] - hb.b.offsets[j]; }
                if (atoms_out) for (int k = 0; k < ns; ++k) atoms_out[first + k] = atoms64[(size_t)k];
                if (n_all == 0) { ret = 0; break; }
                if (extra > 0 &&
                    (hipMemcpyAsync((double *)c->h_xyz.p + 3 * total, hb.b.xyz, 24 * (size_t)extra, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                     hipMemcpyAsync((double *)c->h_radii.p + total, hb.b.radii, 8 * (size_t)extra, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                     hipMemcpyAsync((unsigned char *)c->h_counts.p + total, hb.b.atom_class, (size_t)extra, hipMemcpyHostToDevice, c->stream) != hipSuccess)) {
                    ctx_fail(c, "host-to-device copy failed");
                    break;
                }
                if (ensure(c, c->h_sasa, 8 * (size_t)n_all) || ensure(c, c->h_totals, 8 * 4 * (size_t)nst)) break;
                double *d_tot = (double *)c->h_totals.p, *d_cls = d_tot + nst;
                if (run_batch(c, alg == 0, (double *)c->h_xyz.p, (double *)c->h_radii.p, off.data(), nst, probe, resolution,
                              alg == 1 ? tp.data() : nullptr, (double *)c->h_sasa.p, nullptr, d_tot))
                    break;
                tot.resize((size_t)nst);
                if (want_cls) {
                    cls.resize(3 * (size_t)nst);
                    if (freesasa_gpu_class_sums_dev(c, (double *)c->h_sasa.p, (const unsigned char *)c->h_counts.p, off.data(), nst, d_cls)) break;
                    if (hipMemcpyAsync(cls.data(), d_cls, 8 * 3 * (size_t)nst, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { ctx_fail(c, "device-to-host copy failed"); break; }
                }
                if (hipMemcpyAsync(tot.data(), d_tot, 8 * (size_t)nst, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { ctx_fail(c, "device-to-host copy failed"); break; }
                if (hipStreamSynchronize(c->stream) != hipSuccess) { ctx_fail(c, "stream synchronize failed"); break; }
                /* structure k of the batch is file k; structure ns + j is the j-th file the host read */
                for (int k = 0; k < ns; ++k) totals_out[first + k] = tot[(size_t)k];
                for (size_t j = 0; j < fb.size(); ++j) totals_out[first + fb[j]] = tot[(size_t)ns + j];
                if (want_cls) {
                    for (size_t j = 0; j < fb.size(); ++j) for (int q = 0; q < 3; ++q) cls[3 * (size_t)fb[j] + q] = cls[3 * ((size_t)ns + j) + q];
                    if (class_sums_out) memcpy(class_sums_out + 3 * (size_t)first, cls.data(), 8 * 3 * (size_t)ns);
                }
                if (sprof) tp_run += now_ns() - tq;
                ret = 0;
            } while (0);
            if (ret) (void)hipStreamSynchronize(c->stream);
            long long tr = sprof ? now_ns() : 0;
            if (!ret && fd_done >= 0 && record(c, b, first, ns, atoms64.data(), cls.empty() ? nullptr : cls.data())) ret = -1;
            if (ret) fe.set(c->err[0] ? c->err : "GPU sweep failed");
            if (sprof) { const long long t1 = now_ns(); tp_rec += t1 - tr; tr = t1; }
            loader.join();
            if (sprof) tp_join += now_ns() - tr;
            cur_s.swap(nxt_s);
            ti = tn;
        }
      } catch (...) {
        fe.set_exception();
      }
    };

    auto worker = [&](int w) noexcept {
      if (dev_parse) { worker_dev(w); return; }
      try {
        std::vector<double> cls_tmp;
        Batch cur_b, nxt_b; /* (declared before the context: freed after its stream is idle) */
        freesasa_ingest_batch &cur = cur_b.b, &nxt = nxt_b.b;
        int cur_rc = 0, nxt_rc = 0;
        DeviceNodeScope node(devices[w]); /* this worker - its context's page-locked memory, its loader threads - on the device's NUMA node */
        PoolLease lease(devices[w]);
        freesasa_gpu_ctx *c = lease.c;
        if (!c) { fe.set("could not create a GPU context"); return; }
        auto load = [&](int b, freesasa_ingest_batch *out, int *rc) noexcept { /* (C code: nothing to catch) */
            *rc = freesasa_ingest_pdb_files(paths + cut[b], cut[b + 1] - cut[b], ingest_options, loader_threads, out);
        };
        size_t ti = next.fetch_add(1);
        if (ti < todo.size()) load(todo[ti], &cur, &cur_rc);
        while (ti < todo.size() && !fe.failed.load()) {
            const int b = todo[ti];
            const size_t tn = next.fetch_add(1); /* the batch this worker does next: read while this one computes */
            ThreadGroup loader; /* (joined before nxt can go away, whatever happens below) */
            if (tn < todo.size() && !loader.spawn(load, todo[tn], &nxt, &nxt_rc)) { fe.set("could not start a loader thread"); break; }
            const int first = cut[b], ns = cut[b + 1] - cut[b];
            int ret = 0;
            do {
                if (cur_rc) { ctx_fail(c, "loader failed with code %d", cur_rc); ret = -1; break; }
                for (int k = 0; k < ns; ++k) {
                    status_out[first + k] = cur.status[k];
                    totals_out[first + k] = 0;
                    if (atoms_out) atoms_out[first + k] = cur.offsets[k + 1] - cur.offsets[k];
                    if (class_sums_out) class_sums_out[3 * (first + k)] = class_sums_out[3 * (first + k) + 1] = class_sums_out[3 * (first + k) + 2] = 0;
                }
                const size_t n = (size_t)cur.n_atoms;
                if (n == 0) break;
                ret = -1;
                if (hipSetDevice(c->device) != hipSuccess) { ctx_fail(c, "hipSetDevice failed"); break; }
                if (ensure(c, c->h_xyz, 24 * n) || ensure(c, c->h_radii, 8 * n) || ensure(c, c->h_sasa, 8 * n) ||
                    ensure(c, c->h_counts, n) || ensure(c, c->h_totals, 8 * 4 * (size_t)ns))
                    break;
                if (hipMemcpyAsync(c->h_xyz.p, cur.xyz, 24 * n, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                    hipMemcpyAsync(c->h_radii.p, cur.radii, 8 * n, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
                    ctx_fail(c, "host-to-device copy failed");
                    break;
                }
                double *d_tot = (double *)c->h_totals.p, *d_cls = d_tot + ns;
                if (run_batch(c, alg == 0, (double *)c->h_xyz.p, (double *)c->h_radii.p, cur.offsets, ns, probe, resolution,
                              alg == 1 ? tp.data() : nullptr, (double *)c->h_sasa.p, nullptr, d_tot))
                    break;
                double *cls_dst = class_sums_out ? class_sums_out + 3 * (size_t)first : nullptr;
                if (want_cls) {
                    if (!cls_dst) { cls_tmp.resize(3 * (size_t)ns); cls_dst = cls_tmp.data(); }
                    if (hipMemcpyAsync(c->h_counts.p, cur.atom_class, n, hipMemcpyHostToDevice, c->stream) != hipSuccess) { ctx_fail(c, "host-to-device copy failed"); break; }
                    if (freesasa_gpu_class_sums_dev(c, (double *)c->h_sasa.p, (const unsigned char *)c->h_counts.p, cur.offsets, ns, d_cls)) break;
                    if (hipMemcpyAsync(cls_dst, d_cls, 8 * 3 * (size_t)ns, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { ctx_fail(c, "device-to-host copy failed"); break; }
                }
                if (hipMemcpyAsync(totals_out + first, d_tot, 8 * (size_t)ns, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { ctx_fail(c, "device-to-host copy failed"); break; }
                if (hipStreamSynchronize(c->stream) != hipSuccess) { ctx_fail(c, "stream synchronize failed"); break; }
                ret = 0;
            } while (0);
            if (ret) (void)hipStreamSynchronize(c->stream); /* no copy may still read the batch when it is freed */
            if (!ret && fd_done >= 0) {
                std::vector<long long> at((size_t)ns, 0);
                if (cur.offsets) for (int k = 0; k < ns; ++k) at[(size_t)k] = cur.offsets[k + 1] - cur.offsets[k];
                const double *cls_src = class_sums_out ? class_sums_out + 3 * (size_t)first : (cur.n_atoms > 0 ? cls_tmp.data() : nullptr);
                if (record(c, b, first, ns, at.data(), cls_src)) ret = -1;
            }
            if (ret) fe.set(c->err[0] ? c->err : "GPU sweep failed");
            loader.join();
            cur_b.take(nxt_b);
            cur_rc = nxt_rc;
            ti = tn;
        }
      } catch (...) { /* (an exception that leaves a thread's function ends the process: it ends the sweep instead) */
        fe.set_exception();
      }
    };
    {
        ThreadGroup tg;
        for (int w = 1; w < n_workers; ++w)
            if (!tg.spawn(worker, w)) { fe.set("could not start a worker thread"); break; }
        if (!fe.failed.load()) worker(0);
    }
    if (sprof && dev_parse)
        fprintf(stderr, "sweep profile (%d workers, %zu batches, %d loader threads each; ms summed over the workers): first batch staged %.1f | parse %.1f | host parser %.1f | "
                        "kernels to totals %.1f | done-list %.1f | waiting for the loader %.1f || staging itself (loader threads) %.1f\n",
                n_workers, todo.size(), loader_threads, tp_first / 1e6, tp_parse / 1e6, tp_host / 1e6, tp_run / 1e6, tp_rec / 1e6, tp_join / 1e6, tp_stage / 1e6);
    if (fe.failed.load()) return set_err(err_out, err_len, fe.text);
    return stopped ? 1 : 0;
    });
}

/* ------------------------------------------------------------------ structure sweep: from a binary cache */

/* The sweep of a cache file (freesasa_ingest_save): no parsing, no classification — what is left on the host is to get
 * 33 bytes per atom (coordinates, radius, class) from the file into page-locked memory, which one thread does at
 * ~1e8 atoms/s (pread from the page cache + checksum) against 4.5e8 atoms/s of one GPU at protein density.  So every
 * device gets several lanes (threads), each with its own pooled context and page-locked staging: a lane takes the
 * next batch of structures from the shared counter, reads and verifies exactly its run of atoms
 * (freesasa_ingest_cache_read_atoms: piece checksums) into its staging buffer, copies it to the device and computes,
 * while the other lanes are in another stage. */
int sweep_cache_impl(const char *cache_path, int alg, double probe, int resolution, long long batch_atoms,
                     double *totals_out, double *class_sums_out, long long *atoms_out, int *status_out, int n_out,
                     const int *devices, int n_devices, int lanes_per_device, char *err_out, int err_len)
{
    if (err_out && err_len > 0) err_out[0] = 0;
    if (!cache_path || !totals_out) return set_err(err_out, err_len, "null argument");
    if (alg != 0 && alg != 1) return set_err(err_out, err_len, "unknown algorithm");
    if (resolution <= 0) return set_err(err_out, err_len, "resolution must be > 0");
    if (check_devices(devices, n_devices, err_out, err_len)) return -1;
    return guarded(err_out, err_len, [&]() -> int {
    Cache cache_h; /* (closed on every way out) */
    const int orc = freesasa_ingest_cache_open(cache_path, &cache_h.c);
    if (orc) {
        char msg[96];
        snprintf(msg, sizeof msg, "cannot open the cache file (freesasa_ingest code %d)", orc);
        return set_err(err_out, err_len, msg);
    }
    freesasa_ingest_cache *const cache = cache_h.c;
    const int S = freesasa_ingest_cache_n_structs(cache);
    const int64_t *offs = freesasa_ingest_cache_offsets(cache);
    const int32_t *stat = freesasa_ingest_cache_status(cache);
    if (n_out < S) return set_err(err_out, err_len, "the output arrays are shorter than the cache's structure count");
    if (batch_atoms <= 0) batch_atoms = 1000000; /* (measured, round 5, 1.2e7 protein atoms on one MI355X with 16 CPUs: 8 lanes x 1e6 atoms 3.5e8 atoms/s, 4 x 2e6 3.1e8, 2 x 2e6 2.6e8; the kernels alone run 4.5e8 at this density) */
    if (batch_atoms > (1LL << 30)) batch_atoms = 1LL << 30;
    std::vector<int> cut(1, 0);
    for (int s = 0; s < S; ++s) {
        if (offs[s + 1] - offs[s] > (1LL << 30)) return set_err(err_out, err_len, "a structure of the cache is too large for one batch");
        if (offs[s + 1] - offs[cut.back()] > batch_atoms && s > cut.back()) cut.push_back(s); /* (a batch never exceeds batch_atoms unless one structure does) */
    }
    cut.push_back(S);
    const int n_batches = (int)cut.size() - 1;
    for (int s = 0; s < S; ++s) {
        totals_out[s] = 0;
        if (status_out) status_out[s] = stat[s];
        if (atoms_out) atoms_out[s] = offs[s + 1] - offs[s];
        if (class_sums_out) class_sums_out[3 * s] = class_sums_out[3 * s + 1] = class_sums_out[3 * s + 2] = 0;
    }
    if (lanes_per_device <= 0) {
        /* the granted CPUs divided among the devices, 8 at most; two where they allow (one lane reads while the other
           computes) - but never more lanes in all than twice the CPUs: eight devices on four CPUs get one lane each, not
           sixteen threads that read and checksum in turns (round-5 advisor) */
        const int cpus = process_cpus();
        lanes_per_device = cpus / n_devices;
        if (lanes_per_device > 8) lanes_per_device = 8;
        if (lanes_per_device < 2) lanes_per_device = 2 * cpus >= 2 * n_devices ? 2 : 1;
    }
    if (lanes_per_device > 8) lanes_per_device = 8;
    int n_lanes = lanes_per_device * n_devices;
    if (n_lanes > n_batches) n_lanes = n_batches;
    std::vector<double> tp;
    if (alg == 1) { tp.resize(3 * (size_t)resolution); freesasa_gpu_test_points(resolution, tp.data()); }
    std::atomic<int> next(0);
    FirstError fe;
    auto lane = [&](int id) noexcept {
      try {
        DeviceNodeScope node(devices[id % n_devices]); /* the lane and its page-locked staging on the device's NUMA node */
        PoolLease lease(devices[id % n_devices]);
        freesasa_gpu_ctx *c = lease.c;
        if (!c) { fe.set("could not create a GPU context"); return; }
        std::vector<int64_t> off;
        for (;;) {
            const int b = next.fetch_add(1);
            if (b >= n_batches || fe.failed.load()) break;
            const int s0 = cut[b], ns = cut[b + 1] - cut[b];
            const int64_t a0 = offs[s0];
            const size_t n = (size_t)(offs[s0 + ns] - a0);
            if (n == 0) continue;
            off.resize((size_t)ns + 1);
            for (int i = 0; i <= ns; ++i) off[i] = offs[s0 + i] - a0;
            int rc = -1;
            do {
                if (hipSetDevice(c->device) != hipSuccess) { ctx_fail(c, "hipSetDevice failed"); break; }
                if (ensure(c, c->h_xyz, 24 * n) || ensure(c, c->h_radii, 8 * n) || ensure(c, c->h_sasa, 8 * n) ||
                    ensure(c, c->h_counts, n) || ensure(c, c->h_totals, 8 * 4 * (size_t)ns))
                    break;
                if (ensure_pinned(c, &c->stage_in, &c->stage_in_cap, 33 * n + 64) || ensure_pinned(c, &c->stage_out, &c->stage_out_cap, 8 * 4 * (size_t)ns)) break;
                double *h_xyz = (double *)c->stage_in, *h_r = h_xyz + 3 * n;
                uint8_t *h_cls = (uint8_t *)(h_r + n);
                const int rrc = freesasa_ingest_cache_read_atoms(cache, a0, a0 + (int64_t)n, h_xyz, h_r, class_sums_out ? h_cls : nullptr);
                if (rrc) { ctx_fail(c, "the cache file failed its checksum or could not be read (freesasa_ingest code %d)", rrc); break; }
                if (hipMemcpyAsync(c->h_xyz.p, h_xyz, 24 * n, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                    hipMemcpyAsync(c->h_radii.p, h_r, 8 * n, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                    (class_sums_out && hipMemcpyAsync(c->h_counts.p, h_cls, n, hipMemcpyHostToDevice, c->stream) != hipSuccess)) {
                    ctx_fail(c, "host-to-device copy failed");
                    break;
                }
                double *d_tot = (double *)c->h_totals.p, *d_cls = d_tot + ns;
                if (run_batch(c, alg == 0, (double *)c->h_xyz.p, (double *)c->h_radii.p, off.data(), ns, probe, resolution,
                              alg == 1 ? tp.data() : nullptr, (double *)c->h_sasa.p, nullptr, d_tot))
                    break;
                if (class_sums_out && freesasa_gpu_class_sums_dev(c, (double *)c->h_sasa.p, (const unsigned char *)c->h_counts.p, off.data(), ns, d_cls)) break;
                double *h_out = (double *)c->stage_out;
                if (hipMemcpyAsync(h_out, d_tot, 8 * (size_t)(class_sums_out ? 4 * ns : ns), hipMemcpyDeviceToHost, c->stream) != hipSuccess) { ctx_fail(c, "device-to-host copy failed"); break; }
                if (hipStreamSynchronize(c->stream) != hipSuccess) { ctx_fail(c, "stream synchronize failed"); break; }
                memcpy(totals_out + s0, h_out, 8 * (size_t)ns);
                if (class_sums_out) memcpy(class_sums_out + 3 * (size_t)s0, h_out + ns, 8 * 3 * (size_t)ns);
                rc = 0;
            } while (0);
            if (rc) {
                (void)hipStreamSynchronize(c->stream);
                fe.set(c->err[0] ? c->err : "GPU cache sweep failed");
                break;
            }
        }
      } catch (...) {
        fe.set_exception();
      }
    };
    {
        ThreadGroup tg;
        for (int k = 1; k < n_lanes; ++k)
            if (!tg.spawn(lane, k)) { fe.set("could not start a worker thread"); break; }
        if (n_lanes > 0 && !fe.failed.load()) lane(0);
    }
    if (fe.failed.load()) return set_err(err_out, err_len, fe.text);
    return 0;
    });
}

/* ------------------------------------------------------------------ trajectory driver */

/* Frames of ONE system (same atoms, same radii) are independent structures: a SHARD is a run of frames_per_batch
 * frames that goes through the engine as one batch.  A few host lanes PER DEVICE take shards from a shared counter; a
 * lane owns a pooled context (stream, workspace, page-locked staging) of its device and does, for its shard,
 *     read (memory or frame file) -> host-to-device -> [fp32 frames widened to fp64 on the device: an INPUT format,
 *     the arithmetic stays fp64] -> cell sort + tile kernels -> device-to-host -> write (memory or files)
 * while the other lanes are in another stage.  The radii live once per device context (shared by every frame of
 * a batch).  With a done-list file every finished shard is recorded after its results are on disk; a later call
 * with the same parameters skips the recorded shards: an interrupted run resumes — on any list of devices. */
struct TrajIO {
    const double *mem_in = nullptr; /* frames in host memory (fp64) ... */
    int fd_in = -1;                 /* ... or in a file of raw frames */
    int in_f32 = 0;
    long long in_header = 0;
    double *totals_mem = nullptr, *sasa_mem = nullptr;
    int fd_totals = -1, fd_sasa = -1;
    int fd_done = -1;               /* done-list (append) */
    std::vector<char> done;         /* shards already recorded */
    int out_f32 = 0;                /* per-atom areas written as fp32 (narrowed on the device; an output format) */
};

/* returns 0: all shards done, 1: stopped after max_new shards (more left), -1: error */
int traj_run(TrajIO &io, const double *radii, int n_atoms, long long n_frames, int alg, double probe, int resolution,
             int frames_per_batch, int lanes_per_device, long long max_new, const int *devices, int n_devices, char *err_out, int err_len)
{
    return guarded(err_out, err_len, [&]() -> int {
    const size_t n = (size_t)n_atoms, FB = (size_t)frames_per_batch;
    const long long n_shards = (n_frames + frames_per_batch - 1) / frames_per_batch;
    if (io.done.size() < (size_t)n_shards) io.done.resize((size_t)n_shards, 0);
    if (lanes_per_device <= 0) {
        /* three lanes keep one device's PCIe in, kernels and PCIe out busy (measured, round 2; round 6, from and to files on the
           MI355X box, 600 frames x 100 000 atoms: 3 lanes 2.97e8, 6 lanes 3.10e8 atom-frames/s with two of the three contexts
           cold - the kernel trace shows the tile kernels of the lanes back to back, 2.9 ms per shard of 1.2e6 atoms: the
           driver runs at the rate of the kernels); with several devices the lanes also share the granted
           CPUs (a lane reads, copies and writes on the host): two each at least */
        const int per = process_cpus() / n_devices;
        lanes_per_device = per >= 3 ? 3 : 2;
        if (const char *e = getenv("FREESASA_AMD_TRAJ_LANES")) lanes_per_device = atoi(e) > 0 ? atoi(e) : lanes_per_device; /* tuning aid */
    }
    if (lanes_per_device > 8) lanes_per_device = 8;
    int n_lanes = lanes_per_device * n_devices;
    if (n_lanes > n_shards) n_lanes = (int)n_shards;
    std::vector<double> tp;
    if (alg == 1) { tp.resize(3 * (size_t)resolution); freesasa_gpu_test_points(resolution, tp.data()); }
    std::vector<int64_t> offs(FB + 1);
    for (size_t k = 0; k <= FB; ++k) offs[k] = (int64_t)(k * n);
    const bool in_pinned = io.mem_in && host_pinned(io.mem_in);
    const bool out_pinned = io.totals_mem && host_pinned(io.totals_mem) && (!io.sasa_mem || host_pinned(io.sasa_mem));
    const bool want_sasa = io.sasa_mem || io.fd_sasa >= 0;
    std::atomic<long long> next(0), fresh(0);
    std::atomic<int> stopped(0);
    std::mutex done_mu;
    FirstError fe;
    /* dev aid (FREESASA_AMD_TRAJ_PROFILE): where the lanes' host time goes - read, waiting for the device, write, flush */
    const bool prof = getenv("FREESASA_AMD_TRAJ_PROFILE") != nullptr;
    std::atomic<long long> t_read(0), t_dev(0), t_write(0), t_flush(0);
    auto now_ns = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (long long)ts.tv_sec * 1000000000LL + ts.tv_nsec; };
    auto lane = [&](int id) noexcept {
      try {
        DeviceNodeScope node(devices[id % n_devices]); /* the lane and its page-locked staging on the device's NUMA node */
        PoolLease lease(devices[id % n_devices]); /* lanes 0 .. n_devices-1 open one device each, the next n_devices the second lane of each, ... */
        freesasa_gpu_ctx *c = lease.c;
        if (!c) { fe.set("could not create a GPU context"); return; }
        bool radii_up = false;
        for (;;) {
            const long long k = next.fetch_add(1);
            if (k >= n_shards || fe.failed.load()) break;
            if (io.done[(size_t)k]) continue;
            if (max_new > 0 && fresh.fetch_add(1) >= max_new) { stopped = 1; break; }
            const long long f0 = k * frames_per_batch;
            const int nf = (int)(n_frames - f0 < frames_per_batch ? n_frames - f0 : frames_per_batch);
            const size_t na = n * (size_t)nf;
            const size_t in_bytes = (io.in_f32 ? 12 : 24) * na;
            int rc = -1;
            do {
                if (hipSetDevice(c->device) != hipSuccess) { ctx_fail(c, "hipSetDevice failed"); break; }
                if (ensure(c, c->h_xyz, 24 * n * FB) || ensure(c, c->h_radii, 8 * n) || ensure(c, c->h_sasa, 8 * n * FB) ||
                    ensure(c, c->h_totals, 8 * FB) || ((io.in_f32 || io.out_f32) && ensure(c, c->h_counts, (io.in_f32 ? 12 * n * FB : 0) + (io.out_f32 ? 4 * n * FB : 0))))
                    break;
                if (!radii_up) { /* once per lane: the radii of the system */
                    if (hipMemcpyAsync(c->h_radii.p, radii, 8 * n, hipMemcpyHostToDevice, c->stream) != hipSuccess) { ctx_fail(c, "radii upload failed"); break; }
                    radii_up = true;
                }
                const void *src;
                long long tp0 = prof ? now_ns() : 0;
                if (io.mem_in && in_pinned) {
                    src = io.mem_in + 3 * n * (size_t)f0;
                } else {
                    if (ensure_pinned(c, &c->stage_in, &c->stage_in_cap, in_bytes)) break;
                    if (io.mem_in) memcpy(c->stage_in, io.mem_in + 3 * n * (size_t)f0, in_bytes);
                    else if (!pread_all(io.fd_in, c->stage_in, in_bytes, io.in_header + (long long)(io.in_f32 ? 12 : 24) * (long long)n * f0)) {
                        ctx_fail(c, "could not read frames %lld..%lld of the frame file", f0, f0 + nf - 1);
                        break;
                    }
                    src = c->stage_in;
                }
                if (prof) { const long long t = now_ns(); t_read += t - tp0; tp0 = t; }
                if (io.in_f32) {
                    if (hipMemcpyAsync(c->h_counts.p, src, in_bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) { ctx_fail(c, "host-to-device copy failed"); break; }
                    if (kl_widen_f32((const float *)c->h_counts.p, (double *)c->h_xyz.p, (long long)(3 * na), c->stream) != hipSuccess) { ctx_fail(c, "widening launch failed"); break; }
                } else if (hipMemcpyAsync(c->h_xyz.p, src, in_bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
                    ctx_fail(c, "host-to-device copy failed");
                    break;
                }
                c->shared_radii = true;
                const int rb = run_batch(c, alg == 0, (double *)c->h_xyz.p, (double *)c->h_radii.p, offs.data(), nf, probe, resolution,
                                         alg == 1 ? tp.data() : nullptr, (double *)c->h_sasa.p, nullptr, (double *)c->h_totals.p);
                c->shared_radii = false;
                if (rb) break;
                double *dst_tot = io.totals_mem ? io.totals_mem + f0 : nullptr, *dst_sasa = io.sasa_mem ? io.sasa_mem + n * (size_t)f0 : nullptr;
                const bool staged = !(io.totals_mem && out_pinned);
                const size_t eb = io.out_f32 ? 4 : 8; /* bytes per per-atom area in the result file */
                if (staged) {
                    if (ensure_pinned(c, &c->stage_out, &c->stage_out_cap, 8 * (size_t)nf + (want_sasa ? 8 * na : 0))) break;
                    dst_tot = (double *)c->stage_out;
                    dst_sasa = want_sasa ? (double *)c->stage_out + nf : nullptr;
                }
                const void *d_areas = c->h_sasa.p;
                if (want_sasa && io.out_f32) { /* (file output only) narrowed on the device: half the bytes over PCIe and into the file */
                    if (ensure(c, c->h_counts, 4 * n * FB + (io.in_f32 ? 12 * n * FB : 0))) break;
                    float *d32 = (float *)((char *)c->h_counts.p + (io.in_f32 ? 12 * n * FB : 0));
                    if (kl_narrow_f64((const double *)c->h_sasa.p, d32, (long long)na, c->stream) != hipSuccess) { ctx_fail(c, "narrowing launch failed"); break; }
                    d_areas = d32;
                }
                bool ok = hipMemcpyAsync(dst_tot, c->h_totals.p, 8 * (size_t)nf, hipMemcpyDeviceToHost, c->stream) == hipSuccess;
                if (ok && want_sasa) ok = hipMemcpyAsync(dst_sasa, d_areas, eb * na, hipMemcpyDeviceToHost, c->stream) == hipSuccess;
                if (!ok) { ctx_fail(c, "device-to-host copy failed"); break; }
                if (hipStreamSynchronize(c->stream) != hipSuccess) { ctx_fail(c, "stream synchronize failed"); break; }
                if (prof) { const long long t = now_ns(); t_dev += t - tp0; tp0 = t; }
                const long long sasa_off = (long long)eb * (long long)n * f0;
                if (staged) {
                    if (io.totals_mem) memcpy(io.totals_mem + f0, dst_tot, 8 * (size_t)nf);
                    if (io.sasa_mem) memcpy(io.sasa_mem + n * (size_t)f0, dst_sasa, 8 * na);
                    if (io.fd_totals >= 0 && !pwrite_all(io.fd_totals, dst_tot, 8 * (size_t)nf, 8 * f0)) { ctx_fail(c, "could not write the totals file"); break; }
                    if (io.fd_sasa >= 0 && !pwrite_all(io.fd_sasa, dst_sasa, eb * na, sasa_off)) { ctx_fail(c, "could not write the per-atom file"); break; }
                }
                if (prof) { const long long t = now_ns(); t_write += t - tp0; tp0 = t; }
                if (io.fd_done >= 0) { /* results first, then the record: a shard is listed only when its numbers are on disk */
                    const bool flushed = (io.fd_totals < 0 || fdatasync(io.fd_totals) == 0) && (io.fd_sasa < 0 || fdatasync(io.fd_sasa) == 0);
                    if (!flushed) {
                        ctx_fail(c, "could not flush the result files: the shard is not listed as done"); break;
                    }
                    char line[96];
                    const int len = snprintf(line, sizeof line, "shard %lld %lld %d\n", k, f0, nf);
                    std::lock_guard<std::mutex> lk(done_mu);
                    if (write(io.fd_done, line, (size_t)len) != len || fdatasync(io.fd_done) != 0) { ctx_fail(c, "could not append to the done-list"); break; }
                }
                if (prof) t_flush += now_ns() - tp0;
                io.done[(size_t)k] = 1;
                rc = 0;
            } while (0);
            if (rc) {
                c->shared_radii = false;
                (void)hipStreamSynchronize(c->stream);
                fe.set(c->err[0] ? c->err : "trajectory shard failed");
                break;
            }
        }
      } catch (...) {
        fe.set_exception();
      }
    };
    {
        ThreadGroup tg;
        for (int k = 1; k < n_lanes; ++k)
            if (!tg.spawn(lane, k)) { fe.set("could not start a worker thread"); break; }
        if (!fe.failed.load()) lane(0);
    }
    if (prof) fprintf(stderr, "trajectory lanes %d: per lane, ms: read %.1f  device (copies + kernels) %.1f  write %.1f  flush + done-list %.1f\n", n_lanes,
                      1e-6 * t_read.load() / n_lanes, 1e-6 * t_dev.load() / n_lanes, 1e-6 * t_write.load() / n_lanes, 1e-6 * t_flush.load() / n_lanes);
    if (fe.failed.load()) return set_err(err_out, err_len, fe.text);
    return stopped.load() ? 1 : 0;
    });
}

} /* namespace */

/* ------------------------------------------------------------------ entry points: sweeps */

/* The device-side parser on its own (tests, tools): n files -> coordinates, radii and classes of the atoms it keeps (host
   arrays of `cap` atoms), offsets_out [n + 1], status_out [n] (the loader's codes), host_out [n] (1: the device refuses the
   file - the sweep would hand it to the host parser - and it contributes nothing here).  Returns the atoms written, -1 on
   error, -2 if cap is too small (offsets_out[n] says how many are needed). */
extern "C" long long freesasa_gpu_parse_files(const char *const *paths, int n_paths, int ingest_options, int n_threads, int device,
                                              double *xyz_out, double *radii_out, unsigned char *class_out, long long cap,
                                              long long *offsets_out, int *status_out, int *host_out, char *err_out, int err_len)
{
    if (err_out && err_len > 0) err_out[0] = 0;
    if (!paths || n_paths <= 0 || !offsets_out || !status_out || !host_out) return set_err(err_out, err_len, "bad argument");
    if (check_devices(&device, 1, err_out, err_len)) return -1;
    long long written = -1;
    const int rc = guarded(err_out, err_len, [&]() -> int {
        PoolLease lease(device);
        freesasa_gpu_ctx *c = lease.c;
        if (!c) return set_err(err_out, err_len, "could not create a GPU context");
        Staged s(&c->stage_in, &c->stage_in_cap);
        if (hipSetDevice(c->device) != hipSuccess) return set_err(err_out, err_len, "hipSetDevice failed");
        stage_files(paths, n_paths, ingest_options & ~FREESASA_INGEST_PARSE_ON_DEVICE, threads_per_worker(n_threads, 1), &s);
        if (s.rc) return set_err(err_out, err_len, "could not stage the files");
        std::vector<int> atoms((size_t)n_paths);
        long long total = 0;
        if (parse_batch_dev_begin(c, s.text, s.T, s.files.data(), n_paths, ingest_options & ~FREESASA_INGEST_PARSE_ON_DEVICE, atoms.data(), status_out, host_out, &total) ||
            parse_batch_dev_finish(c, 0))
            return set_err(err_out, err_len, c->err);
        offsets_out[0] = 0;
        for (int k = 0; k < n_paths; ++k) offsets_out[k + 1] = offsets_out[k] + atoms[(size_t)k];
        if (total > cap) { written = -2; return 0; }
        if (total > 0 &&
            ((xyz_out && hipMemcpyAsync(xyz_out, c->h_xyz.p, 24 * (size_t)total, hipMemcpyDeviceToHost, c->stream) != hipSuccess) ||
             (radii_out && hipMemcpyAsync(radii_out, c->h_radii.p, 8 * (size_t)total, hipMemcpyDeviceToHost, c->stream) != hipSuccess) ||
             (class_out && hipMemcpyAsync(class_out, c->h_counts.p, (size_t)total, hipMemcpyDeviceToHost, c->stream) != hipSuccess) ||
             hipStreamSynchronize(c->stream) != hipSuccess))
            return set_err(err_out, err_len, "device-to-host copy failed");
        written = total;
        return 0;
    });
    return rc ? -1 : written;
}

/* files the sweeps of this process parsed on the device / left to the host parser since the last call (FREESASA_INGEST_PARSE_ON_DEVICE) */
extern "C" void freesasa_gpu_sweep_parse_stats(long long *device_files, long long *host_files)
{
    if (device_files) *device_files = g_parse_dev_files.exchange(0);
    if (host_files) *host_files = g_parse_host_files.exchange(0);
}

extern "C" int freesasa_gpu_sweep_files(const char *const *paths, int n_paths, int ingest_options, int n_threads,
                                        int alg, double probe, int resolution, long long batch_atoms,
                                        double *totals_out, double *class_sums_out, long long *atoms_out, int *status_out,
                                        int device, char *err_out, int err_len)
{
    return sweep_impl(paths, n_paths, ingest_options, n_threads, alg, probe, resolution, batch_atoms, totals_out, class_sums_out,
                      atoms_out, status_out, nullptr, 0, &device, 1, err_out, err_len);
}

extern "C" int freesasa_gpu_sweep_files_resumable(const char *const *paths, int n_paths, int ingest_options, int n_threads,
                                                  int alg, double probe, int resolution, long long batch_atoms,
                                                  double *totals_out, double *class_sums_out, long long *atoms_out, int *status_out,
                                                  const char *done_path, long long max_new_batches, int device, char *err_out, int err_len)
{
    return sweep_impl(paths, n_paths, ingest_options, n_threads, alg, probe, resolution, batch_atoms, totals_out, class_sums_out,
                      atoms_out, status_out, done_path, max_new_batches, &device, 1, err_out, err_len);
}

extern "C" int freesasa_gpu_sweep_files_devices(const char *const *paths, int n_paths, int ingest_options, int n_threads,
                                                int alg, double probe, int resolution, long long batch_atoms,
                                                double *totals_out, double *class_sums_out, long long *atoms_out, int *status_out,
                                                const char *done_path, long long max_new_batches, const int *devices, int n_devices,
                                                char *err_out, int err_len)
{
    return sweep_impl(paths, n_paths, ingest_options, n_threads, alg, probe, resolution, batch_atoms, totals_out, class_sums_out,
                      atoms_out, status_out, done_path, max_new_batches, devices, n_devices, err_out, err_len);
}

extern "C" int freesasa_gpu_sweep_cache_devices(const char *cache_path, int alg, double probe, int resolution, long long batch_atoms,
                                                double *totals_out, double *class_sums_out, long long *atoms_out, int *status_out, int n_out,
                                                const int *devices, int n_devices, int lanes_per_device, char *err_out, int err_len)
{
    return sweep_cache_impl(cache_path, alg, probe, resolution, batch_atoms, totals_out, class_sums_out, atoms_out, status_out, n_out,
                            devices, n_devices, lanes_per_device, err_out, err_len);
}

/* ------------------------------------------------------------------ entry points: trajectories */

static int trajectory_mem(const double *xyz_frames, const double *radii, int n_atoms, int n_frames, int alg, double probe, int resolution,
                          int frames_per_batch, double *totals_out, double *sasa_out, const int *devices, int n_devices, char *err_out, int err_len)
{
    if (err_out && err_len > 0) err_out[0] = 0;
    if (!xyz_frames || !radii || !totals_out) return set_err(err_out, err_len, "null argument");
    if (n_atoms <= 0 || n_frames <= 0) return set_err(err_out, err_len, "n_atoms and n_frames must be > 0");
    if (alg != 0 && alg != 1) return set_err(err_out, err_len, "unknown algorithm");
    if (resolution <= 0) return set_err(err_out, err_len, "resolution must be > 0");
    if (check_devices(devices, n_devices, err_out, err_len)) return -1;
    if (frames_per_batch <= 0) frames_per_batch = (int)(1250000 / n_atoms) + 1;
    if (frames_per_batch > n_frames) frames_per_batch = n_frames;
    if ((long long)frames_per_batch * n_atoms > (1LL << 30)) return set_err(err_out, err_len, "batch too large");
    return guarded(err_out, err_len, [&]() -> int {
        TrajIO io;
        io.mem_in = xyz_frames; io.totals_mem = totals_out; io.sasa_mem = sasa_out;
        return traj_run(io, radii, n_atoms, n_frames, alg, probe, resolution, frames_per_batch, 0, 0, devices, n_devices, err_out, err_len) < 0 ? -1 : 0;
    });
}

extern "C" int freesasa_gpu_trajectory(const double *xyz_frames, const double *radii, int n_atoms, int n_frames,
                                       int alg, double probe, int resolution, int frames_per_batch,
                                       double *totals_out, double *sasa_out, int device, char *err_out, int err_len)
{
    return trajectory_mem(xyz_frames, radii, n_atoms, n_frames, alg, probe, resolution, frames_per_batch, totals_out, sasa_out, &device, 1, err_out, err_len);
}

extern "C" int freesasa_gpu_trajectory_devices(const double *xyz_frames, const double *radii, int n_atoms, int n_frames,
                                               int alg, double probe, int resolution, int frames_per_batch,
                                               double *totals_out, double *sasa_out, const int *devices, int n_devices, char *err_out, int err_len)
{
    return trajectory_mem(xyz_frames, radii, n_atoms, n_frames, alg, probe, resolution, frames_per_batch, totals_out, sasa_out, devices, n_devices, err_out, err_len);
}

/* Frame file -> result files, resumable (include/freesasa_gpu.h has the formats).  The done-list names its run: the
 * parameters, the frame file's size and modification time and a checksum of the radii — NOT the devices: a run
 * interrupted on eight GPUs may be finished on one, with the same files byte for byte. */
extern "C" int freesasa_gpu_trajectory_file_devices(const char *frames_path, int frames_f32, long long header_bytes, const double *radii,
                                                    int n_atoms, long long n_frames, int alg, double probe, int resolution,
                                                    int frames_per_batch, const char *totals_path, const char *sasa_path,
                                                    const char *done_path, long long max_new_shards, const int *devices, int n_devices,
                                                    long long *frames_total_out, char *err_out, int err_len)
{
    if (err_out && err_len > 0) err_out[0] = 0;
    if (!frames_path || !radii || !totals_path) return set_err(err_out, err_len, "null argument");
    if (n_atoms <= 0 || header_bytes < 0) return set_err(err_out, err_len, "bad argument");
    if (alg != 0 && alg != 1) return set_err(err_out, err_len, "unknown algorithm");
    if (resolution <= 0) return set_err(err_out, err_len, "resolution must be > 0");
    if (check_devices(devices, n_devices, err_out, err_len)) return -1;
    return guarded(err_out, err_len, [&]() -> int {
    TrajIO io;
    struct Closer { /* (the descriptors are closed on every way out) */
        TrajIO &io;
        ~Closer() { for (int fd : {io.fd_in, io.fd_totals, io.fd_sasa, io.fd_done}) if (fd >= 0) close(fd); }
    } closer{io};
    int ret = -1;
    do {
        io.fd_in = open(frames_path, O_RDONLY);
        if (io.fd_in < 0) { set_err(err_out, err_len, "cannot open the frame file"); break; }
        struct stat st;
        if (fstat(io.fd_in, &st) != 0) { set_err(err_out, err_len, "cannot stat the frame file"); break; }
        const long long frame_bytes = ((frames_f32 & 1) ? 12LL : 24LL) * n_atoms;
        const long long in_file = ((long long)st.st_size - header_bytes) / frame_bytes;
        if (n_frames <= 0) n_frames = in_file;
        if (n_frames <= 0 || n_frames > in_file) { set_err(err_out, err_len, "the frame file holds fewer frames than asked for"); break; }
        if (frames_total_out) *frames_total_out = n_frames;
        if (frames_per_batch <= 0) frames_per_batch = (int)(1250000 / n_atoms) + 1;
        if (frames_per_batch > n_frames) frames_per_batch = (int)n_frames;
        if ((long long)frames_per_batch * n_atoms > (1LL << 30)) { set_err(err_out, err_len, "batch too large"); break; }
        io.in_f32 = (frames_f32 & 1) ? 1 : 0; io.in_header = header_bytes;
        io.out_f32 = (frames_f32 & 2) ? 1 : 0;
        const long long n_shards = (n_frames + frames_per_batch - 1) / frames_per_batch;
        io.done.assign((size_t)n_shards, 0);
        unsigned long long hr = 1469598103934665603ULL; /* FNV-1a over the radii */
        for (size_t q = 0; q < 8 * (size_t)n_atoms; ++q) hr = (hr ^ ((const unsigned char *)radii)[q]) * 1099511628211ULL;
        char head[384];
        snprintf(head, sizeof head, "freesasa_amd trajectory done-list v2 n_atoms=%d n_frames=%lld frames_per_batch=%d alg=%d resolution=%d probe=%.17g f32=%d "
                 "header_bytes=%lld frames_size=%lld frames_mtime=%lld.%09ld radii=%016llx\n",
                 n_atoms, n_frames, frames_per_batch, alg, resolution, probe, io.in_f32 | (io.out_f32 << 1), header_bytes, (long long)st.st_size,
                 (long long)st.st_mtim.tv_sec, (long)st.st_mtim.tv_nsec, hr);
        bool resume = false;
        if (done_path) {
            FILE *fp = fopen(done_path, "r");
            if (fp) {
                char line[384];
                if (fgets(line, sizeof line, fp)) {
                    if (strcmp(line, head) != 0) { fclose(fp); set_err(err_out, err_len, "the done-list belongs to a run with other parameters, radii or frame file"); break; }
                    resume = true;
                    long long k, f0; int nf;
                    while (fgets(line, sizeof line, fp))
                        if (sscanf(line, "shard %lld %lld %d", &k, &f0, &nf) == 3 && k >= 0 && k < n_shards && f0 == k * frames_per_batch &&
                            line[strlen(line) - 1] == '\n') /* (a record cut short by a crash does not count) */
                            io.done[(size_t)k] = 1;
                }
                fclose(fp);
            }
        }
        io.fd_totals = open(totals_path, resume ? O_WRONLY | O_CREAT : O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (io.fd_totals < 0) { set_err(err_out, err_len, "cannot open the totals file"); break; }
        if (sasa_path) {
            io.fd_sasa = open(sasa_path, resume ? O_WRONLY | O_CREAT : O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (io.fd_sasa < 0) { set_err(err_out, err_len, "cannot open the per-atom file"); break; }
        }
        if (done_path) {
            io.fd_done = open(done_path, resume ? O_WRONLY | O_APPEND : O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (io.fd_done < 0) { set_err(err_out, err_len, "cannot open the done-list"); break; }
            if (!resume && write(io.fd_done, head, strlen(head)) != (ssize_t)strlen(head)) { set_err(err_out, err_len, "cannot write the done-list"); break; }
        }
        ret = traj_run(io, radii, n_atoms, n_frames, alg, probe, resolution, frames_per_batch, 0, max_new_shards, devices, n_devices, err_out, err_len);
    } while (0);
    return ret;
    });
}

extern "C" int freesasa_gpu_trajectory_file(const char *frames_path, int frames_f32, long long header_bytes, const double *radii,
                                            int n_atoms, long long n_frames, int alg, double probe, int resolution,
                                            int frames_per_batch, const char *totals_path, const char *sasa_path,
                                            const char *done_path, long long max_new_shards, int device,
                                            long long *frames_total_out, char *err_out, int err_len)
{
    return freesasa_gpu_trajectory_file_devices(frames_path, frames_f32, header_bytes, radii, n_atoms, n_frames, alg, probe, resolution,
                                                frames_per_batch, totals_path, sasa_path, done_path, max_new_shards, &device, 1,
                                                frames_total_out, err_out, err_len);
}
