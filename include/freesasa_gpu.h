/*
 * freesasa_gpu.h — ADDITIVE C-ABI of the MI355X engine (nothing here exists in the
 * reference; freesasa_amd.h stays byte-compatible with the reference's freesasa.h).
 *
 * The reference computes one structure per call (src/freesasa.c:76-120) and parallelises
 * with <= 16 pthreads inside it (src/sasa_lr.c:219-253).  A GPU needs many structures per
 * launch, so the engine's native unit is a BATCH of independent structures in CSR form.
 * Plain pointers and sizes only; no torch / HIP types in any signature (a stream is passed
 * as void*).
 *
 * NUMERIC DOMAIN (what the parity tests prove; north_star's contract is 1e-4 A^2 per atom against the reference).
 * Inputs: finite coordinates and radii (anything else is FREESASA_FAIL with a message, never garbage), radius + probe
 * > 0, at most 2^30 atoms per batch and 2^30 cells (cell edge 2 max(R + probe), ref: src/nb.c:543).
 *   Shrake-Rupley: test-point counts and areas are the reference's bit for bit, for any coordinates and any number of
 *   points (tests/test_gpu_parity.py, tests/test_deep_parity.py: no atom of 1e6 differs).
 *   Lee-Richards, |coordinate| <= 1e5 A (tested to 5e4 A), radii 0.1 .. 30 A, probe 0 .. 5 A, 1 .. 20000 slices:
 *   per-atom |dSASA| <= 1e-8 A^2 on ordinary structures (asserted; measured <= 1e-9 over 6e6 atoms) - slice planes in
 *   closed form for atoms with |z| <= 1024 A, walked exactly as the reference walks them (src/sasa_lr.c:304-307)
 *   beyond, so the accuracy does not depend on the distance from the origin.  Inputs CONSTRUCTED so that two slice
 *   circles are tangent to the last bits - where the reference's own three comparisons and its acos argument disagree
 *   and it returns NaN or a full circle for a covered one (src/sasa_lr.c:324-351) - get the value of the reference
 *   just outside that band: <= 3e-5 A^2 for |z| <= 1024 A, <= 1e-6 beyond (tests/test_adversarial.py).
 *   Two atoms at the same position with equal radii: NaN, as the reference (0 / 0 in its acos argument).
 */
#ifndef FREESASA_GPU_H
#define FREESASA_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct freesasa_gpu_ctx freesasa_gpu_ctx;

/* Per-call statistics of the last batch run on a context. */
typedef struct freesasa_gpu_stats {
    long long n_atoms;      /* atoms processed */
    long long n_cells;      /* cells of the batch-wide cell list */
    int n_structs;
    int max_neighbors;      /* largest neighbor count of any atom */
    int fallback_tiles;     /* tiles re-done by the large-capacity fallback launch */
    int tile_atoms;         /* launch configuration of the fused kernel */
    int block_threads;
    int lds_bytes;
    double ms_prep;         /* HIP-event time of the cell-sort pipeline (0 unless timing on) */
    double ms_kernel;       /* HIP-event time of the fused L&R / S&R kernel */
    double ms_total;        /* HIP-event time of the whole call on the stream */
} freesasa_gpu_stats;

/* Number of usable HIP devices (0 when there is none; never fails). */
int freesasa_gpu_device_count(void);

/* A context owns one device's workspace.  stream: a hipStream_t to launch on (e.g. torch's
   current stream) or NULL for a private non-blocking stream.  Work is ordered only with respect
   to THAT stream: device inputs produced asynchronously on another stream must be complete
   (synchronise, or create the context on the producing stream).  Returns NULL on failure. */
freesasa_gpu_ctx *freesasa_gpu_ctx_create(int device, void *stream);
void freesasa_gpu_ctx_destroy(freesasa_gpu_ctx *ctx);
/* Record HIP events around the pipeline stages (adds two syncs per call). */
void freesasa_gpu_ctx_set_timing(freesasa_gpu_ctx *ctx, int enable);
void freesasa_gpu_ctx_get_stats(const freesasa_gpu_ctx *ctx, freesasa_gpu_stats *out);
/* Text of the last error on this context ("" if none). */
const char *freesasa_gpu_ctx_last_error(const freesasa_gpu_ctx *ctx);

/* Device-resident batch.  d_* are DEVICE pointers, offsets is a HOST array [n_structs+1]
   (first atom of each structure; offsets[0] == 0).  d_xyz: x1,y1,z1,... (3 * n_atoms),
   d_radii without probe.  d_sasa [n_atoms] per-atom areas in input order; d_totals
   [n_structs] per-structure sums in atom order (may be NULL).  Work is enqueued on the
   context's stream; the call returns after the results are complete on that stream
   (it synchronises the stream once to size the cell list and once to read the status).
   Returns FREESASA_SUCCESS (0) or FREESASA_FAIL (-1). */
int freesasa_gpu_lr_batch_dev(freesasa_gpu_ctx *ctx, const double *d_xyz, const double *d_radii,
                              const int64_t *offsets, int n_structs, double probe_radius,
                              int n_slices, double *d_sasa, double *d_totals);
/* The same batch, enqueued only: the call returns as soon as the batch is on the context's stream, so that the host
   side of the next batch (argument checks, launches; and on the device its cell sort) follows the tile kernel of
   this one without a gap.  Up to two batches may be in flight on a context; a third call first collects the oldest.
   The inputs of a batch and its offsets' VALUES must stay valid, and its outputs are complete, only after
   freesasa_gpu_wait (or the call that collects it) has returned 0; a failed batch is reported there, with the
   context's error text.  A batch whose cell table turns out too small (a first, very sparse batch) is redone by the
   collecting call, synchronously.  Results are bit-identical to freesasa_gpu_lr_batch_dev's.  Every synchronous
   entry point of the context collects what is in flight first.  Returns 0 (enqueued) / -1. */
int freesasa_gpu_lr_batch_dev_async(freesasa_gpu_ctx *ctx, const double *d_xyz, const double *d_radii,
                                    const int64_t *offsets, int n_structs, double probe_radius,
                                    int n_slices, double *d_sasa, double *d_totals);
/* Collect every batch in flight on the context.  Returns 0, or -1 if one of them failed. */
int freesasa_gpu_wait(freesasa_gpu_ctx *ctx);
/* unit_points: HOST array [3*n_points] of unit test points (generate with
   freesasa_gpu_test_points for bit-exact parity with the reference).  d_counts [n_atoms]
   exposed points per atom (may be NULL).
   Any number of points is accepted (ref: src/sasa_sr.c:56-90, :168-224).  Up to 128 points that are unit vectors (to
   4e-15 in |u|^2: what freesasa_gpu_test_points produces) run the round-6 arrangement: when a context first sees a set of
   points it builds a table of cap masks for them (1.5 MB, a few milliseconds of host time, once per point set), the points
   a neighbor covers are looked up, and only the doubtful (neighbor, point) pairs are put to the reference's test,
   operand for operand (freesasa_amd/csrc/sr_caps.h: counts and areas identical by construction; checked against the
   reference).  Other point sets run the arrangement that tests every point.  The choice changes no result. */
int freesasa_gpu_sr_batch_dev(freesasa_gpu_ctx *ctx, const double *d_xyz, const double *d_radii,
                              const int64_t *offsets, int n_structs, double probe_radius,
                              int n_points, const double *unit_points, double *d_sasa,
                              int *d_counts, double *d_totals);

/* Segmented sums of per-atom areas on the device: out[k] = sum of d_sasa[seg[k] .. seg[k+1]) in
   atom order, for k < n_segs.  seg is a HOST array [n_segs+1] of atom offsets (residues, chains
   or structures: the per-residue / per-chain totals that the reference's result tree computes on
   the host, src/node.c:150-176).  d_out [n_segs] is a device pointer.  Returns 0 / -1. */
int freesasa_gpu_segment_sums_dev(freesasa_gpu_ctx *ctx, const double *d_sasa, const int64_t *seg,
                                  int n_segs, double *d_out);

/* Per-structure sums by atom class: d_out[3*s + c] = sum of d_sasa over the atoms of structure s
   whose d_class byte is c (0 apolar, 1 polar, 2 unknown — the reference's freesasa_atom_class,
   src/freesasa.h:163-167; what freesasa_result_classes adds up on the host,
   src/classifier.c:830-866, and the CLI prints as Apolar / Polar).  offsets is a HOST array
   [n_structs+1]; d_class [n_atoms] and d_out [3*n_structs] are device pointers.  Returns 0 / -1. */
int freesasa_gpu_class_sums_dev(freesasa_gpu_ctx *ctx, const double *d_sasa, const unsigned char *d_class,
                                const int64_t *offsets, int n_structs, double *d_out);

/* Per-residue areas as the reference's result tree holds them (freesasa_nodearea of a residue node,
   src/node.c:717-764): d_abs[6*r + {0..5}] = total, main chain, side chain, polar, apolar, unknown,
   summed in atom order over residue r = atoms [res_first[r], res_first[r+1]); and, if d_rel is not
   NULL, the relative areas of the RSA output (src/rsa.c:14-25): d_rel[5*r + {0..4}] =
   100 * {total, main chain, side chain, polar, apolar} / reference, NaN where the residue has no
   reference values (ref_row[r] < 0; the reference prints N/A).  d_class / d_backbone [n_atoms] are
   device byte arrays (freesasa_ingest_batch.atom_class / .atom_backbone); res_first [n_res+1],
   ref_row [n_res] (freesasa_ingest_batch.res_ref) and ref_table [5*ref_rows]
   (freesasa_ingest_residue_reference_table) are HOST arrays.  Returns 0 / -1. */
int freesasa_gpu_residue_areas_dev(freesasa_gpu_ctx *ctx, const double *d_sasa, const unsigned char *d_class,
                                   const unsigned char *d_backbone, const int64_t *res_first, int n_res,
                                   const short *ref_row, const double *ref_table, int ref_rows,
                                   double *d_abs, double *d_rel);

/* Golden-spiral unit test points on the host, host libm (src/sasa_sr.c:56-90). */
void freesasa_gpu_test_points(int n_points, double *unit_points);

/* The same over several GPUs of the node from one process: the structures are cut into contiguous runs
   of about equal atom count (freesasa_gpu_shard_cuts), one per entry of devices[] — a device may
   appear more than once, its runs then overlap their copies and kernels — each run on its own host
   thread, context and stream; no exchange between devices (independent structures).  Arrays as in
   freesasa_gpu_calc_batch.  _multi takes a bit mask instead (bit d = device d).  Return 0 / -1. */
int freesasa_gpu_calc_batch_devices(const double *xyz, const double *radii, const int64_t *offsets, int n_structs,
                                    int alg, double probe_radius, int resolution, double *sasa_out, int *counts_out,
                                    double *totals_out, const int *devices, int n_devices, char *err, int err_len);
int freesasa_gpu_calc_batch_multi(const double *xyz, const double *radii, const int64_t *offsets, int n_structs,
                                  int alg, double probe_radius, int resolution, double *sasa_out, int *counts_out,
                                  double *totals_out, unsigned device_mask, char *err, int err_len);
/* Host arrays in, host arrays out, with the PCIe copies under the kernels: the batch is cut into chunks of whole
   structures (about chunk_atoms atoms, <= 0: 1.25e6) that n_lanes host threads (<= 0: 3 for page-locked arrays, 4 for
   pageable ones; at most 8) take from a
   shared counter, each lane on its own pooled context and stream, so that the upload of one chunk, the kernels of
   another and the download of a third overlap.  Page-locked caller arrays (hipHostMalloc / hipHostRegister, a
   pinned tensor) are copied by DMA in place; pageable ones go through page-locked staging buffers of the lanes.
   Arrays and results as in freesasa_gpu_calc_batch (bit-identical: chunks are independent structures).
   Return 0 / -1. */
int freesasa_gpu_calc_batch_pipelined(const double *xyz, const double *radii, const int64_t *offsets, int n_structs,
                                      int alg, double probe_radius, int resolution, double *sasa_out, int *counts_out,
                                      double *totals_out, int device, int n_lanes, long long chunk_atoms,
                                      char *err, int err_len);
int freesasa_gpu_trajectory_file(const char *frames_path, int frames_f32, long long header_bytes, const double *radii,
                                 int n_atoms, long long n_frames, int alg, double probe_radius, int resolution,
                                 int frames_per_batch, const char *totals_path, const char *sasa_path, const char *done_path,
                                 long long max_new_shards, int device, long long *frames_total_out, char *err, int err_len);

/* The host-pointer entries (freesasa_calc_coord, freesasa_gpu_calc_batch*, _trajectory, _sweep_files) keep their
   contexts — stream, device workspace, staging buffers — in a pool between calls.  This destroys the idle ones and
   returns their device memory. */
void freesasa_gpu_release_pool(void);

/* Test hook: fault injection.  The n-th device or page-locked-host allocation made by this library from now on
   fails (n <= 0: off), the way the reference's test suite makes its n-th malloc fail (tests/tools.c:10-48,
   tests/test_freesasa.c:475-514): every entry point must then return its failure value with a message, leave
   nothing running on its stream, and work again on the next call. */
void freesasa_gpu_test_fail_after(int n);
/* ... and its HOST-side twin: the n-th allocation (malloc / calloc / realloc of the C sources, operator new of the C++
   sources) or thread creation of the library's own host code from now on fails - the loaders (freesasa_ingest_*), the
   cache reader, the selection parser, freesasa_calc / result_new and every driver below.  n <= 0: off.  Returns what
   was left of the previous countdown (0: it fired, or was not armed).  Every extern "C" entry point catches what its
   C++ code throws (std::bad_alloc, std::system_error from a thread that does not start) and returns its failure value
   with a message: no exception crosses this boundary (ref: src/util.c:89-113, "never exit()"). */
int freesasa_host_test_fail_after(int n);
/* Test hook: the mapping behind the drivers' NUMA placement (a lane, its page-locked staging and its loader threads run on
   the CPUs of the socket its GPU hangs off; FREESASA_AMD_NO_AFFINITY=1 turns it off): the CPUs of the NUMA node of the PCI
   device `pci_address` ("0000:c1:00.0", as hipDeviceGetPCIBusId names it) under the sysfs tree `sysfs_root` ("/sys"; a
   made-up tree in the tests).  Returns how many CPUs the node has (the first `cap` are stored in cpus_out), 0 when the
   platform names no node for the device (numa_node -1), -1 when the tree cannot be read. */
int freesasa_gpu_test_node_cpus(const char *sysfs_root, const char *pci_address, int *cpus_out, int cap);

/* Test hooks: the integer / exact parts of the Lee-Richards kernel, run on the device on their own.
   _lr_neighbors_dev: the neighbor sets it finds (what freesasa_nb_new builds, src/nb.c:524-557; the reference's
   tests/test_nb.c): d_nn[n] = neighbors per atom, d_nb[n * nb_cap] (may be NULL) = the first nb_cap neighbors of
   every atom (original indices); device pointers, original atom order.
   _arc_union_dev: exposed arc length of n_sets (<= 64) sets of arcs given as (start, end) pairs in [0, 2 pi]
   (set k = pairs first[k] .. first[k+1]), through the kernel's arc union and sweep (exposed_arc_length,
   src/sasa_lr.c:389-408; its KATs :455-475); host arrays.  Return 0 / -1. */
int freesasa_gpu_lr_neighbors_dev(freesasa_gpu_ctx *ctx, const double *d_xyz, const double *d_radii, const int64_t *offsets,
                                  int n_structs, double probe_radius, int *d_nn, int *d_nb, int nb_cap);
int freesasa_gpu_arc_union_dev(freesasa_gpu_ctx *ctx, const double *arcs, const int *first, int n_sets, double *out);
/* cuts[0..n_parts]: first structure of every run for the partition above (host-only helper) */
void freesasa_gpu_shard_cuts(const int64_t *offsets, int n_structs, int n_parts, int *cuts);

/* Structure sweep (BASELINE configs[3]): PDB / mmCIF files -> per-structure totals.  The files are
   read in batches of about batch_atoms atoms (<= 0: 1e6) by n_threads host threads
   (include/freesasa_ingest.h; ingest_options are its option bits) while the previous batch is on
   the GPU.  totals_out[n_paths]; class_sums_out[3*n_paths] (apolar, polar, unknown) and
   atoms_out[n_paths] may be NULL; status_out[n_paths] receives the loader's per-input status
   (non-zero: the input contributed nothing and its total is 0).  alg: 0 Lee-Richards (resolution =
   slices), 1 Shrake-Rupley (test points).  Returns 0, or -1 on a GPU error (message in err). */
int freesasa_gpu_sweep_files(const char *const *paths, int n_paths, int ingest_options, int n_threads,
                             int alg, double probe_radius, int resolution, long long batch_atoms,
                             double *totals_out, double *class_sums_out, long long *atoms_out, int *status_out,
                             int device, char *err, int err_len);

/* The same sweep with a done-list (see freesasa_gpu_trajectory_file): done_path holds a first line with the sweep's
   parameters and one line "shard <batch> <first file> <files>" per finished batch; <done_path>.bin holds the
   results of the finished batches (per file a 48-byte record: total, three class sums, atoms, status), written
   before the batch is listed.  A later call with the same files and parameters takes the listed batches' results
   from there and computes only the others.  max_new_batches > 0: stop after that many batches.
   Returns 0 done, 1 stopped early, -1 error. */
int freesasa_gpu_sweep_files_resumable(const char *const *paths, int n_paths, int ingest_options, int n_threads,
                                       int alg, double probe_radius, int resolution, long long batch_atoms,
                                       double *totals_out, double *class_sums_out, long long *atoms_out, int *status_out,
                                       const char *done_path, long long max_new_batches, int device, char *err, int err_len);

/* The drivers over a LIST of devices of the node (BASELINE configs[3] "sharded across 8 x MI355X", configs[4] "frames
   sharded 8 x MI355X with per-frame stream-out"), from one process.  Work units — batches of files, shards of frames —
   are independent, so there is no exchange between devices: the workers of all devices take units from one shared
   list (the file sweep: largest batch first, LPT on the file sizes), every result lands at its own place (the caller's
   arrays; pwrite at the unit's offset of the result files) and ONE done-list serves all devices, so a run interrupted
   on eight devices can be finished on one (or the other way round) with the same files byte for byte.  devices[]: 1 to
   64 entries, each an existing device, repeats allowed (the entries of one device overlap their copies and kernels; a
   one-GPU box runs [0, 0, 0]).  The host threads the CGROUP grants (freesasa_ingest_usable_cpus) are divided among the
   devices' loaders / lanes.  Results are bit-identical to the single-device drivers', which are these with one entry.
   _sweep_files_devices: as freesasa_gpu_sweep_files_resumable (done_path may be NULL, max_new_batches <= 0: all).
   _sweep_cache_devices: the sweep of a binary cache (freesasa_ingest_save): lanes_per_device threads per device
   (<= 0: the granted CPUs divided by the devices, 2 .. 8; batch_atoms <= 0: 1e6) read, VERIFY (1 MiB piece checksums) and upload exactly the
   coordinates, radii and classes of their batch through page-locked staging; n_out = length of the output arrays
   (>= the cache's structure count); class_sums_out / atoms_out / status_out may be NULL.  Returns 0 / -1.
   _trajectory_devices, _trajectory_file_devices: as freesasa_gpu_trajectory / _trajectory_file. */
/* ingest_options | FREESASA_INGEST_PARSE_ON_DEVICE (include/freesasa_ingest.h): the files' text is parsed ON THE DEVICE (round 6;
   the host reads bytes, kernels do what src/structure.c:644-722, src/pdb.c:176-283, src/cif.cc:113-240, src/classifier.c:781-796
   do); results are those of the host parser bit for bit, files the device refuses are read by it.  _sweep_parse_stats: files
   parsed on the device / by the host in this process's sweeps since the last call. */
void freesasa_gpu_sweep_parse_stats(long long *device_files, long long *host_files);
/* The device-side parser on its own: n files -> coordinates [3 * atoms], radii, classes of the atoms it keeps (host arrays with
   room for `cap` atoms; each may be NULL), offsets_out [n + 1], status_out [n] (FREESASA_INGEST_* codes), host_out [n] (1: the
   device refuses the file, which then contributes nothing here; the sweep hands such a file to the host parser).  Returns the
   atoms written, -1 on error, -2 when cap is too small (offsets_out[n] = atoms needed). */
long long freesasa_gpu_parse_files(const char *const *paths, int n_paths, int ingest_options, int n_threads, int device,
                                   double *xyz_out, double *radii_out, unsigned char *class_out, long long cap,
                                   long long *offsets_out, int *status_out, int *host_out, char *err, int err_len);
int freesasa_gpu_sweep_files_devices(const char *const *paths, int n_paths, int ingest_options, int n_threads,
                                     int alg, double probe_radius, int resolution, long long batch_atoms,
                                     double *totals_out, double *class_sums_out, long long *atoms_out, int *status_out,
                                     const char *done_path, long long max_new_batches, const int *devices, int n_devices,
                                     char *err, int err_len);
int freesasa_gpu_sweep_cache_devices(const char *cache_path, int alg, double probe_radius, int resolution, long long batch_atoms,
                                     double *totals_out, double *class_sums_out, long long *atoms_out, int *status_out, int n_out,
                                     const int *devices, int n_devices, int lanes_per_device, char *err, int err_len);
int freesasa_gpu_trajectory_devices(const double *xyz_frames, const double *radii, int n_atoms, int n_frames,
                                    int alg, double probe_radius, int resolution, int frames_per_batch,
                                    double *totals_out, double *sasa_out, const int *devices, int n_devices,
                                    char *err_out, int err_len);
int freesasa_gpu_trajectory_file_devices(const char *frames_path, int frames_f32, long long header_bytes, const double *radii,
                                         int n_atoms, long long n_frames, int alg, double probe_radius, int resolution,
                                         int frames_per_batch, const char *totals_path, const char *sasa_path, const char *done_path,
                                         long long max_new_shards, const int *devices, int n_devices,
                                         long long *frames_total_out, char *err, int err_len);

/* Host-pointer batch on a pooled per-thread context of `device` (-1: current default).
   alg/probe/resolution as in freesasa_parameters; counts_out may be NULL (S&R only);
   totals_out may be NULL.  Thread-safe.  Returns 0 / -1; message via err_out (>= len 1). */
int freesasa_gpu_calc_batch(const double *xyz, const double *radii, const int64_t *offsets,
                            int n_structs, int alg, double probe_radius, int resolution,
                            double *sasa_out, int *counts_out, double *totals_out,
                            int device, char *err_out, int err_len);

/* Trajectory drivers (BASELINE configs[4]): frames of the SAME n_atoms atoms, radii constant.
   Frames are independent structures; a SHARD = frames_per_batch frames (<= 0: about 1.25e6 atoms) goes through the
   engine as one batch.  A few host lanes take shards from a shared counter, each on its own pooled context and
   stream, so that reading / uploading one shard, the kernels of another and the download / writing of a third
   overlap.  The radii are stored ONCE per device context, not once per frame.

   freesasa_gpu_trajectory: frames in HOST memory (frame f at xyz_frames + f*3*n_atoms; DMA in place when the
   array is page-locked), totals_out [n_frames], sasa_out NULL or [n_frames*n_atoms].  Returns 0 / -1.

   freesasa_gpu_trajectory_file: frames from a file of raw little-endian frames (3*n_atoms doubles, or floats when
   bit 0 of frames_f32 is set — an input format: they are widened on the device and all arithmetic is fp64 — at byte
   header_bytes + f * frame size), results to files: totals_path (one double per frame at byte 8*f) and, unless
   NULL, sasa_path (n_atoms doubles per frame; n_atoms FLOATS per frame when bit 1 of frames_f32 is set — an output
   format, round 6: the areas are computed in fp64 and narrowed on the device, half the bytes over PCIe and on disk).
   done_path (may be NULL) is the done-list: a text file whose first
   line holds the run's parameters, followed by one line "shard <k> <first frame> <frames>" per finished shard,
   appended after that shard's results are on disk.  A call that finds the done-list of the same run skips the
   shards listed there, so an interrupted run (crash, kill, max_new_shards) resumes where it stopped and ends
   with the same files, bit for bit, as an uninterrupted one; a done-list with other parameters is an error.
   n_frames <= 0: all whole frames of the file; *frames_total_out (may be NULL) receives the count.
   max_new_shards > 0: stop after that many shards.  Returns 0 all done, 1 stopped early, -1 error. */
int freesasa_gpu_trajectory(const double *xyz_frames, const double *radii, int n_atoms, int n_frames,
                            int alg, double probe_radius, int resolution, int frames_per_batch,
                            double *totals_out, double *sasa_out, int device,
                            char *err_out, int err_len);

#ifdef __cplusplus
}
#endif
#endif /* FREESASA_GPU_H */
