/* freesasa_ingest.h — batched PDB / mmCIF -> (xyz, radius) ingestion for structure sweeps.
 *
 * once the SASA kernels run at 1e8-1e9 atoms/s, reading the structures is the
 * bottleneck of a whole-PDB sweep.  This is a host-side, multi-threaded loader that produces the
 * packed batch the GPU entry points take (include/freesasa_gpu.h: concatenated xyz / radii plus
 * CSR offsets) and, for per-residue results, the residue segments freesasa_gpu_segment_sums_dev
 * reduces over.  It is additive: the reference has no batch reader (its CLI reads one file at a
 * time, src/main.cc:763-779), and the drop-in library keeps using the reference's own parser.
 *
 * What a file contributes is what the reference's freesasa_structure_from_pdb() would hold for it
 * (src/structure.c:644-722 with src/pdb.c:13-283 and the default ProtOr classifier,
 * src/classifier.c:738-796, 1002-1017): same atoms in the same order, same coordinates, same
 * radii, same polar/apolar classes, same residue boundaries.  mmCIF inputs (recognised by their
 * leading data_ block) contribute what freesasa_structure_from_cif() would hold (src/cif.cc:113-240:
 * the _atom_site loop, auth_* columns, lowest model number).  Pinned by tests/test_ingest.py
 * against vectors minted from the reference library for every PDB and mmCIF file of its test suite.
 */
#ifndef FREESASA_INGEST_H
#define FREESASA_INGEST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The reference's freesasa_structure_options bits that apply to one-structure-per-file reading
 * (ref: src/freesasa.h:182-191; same values).  SEPARATE_MODELS / SEPARATE_CHAINS are not
 * supported here: FREESASA_INGEST_EOPTION. */
enum {
    FREESASA_INGEST_INCLUDE_HETATM = 1,
    FREESASA_INGEST_INCLUDE_HYDROGEN = 1 << 2,
    FREESASA_INGEST_JOIN_MODELS = 1 << 5,
    FREESASA_INGEST_HALT_AT_UNKNOWN = 1 << 6,
    FREESASA_INGEST_SKIP_UNKNOWN = 1 << 7,
    FREESASA_INGEST_RADIUS_FROM_OCCUPANCY = 1 << 8
};

/* per-input status */
enum {
    FREESASA_INGEST_OK = 0,
    FREESASA_INGEST_EIO = 1,      /* cannot open / read the file */
    FREESASA_INGEST_EFORMAT = 2,  /* an ATOM line too short for coordinates, or unreadable numbers
                                     (the reference returns NULL: src/structure.c:683-685) */
    FREESASA_INGEST_EEMPTY = 3,   /* no valid ATOM/HETATM line (ref: src/structure.c:710-713) */
    FREESASA_INGEST_EUNKNOWN = 4, /* HALT_AT_UNKNOWN and an atom the classifier does not know */
    FREESASA_INGEST_EOPTION = 5,  /* unsupported option bits */
    FREESASA_INGEST_ENOMEM = 6,
    FREESASA_INGEST_EVERSION = 7  /* a cache file written by an earlier format version (1: rounds 2-4): not damaged, but
                                     this build reads version 2 only - save the batch again (freesasa_ingest_save) */
};

/* atom classes (ref: src/freesasa.h:163-167) */
enum { FREESASA_INGEST_APOLAR = 0, FREESASA_INGEST_POLAR = 1, FREESASA_INGEST_UNKNOWN = 2 };

/* A packed batch.  Structure s owns atoms [offsets[s], offsets[s+1]) and residues
 * [res_offsets[s], res_offsets[s+1]); residue r owns atoms [res_first[r], res_first[r+1]).
 * An input that failed contributes an empty structure (status[s] != 0).  All arrays are
 * allocated by the library (ONE block per batch) and released by freesasa_ingest_free() only. */
typedef struct freesasa_ingest_batch {
    int32_t n_structs;
    int64_t n_atoms;
    int64_t n_residues;
    double *xyz;          /* [3 * n_atoms] x1,y1,z1,... (ref: src/coord.h:26-38 layout) */
    double *radii;        /* [n_atoms] */
    uint8_t *atom_class;  /* [n_atoms] FREESASA_INGEST_APOLAR / POLAR / UNKNOWN */
    uint8_t *atom_backbone; /* [n_atoms] 1 for main-chain atoms (ref: freesasa_atom_is_backbone, src/classifier.c:1090-1109) */
    char *atom_name;      /* [4 * n_atoms] atom names without padding, NUL padded (no terminator when 4 long) */
    char *atom_symbol;    /* [2 * n_atoms] element symbols without padding, NUL padded */
    int64_t *offsets;     /* [n_structs + 1] */
    int64_t *res_first;   /* [n_residues + 1] batch-wide atom index of each residue's first atom */
    int64_t *res_offsets; /* [n_structs + 1] */
    int16_t *res_ref;     /* [n_residues] row of the reference-area table for relative SASA, -1: the classifier
                             does not know the residue (ref: src/classifier.c:853-861) */
    char *res_name;       /* [4 * n_residues] residue names, NUL padded ("ALA\0") */
    char *res_number;     /* [6 * n_residues] residue number incl. insertion code (" 123A\0") */
    char *res_chain;      /* [4 * n_residues] chain label, NUL padded (one character from PDB files, up to
                             three from mmCIF, as in the reference's structure) */
    int32_t *status;      /* [n_structs] */
} freesasa_ingest_batch;

/* CPUs this process may use at once: the affinity mask capped by the cgroup's CPU quota (a GPU box may show 256
 * logical CPUs and grant 16).  What every "n_threads <= 0" default below, and the division of host threads among
 * the devices of the multi-device drivers (include/freesasa_gpu.h), start from. */
int freesasa_ingest_usable_cpus(void);

/* Option bit of the SWEEP drivers only (include/freesasa_gpu.h, freesasa_gpu_sweep_files*; outside the reference's
 * freesasa_structure_options values): parse the files' text ON THE DEVICE.  Host threads only read the bytes into page-locked
 * staging (and find an mmCIF file's _atom_site loop header); kernels find the lines, filter the records, convert the
 * coordinates, classify the atoms and write the batch the tile kernels read (freesasa_amd/csrc/gpu_parse.hip).  Files the
 * device refuses - coordinates that need strtod, lines beyond the reference's 119-byte chunks, RADIUS_FROM_OCCUPANCY,
 * mmCIF outside its everyday one-block, one-row-per-line loop form - are read by the host parser, one by one. */
#define FREESASA_INGEST_PARSE_ON_DEVICE (1 << 16)
/* (what the device parser asks of the host per mmCIF file; see ingest.c) */
int freesasa_ingest_cif_locate(const char *text, size_t len, int *ncol_out, signed char slot_out[12], size_t *row0_out);

/* Read n_paths PDB or mmCIF files with n_threads host threads (<= 0: one per usable CPU -- divided by
 * LOCAL_WORLD_SIZE when a launcher exports it, so the ranks of a node share the cores --, at most 64
 * and at most one per four inputs) into one batch.
 * Returns 0 if the batch could be built (individual failures are in status[]), a
 * FREESASA_INGEST_E* code otherwise (out is zeroed). */
int freesasa_ingest_pdb_files(const char *const *paths, int n_paths, int options, int n_threads,
                              freesasa_ingest_batch *out);

/* Same for PDB / mmCIF texts already in memory (texts[k] has lens[k] bytes, no terminator needed). */
int freesasa_ingest_pdb_texts(const char *const *texts, const size_t *lens, int n_texts, int options,
                              int n_threads, freesasa_ingest_batch *out);

/* Releases a batch THIS LIBRARY built (freesasa_ingest_pdb_files / _pdb_texts / _load / _load_mt) and zeroes the
 * struct.  The arrays of such a batch lie in one block whose header sits 16 bytes before `xyz`; a struct filled in by
 * the caller with arrays of its own must NOT be passed here (the header would be read out of the caller's bounds) - the
 * caller frees those itself.  Freed blocks of 1 MiB and more are kept for the next batches (a sweep builds one every few
 * milliseconds; fresh memory costs it a third of the loader's time in page faults): at most 24 blocks and 1 GiB
 * (environment FREESASA_INGEST_KEEP_MB overrides the cap; 0 keeps nothing).  freesasa_ingest_trim(keep_bytes) gives kept
 * blocks back to the allocator until at most keep_bytes remain and returns the bytes released; unloading the library
 * releases all of them. */
void freesasa_ingest_free(freesasa_ingest_batch *batch);
size_t freesasa_ingest_trim(size_t keep_bytes);

/* A batch on disk (the "binary cache" of SURVEY.md 8(f) N1): a sweep that is run again - another probe radius,
 * another resolution, the other algorithm - starts from one sequential read instead of parsing and classifying
 * every file again.  freesasa_ingest_save() writes every array of the batch (little endian, a 128-byte header
 * with the counts and a checksum; to a temporary name, then renamed); freesasa_ingest_load() gives back an equal
 * batch (released with freesasa_ingest_free) or refuses the file: FREESASA_INGEST_EIO if it cannot be opened or
 * written, FREESASA_INGEST_EFORMAT if it is not a cache file, is truncated, fails its checksum or holds inconsistent
 * offsets (save: if the batch itself is inconsistent), FREESASA_INGEST_EVERSION if it is a cache file of an earlier
 * format version (there is no reader for version 1 in this build: re-save), FREESASA_INGEST_ENOMEM. */
int freesasa_ingest_save(const freesasa_ingest_batch *batch, const char *path);
int freesasa_ingest_load(const char *path, freesasa_ingest_batch *out);
/* The same load with n_threads readers (<= 0: the usable CPUs, at most 8): since version 2 of the file every array is
 * checksummed in 1 MiB pieces, which are read (pread) and verified independently. */
int freesasa_ingest_load_mt(const char *path, int n_threads, freesasa_ingest_batch *out);

/* A cache file read PARTIALLY: what a sweep needs of a run of structures - coordinates, radii, classes - and nothing
 * else, verified piece by piece, straight into the caller's (e.g. page-locked) buffers; several threads may read from
 * one handle at once.  _open reads and verifies the header, the checksum table, the structure offsets and the
 * per-input status values; _read_atoms fills xyz [3 * (a1 - a0)], radii [a1 - a0], atom_class [a1 - a0] (each may be
 * NULL) with atoms [a0, a1) of the batch.  Return codes as for freesasa_ingest_load. */
typedef struct freesasa_ingest_cache freesasa_ingest_cache;
int freesasa_ingest_cache_open(const char *path, freesasa_ingest_cache **out);
void freesasa_ingest_cache_close(freesasa_ingest_cache *cache);
int32_t freesasa_ingest_cache_n_structs(const freesasa_ingest_cache *cache);
int64_t freesasa_ingest_cache_n_atoms(const freesasa_ingest_cache *cache);
const int64_t *freesasa_ingest_cache_offsets(const freesasa_ingest_cache *cache); /* [n_structs + 1], owned by the handle */
const int32_t *freesasa_ingest_cache_status(const freesasa_ingest_cache *cache);  /* [n_structs] */
int freesasa_ingest_cache_read_atoms(const freesasa_ingest_cache *cache, int64_t a0, int64_t a1, double *xyz, double *radii,
                                     uint8_t *atom_class);

/* The reference's selection language ("name, resn ala+arg and not chain B", src/selection.c,
 * src/parser.y, src/lexer.l) on structure `structure` of a batch: mask_out[i] = 1 for the selected
 * atoms ([offsets[s+1] - offsets[s]] bytes), name_out = the selection's name.  Returns the number of
 * atoms of the structure like freesasa_select_area (src/selection.c:683-742), FREESASA_INGEST_SELECT_WARN
 * when parts of the command were ignored (mask still valid), FREESASA_INGEST_SELECT_FAIL on a syntax
 * error.  The area of the selection is the masked sum of the per-atom SASA. */
#define FREESASA_INGEST_MAX_SELECTION_NAME 50 /* ref: src/freesasa.h:226 */
#define FREESASA_INGEST_SELECT_FAIL (-1)
#define FREESASA_INGEST_SELECT_WARN (-2)
int freesasa_ingest_select(const freesasa_ingest_batch *batch, int structure, const char *command,
                           char name_out[FREESASA_INGEST_MAX_SELECTION_NAME + 1], unsigned char *mask_out);

/* The classifier on its own (ref: freesasa_classifier_radius / _class with the ProtOr classifier,
 * src/classifier.c:781-813): radius in A or -1.0 if unknown; *cls (may be NULL) receives the class. */
double freesasa_ingest_protor_radius(const char *res_name, const char *atom_name, int *cls);
/* ref: freesasa_guess_radius, src/classifier.c:1002-1017 */
double freesasa_ingest_guess_radius(const char *symbol);
/* ref: freesasa_atom_is_backbone, src/classifier.c:1090-1109 */
int freesasa_ingest_is_backbone(const char *atom_name);
/* Reference areas of residue res_name for relative SASA (total, main chain, side chain, polar,
 * apolar; ref: freesasa_classifier_residue_reference with the ProtOr classifier).  Returns the row
 * of the table, -1 if unknown.  freesasa_ingest_residue_reference_table copies the whole table
 * ([5 * rows], may be NULL) and returns its number of rows: res_ref indexes it. */
int freesasa_ingest_residue_reference(const char *res_name, double ref[5]);
int freesasa_ingest_residue_reference_table(double *table);

#ifdef __cplusplus
}
#endif
#endif
