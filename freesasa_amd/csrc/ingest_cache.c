/* ingest_cache.c — a packed batch on disk ("... and a binary cache").
 *
 * Reading and classifying 2e5 mmCIF files costs orders of magnitude more than their SASA on the GPU, and a sweep is
 * rarely run once (other probe radii, other resolutions, the other algorithm).  freesasa_ingest_save() writes a
 * freesasa_ingest_batch as it is in memory; freesasa_ingest_load() gives it back array for array, so that a second
 * sweep starts from one sequential read.  No counterpart in the reference (its CLI parses one file per run,
 * src/main.cc:763-779); what the batch holds is what the reference's structures hold (include/freesasa_ingest.h).
 *
 * File = header (128 bytes) + the arrays of the batch in the order of section_bytes() below, each padded to 16
 * bytes.  Little endian, IEEE doubles; the header carries a byte-order mark, the array lengths and a checksum of
 * the payload, and a file is only accepted when all of it adds up (truncated copies and edited files are refused,
 * not half-loaded).  Written to "<path>.tmp<pid>" and renamed, so a reader never sees a partial file under the
 * final name.
 */
#include "freesasa_ingest.h"

#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#define CACHE_MAGIC "FSASABAT"
#define CACHE_VERSION 1u
#define CACHE_BOM 0x01020304u
#define CACHE_SECTIONS 14

typedef struct cache_header {
    char magic[8];
    uint32_t version, bom;
    int32_t n_structs, reserved0;
    int64_t n_atoms, n_residues;
    uint64_t payload_bytes, checksum;
    uint64_t section_bytes[CACHE_SECTIONS > 8 ? 8 : CACHE_SECTIONS]; /* the first eight section lengths, a redundancy check */
} cache_header; /* 8 + 8 + 8 + 16 + 16 + 64 = 120 bytes, padded to 128 on disk */
#define CACHE_HEADER_BYTES 128

static uint64_t pad16(uint64_t v) { return (v + 15u) & ~(uint64_t)15u; }

/* byte length of every array, in file order */
static void section_bytes(int32_t S, int64_t N, int64_t R, uint64_t len[CACHE_SECTIONS])
{
    const uint64_t s = (uint64_t)S, n = (uint64_t)N, r = (uint64_t)R;
    len[0] = 8 * (s + 1);  /* offsets */
    len[1] = 8 * (s + 1);  /* res_offsets */
    len[2] = 4 * s;        /* status */
    len[3] = 24 * n;       /* xyz */
    len[4] = 8 * n;        /* radii */
    len[5] = n;            /* atom_class */
    len[6] = n;            /* atom_backbone */
    len[7] = 4 * n;        /* atom_name */
    len[8] = 2 * n;        /* atom_symbol */
    len[9] = 8 * (r + 1);  /* res_first */
    len[10] = 2 * r;       /* res_ref */
    len[11] = 4 * r;       /* res_name */
    len[12] = 6 * r;       /* res_number */
    len[13] = 4 * r;       /* res_chain */
}

static void section_ptrs(const freesasa_ingest_batch *b, const void *ptr[CACHE_SECTIONS])
{
    ptr[0] = b->offsets; ptr[1] = b->res_offsets; ptr[2] = b->status; ptr[3] = b->xyz; ptr[4] = b->radii;
    ptr[5] = b->atom_class; ptr[6] = b->atom_backbone; ptr[7] = b->atom_name; ptr[8] = b->atom_symbol;
    ptr[9] = b->res_first; ptr[10] = b->res_ref; ptr[11] = b->res_name; ptr[12] = b->res_number; ptr[13] = b->res_chain;
}

/* checksum of a byte run, 8 bytes at a time (multiply-xorshift; the tail is zero-padded); chained through h */
static uint64_t mix_bytes(uint64_t h, const void *p, uint64_t n)
{
    const unsigned char *q = (const unsigned char *)p;
    uint64_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, q + i, 8);
        h = (h ^ w) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
    }
    if (i < n) {
        uint64_t w = 0;
        memcpy(&w, q + i, (size_t)(n - i));
        h = (h ^ w) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
    }
    return (h ^ n) * 0xC2B2AE3D27D4EB4Full;
}

static int write_all(int fd, const void *p, uint64_t n)
{
    const char *q = (const char *)p;
    while (n > 0) {
        const ssize_t w = write(fd, q, n > ((uint64_t)1 << 30) ? (size_t)1 << 30 : (size_t)n);
        if (w < 0) { if (errno == EINTR) continue; return 0; }
        if (w == 0) return 0;
        q += w; n -= (uint64_t)w;
    }
    return 1;
}
static int read_all(int fd, void *p, uint64_t n)
{
    char *q = (char *)p;
    while (n > 0) {
        const ssize_t r = read(fd, q, n > ((uint64_t)1 << 30) ? (size_t)1 << 30 : (size_t)n);
        if (r < 0) { if (errno == EINTR) continue; return 0; }
        if (r == 0) return 0; /* short file */
        q += r; n -= (uint64_t)r;
    }
    return 1;
}

static int batch_shape_ok(const freesasa_ingest_batch *b)
{
    if (b->n_structs < 0 || b->n_atoms < 0 || b->n_residues < 0) return 0;
    if (!b->offsets || !b->res_offsets || !b->res_first) return 0;
    if (b->n_structs > 0 && !b->status) return 0;
    if (b->n_atoms > 0 && (!b->xyz || !b->radii || !b->atom_class || !b->atom_backbone || !b->atom_name || !b->atom_symbol)) return 0;
    if (b->n_residues > 0 && (!b->res_ref || !b->res_name || !b->res_number || !b->res_chain)) return 0;
    return 1;
}

/* the index arrays of a batch say what its headers say: CSR offsets from 0 to the counts, never decreasing, the
   residues of a structure inside that structure's atoms */
static int batch_indices_ok(const freesasa_ingest_batch *b)
{
    const int32_t S = b->n_structs;
    if (b->offsets[0] != 0 || b->offsets[S] != b->n_atoms) return 0;
    if (b->res_offsets[0] != 0 || b->res_offsets[S] != b->n_residues) return 0;
    if (b->res_first[0] != 0 || b->res_first[b->n_residues] != b->n_atoms) return 0;
    /* first pass: every offset inside its array BEFORE anything is indexed with it (never decreasing and ending at
       the count is not enough: {0, 1 << 20, 2} passes both) */
    for (int32_t s = 0; s < S; ++s) {
        if (b->offsets[s + 1] < b->offsets[s] || b->offsets[s + 1] > b->n_atoms) return 0;
        if (b->res_offsets[s + 1] < b->res_offsets[s] || b->res_offsets[s + 1] > b->n_residues) return 0;
        if (b->status[s] < FREESASA_INGEST_OK || b->status[s] > FREESASA_INGEST_ENOMEM) return 0;
    }
    const int ref_rows = freesasa_ingest_residue_reference_table(NULL);
    for (int64_t r = 0; r < b->n_residues; ++r)
        if (b->res_ref[r] < -1 || b->res_ref[r] >= ref_rows) return 0; /* (indexes the reference table on the device) */
    for (int64_t i = 0; i < b->n_atoms; ++i)
        if (b->atom_class[i] > FREESASA_INGEST_UNKNOWN || b->atom_backbone[i] > 1) return 0;
    for (int32_t s = 0; s < S; ++s) {
        if (b->res_offsets[s + 1] > b->res_offsets[s]) { /* its residues start at its first atom and end at its last */
            if (b->res_first[b->res_offsets[s]] != b->offsets[s]) return 0;
            if (b->res_first[b->res_offsets[s + 1]] != b->offsets[s + 1]) return 0;
        } else if (b->offsets[s + 1] != b->offsets[s]) {
            return 0; /* atoms without a residue */
        }
    }
    for (int64_t r = 0; r < b->n_residues; ++r)
        if (b->res_first[r + 1] <= b->res_first[r]) return 0; /* a residue owns at least one atom */
    return 1;
}

int freesasa_ingest_save(const freesasa_ingest_batch *b, const char *path)
{
    if (!b || !path || !batch_shape_ok(b) || !batch_indices_ok(b)) return FREESASA_INGEST_EFORMAT;
    uint64_t len[CACHE_SECTIONS];
    const void *ptr[CACHE_SECTIONS];
    section_bytes(b->n_structs, b->n_atoms, b->n_residues, len);
    section_ptrs(b, ptr);
    cache_header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, CACHE_MAGIC, 8);
    h.version = CACHE_VERSION; h.bom = CACHE_BOM;
    h.n_structs = b->n_structs; h.n_atoms = b->n_atoms; h.n_residues = b->n_residues;
    uint64_t sum = 0x243F6A8885A308D3ull;
    for (int k = 0; k < CACHE_SECTIONS; ++k) {
        h.payload_bytes += pad16(len[k]);
        sum = mix_bytes(sum, ptr[k], len[k]);
        if (k < 8) h.section_bytes[k] = len[k];
    }
    h.checksum = sum;

    const size_t pl = strlen(path);
    char *tmp = (char *)malloc(pl + 32);
    if (!tmp) return FREESASA_INGEST_ENOMEM;
    snprintf(tmp, pl + 32, "%s.tmp%ld", path, (long)getpid());
    const int fd = open(tmp, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { free(tmp); return FREESASA_INGEST_EIO; }
    unsigned char head[CACHE_HEADER_BYTES];
    memset(head, 0, sizeof head);
    memcpy(head, &h, sizeof h);
    static const unsigned char zeros[16] = {0};
    int ok = write_all(fd, head, sizeof head);
    for (int k = 0; ok && k < CACHE_SECTIONS; ++k) {
        ok = write_all(fd, ptr[k], len[k]);
        if (ok) ok = write_all(fd, zeros, pad16(len[k]) - len[k]);
    }
    if (ok && fsync(fd) != 0) ok = 0;
    if (close(fd) != 0) ok = 0;
    if (ok && rename(tmp, path) != 0) ok = 0;
    if (!ok) (void)unlink(tmp);
    free(tmp);
    return ok ? FREESASA_INGEST_OK : FREESASA_INGEST_EIO;
}

int freesasa_ingest_load(const char *path, freesasa_ingest_batch *out)
{
    if (!out) return FREESASA_INGEST_EFORMAT;
    memset(out, 0, sizeof *out);
    if (!path) return FREESASA_INGEST_EIO;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return FREESASA_INGEST_EIO;
    int rc = FREESASA_INGEST_EFORMAT;
    void *arr[CACHE_SECTIONS] = {0};
    do {
        struct stat st;
        if (fstat(fd, &st) != 0) { rc = FREESASA_INGEST_EIO; break; }
        unsigned char head[CACHE_HEADER_BYTES];
        if (st.st_size < (off_t)sizeof head || !read_all(fd, head, sizeof head)) break;
        cache_header h;
        memcpy(&h, head, sizeof h);
        int tail_zero = 1;
        for (size_t q = sizeof h; q < sizeof head; ++q) if (head[q]) tail_zero = 0;
        if (!tail_zero || h.reserved0 != 0) break;
        if (memcmp(h.magic, CACHE_MAGIC, 8) != 0 || h.version != CACHE_VERSION || h.bom != CACHE_BOM) break;
        if (h.n_structs < 0 || h.n_atoms < 0 || h.n_residues < 0 || h.n_atoms > ((int64_t)1 << 40) || h.n_residues > h.n_atoms) break;
        uint64_t len[CACHE_SECTIONS], payload = 0;
        section_bytes(h.n_structs, h.n_atoms, h.n_residues, len);
        int same = 1;
        for (int k = 0; k < CACHE_SECTIONS; ++k) {
            payload += pad16(len[k]);
            if (k < 8 && h.section_bytes[k] != len[k]) same = 0;
        }
        if (!same || payload != h.payload_bytes || (uint64_t)st.st_size != CACHE_HEADER_BYTES + payload) break;
        uint64_t sum = 0x243F6A8885A308D3ull;
        int ok = 1;
        for (int k = 0; ok && k < CACHE_SECTIONS; ++k) {
            arr[k] = malloc(len[k] > 0 ? (size_t)len[k] : 1);
            if (!arr[k]) { rc = FREESASA_INGEST_ENOMEM; ok = 0; break; }
            unsigned char padding[16] = {0};
            if (!read_all(fd, arr[k], len[k]) || !read_all(fd, padding, pad16(len[k]) - len[k])) { ok = 0; break; }
            for (int q = 0; q < 16; ++q) if (padding[q]) ok = 0; /* (the padding is written as zeros) */
            if (!ok) break;
            sum = mix_bytes(sum, arr[k], len[k]);
        }
        if (!ok) break;
        if (sum != h.checksum) break;
        freesasa_ingest_batch b;
        memset(&b, 0, sizeof b);
        b.n_structs = h.n_structs; b.n_atoms = h.n_atoms; b.n_residues = h.n_residues;
        b.offsets = (int64_t *)arr[0]; b.res_offsets = (int64_t *)arr[1]; b.status = (int32_t *)arr[2];
        b.xyz = (double *)arr[3]; b.radii = (double *)arr[4]; b.atom_class = (uint8_t *)arr[5]; b.atom_backbone = (uint8_t *)arr[6];
        b.atom_name = (char *)arr[7]; b.atom_symbol = (char *)arr[8]; b.res_first = (int64_t *)arr[9]; b.res_ref = (int16_t *)arr[10];
        b.res_name = (char *)arr[11]; b.res_number = (char *)arr[12]; b.res_chain = (char *)arr[13];
        if (!batch_indices_ok(&b)) break;
        *out = b;
        memset(arr, 0, sizeof arr);
        rc = FREESASA_INGEST_OK;
    } while (0);
    for (int k = 0; k < CACHE_SECTIONS; ++k) free(arr[k]);
    (void)close(fd);
    return rc;
}
