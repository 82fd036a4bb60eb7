/* ingest_cache.c — a packed batch on disk ("... and a binary cache").
 *
 * Reading and classifying 2e5 mmCIF files costs orders of magnitude more than their SASA on the GPU, and a sweep is
 * rarely run once (other probe radii, other resolutions, the other algorithm).  freesasa_ingest_save() writes a
 * freesasa_ingest_batch as it is in memory; freesasa_ingest_load() gives it back array for array, so that a second
 * sweep starts from one sequential read.  No counterpart in the reference (its CLI parses one file per run,
 * src/main.cc:763-779); what the batch holds is what the reference's structures hold (include/freesasa_ingest.h).
 *
 * File = header (128 bytes) + the arrays of the batch in the order of section_bytes() below, each padded to 16
 * bytes, + a table with one checksum per 1 MiB piece of every array (version 2: pieces are read and verified by
 * several threads at once, and a sweep reads - and verifies - only the coordinates, radii and classes of the
 * structures it is about to compute, straight into page-locked memory).  Little endian, IEEE doubles; the header
 * carries a byte-order mark, the array lengths and a checksum of the table, and a file is only accepted when all of
 * it adds up (truncated copies and edited files are refused, not half-loaded).  Written to "<path>.tmp<pid>" and renamed, so a reader never sees a partial file under the
 * final name.
 */
#include "freesasa_ingest.h"

int ingest_batch_alloc__(freesasa_ingest_batch *b, int32_t ns, int64_t na, int64_t nr); /* ingest.c: a batch's arrays in one pooled block; freesasa_ingest_free gives it back */

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include "hostfault.h"

#define CACHE_MAGIC "FSASABAT"
#define CACHE_VERSION 2u
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

/* Checksums.  Version 2 cuts every section into PIECES of CACHE_PIECE bytes (the last one shorter) and keeps one
 * checksum per piece in a table behind the payload; the header's checksum covers that table.  Pieces are independent:
 * several threads read and verify them at once (freesasa_ingest_load_mt), and a reader that wants a run of atoms only
 * (freesasa_ingest_cache_read_atoms: a sweep needs coordinates, radii and classes, not names) verifies exactly the
 * pieces it touches.  Version 1 chained ONE checksum through the whole payload: one thread, 2.2e7 atoms/s.
 * A piece: four interleaved multiply-xorshift lanes over 32 bytes a step (the lanes' dependent multiplications
 * overlap: ~3 x the throughput of one chain), the tail zero-padded, the lanes folded with the length at the end. */
#define CACHE_PIECE ((uint64_t)1 << 20)
#define MIX_K 0x9E3779B97F4A7C15ull
static inline uint64_t mix_step(uint64_t h, uint64_t w)
{
    h = (h ^ w) * MIX_K;
    return h ^ (h >> 29);
}
static uint64_t mix_piece(const void *p, uint64_t n)
{
    const unsigned char *q = (const unsigned char *)p;
    uint64_t h[4] = {0x243F6A8885A308D3ull, 0x13198A2E03707344ull, 0xA4093822299F31D0ull, 0x082EFA98EC4E6C89ull};
    uint64_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        memcpy(w, q + i, 32);
        h[0] = mix_step(h[0], w[0]); h[1] = mix_step(h[1], w[1]); h[2] = mix_step(h[2], w[2]); h[3] = mix_step(h[3], w[3]);
    }
    if (i < n) {
        uint64_t w[4] = {0, 0, 0, 0};
        memcpy(w, q + i, (size_t)(n - i));
        h[0] = mix_step(h[0], w[0]); h[1] = mix_step(h[1], w[1]); h[2] = mix_step(h[2], w[2]); h[3] = mix_step(h[3], w[3]);
    }
    uint64_t r = mix_step(mix_step(mix_step(h[0], h[1]), h[2]), h[3]);
    return (r ^ n) * 0xC2B2AE3D27D4EB4Full;
}
/* the header's checksum: the counts and every piece checksum, chained */
static uint64_t mix_table(int32_t S, int64_t N, int64_t R, const uint64_t *table, uint64_t n_pieces)
{
    uint64_t h = mix_step(mix_step(mix_step(0x452821E638D01377ull, (uint64_t)(uint32_t)S), (uint64_t)N), (uint64_t)R);
    for (uint64_t k = 0; k < n_pieces; ++k) h = mix_step(h, table[k]);
    return (h ^ n_pieces) * 0xC2B2AE3D27D4EB4Full;
}
static uint64_t pieces_of(uint64_t len) { return (len + CACHE_PIECE - 1) / CACHE_PIECE; }

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
static int pread_all(int fd, void *p, uint64_t n, uint64_t off)
{
    char *q = (char *)p;
    while (n > 0) {
        const ssize_t r = pread(fd, q, n > ((uint64_t)1 << 30) ? (size_t)1 << 30 : (size_t)n, (off_t)off);
        if (r < 0) { if (errno == EINTR) continue; return 0; }
        if (r == 0) return 0; /* short file */
        q += r; n -= (uint64_t)r; off += (uint64_t)r;
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
    uint64_t n_pieces = 0;
    for (int k = 0; k < CACHE_SECTIONS; ++k) {
        h.payload_bytes += pad16(len[k]);
        n_pieces += pieces_of(len[k]);
        if (k < 8) h.section_bytes[k] = len[k];
    }
    uint64_t *table = (uint64_t *)hf_calloc((size_t)(n_pieces + 2), 8); /* (+ the padding to 16 bytes) */
    if (!table) return FREESASA_INGEST_ENOMEM;
    for (uint64_t k = 0, t = 0; k < CACHE_SECTIONS; ++k)
        for (uint64_t o = 0; o < len[k]; o += CACHE_PIECE)
            table[t++] = mix_piece((const char *)ptr[k] + o, len[k] - o < CACHE_PIECE ? len[k] - o : CACHE_PIECE);
    h.checksum = mix_table(h.n_structs, h.n_atoms, h.n_residues, table, n_pieces);

    const size_t pl = strlen(path);
    char *tmp = (char *)hf_malloc(pl + 32);
    if (!tmp) { free(table); return FREESASA_INGEST_ENOMEM; }
    snprintf(tmp, pl + 32, "%s.tmp%ld", path, (long)getpid());
    const int fd = open(tmp, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { free(tmp); free(table); return FREESASA_INGEST_EIO; }
    unsigned char head[CACHE_HEADER_BYTES];
    memset(head, 0, sizeof head);
    memcpy(head, &h, sizeof h);
    static const unsigned char zeros[16] = {0};
    int ok = write_all(fd, head, sizeof head);
    for (int k = 0; ok && k < CACHE_SECTIONS; ++k) {
        ok = write_all(fd, ptr[k], len[k]);
        if (ok) ok = write_all(fd, zeros, pad16(len[k]) - len[k]);
    }
    if (ok) ok = write_all(fd, table, pad16(8 * n_pieces));
    if (ok && fsync(fd) != 0) ok = 0;
    if (close(fd) != 0) ok = 0;
    if (ok && rename(tmp, path) != 0) ok = 0;
    if (!ok) (void)unlink(tmp);
    free(tmp);
    free(table);
    return ok ? FREESASA_INGEST_OK : FREESASA_INGEST_EIO;
}

/* ------------------------------------------------------------------ reading */

/* An open cache file: the header, the piece table (verified against the header) and where every section starts.
   Shared by the whole-batch loader and the partial reader. */
struct freesasa_ingest_cache {
    int fd;
    cache_header h;
    uint64_t len[CACHE_SECTIONS], start[CACHE_SECTIONS], first_piece[CACHE_SECTIONS];
    uint64_t n_pieces, *table;
    int64_t *offsets; /* [n_structs + 1], verified (the partial reader's index) */
    int32_t *status;  /* [n_structs] */
};

static int read_section_verified(const freesasa_ingest_cache *c, int k, uint64_t p0, uint64_t p1, void *dst);

static int cache_open(const char *path, freesasa_ingest_cache **out)
{
    *out = NULL;
    if (!path) return FREESASA_INGEST_EIO;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return FREESASA_INGEST_EIO;
    freesasa_ingest_cache *c = (freesasa_ingest_cache *)hf_calloc(1, sizeof *c);
    if (!c) { (void)close(fd); return FREESASA_INGEST_ENOMEM; }
    c->fd = fd;
    int rc = FREESASA_INGEST_EFORMAT;
    do {
        struct stat st;
        if (fstat(fd, &st) != 0) { rc = FREESASA_INGEST_EIO; break; }
        unsigned char head[CACHE_HEADER_BYTES];
        if (st.st_size < (off_t)sizeof head || !pread_all(fd, head, sizeof head, 0)) break;
        memcpy(&c->h, head, sizeof c->h);
        const cache_header *h = &c->h;
        int tail_zero = 1;
        for (size_t q = sizeof *h; q < sizeof head; ++q) if (head[q]) tail_zero = 0;
        if (!tail_zero || h->reserved0 != 0) break;
        if (memcmp(h->magic, CACHE_MAGIC, 8) != 0 || h->bom != CACHE_BOM) break;
        /* a cache file of an EARLIER format version (version 1: one chained checksum, rounds 2-4) is told apart from a
           damaged one: its batch has to be saved again - from the parsed files, or by a build of that round (round-5 advisor) */
        if (h->version != CACHE_VERSION) { if (h->version >= 1 && h->version < CACHE_VERSION) rc = FREESASA_INGEST_EVERSION; break; }
        if (h->n_structs < 0 || h->n_atoms < 0 || h->n_residues < 0 || h->n_atoms > ((int64_t)1 << 40) || h->n_residues > h->n_atoms) break;
        uint64_t payload = 0;
        section_bytes(h->n_structs, h->n_atoms, h->n_residues, c->len);
        int same = 1;
        for (int k = 0; k < CACHE_SECTIONS; ++k) {
            c->start[k] = CACHE_HEADER_BYTES + payload;
            c->first_piece[k] = c->n_pieces;
            payload += pad16(c->len[k]);
            c->n_pieces += pieces_of(c->len[k]);
            if (k < 8 && h->section_bytes[k] != c->len[k]) same = 0;
        }
        if (!same || payload != h->payload_bytes || (uint64_t)st.st_size != CACHE_HEADER_BYTES + payload + pad16(8 * c->n_pieces)) break;
        c->table = (uint64_t *)hf_malloc((size_t)pad16(8 * c->n_pieces) + 16);
        if (!c->table) { rc = FREESASA_INGEST_ENOMEM; break; }
        if (!pread_all(fd, c->table, pad16(8 * c->n_pieces), CACHE_HEADER_BYTES + payload)) break;
        if ((c->n_pieces & 1) && c->table[c->n_pieces] != 0) break; /* (the table's padding is written as zeros) */
        if (mix_table(h->n_structs, h->n_atoms, h->n_residues, c->table, c->n_pieces) != h->checksum) break;
        *out = c;
        return FREESASA_INGEST_OK;
    } while (0);
    free(c->table);
    (void)close(fd);
    free(c);
    return rc;
}

/* pieces [p0, p1) of section k into dst (the section's byte p0 * CACHE_PIECE lands on dst[0]), each checked */
static int read_section_verified(const freesasa_ingest_cache *c, int k, uint64_t p0, uint64_t p1, void *dst)
{
    for (uint64_t p = p0; p < p1; ++p) {
        const uint64_t o = p * CACHE_PIECE, n = c->len[k] - o < CACHE_PIECE ? c->len[k] - o : CACHE_PIECE;
        char *d = (char *)dst + (o - p0 * CACHE_PIECE);
        if (!pread_all(c->fd, d, n, c->start[k] + o)) return 0;
        if (mix_piece(d, n) != c->table[c->first_piece[k] + p]) return 0;
    }
    return 1;
}

typedef struct load_job {
    const freesasa_ingest_cache *c;
    void **arr;
    uint64_t next; /* next piece (over all sections) */
    int failed;
    pthread_mutex_t mu;
} load_job;
static void *load_worker(void *arg)
{
    load_job *j = (load_job *)arg;
    const freesasa_ingest_cache *c = j->c;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        const uint64_t t = j->failed ? c->n_pieces : j->next++;
        pthread_mutex_unlock(&j->mu);
        if (t >= c->n_pieces) break;
        int k = CACHE_SECTIONS - 1;
        while (k > 0 && c->first_piece[k] > t) --k;
        while (pieces_of(c->len[k]) == 0 && k > 0) --k; /* (never: an empty section owns no piece) */
        const uint64_t p = t - c->first_piece[k];
        if (!read_section_verified(c, k, p, p + 1, (char *)j->arr[k] + p * CACHE_PIECE)) {
            pthread_mutex_lock(&j->mu);
            j->failed = 1;
            pthread_mutex_unlock(&j->mu);
        }
    }
    return NULL;
}

int freesasa_ingest_load_mt(const char *path, int n_threads, freesasa_ingest_batch *out)
{
    if (!out) return FREESASA_INGEST_EFORMAT;
    memset(out, 0, sizeof *out);
    freesasa_ingest_cache *c = NULL;
    int rc = cache_open(path, &c);
    if (rc) return rc;
    void *arr[CACHE_SECTIONS] = {0};
    freesasa_ingest_batch b;
    /* the arrays of the batch, in one pooled block like the parser's (ingest.c); section k of the file is array k */
    if (ingest_batch_alloc__(&b, c->h.n_structs, c->h.n_atoms, c->h.n_residues)) { freesasa_ingest_cache_close(c); return FREESASA_INGEST_ENOMEM; }
    arr[0] = b.offsets; arr[1] = b.res_offsets; arr[2] = b.status; arr[3] = b.xyz; arr[4] = b.radii; arr[5] = b.atom_class; arr[6] = b.atom_backbone;
    arr[7] = b.atom_name; arr[8] = b.atom_symbol; arr[9] = b.res_first; arr[10] = b.res_ref; arr[11] = b.res_name; arr[12] = b.res_number; arr[13] = b.res_chain;
    rc = FREESASA_INGEST_EFORMAT;
    do {
        int ok = 1;
        for (int k = 0; k < CACHE_SECTIONS; ++k) {
            unsigned char padding[16] = {0};
            const uint64_t np = pad16(c->len[k]) - c->len[k];
            if (np && !pread_all(c->fd, padding, np, c->start[k] + c->len[k])) { ok = 0; break; }
            for (int q = 0; q < 16; ++q) if (padding[q]) ok = 0; /* (the padding is written as zeros) */
            if (!ok) break;
        }
        if (!ok) break;
        if (n_threads <= 0) { n_threads = freesasa_ingest_usable_cpus(); if (n_threads > 8) n_threads = 8; }
        if ((uint64_t)n_threads > c->n_pieces) n_threads = (int)(c->n_pieces > 0 ? c->n_pieces : 1);
        if (n_threads > 64) n_threads = 64;
        load_job j;
        memset(&j, 0, sizeof j);
        j.c = c; j.arr = arr;
        pthread_mutex_init(&j.mu, NULL);
        pthread_t th[64];
        int started = 0;
        for (; started < n_threads - 1; ++started)
            if (hf_thread_create(&th[started], NULL, load_worker, &j)) break;
        load_worker(&j); /* the calling thread works too */
        for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
        pthread_mutex_destroy(&j.mu);
        if (j.failed) break;
        if (!batch_indices_ok(&b)) break;
        *out = b;
        memset(&b, 0, sizeof b);
        rc = FREESASA_INGEST_OK;
    } while (0);
    freesasa_ingest_free(&b); /* (nothing, when the batch went to the caller) */
    freesasa_ingest_cache_close(c);
    return rc;
}

int freesasa_ingest_load(const char *path, freesasa_ingest_batch *out) { return freesasa_ingest_load_mt(path, 0, out); }

/* ------------------------------------------------------------------ partial reader (sweeps from a cache) */

int freesasa_ingest_cache_open(const char *path, freesasa_ingest_cache **out)
{
    if (!out) return FREESASA_INGEST_EFORMAT;
    freesasa_ingest_cache *c = NULL;
    int rc = cache_open(path, &c);
    if (rc) return rc;
    rc = FREESASA_INGEST_EFORMAT;
    do {
        const int32_t S = c->h.n_structs;
        c->offsets = (int64_t *)hf_malloc(c->len[0] > 0 ? (size_t)c->len[0] : 8);
        c->status = (int32_t *)hf_malloc(c->len[2] > 0 ? (size_t)c->len[2] : 4);
        if (!c->offsets || !c->status) { rc = FREESASA_INGEST_ENOMEM; break; }
        if (!read_section_verified(c, 0, 0, pieces_of(c->len[0]), c->offsets) || !read_section_verified(c, 2, 0, pieces_of(c->len[2]), c->status)) break;
        int ok = c->offsets[0] == 0 && c->offsets[S] == c->h.n_atoms;
        for (int32_t s = 0; ok && s < S; ++s)
            if (c->offsets[s + 1] < c->offsets[s] || c->offsets[s + 1] > c->h.n_atoms || c->status[s] < FREESASA_INGEST_OK || c->status[s] > FREESASA_INGEST_ENOMEM) ok = 0;
        if (!ok) break;
        *out = c;
        return FREESASA_INGEST_OK;
    } while (0);
    freesasa_ingest_cache_close(c);
    *out = NULL;
    return rc;
}
void freesasa_ingest_cache_close(freesasa_ingest_cache *c)
{
    if (!c) return;
    free(c->table); free(c->offsets); free(c->status);
    if (c->fd >= 0) (void)close(c->fd);
    free(c);
}
int32_t freesasa_ingest_cache_n_structs(const freesasa_ingest_cache *c) { return c->h.n_structs; }
int64_t freesasa_ingest_cache_n_atoms(const freesasa_ingest_cache *c) { return c->h.n_atoms; }
const int64_t *freesasa_ingest_cache_offsets(const freesasa_ingest_cache *c) { return c->offsets; }
const int32_t *freesasa_ingest_cache_status(const freesasa_ingest_cache *c) { return c->status; }

/* bytes [b0, b1) of section k into dst: whole pieces that lie inside the range are read and verified in place, the
   one or two pieces that straddle its ends go through `scratch` (CACHE_PIECE bytes) */
static int read_range_verified(const freesasa_ingest_cache *c, int k, uint64_t b0, uint64_t b1, void *dst, void *scratch)
{
    if (b1 > c->len[k] || b0 > b1) return 0;
    char *d = (char *)dst;
    uint64_t pos = b0;
    while (pos < b1) {
        const uint64_t p = pos / CACHE_PIECE, po = p * CACHE_PIECE;
        const uint64_t pn = c->len[k] - po < CACHE_PIECE ? c->len[k] - po : CACHE_PIECE;
        if (pos == po && po + pn <= b1) { /* a whole piece */
            if (!read_section_verified(c, k, p, p + 1, d + (pos - b0))) return 0;
            pos = po + pn;
        } else {
            if (!pread_all(c->fd, scratch, pn, c->start[k] + po) || mix_piece(scratch, pn) != c->table[c->first_piece[k] + p]) return 0;
            const uint64_t e = po + pn < b1 ? po + pn : b1;
            memcpy(d + (pos - b0), (const char *)scratch + (pos - po), (size_t)(e - pos));
            pos = e;
        }
    }
    return 1;
}

int freesasa_ingest_cache_read_atoms(const freesasa_ingest_cache *c, int64_t a0, int64_t a1, double *xyz, double *radii, uint8_t *atom_class)
{
    if (!c || a0 < 0 || a1 < a0 || a1 > c->h.n_atoms) return FREESASA_INGEST_EFORMAT;
    if (a1 == a0) return FREESASA_INGEST_OK;
    void *scratch = hf_malloc((size_t)CACHE_PIECE);
    if (!scratch) return FREESASA_INGEST_ENOMEM;
    int ok = 1;
    if (xyz) ok = read_range_verified(c, 3, 24 * (uint64_t)a0, 24 * (uint64_t)a1, xyz, scratch);
    if (ok && radii) ok = read_range_verified(c, 4, 8 * (uint64_t)a0, 8 * (uint64_t)a1, radii, scratch);
    if (ok && atom_class) {
        ok = read_range_verified(c, 5, (uint64_t)a0, (uint64_t)a1, atom_class, scratch);
        for (int64_t i = 0; ok && i < a1 - a0; ++i)
            if (atom_class[i] > FREESASA_INGEST_UNKNOWN) ok = 0; /* (indexes the class sums on the device) */
    }
    free(scratch);
    return ok ? FREESASA_INGEST_OK : FREESASA_INGEST_EFORMAT;
}
