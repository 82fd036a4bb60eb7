/* ingest.c — multi-threaded PDB -> packed (xyz, radius, class, residue segments) batches.
 * Contract and scope: include/freesasa_ingest.h.  Host code only (gcc); no GPU involved.
 *
 * The rules that decide which atoms a file contributes, their order, coordinates, radii and
 * residue boundaries restate the reference's reader so that a sweep through this loader computes
 * on exactly the arrays `freesasa_structure_from_pdb` + `freesasa_structure_radius` would give:
 *   record / hydrogen / alt-loc / model rules   ref: src/structure.c:644-722, src/pdb.c:259-281
 *   fixed-column fields and length checks       ref: src/pdb.c:13-24, 148-237
 *   element guess from the atom name            ref: src/structure.c:420-446
 *   radius: classifier, else element, else 0    ref: src/structure.c:519-550, src/classifier.c:738-796, 1002-1017
 *   residue boundaries                          ref: src/structure.c:473-512
 * Lines are consumed the way fgets(line, 120) does (src/structure.c:654, src/pdb.h:22): physical
 * lines longer than 119 characters continue as a new "line".
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE /* sched_getaffinity, CPU_COUNT */
#endif
#include "freesasa_ingest.h"

#include <ctype.h>
#include <sched.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "protor_table.h"
#include "hostfault.h"

#define CHUNK 119 /* fgets(line, PDB_MAX_LINE_STRL = 120) */

int ingest_batch_alloc__(freesasa_ingest_batch *b, int32_t ns, int64_t na, int64_t nr); /* (below: batches live in pooled blocks) */

/* ------------------------------------------------------------------ classifier */

/* isspace() of the C locale without the call (the parsers look at every character of every atom line) */
static inline int is_sp(char c) { return c == ' ' || (unsigned)((unsigned char)c - 9u) < 5u; }

/* first whitespace-delimited token of a field, like sscanf("%s") (ref: src/classifier.c:126-155) */
static int first_token(const char *s, const char **tok)
{
    while (*s && is_sp(*s)) ++s;
    *tok = s;
    int n = 0;
    while (s[n] && !is_sp(s[n])) ++n;
    return n;
}

static uint64_t name_key(const char *res, int rl, const char *atom, int al)
{
    if (rl < 1 || rl > 3 || al < 1 || al > 4) return 0;
    unsigned char b[7] = {' ', ' ', ' ', ' ', ' ', ' ', ' '};
    memcpy(b, res, (size_t)rl);
    memcpy(b + 3, atom, (size_t)al);
    uint64_t k = 0;
    for (int i = 0; i < 7; ++i) k = (k << 8) | b[i];
    return k;
}

/* The 506 (residue, atom) keys in an open-addressing table (built once, read-only afterwards): a
 * lookup is one or two probes instead of a nine-step binary search per atom. */
#define PROTOR_HASH_BITS 11
static int16_t protor_hash[1 << PROTOR_HASH_BITS];
static pthread_once_t protor_hash_once = PTHREAD_ONCE_INIT;
static unsigned protor_slot(uint64_t k) { return (unsigned)((k * 0x9E3779B97F4A7C15ULL) >> (64 - PROTOR_HASH_BITS)); }
static void protor_hash_build(void)
{
    for (int i = 0; i < (1 << PROTOR_HASH_BITS); ++i) protor_hash[i] = -1;
    for (int i = 0; i < PROTOR_N; ++i) {
        unsigned h = protor_slot(protor_table[i].key);
        while (protor_hash[h] >= 0) h = (h + 1) & ((1 << PROTOR_HASH_BITS) - 1);
        protor_hash[h] = (int16_t)i;
    }
}

/* the lookup on tokens that are already trimmed (rl, al: their lengths) */
static inline double protor_lookup(const char *rt, int rl, const char *at, int al, int *cls)
{
    *cls = FREESASA_INGEST_UNKNOWN;
    if (rl < 1 || rl > 3 || al < 1 || al > 4) return -1.0;
    unsigned char b[7] = {' ', ' ', ' ', ' ', ' ', ' ', ' '};
    for (int i = 0; i < rl; ++i) b[i] = (unsigned char)rt[i];
    for (int i = 0; i < al; ++i) b[3 + i] = (unsigned char)at[i];
    uint64_t k = 0;
    for (int i = 0; i < 7; ++i) k = (k << 8) | b[i];
    pthread_once(&protor_hash_once, protor_hash_build);
    for (unsigned h = protor_slot(k);; h = (h + 1) & ((1 << PROTOR_HASH_BITS) - 1)) {
        const int i = protor_hash[h];
        if (i < 0) return -1.0;
        if (protor_table[i].key == k) {
            *cls = protor_table[i].cls;
            return protor_table[i].radius;
        }
    }
}

double freesasa_ingest_protor_radius(const char *res_name, const char *atom_name, int *cls)
{
    const char *rt, *at;
    const int rl = first_token(res_name, &rt), al = first_token(atom_name, &at);
    const uint64_t k = name_key(rt, rl, at, al);
    if (cls) *cls = FREESASA_INGEST_UNKNOWN;
    if (!k) return -1.0;
    pthread_once(&protor_hash_once, protor_hash_build);
    for (unsigned h = protor_slot(k);; h = (h + 1) & ((1 << PROTOR_HASH_BITS) - 1)) {
        const int i = protor_hash[h];
        if (i < 0) return -1.0;
        if (protor_table[i].key == k) {
            if (cls) *cls = protor_table[i].cls;
            return protor_table[i].radius;
        }
    }
}

/* ref: freesasa_atom_is_backbone, src/classifier.c:1090-1109 (the name is trimmed first) */
static inline int backbone_tok(const char *t, int n)
{
    if (n < 1 || n > 3) return 0;
    const unsigned key = (unsigned)(unsigned char)t[0] | (n > 1 ? (unsigned)(unsigned char)t[1] << 8 : 0) |
                         (n > 2 ? (unsigned)(unsigned char)t[2] << 16 : 0);
    for (int i = 0; i < BACKBONE_N; ++i) {
        const char *b = backbone_names[i]; /* at most 3 characters */
        const unsigned kb = (unsigned)(unsigned char)b[0] | (b[1] ? (unsigned)(unsigned char)b[1] << 8 : 0) |
                            (b[1] && b[2] ? (unsigned)(unsigned char)b[2] << 16 : 0);
        if (kb == key) return 1;
    }
    return 0;
}

int freesasa_ingest_is_backbone(const char *atom_name)
{
    const char *t;
    const int n = first_token(atom_name, &t);
    if (n < 1 || n > 3) return 0;
    const unsigned key = (unsigned)(unsigned char)t[0] | (n > 1 ? (unsigned)(unsigned char)t[1] << 8 : 0) |
                         (n > 2 ? (unsigned)(unsigned char)t[2] << 16 : 0);
    for (int i = 0; i < BACKBONE_N; ++i) {
        const char *b = backbone_names[i]; /* at most 3 characters */
        const unsigned kb = (unsigned)(unsigned char)b[0] | (b[1] ? (unsigned)(unsigned char)b[1] << 8 : 0) |
                            (b[1] && b[2] ? (unsigned)(unsigned char)b[2] << 16 : 0);
        if (kb == key) return 1;
    }
    return 0;
}

static int residue_ref_index(const char *res_name)
{
    const char *t;
    const int n = first_token(res_name, &t);
    if (n < 1 || n > 3) return -1;
    for (int i = 0; i < RESIDUE_REF_N; ++i)
        if ((int)strlen(residue_ref_table[i].res) == n && memcmp(residue_ref_table[i].res, t, (size_t)n) == 0) return i;
    return -1;
}

int freesasa_ingest_residue_reference(const char *res_name, double ref[5])
{
    const int i = residue_ref_index(res_name);
    if (i < 0) return -1;
    ref[0] = residue_ref_table[i].total; ref[1] = residue_ref_table[i].main_chain; ref[2] = residue_ref_table[i].side_chain;
    ref[3] = residue_ref_table[i].polar; ref[4] = residue_ref_table[i].apolar;
    return i;
}

int freesasa_ingest_residue_reference_table(double *table)
{
    if (table)
        for (int i = 0; i < RESIDUE_REF_N; ++i) {
            table[5 * i] = residue_ref_table[i].total; table[5 * i + 1] = residue_ref_table[i].main_chain;
            table[5 * i + 2] = residue_ref_table[i].side_chain; table[5 * i + 3] = residue_ref_table[i].polar;
            table[5 * i + 4] = residue_ref_table[i].apolar;
        }
    return RESIDUE_REF_N;
}

double freesasa_ingest_guess_radius(const char *symbol)
{
    char s[3];
    snprintf(s, 3, "%2s", symbol); /* right-justified, at most 2 characters */
    for (int i = 0; i < ELEMENT_N; ++i)
        if (strcmp(s, element_table[i].sym) == 0) return element_table[i].radius;
    return -1.0;
}

/* first token of a field, NUL padded to w bytes (no terminator when the token fills the field) */
static void store_token(char *dst, int w, const char *field)
{
    const char *t;
    int n = first_token(field, &t);
    if (n > w) n = w;
    memset(dst, 0, (size_t)w);
    memcpy(dst, t, (size_t)n);
}

/* ------------------------------------------------------------------ one file */

typedef struct {
    int64_t n, cap;
    double *xyz, *rad;
    uint8_t *cls, *bb;
    char *aname, *asym; /* [4n] trimmed atom names, [2n] trimmed element symbols, NUL padded */
    int64_t nres, rescap;
    int64_t *res_first;
    int16_t *res_ref;
    char *res_name, *res_number, *res_chain;
    int status;
    int scratch_model; /* mmCIF: lowest model number found by the first pass */
    int64_t n0, nres0; /* where the input being parsed starts: a worker appends all its inputs to one arena */
} parsed;

static void parsed_free(parsed *p)
{
    free(p->xyz); free(p->rad); free(p->cls); free(p->bb); free(p->aname); free(p->asym);
    free(p->res_first); free(p->res_ref); free(p->res_name); free(p->res_number); free(p->res_chain);
    memset(p, 0, sizeof *p);
}

static int grow_atoms(parsed *p)
{
    if (p->n < p->cap) return 0;
    const int64_t cap = p->cap ? 2 * p->cap : 4096;
    double *x = hf_realloc(p->xyz, sizeof(double) * 3 * (size_t)cap);
    if (!x) return -1;
    p->xyz = x;
    double *r = hf_realloc(p->rad, sizeof(double) * (size_t)cap);
    if (!r) return -1;
    p->rad = r;
    uint8_t *c = hf_realloc(p->cls, (size_t)cap);
    if (!c) return -1;
    p->cls = c;
    uint8_t *b = hf_realloc(p->bb, (size_t)cap);
    if (!b) return -1;
    p->bb = b;
    char *an = hf_realloc(p->aname, 4 * (size_t)cap);
    if (!an) return -1;
    p->aname = an;
    char *as = hf_realloc(p->asym, 2 * (size_t)cap);
    if (!as) return -1;
    p->asym = as;
    p->cap = cap;
    return 0;
}

static int grow_res(parsed *p)
{
    if (p->nres < p->rescap) return 0;
    const int64_t cap = p->rescap ? 2 * p->rescap : 512;
    int64_t *f = hf_realloc(p->res_first, sizeof(int64_t) * (size_t)cap);
    if (!f) return -1;
    p->res_first = f;
    int16_t *rr = hf_realloc(p->res_ref, sizeof(int16_t) * (size_t)cap);
    if (!rr) return -1;
    p->res_ref = rr;
    char *a = hf_realloc(p->res_name, 4 * (size_t)cap);
    if (!a) return -1;
    p->res_name = a;
    char *b = hf_realloc(p->res_number, 6 * (size_t)cap);
    if (!b) return -1;
    p->res_number = b;
    char *c = hf_realloc(p->res_chain, 4 * (size_t)cap);
    if (!c) return -1;
    p->res_chain = c;
    p->rescap = cap;
    return 0;
}

/* ref: src/structure.c:420-446 */
static void guess_symbol(char *symbol, const char *name)
{
    if (name[0] == ' ' || (name[0] >= '1' && name[0] <= '9')) {
        symbol[0] = ' '; symbol[1] = name[1];
    } else if (name[3] == ' ') {
        symbol[0] = name[0]; symbol[1] = name[1];
    } else {
        symbol[0] = ' '; symbol[1] = name[0];
    }
    symbol[2] = '\0';
}

static const double pow10_tab[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};

/* One number of the coordinate section with sscanf("%lf") semantics.  Plain decimals with at
 * most 15 digits are converted exactly (integer / power of ten, both exact, one correctly rounded
 * division = what strtod returns); anything else goes to strtod. */
static int scan_double(const char **pp, double *out)
{
    const char *p = *pp;
    while (*p && is_sp(*p)) ++p;
    const char *q = p;
    int neg = 0;
    if (*q == '-' || *q == '+') { neg = *q == '-'; ++q; }
    uint64_t m = 0;
    int digits = 0, frac = 0;
    while (*q >= '0' && *q <= '9') { m = m * 10 + (uint64_t)(*q - '0'); ++digits; ++q; }
    if (*q == '.') {
        ++q;
        while (*q >= '0' && *q <= '9') { m = m * 10 + (uint64_t)(*q - '0'); ++digits; ++frac; ++q; }
    }
    const int plain = digits >= 1 && digits <= 15 &&
                      (*q == '\0' || *q == '-' || *q == '+' || is_sp(*q));
    if (plain) {
        const double v = (double)m / pow10_tab[frac];
        *out = neg ? -v : v;
        *pp = q;
        return 1;
    }
    char *end;
    const double v = strtod(p, &end);
    if (end == p) return 0;
    *out = v;
    *pp = end;
    return 1;
}

/* the first token of a field of w characters that is known to hold no NUL (a slice of an atom line): its start and length */
static inline int field_token(const char *f, int w, const char **tok)
{
    int i = 0;
    while (i < w && is_sp(f[i])) ++i;
    *tok = f + i;
    int n = 0;
    while (i + n < w && !is_sp(f[i + n])) ++n;
    return n;
}

/* The coordinate section as nearly every file writes it: three fields of the form "%8.3f" - blanks, an optional '-',
 * digits, '.', three digits - each beginning with a blank or its '-' (so that sscanf's whitespace-delimited numbers,
 * ref: src/pdb.c:176-197, ARE the three columns: digits that run into the next field would be one number for it).
 * 1: v[] holds what scan_double would have returned (an integer below 2^53 divided by 1000: the correctly rounded
 * value of the decimal); 0: not of this form, nothing decided. */
static inline int coords_8_3(const char *sec, double v[3])
{
    for (int f = 0; f < 3; ++f) {
        const char *c = sec + 8 * f;
        const unsigned d3 = (unsigned)(c[3] - '0'), d5 = (unsigned)(c[5] - '0'), d6 = (unsigned)(c[6] - '0'), d7 = (unsigned)(c[7] - '0');
        if (c[4] != '.' || d3 > 9 || d5 > 9 || d6 > 9 || d7 > 9 || !(c[0] == ' ' || c[0] == '-')) return 0;
        unsigned m = d3, mul = 10;
        int i = 2;
        for (; i >= 0 && (unsigned)(c[i] - '0') <= 9; --i) { m += (unsigned)(c[i] - '0') * mul; mul *= 10; }
        int neg = 0;
        if (i >= 0 && c[i] == '-') { neg = 1; --i; }
        for (; i >= 0; --i)
            if (c[i] != ' ') return 0;
        const double x = (double)(m * 1000u + d5 * 100u + d6 * 10u + d7) / 1000.0;
        v[f] = neg ? -x : x;
    }
    return 1;
}

/* Parses into p's buffers, which are reused from call to call (only the counters are reset).
 * Round 5: an atom line is read where it lies (no copy into a line buffer: an embedded NUL ends the line, as it does for
 * the reference's strlen on its fgets buffer, by a memchr), its name fields are trimmed once and serve the radius
 * lookup, the backbone test and the stored names alike, and no character test is a library call - 157 -> ~95 ns per
 * atom on 1a0q, one thread. */
static void parse_pdb(const char *text, size_t len, int options, parsed *p)
{
    p->n = p->n0; p->nres = p->nres0; p->status = 0;
    if ((options & FREESASA_INGEST_SKIP_UNKNOWN) && (options & FREESASA_INGEST_HALT_AT_UNKNOWN))
        options &= ~FREESASA_INGEST_SKIP_UNKNOWN; /* the stricter one wins (ref: src/structure.c:596-597) */
    const int want_het = (options & FREESASA_INGEST_INCLUDE_HETATM) != 0, want_h = (options & FREESASA_INGEST_INCLUDE_HYDROGEN) != 0;
    char the_alt = ' ';
    char prev_number[6] = "", prev_chain = 0;
    size_t pos = 0;
    while (pos < len) {
        const char *line = text + pos;
        const size_t room = len - pos < CHUNK ? len - pos : CHUNK;
        const char *nl = memchr(line, '\n', room);
        size_t n = nl ? (size_t)(nl - line) + 1 : room;
        pos += n;
        if (n < 4 || (line[0] != 'A' && line[0] != 'H' && line[0] != 'E')) continue; /* not ATOM, HETATM or ENDMDL */
        { const char *z = memchr(line, '\0', n); if (z) n = (size_t)(z - line); } /* an embedded NUL ends the line, as it does for the reference's strlen */

        const int is_atom = n >= 4 && memcmp(line, "ATOM", 4) == 0;
        if (is_atom || (want_het && n >= 6 && memcmp(line, "HETATM", 6) == 0)) {
            /* ref: src/pdb.c:259-281 (is_hydrogen), including what it does with lines that lack the element columns */
            const int has_sym = n >= 78;
            const char s0 = has_sym ? line[76] : '\0', s1 = has_sym ? line[77] : '\0';
            int hyd;
            if (n < 13) hyd = -1;
            else if (has_sym && s0 == ' ' && (s1 == 'H' || s1 == 'D')) hyd = 1;
            else if (!(has_sym && s0 == ' ' && s1 == ' ')) hyd = 0; /* (a line without the columns compares unequal to "  ": no hydrogen) */
            else if (!(line[12] == ' ' || (line[12] >= '1' && line[12] <= '9'))) hyd = 0;
            else hyd = (line[12] == 'H' || line[13] == 'H' || line[12] == 'D' || line[13] == 'D') ? 1 : 0;
            if (hyd && !want_h) continue;

            /* the fields atom_new_from_line takes (ref: src/structure.c:199-235) */
            const int has_name = n >= 16;
            const char alt = has_name ? line[16] : '\0';
            char aname[5] = "", rname[4] = "", rnumber[6] = "", symbol[3] = "";
            if (has_name) memcpy(aname, line + 12, 4);
            if (n >= 20) memcpy(rname, line + 17, 3);
            if (n >= 27) memcpy(rnumber, line + 22, 5);
            const char chain = n >= 21 ? line[21] : '\0';
            if (has_sym) { symbol[0] = s0; symbol[1] = s1; }
            if (!has_sym || (s0 == ' ' && s1 == ' ')) {
                /* (a line without a name column is either skipped by the alt-loc rule or fails at
                   the coordinates below, so its symbol never matters) */
                if (has_name) guess_symbol(symbol, aname);
            }

            /* alternate locations: keep blank ones and the first label seen (ref: :675-681) */
            if ((alt != ' ' && the_alt == ' ') || alt == ' ') the_alt = alt;
            else if (alt != ' ' && alt != the_alt) continue;

            /* coordinates (ref: src/pdb.c:176-197) */
            if (n < 54) { p->status = FREESASA_INGEST_EFORMAT; return; }
            double v[3];
            if (!coords_8_3(line + 30, v)) {
                char sec[25];
                memcpy(sec, line + 30, 24);
                sec[24] = '\0';
                const char *sp = sec;
                if (!scan_double(&sp, &v[0]) || !scan_double(&sp, &v[1]) || !scan_double(&sp, &v[2])) {
                    p->status = FREESASA_INGEST_EFORMAT;
                    return;
                }
            }

            /* radius (ref: src/structure.c:519-550, 606-612); the names trimmed once */
            const char *at, *rt, *st;
            const int al = field_token(aname, has_name ? 4 : 0, &at), rl = field_token(rname, n >= 20 ? 3 : 0, &rt);
            double r;
            int cls;
            const double rc = protor_lookup(rt, rl, at, al, &cls);
            if (options & FREESASA_INGEST_RADIUS_FROM_OCCUPANCY) {
                r = 1;
            } else if (rc >= 0) {
                r = rc;
            } else if (options & FREESASA_INGEST_HALT_AT_UNKNOWN) {
                p->status = FREESASA_INGEST_EUNKNOWN;
                return;
            } else if (options & FREESASA_INGEST_SKIP_UNKNOWN) {
                continue;
            } else {
                r = freesasa_ingest_guess_radius(symbol);
                if (r < 0) r = +0.;
            }
            if (options & FREESASA_INGEST_RADIUS_FROM_OCCUPANCY) { /* ref: :696-701, src/pdb.c:31-49, 239-247 */
                if (n < 55) { p->status = FREESASA_INGEST_EFORMAT; return; }
                char buf[8];
                const size_t w = n - 54 < 6 ? n - 54 : 6;
                memcpy(buf, line + 54, w);
                buf[w] = '\0';
                float occ;
                if (sscanf(buf, "%f", &occ) != 1) { p->status = FREESASA_INGEST_EFORMAT; return; }
                r = occ;
            }

            if (grow_atoms(p)) { p->status = FREESASA_INGEST_ENOMEM; return; }
            /* a new residue starts when the residue number or the chain changes (ref: :488-496) */
            if (p->n == p->n0 || memcmp(rnumber, prev_number, 6) != 0 || chain != prev_chain) {
                if (grow_res(p)) { p->status = FREESASA_INGEST_ENOMEM; return; }
                p->res_first[p->nres] = p->n - p->n0;
                p->res_ref[p->nres] = (int16_t)residue_ref_index(rname);
                memset(p->res_name + 4 * p->nres, 0, 4);
                memcpy(p->res_name + 4 * p->nres, rname, 3);
                memcpy(p->res_number + 6 * p->nres, rnumber, 6);
                memset(p->res_chain + 4 * p->nres, 0, 4);
                p->res_chain[4 * p->nres] = chain;
                ++p->nres;
                memcpy(prev_number, rnumber, 6);
                prev_chain = chain;
            }
            p->xyz[3 * p->n] = v[0]; p->xyz[3 * p->n + 1] = v[1]; p->xyz[3 * p->n + 2] = v[2];
            p->rad[p->n] = r;
            p->cls[p->n] = (uint8_t)cls;
            p->bb[p->n] = (uint8_t)backbone_tok(at, al);
            { char *d = p->aname + 4 * p->n; d[0] = d[1] = d[2] = d[3] = 0; for (int k = 0; k < al; ++k) d[k] = at[k]; }
            { const int sl = field_token(symbol, 2, &st); char *d = p->asym + 2 * p->n; d[0] = d[1] = 0; for (int k = 0; k < sl; ++k) d[k] = st[k]; }
            ++p->n;
        }
        if (!(options & FREESASA_INGEST_JOIN_MODELS) && n >= 6 && memcmp(line, "ENDMDL", 6) == 0) break; /* ref: :705-708 */
    }
    if (p->n == p->n0) p->status = FREESASA_INGEST_EEMPTY;
}

/* ------------------------------------------------------------------ mmCIF */

/* What the reference takes from an mmCIF file (ref: src/cif.cc:113-200, src/structure.c:724-836):
 * the rows of the _atom_site loop of every data block, columns group_PDB, auth_asym_id,
 * auth_seq_id, pdbx_PDB_ins_code, auth_comp_id, auth_atom_id, label_alt_id, type_symbol,
 * Cartn_x/y/z, pdbx_PDB_model_num; only the lowest model number unless JOIN_MODELS; "ATOM" rows
 * (HETATM with the option); type_symbol "H" skipped unless INCLUDE_HYDROGEN; first alt-loc label
 * wins ('.' = none); double quotes stripped from the atom name; names cut to the widths of the
 * reference's atom record (3 / 4 / 5 / 2 / 3 characters); coordinates by strtod; the element is
 * the file's type_symbol, never guessed.  An atom the classifier does not know is guessed by
 * element, or dropped under SKIP_UNKNOWN and HALT_AT_UNKNOWN alike (the reference ignores the
 * failure of a single atom here, src/cif.cc:186).
 * The tokenizer follows the CIF 1.1 lexical rules the reference's parser (gemmi) implements:
 * whitespace-separated values, '...' and "..." strings closed by a quote followed by whitespace,
 * ;-delimited text fields at line starts, # comments, case-insensitive data_/loop_/save_ keywords. */
enum { T_END, T_TAG, T_VALUE, T_LOOP, T_DATA, T_SAVE };
typedef struct { const char *p; size_t n; int type; } cif_tok;
typedef struct { const char *cur, *end; int bol; } cif_lex; /* bol: cur is at the beginning of a line */

static int ieq_n(const char *a, const char *b, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        if (tolower((unsigned char)a[i]) != tolower((unsigned char)b[i])) return 0;
    return 1;
}

/* 1 for the CIF whitespace characters */
static const unsigned char cif_ws[256] = {['\t'] = 1, ['\n'] = 1, ['\r'] = 1, [' '] = 1};

/* inlined into the row loop of cif_walk (21 tokens per atom: the call and the struct return cost as
 * much as the scanning); cif_next is the out-of-line copy for everything else */
static inline __attribute__((always_inline)) cif_tok cif_next_inl(cif_lex *lx)
{
    cif_tok t = {NULL, 0, T_END};
    const unsigned char *p = (const unsigned char *)lx->cur, *end = (const unsigned char *)lx->end;
    for (;;) { /* whitespace and comments */
        while (p < end && cif_ws[*p]) ++p;
        if (p < end && *p == '#') {
            const unsigned char *nl = memchr(p, '\n', (size_t)(end - p));
            p = nl ? nl : end;
            continue;
        }
        break;
    }
    if (p >= end) { lx->cur = (const char *)p; return t; }
    const unsigned char *start = p;
    const int bol = (const char *)p == lx->cur ? lx->bol : p[-1] == '\n'; /* whitespace was skipped: look behind */
    if (*p == ';' && bol) { /* text field: up to a line that starts with ';' */
        ++p;
        while (p < end && !(*p == '\n' && p + 1 < end && p[1] == ';')) ++p;
        if (p < end) p += 2;
        t.type = T_VALUE;
    } else if (*p == '\'' || *p == '"') {
        const unsigned char q = *p++;
        while (p < end && *p != '\n') {
            if (*p == q && (p + 1 >= end || cif_ws[p[1]])) { ++p; break; }
            ++p;
        }
        t.type = T_VALUE; /* raw, quotes included, like gemmi */
    } else {
        while (p < end && !cif_ws[*p]) ++p;
        const size_t n = (size_t)(p - start);
        if (*start == '_') t.type = T_TAG;
        else if (n == 5 && ieq_n((const char *)start, "loop_", 5)) t.type = T_LOOP;
        else if (n >= 5 && (start[4] == '_') && ieq_n((const char *)start, "data_", 5)) t.type = T_DATA;
        else if (n >= 5 && (start[4] == '_') && ieq_n((const char *)start, "save_", 5)) t.type = T_SAVE;
        else t.type = T_VALUE;
    }
    t.p = (const char *)start;
    t.n = (size_t)(p - start);
    lx->bol = 0;
    lx->cur = (const char *)p;
    return t;
}

static cif_tok cif_next(cif_lex *lx) { return cif_next_inl(lx); }

/* The values of an _atom_site loop are almost all plain tokens (21 per atom, a few characters each,
 * padded with blanks): scanning them a byte at a time costs a mispredicted branch at every token
 * boundary.  cif_row_next finds the boundaries in a 64-byte whitespace bit mask (SSE2 compares +
 * count-trailing-zeros) and hands anything that is not a plain value -- quotes, comments, text
 * fields, tags and keywords, the last bytes of the text -- to the general tokenizer, so the token
 * stream is the same. */
#if defined(__SSE2__) && !defined(FREESASA_INGEST_NO_SIMD) /* NO_SIMD: the test suite's byte-at-a-time twin */
#include <emmintrin.h>
/* base: start of the current 64-byte block (NULL: none); starts / ends: the not yet consumed token
 * starts (a non-blank after a blank) and token ends (a blank after a non-blank) of the block;
 * last_ws: whether the byte before the next block is whitespace */
typedef struct { const unsigned char *base; uint64_t starts, ends; int last_ws; } cif_fast;
static inline uint64_t cif_ws_mask64(const unsigned char *p)
{
    const __m128i sp = _mm_set1_epi8(' '), nl = _mm_set1_epi8('\n'), tb = _mm_set1_epi8('\t'), cr = _mm_set1_epi8('\r');
    uint64_t m = 0;
    for (int k = 0; k < 4; ++k) {
        const __m128i v = _mm_loadu_si128((const __m128i *)(p + 16 * k));
        const __m128i w = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, sp), _mm_cmpeq_epi8(v, nl)),
                                       _mm_or_si128(_mm_cmpeq_epi8(v, tb), _mm_cmpeq_epi8(v, cr)));
        m |= (uint64_t)(unsigned)_mm_movemask_epi8(w) << (16 * k);
    }
    return m;
}
/* load the block at p; prev_ws: the byte before p is whitespace (or p is the start of the text) */
static inline void cif_fast_load(cif_fast *f, const unsigned char *p, int prev_ws)
{
    const uint64_t ws = cif_ws_mask64(p);
    f->base = p;
    f->starts = ~ws & ((ws << 1) | (uint64_t)(prev_ws != 0));
    f->ends = ws & ((~ws << 1) | (uint64_t)(prev_ws == 0));
    f->last_ws = (int)(ws >> 63);
}
static inline __attribute__((always_inline)) cif_tok cif_row_next(cif_lex *lx, cif_fast *f)
{
    const unsigned char *const end = (const unsigned char *)lx->end;
    const unsigned char *const cur = (const unsigned char *)lx->cur;
    if (!f->base) { /* (re)start behind the token the general tokenizer just returned */
        if (end - cur < 64) return cif_next_inl(lx);
        cif_fast_load(f, cur, 1); /* cur is the blank that ended that token: not an end to report again */
    }
    while (!f->starts) { /* next block with a token start */
        if (end - (f->base + 64) < 64) goto slow;
        cif_fast_load(f, f->base + 64, f->last_ws);
    }
    const unsigned char *const start = f->base + __builtin_ctzll(f->starts);
    {
        const unsigned char c = *start;
        if (c == '#' || c == '\'' || c == '"' || c == '_' || c == ';') goto slow;
    }
    f->starts &= f->starts - 1;
    while (!f->ends) { /* the token runs into the next block(s); no start can precede its end there */
        if (end - (f->base + 64) < 64) goto slow;
        cif_fast_load(f, f->base + 64, f->last_ws);
    }
    const unsigned char *const stop = f->base + __builtin_ctzll(f->ends);
    f->ends &= f->ends - 1;
    if (stop - start >= 5 && start[4] == '_') goto slow; /* loop_, data_..., save_... */
    lx->cur = (const char *)stop;
    lx->bol = 0;
    return (cif_tok){(const char *)start, (size_t)(stop - start), T_VALUE};
slow:
    /* nothing has been consumed as far as the general tokenizer is concerned: it redoes this token
       from the end of the previous one, and the bit masks are rebuilt behind whatever it returns */
    f->base = NULL;
    return cif_next_inl(lx);
}
/* Rows by TEMPLATE (round 5).  The files of the wwPDB pad the values of an _atom_site loop so that every row has its
 * tokens in the same columns (left-aligned: the lengths differ from row to row, the starts do not).  When two
 * consecutive rows have been tokenized one token at a time, the second one becomes a template: the stride S to the
 * next row's first token and where each of its tokens starts.  A following row whose S bytes show token starts -
 * a non-blank behind a blank - in exactly those places has its tokens there, each as long as its run of non-blanks:
 * provided none of them is anything but a plain value, i.e. none starts with a quote, '#', ';' or '_' and none is a
 * keyword (loop_, data_..., save_...: an underscore as fifth character, as cif_row_next tests it - here refused for
 * a token of any length).  A quoted value with a blank inside shows the starts of two plain ones, so the first
 * characters of ALL tokens are looked at, not only of the twelve wanted.  Three bit masks of the row (SSE2 compares)
 * replace 21 trips through the token scanner; any difference - the row behind the loop's last, a value that outgrew
 * its column, a comment - hands the row to the scanner, and the next two rows it tokenizes make the next template. */
#define CIF_TPL_MAX 256 /* bytes of a row's stride */
typedef struct { uint64_t w[CIF_TPL_MAX / 64]; } cif_bits;
static inline void cif_row_masks(const unsigned char *p, int nblk, cif_bits *ws, cif_bits *special, cif_bits *under)
{
    const __m128i sp = _mm_set1_epi8(' '), nl = _mm_set1_epi8('\n'), tb = _mm_set1_epi8('\t'), cr = _mm_set1_epi8('\r');
    const __m128i q1 = _mm_set1_epi8('\''), q2 = _mm_set1_epi8('"'), hs = _mm_set1_epi8('#'), sc = _mm_set1_epi8(';'), us = _mm_set1_epi8('_');
    for (int b = 0; b < nblk; ++b) {
        uint64_t mw = 0, mq = 0, mu = 0;
        for (int k = 0; k < 4; ++k) {
            const __m128i v = _mm_loadu_si128((const __m128i *)(p + 64 * b + 16 * k));
            const __m128i w = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, sp), _mm_cmpeq_epi8(v, nl)), _mm_or_si128(_mm_cmpeq_epi8(v, tb), _mm_cmpeq_epi8(v, cr)));
            const __m128i u = _mm_cmpeq_epi8(v, us);
            const __m128i q = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, q1), _mm_cmpeq_epi8(v, q2)), _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, hs), _mm_cmpeq_epi8(v, sc)), u));
            mw |= (uint64_t)(unsigned)_mm_movemask_epi8(w) << (16 * k);
            mq |= (uint64_t)(unsigned)_mm_movemask_epi8(q) << (16 * k);
            mu |= (uint64_t)(unsigned)_mm_movemask_epi8(u) << (16 * k);
        }
        ws->w[b] = mw; special->w[b] = mq; under->w[b] = mu;
    }
}
#if defined(__x86_64__) && defined(__GNUC__)
/* the same with 32-byte registers where the CPU has them (decided once, at the first row) */
#include <immintrin.h>
__attribute__((target("avx2"))) static void cif_row_masks_avx2(const unsigned char *p, int nblk, cif_bits *ws, cif_bits *special, cif_bits *under)
{
    const __m256i sp = _mm256_set1_epi8(' '), nl = _mm256_set1_epi8('\n'), tb = _mm256_set1_epi8('\t'), cr = _mm256_set1_epi8('\r');
    const __m256i q1 = _mm256_set1_epi8('\''), q2 = _mm256_set1_epi8('"'), hs = _mm256_set1_epi8('#'), sc = _mm256_set1_epi8(';'), us = _mm256_set1_epi8('_');
    for (int b = 0; b < nblk; ++b) {
        uint64_t mw = 0, mq = 0, mu = 0;
        for (int k = 0; k < 2; ++k) {
            const __m256i v = _mm256_loadu_si256((const __m256i *)(p + 64 * b + 32 * k));
            const __m256i w = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, sp), _mm256_cmpeq_epi8(v, nl)), _mm256_or_si256(_mm256_cmpeq_epi8(v, tb), _mm256_cmpeq_epi8(v, cr)));
            const __m256i u = _mm256_cmpeq_epi8(v, us);
            const __m256i q = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, q1), _mm256_cmpeq_epi8(v, q2)), _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, hs), _mm256_cmpeq_epi8(v, sc)), u));
            mw |= (uint64_t)(unsigned)_mm256_movemask_epi8(w) << (32 * k);
            mq |= (uint64_t)(unsigned)_mm256_movemask_epi8(q) << (32 * k);
            mu |= (uint64_t)(unsigned)_mm256_movemask_epi8(u) << (32 * k);
        }
        ws->w[b] = mw; special->w[b] = mq; under->w[b] = mu;
    }
}
static int cif_have_avx2 = -1;
static inline void cif_row_masks_best(const unsigned char *p, int nblk, cif_bits *ws, cif_bits *special, cif_bits *under)
{
    if (cif_have_avx2 < 0) cif_have_avx2 = __builtin_cpu_supports("avx2") ? 1 : 0; /* (a benign race: every thread stores the same value) */
    if (cif_have_avx2) cif_row_masks_avx2(p, nblk, ws, special, under);
    else cif_row_masks(p, nblk, ws, special, under);
}
#else
#define cif_row_masks_best cif_row_masks
#endif
typedef struct {
    int S, nblk;          /* stride in bytes, 64-byte blocks of it */
    uint64_t last;        /* the bits of the last block that belong to the stride */
    cif_bits starts, t4;  /* first characters of the tokens; their fifth characters */
} cif_tpl;
/* do the S bytes at `row` hold plain tokens starting exactly where the template's do?  ws: the row's blanks (bits behind
   the stride set), for the lengths */
static inline int cif_tpl_match(const cif_tpl *T, const unsigned char *row, cif_bits *ws)
{
    cif_bits sp, un;
    cif_row_masks_best(row, T->nblk, ws, &sp, &un);
    uint64_t bad = 0, carry = 1; /* (the byte before the row is a blank: the last of the row before, or where the scanner came from) */
    for (int b = 0; b < T->nblk; ++b) {
        const uint64_t m = b == T->nblk - 1 ? T->last : ~(uint64_t)0;
        const uint64_t w = ws->w[b] | ~m; /* (behind the stride: as if blank) */
        ws->w[b] = w;
        bad |= ((~w & ((w << 1) | carry)) ^ T->starts.w[b]) | (sp.w[b] & T->starts.w[b]) | (un.w[b] & T->t4.w[b]);
        carry = w >> 63;
    }
    const int e = T->S - 1;
    return bad == 0 && ((ws->w[e >> 6] >> (e & 63)) & 1); /* ... and a blank before the next row */
}
/* length of the token that starts at bit o: up to the next blank (there is one inside the stride) */
static inline int cif_tpl_len(const cif_bits *ws, int o)
{
    int b = o >> 6;
    uint64_t w = ws->w[b] >> (o & 63);
    if (w) return __builtin_ctzll(w);
    int n = 64 - (o & 63);
    for (++b; !ws->w[b]; ++b) n += 64;
    return n + __builtin_ctzll(ws->w[b]);
}
/* the row at `row` (its tokens at off[k], ncol of them), S bytes to the next: a template, or 0 */
static int cif_tpl_make(cif_tpl *T, const unsigned char *row, int S, const unsigned short *off, int ncol)
{
    if (S < 2 || S > CIF_TPL_MAX) return 0;
    T->S = S; T->nblk = (S + 63) / 64;
    T->last = (S & 63) ? (((uint64_t)1 << (S & 63)) - 1) : ~(uint64_t)0;
    memset(&T->starts, 0, sizeof T->starts); memset(&T->t4, 0, sizeof T->t4);
    for (int k = 0; k < ncol; ++k) {
        if ((int)off[k] >= S) return 0;
        T->starts.w[off[k] >> 6] |= (uint64_t)1 << (off[k] & 63);
        if (off[k] + 4 < S) T->t4.w[(off[k] + 4) >> 6] |= (uint64_t)1 << ((off[k] + 4) & 63);
    }
    cif_bits ws;
    return cif_tpl_match(T, row, &ws); /* the row itself: its tokens are exactly its runs of non-blanks, all plain values */
}
#define CIF_TEMPLATES 1
#else
typedef struct { int unused; } cif_fast;
static inline cif_tok cif_row_next(cif_lex *lx, cif_fast *f) { (void)f; return cif_next_inl(lx); }
#endif

static const char *const cif_cols[12] = {"group_PDB", "auth_asym_id", "auth_seq_id", "pdbx_PDB_ins_code", "auth_comp_id",
                                         "auth_atom_id", "label_alt_id", "type_symbol", "Cartn_x", "Cartn_y", "Cartn_z",
                                         "pdbx_PDB_model_num"};

/* copy at most w characters of a token into a NUL-terminated field */
static inline void cut(char *dst, size_t w, const char *p, size_t n)
{
    if (n > w) n = w;
    for (size_t i = 0; i < n; ++i) dst[i] = p[i]; /* a few bytes: cheaper than a call to memcpy */
    dst[n] = '\0';
}

typedef struct { cif_lex at; int ncol; int col[12]; } cif_loop; /* an _atom_site loop: lexer state at its first value */

/* Walk the document; calls visit(row tokens) for the rows of the _atom_site category of every data block, found
 * the way the reference finds it (gemmi's block.find("_atom_site.", columns), src/cif.cc:113-126,161-200): the
 * FIRST item of the block that carries _atom_site.group_PDB decides — a loop (it must hold all twelve columns,
 * otherwise the block has no atoms; later _atom_site loops of the same block are not read) or a tag-value pair
 * (then the category is one row made of the block's _atom_site.* pairs, all twelve needed). */
typedef int (*cif_row_fn)(const cif_tok *row, void *ctx);
static int cif_col_of(const cif_tok *t)
{
    if (t->n > 11 && ieq_n(t->p, "_atom_site.", 11))
        for (int k = 0; k < 12; ++k)
            if (strlen(cif_cols[k]) == t->n - 11 && ieq_n(t->p + 11, cif_cols[k], t->n - 11)) return k;
    return -1;
}
static int cif_walk(const char *text, size_t len, cif_row_fn visit, void *ctx)
{
    cif_lex lx = {text, text + len, 1};
    cif_tok t = cif_next(&lx);
    int decided = 0;    /* this block's _atom_site category has been located: 1 a loop (done), 2 pairs */
    cif_tok prow[12];   /* pair form: the values collected so far */
    int phave = 0;      /* bit k: prow[k] is set */
    while (1) {
        if (t.type == T_END || t.type == T_DATA) { /* a block ends: a complete row of pairs is its one atom */
            if (decided == 2 && phave == 0xfff) {
                const int rc = visit(prow, ctx);
                if (rc) return rc;
            }
            decided = 0; phave = 0;
            if (t.type == T_END) break;
            t = cif_next(&lx);
            continue;
        }
        if (t.type == T_TAG) { /* tag-value pair */
            const int k = cif_col_of(&t);
            cif_tok v = cif_next(&lx);
            if (v.type != T_VALUE) { t = v; continue; } /* (a tag without a value: malformed, skip the tag) */
            if (k >= 0) {
                if (k == 0 && !decided) decided = 2;
                if (!(phave >> k & 1)) { prow[k] = v; phave |= 1 << k; } /* first occurrence, like a lookup by tag */
            }
            t = cif_next(&lx);
            continue;
        }
        if (t.type != T_LOOP) { /* save_, stray values */
            t = cif_next(&lx);
            continue;
        }
        int ncol = 0, col[12], has_first = 0;
        signed char slot_of[256]; /* column of the loop -> wanted field, -1: not wanted */
        memset(slot_of, -1, sizeof slot_of);
        for (int k = 0; k < 12; ++k) col[k] = -1;
        t = cif_next(&lx);
        while (t.type == T_TAG) {
            const int k = cif_col_of(&t);
            if (k >= 0) {
                if (k == 0) has_first = 1;
                if (col[k] < 0) { col[k] = ncol; if (ncol < 256) slot_of[ncol] = (signed char)k; }
            }
            ++ncol;
            t = cif_next(&lx);
        }
        int complete = has_first && !decided && ncol > 0 && ncol <= 256;
        for (int k = 0; k < 12; ++k) complete = complete && col[k] >= 0;
        if (has_first && !decided) decided = 1; /* this loop is the category, complete or not */
        cif_tok row[12];
        int c = 0;
        cif_fast fast = {0};
#ifdef CIF_TEMPLATES
        unsigned short toff[64], tlen[64]; /* the row being tokenized: where its tokens lie behind its first, their lengths */
        const char *row0 = NULL, *prev0 = NULL; /* its first token; the first token of the row before it (NULL: not a row to learn from) */
        int fits = 0;
        const int learn = complete && ncol <= 64;
#endif
        while (t.type == T_VALUE) {
            if (complete && slot_of[c] >= 0) row[slot_of[c]] = t;
#ifdef CIF_TEMPLATES
            if (learn) {
                if (c == 0) { row0 = t.p; fits = 1; }
                if (t.p - row0 < CIF_TPL_MAX && t.n < CIF_TPL_MAX) { toff[c] = (unsigned short)(t.p - row0); tlen[c] = (unsigned short)t.n; } else fits = 0;
            }
#endif
            if (++c == ncol) {
                c = 0;
                if (complete) {
                    const int rc = visit(row, ctx);
                    if (rc) return rc;
                }
#ifdef CIF_TEMPLATES
                if (learn) {
                    cif_tpl T;
                    const char *const last_end = row0 + toff[ncol - 1] + tlen[ncol - 1];
                    /* (the 64-byte blocks read from a row reach at most 63 bytes behind its stride) */
                    if (fits && prev0 && lx.cur == last_end && row0 - prev0 <= CIF_TPL_MAX && lx.end - row0 >= 2 * (row0 - prev0) + 128 &&
                        cif_tpl_make(&T, (const unsigned char *)row0, (int)(row0 - prev0), toff, ncol)) {
                        const char *r = row0 + T.S, *done = NULL; /* candidate row; the last row taken by template */
                        cif_bits ws;
                        int last_len = 0;
                        while (lx.end - r >= T.S + 128 && cif_tpl_match(&T, (const unsigned char *)r, &ws)) {
                            for (int k = 0; k < ncol; ++k)
                                if (slot_of[k] >= 0) { cif_tok v = {r + toff[k], (size_t)cif_tpl_len(&ws, toff[k]), T_VALUE}; row[slot_of[k]] = v; }
                            last_len = cif_tpl_len(&ws, toff[ncol - 1]);
                            const int rc = visit(row, ctx);
                            if (rc) return rc;
                            done = r;
                            r += T.S;
                        }
                        if (done) { /* the scanner goes on behind the last token of the last row taken */
                            lx.cur = done + toff[ncol - 1] + last_len;
                            lx.bol = 0;
                            fast.base = NULL;
                        }
                        prev0 = NULL; /* (two more rows by the scanner before the next template) */
                    } else {
                        prev0 = fits ? row0 : NULL;
                    }
                }
#endif
            }
            t = cif_row_next(&lx, &fast);
        }
    }
    return 0;
}

/* strtol(., 10) of a token, in place (the token is followed by whitespace or the end of the text) */
static int tok_int(const cif_tok *t)
{
    const char *p = t->p, *e = t->p + t->n;
    int neg = 0;
    if (p < e && (*p == '-' || *p == '+')) neg = *p++ == '-';
    long v = 0;
    while (p < e && *p >= '0' && *p <= '9' && v < 100000000L) v = v * 10 + (*p++ - '0');
    return (int)(neg ? -v : v);
}

/* [+-]digits[.digits] filling the whole token, at most 15 digits: integer / power of ten, both exact,
 * one correctly rounded division = what atof returns (see scan_double) */
static int tok_plain_double(const cif_tok *t, double *out)
{
    const char *q = t->p, *e = t->p + t->n;
    int neg = 0;
    if (q < e && (*q == '-' || *q == '+')) neg = *q++ == '-';
    uint64_t m = 0;
    int digits = 0, frac = 0;
    while (q < e && *q >= '0' && *q <= '9') { m = m * 10 + (uint64_t)(*q++ - '0'); ++digits; }
    if (q < e && *q == '.') {
        ++q;
        while (q < e && *q >= '0' && *q <= '9') { m = m * 10 + (uint64_t)(*q++ - '0'); ++digits; ++frac; }
    }
    if (q != e || digits < 1 || digits > 15) return 0;
    const double v = (double)m / pow10_tab[frac];
    *out = neg ? -v : v;
    return 1;
}

typedef struct {
    parsed *p;
    int options;
    int have_model, model, min_model; /* model being read (the first one met) and the lowest seen */
    char prev_alt;
    char prev_number[6], prev_chain[4];
} cif_atoms;

static int cif_visit_atom(const cif_tok *row, void *ctx)
{
    cif_atoms *c = (cif_atoms *)ctx;
    parsed *p = c->p;
    /* (the model set is built from every row, whatever its record type) */
    const int model = tok_int(&row[11]);
    if (!c->have_model) { c->have_model = 1; c->model = c->min_model = model; }
    if (model < c->min_model) c->min_model = model;
    if (!(row[0].n == 4 && memcmp(row[0].p, "ATOM", 4) == 0) && !(c->options & FREESASA_INGEST_INCLUDE_HETATM)) return 0;
    if (!(c->options & FREESASA_INGEST_JOIN_MODELS) && model != c->model) return 0;
    if (!(c->options & FREESASA_INGEST_INCLUDE_HYDROGEN) && row[7].n == 1 && row[7].p[0] == 'H') return 0;
    const char alt = row[6].p[0];
    if ((alt != '.' && c->prev_alt == '.') || alt == '.') c->prev_alt = alt;
    else if (alt != '.' && alt != c->prev_alt) return 0;

    char aname[5], rname[4], rnumber[6], symbol[3], chain[4], num[24];
    if (row[5].n >= 2 && row[5].p[0] == '"') cut(aname, 4, row[5].p + 1, row[5].n - 2);
    else cut(aname, 4, row[5].p, row[5].n);
    cut(rname, 3, row[4].p, row[4].n);
    /* residue number = auth_seq_id (at most 15 characters of it, up to a NUL) + the insertion code
       unless it is '?', cut to 5 characters (ref: src/cif.cc:139-147, the atom record's field width) */
    size_t nn = row[2].n < 15 ? row[2].n : 15;
    memcpy(num, row[2].p, nn);
    { const char *z = memchr(num, '\0', nn); if (z) nn = (size_t)(z - num); }
    if (row[3].p[0] != '?') num[nn++] = row[3].p[0];
    num[nn] = '\0';
    cut(rnumber, 5, num, strlen(num));
    cut(symbol, 2, row[7].p, row[7].n);
    cut(chain, 3, row[1].p, row[1].n);
    double v[3];
    for (int k = 0; k < 3; ++k) { /* atof: plain decimals exactly like the PDB reader's fast path, else strtod */
        if (tok_plain_double(&row[8 + k], &v[k])) continue;
        char buf[40];
        cut(buf, sizeof buf - 1, row[8 + k].p, row[8 + k].n);
        const char *sp = buf;
        if (!scan_double(&sp, &v[k])) v[k] = 0.0;
    }

    /* names without blanks or NULs (all but quoted oddities) are their own first token: classify and
       store them without the sscanf("%s")-style rescans of the general path */
    int al = 0, rl = 0, sl = 0, plain = 1;
    while (aname[al]) { plain &= !isspace((unsigned char)aname[al]); ++al; }
    while (rname[rl]) { plain &= !isspace((unsigned char)rname[rl]); ++rl; }
    while (symbol[sl]) { plain &= !isspace((unsigned char)symbol[sl]); ++sl; }
    plain = plain && (size_t)al == (row[5].n >= 2 && row[5].p[0] == '"' ? (row[5].n - 2 < 4 ? row[5].n - 2 : 4) : (row[5].n < 4 ? row[5].n : 4)) &&
            (size_t)rl == (row[4].n < 3 ? row[4].n : 3) && (size_t)sl == (row[7].n < 2 ? row[7].n : 2);
    int cls;
    double r = plain ? protor_lookup(rname, rl, aname, al, &cls) : freesasa_ingest_protor_radius(rname, aname, &cls);
    if (r < 0) {
        if (c->options & (FREESASA_INGEST_HALT_AT_UNKNOWN | FREESASA_INGEST_SKIP_UNKNOWN)) return 0;
        r = freesasa_ingest_guess_radius(symbol);
        if (r < 0) r = +0.;
    }
    if (grow_atoms(p)) { p->status = FREESASA_INGEST_ENOMEM; return -1; }
    /* zero-padded copies of the residue labels: compared and stored as whole words */
    char znum[8] = {0}, zchain[4] = {0};
    for (int i = 0; i < 5 && rnumber[i]; ++i) znum[i] = rnumber[i];
    for (int i = 0; i < 3 && chain[i]; ++i) zchain[i] = chain[i];
    if (p->n == p->n0 || memcmp(znum, c->prev_number, 6) != 0 || memcmp(zchain, c->prev_chain, 4) != 0) {
        if (grow_res(p)) { p->status = FREESASA_INGEST_ENOMEM; return -1; }
        p->res_first[p->nres] = p->n - p->n0;
        p->res_ref[p->nres] = (int16_t)residue_ref_index(rname);
        char zname[4] = {0};
        for (int i = 0; i < 3 && rname[i]; ++i) zname[i] = rname[i];
        memcpy(p->res_name + 4 * p->nres, zname, 4);
        memcpy(p->res_number + 6 * p->nres, znum, 6);
        memcpy(p->res_chain + 4 * p->nres, zchain, 4);
        ++p->nres;
        memcpy(c->prev_number, znum, 6);
        memcpy(c->prev_chain, zchain, 4);
    }
    p->xyz[3 * p->n] = v[0]; p->xyz[3 * p->n + 1] = v[1]; p->xyz[3 * p->n + 2] = v[2];
    p->rad[p->n] = r;
    p->cls[p->n] = (uint8_t)cls;
    if (plain) {
        p->bb[p->n] = (uint8_t)backbone_tok(aname, al);
        char *dn = p->aname + 4 * p->n, *ds = p->asym + 2 * p->n;
        for (int i = 0; i < 4; ++i) dn[i] = i < al ? aname[i] : '\0';
        for (int i = 0; i < 2; ++i) ds[i] = i < sl ? symbol[i] : '\0';
    } else {
        p->bb[p->n] = (uint8_t)freesasa_ingest_is_backbone(aname);
        store_token(p->aname + 4 * p->n, 4, aname);
        store_token(p->asym + 2 * p->n, 2, symbol);
    }
    ++p->n;
    return 0;
}

static void parse_cif(const char *text, size_t len, int options, parsed *p)
{
    /* The reference collects the model numbers first and keeps the lowest (src/cif.cc:78-87,
       225-234).  Files list their models in ascending order, so one pass suffices: read the atoms
       of the first model met while tracking the minimum, and start over only if a lower number
       shows up later. */
    cif_atoms c;
    for (int pass = 0; pass < 2; ++pass) {
        p->n = p->n0; p->nres = p->nres0; p->status = 0;
        memset(&c, 0, sizeof c);
        c.p = p; c.options = options; c.prev_alt = '.';
        if (pass == 1) { c.have_model = 1; c.model = c.min_model = p->scratch_model; }
        cif_walk(text, len, cif_visit_atom, &c);
        if (!c.have_model || c.min_model == c.model || p->status) break;
        p->scratch_model = c.min_model;
    }
    if (p->status == 0 && p->n == p->n0) p->status = FREESASA_INGEST_EEMPTY;
}

/* A file whose first token is data_... is mmCIF; anything else is read as PDB. */
static int looks_like_cif(const char *text, size_t len)
{
    cif_lex lx = {text, text + (len < 4096 ? len : 4096), 1};
    const cif_tok t = cif_next(&lx);
    return t.type == T_DATA;
}

/* For the device-side parser (gpu_parse.hip): what kind of text this is and, for mmCIF in its everyday form, where the rows
 * of its _atom_site loop begin and which of the loop's columns are the twelve the reader wants (cif_cols' order).
 * 0: read as PDB; 1: mmCIF, ONE data block so far, whose first item carrying _atom_site.group_PDB is a loop with all twelve
 * columns, one tag per header line (*row0 = offset of the line behind the header); 2: mmCIF that the host parser has to
 * read (pair form, a block without the category before another block, several tokens on a header line, no loop at all).
 * A LINE scan of the text before the loop - text fields (';' in column one) are stepped over, tags, values and other
 * loops are not looked into -: what is left of the host's share of an mmCIF file when the device parses it. */
int freesasa_ingest_cif_locate(const char *text, size_t len, int *ncol_out, signed char slot_out[12], size_t *row0_out)
{
    if (!looks_like_cif(text, len)) return 0;
    size_t pos = 0;
    int in_text = 0, n_data = 0;
    while (pos < len) {
        const char *line = text + pos;
        const char *nl = memchr(line, '\n', len - pos);
        const size_t n = nl ? (size_t)(nl - line) : len - pos;
        const size_t next = pos + n + (nl ? 1 : 0);
        if (in_text) {
            if (n > 0 && line[0] == ';') {
                in_text = 0;
                for (size_t i = 1; i < n; ++i) if (!cif_ws[(unsigned char)line[i]]) return 2; /* something behind the closing ';': host */
            }
            pos = next;
            continue;
        }
        if (n > 0 && line[0] == ';') { in_text = 1; pos = next; continue; }
        size_t s = 0;
        while (s < n && cif_ws[(unsigned char)line[s]]) ++s;
        if (s == n || line[s] == '#') { pos = next; continue; }
        size_t e = s;
        while (e < n && !cif_ws[(unsigned char)line[e]]) ++e;
        const size_t tl = e - s;
        if (line[s] == '_') {
            if (tl == 20 && ieq_n(line + s, "_atom_site.group_pdb", 20)) return 2; /* the pair form decides: host */
            pos = next;
            continue;
        }
        if (tl >= 5 && line[s + 4] == '_' && ieq_n(line + s, "data_", 5)) {
            if (++n_data > 1) return 2; /* a second block (the first had no _atom_site loop): host */
            pos = next;
            continue;
        }
        if (!(tl == 5 && ieq_n(line + s, "loop_", 5))) { pos = next; continue; } /* values, save_ frames */
        for (size_t i = e; i < n; ++i) if (!cif_ws[(unsigned char)line[i]]) return 2; /* tags on the loop_ line: host */
        /* the loop's header: one tag per line (blank lines and comments between them are stepped over) */
        int ncol = 0, col[12], has_first = 0;
        for (int k = 0; k < 12; ++k) col[k] = -1;
        size_t q = next;
        while (q < len) {
            const char *hl = text + q;
            const char *hn = memchr(hl, '\n', len - q);
            const size_t m = hn ? (size_t)(hn - hl) : len - q;
            size_t a = 0;
            while (a < m && cif_ws[(unsigned char)hl[a]]) ++a;
            if (a == m || hl[a] == '#') { q += m + (hn ? 1 : 0); continue; }
            if (hl[a] != '_') break; /* the first row (or whatever follows) */
            size_t b = a;
            while (b < m && !cif_ws[(unsigned char)hl[b]]) ++b;
            for (size_t i = b; i < m; ++i) if (!cif_ws[(unsigned char)hl[i]]) return 2; /* two tokens on a header line: host */
            const cif_tok t = {hl + a, b - a, T_TAG};
            const int k = cif_col_of(&t);
            if (k >= 0) {
                if (k == 0) has_first = 1;
                if (col[k] < 0) col[k] = ncol;
            }
            ++ncol;
            q += m + (hn ? 1 : 0);
        }
        if (has_first) { /* this loop is the block's _atom_site category */
            for (int k = 0; k < 12; ++k) if (col[k] < 0 || col[k] >= 64) return 2; /* incomplete (no atoms for the reference), or wider than the device's table: host */
            if (ncol < 1 || ncol > 64) return 2;
            *ncol_out = ncol;
            for (int k = 0; k < 12; ++k) slot_out[k] = (signed char)col[k];
            *row0_out = q;
            return 1;
        }
        pos = q; /* another category's loop: its rows are stepped over line by line */
    }
    return 2;
}

static void parse_any(const char *text, size_t len, int options, parsed *p)
{
    if (looks_like_cif(text, len)) parse_cif(text, len, options, p);
    else parse_pdb(text, len, options, p);
}

/* Whole file into a buffer that is reused from call to call.  0 on success, else the input's status
   (FREESASA_INGEST_EIO, or FREESASA_INGEST_ENOMEM when the buffer could not grow: round 6, found by tests/test_hostfault.py -
   it used to be reported as an unreadable file). */
static int read_file(const char *path, char **buf, size_t *cap, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) return FREESASA_INGEST_EIO;
    size_t n = 0;
    for (;;) {
        if (n == *cap) {
            const size_t nc = *cap ? 2 * *cap : (size_t)1 << 20;
            char *nb = hf_realloc(*buf, nc);
            if (!nb) { fclose(f); return FREESASA_INGEST_ENOMEM; }
            *buf = nb;
            *cap = nc;
        }
        const size_t got = fread(*buf + n, 1, *cap - n, f);
        n += got;
        if (got == 0) break;
    }
    const int bad = ferror(f);
    fclose(f);
    *len = n;
    return bad ? FREESASA_INGEST_EIO : 0;
}

/* ------------------------------------------------------------------ the batch */

/* Every worker appends the inputs it takes to its own arena (one set of growing arrays, no per-file
 * allocation); when all inputs are parsed, one thread sizes the output and all workers copy their
 * inputs' slices into place (parallel first touch of the output pages). */
typedef struct { int worker; int status; int64_t a0, na, r0, nr; } slot; /* where input k lives */

typedef struct {
    const char *const *paths;
    const char *const *texts;
    const size_t *lens;
    int n, options, n_workers;
    slot *slots;
    parsed *arena; /* [n_workers] */
    int64_t *a_off, *r_off; /* [n + 1] output offsets of every input */
    freesasa_ingest_batch *out;
    int rc;
    int next, next2; /* work counters of the two phases */
    pthread_mutex_t mu;
    pthread_barrier_t bar;
} job;

/* arenas are kept between calls (a fresh one costs its page faults again) */
#define ARENA_POOL 64
static parsed g_pool[ARENA_POOL];
static int g_pool_n = 0;
static pthread_mutex_t g_pool_mu = PTHREAD_MUTEX_INITIALIZER;

static void arena_get(parsed *a)
{
    memset(a, 0, sizeof *a);
    pthread_mutex_lock(&g_pool_mu);
    if (g_pool_n > 0) *a = g_pool[--g_pool_n];
    pthread_mutex_unlock(&g_pool_mu);
    a->n = a->nres = a->n0 = a->nres0 = 0;
    a->status = 0;
}
static void arena_put(parsed *a)
{
    pthread_mutex_lock(&g_pool_mu);
    if (g_pool_n < ARENA_POOL && a->cap <= ((int64_t)1 << 21)) { g_pool[g_pool_n++] = *a; memset(a, 0, sizeof *a); }
    pthread_mutex_unlock(&g_pool_mu);
    parsed_free(a);
}

/* the library is usually dlopen'd: give the cached arenas back when it goes */
__attribute__((destructor)) static void arena_pool_release(void)
{
    pthread_mutex_lock(&g_pool_mu);
    while (g_pool_n > 0) parsed_free(&g_pool[--g_pool_n]);
    pthread_mutex_unlock(&g_pool_mu);
}

static int take(job *j, int *counter)
{
    pthread_mutex_lock(&j->mu);
    const int k = (*counter)++;
    pthread_mutex_unlock(&j->mu);
    return k;
}

static void assemble_sizes(job *j)
{
    freesasa_ingest_batch *out = j->out;
    int64_t na = 0, nr = 0;
    for (int k = 0; k < j->n; ++k) {
        j->a_off[k] = na; j->r_off[k] = nr;
        na += j->slots[k].na; nr += j->slots[k].nr;
    }
    j->a_off[j->n] = na; j->r_off[j->n] = nr;
    if (ingest_batch_alloc__(out, j->n, na, nr)) {
        j->rc = FREESASA_INGEST_ENOMEM;
        return;
    }
    memcpy(out->offsets, j->a_off, sizeof(int64_t) * ((size_t)j->n + 1));
    memcpy(out->res_offsets, j->r_off, sizeof(int64_t) * ((size_t)j->n + 1));
    out->res_first[nr] = na;
}

static void *worker(void *arg)
{
    job *j = (job *)arg;
    /* which worker am I: the order of arrival */
    const int w = take(j, &j->n_workers);
    /* the worker's arena is filled through a copy of its descriptor on the worker's own stack: the descriptors of all
       workers lie side by side in j->arena, the parser bumps a counter in it for every atom, and two workers on one
       cache line ran no faster than one (round 5: 2 threads 9.3e6 atoms/s where 1 did 9.3e6) */
    parsed local, *A = &local;
    arena_get(A);
    char *text = NULL; /* file text, reused from input to input */
    size_t cap = 0;
    for (;;) {
        const int k = take(j, &j->next);
        if (k >= j->n) break;
        A->n0 = A->n; A->nres0 = A->nres;
        if (j->texts) {
            parse_any(j->texts[k], j->lens[k], j->options, A);
        } else {
            size_t len = 0;
            if ((A->status = read_file(j->paths[k], &text, &cap, &len)) != 0) { /* (EIO or ENOMEM) */ }
            else parse_any(text, len, j->options, A);
        }
        slot *s = &j->slots[k];
        s->worker = w;
        s->status = A->status;
        if (A->status) { A->n = A->n0; A->nres = A->nres0; } /* failed inputs contribute an empty structure */
        s->a0 = A->n0; s->na = A->n - A->n0;
        s->r0 = A->nres0; s->nr = A->nres - A->nres0;
    }
    free(text);
    j->arena[w] = local; /* (read by every worker behind the barrier) */
    A = &j->arena[w];
    if (pthread_barrier_wait(&j->bar) == PTHREAD_BARRIER_SERIAL_THREAD) assemble_sizes(j);
    pthread_barrier_wait(&j->bar);
    if (!j->rc) {
        freesasa_ingest_batch *out = j->out;
        for (;;) {
            const int k = take(j, &j->next2);
            if (k >= j->n) break;
            const slot *s = &j->slots[k];
            const parsed *S = &j->arena[s->worker];
            const int64_t a = j->a_off[k], r = j->r_off[k];
            out->status[k] = s->status;
            if (s->na) {
                memcpy(out->xyz + 3 * a, S->xyz + 3 * s->a0, sizeof(double) * 3 * (size_t)s->na);
                memcpy(out->radii + a, S->rad + s->a0, sizeof(double) * (size_t)s->na);
                memcpy(out->atom_class + a, S->cls + s->a0, (size_t)s->na);
                memcpy(out->atom_backbone + a, S->bb + s->a0, (size_t)s->na);
                memcpy(out->atom_name + 4 * a, S->aname + 4 * s->a0, 4 * (size_t)s->na);
                memcpy(out->atom_symbol + 2 * a, S->asym + 2 * s->a0, 2 * (size_t)s->na);
            }
            for (int64_t i = 0; i < s->nr; ++i) out->res_first[r + i] = a + S->res_first[s->r0 + i];
            if (s->nr) {
                memcpy(out->res_ref + r, S->res_ref + s->r0, sizeof(int16_t) * (size_t)s->nr);
                memcpy(out->res_name + 4 * r, S->res_name + 4 * s->r0, 4 * (size_t)s->nr);
                memcpy(out->res_number + 6 * r, S->res_number + 6 * s->r0, 6 * (size_t)s->nr);
                memcpy(out->res_chain + 4 * r, S->res_chain + 4 * s->r0, 4 * (size_t)s->nr);
            }
        }
    }
    pthread_barrier_wait(&j->bar); /* nobody's arena may go while others still copy from it */
    arena_put(A);
    return NULL;
}

/* A batch's arrays lie in ONE block (16-byte header, then every array at a multiple of 16), and the blocks of freed
 * batches are kept for the next ones (round 5).  A sweep builds a batch of ~40 MB per million atoms every few
 * milliseconds; fresh from malloc that is an mmap, 10 000 first-touch page faults taken by all loader threads at once
 * on one address space, and a munmap - a third of the loader's time on one thread, more on sixteen.  A kept block
 * has its pages.  (freesasa_ingest_load builds its batches here too: freesasa_ingest_free is the one way back.) */
#define BLOCK_POOL 24
#define BLOCK_MAGIC 0x66736162636b3031ULL
#define BLOCK_KEEP_MIN ((size_t)1 << 20)       /* smaller blocks are not worth keeping */
#define BLOCK_KEEP_TOTAL ((size_t)1 << 30)     /* bytes kept at most (round 6: 1 GiB, was 3; FREESASA_INGEST_KEEP_MB overrides; freesasa_ingest_trim gives them back) */
static struct { void *p; size_t cap; } g_blocks[BLOCK_POOL];
static int g_nblocks = 0;
static size_t g_block_bytes = 0;
static pthread_mutex_t g_block_mu = PTHREAD_MUTEX_INITIALIZER;

static size_t block_keep_total(void)
{
    static size_t cap = 0; /* (read once; racing threads compute the same value) */
    if (!cap) {
        const char *e = getenv("FREESASA_INGEST_KEEP_MB");
        const long long mb = e ? atoll(e) : -1;
        cap = mb >= 0 ? ((size_t)mb << 20) + 1 : BLOCK_KEEP_TOTAL;
    }
    return cap;
}

static void *block_get(size_t bytes, size_t *cap_out)
{
    void *p = NULL;
    pthread_mutex_lock(&g_block_mu);
    int best = -1;
    for (int k = 0; k < g_nblocks; ++k) /* the smallest that fits, and not one four times too large */
        if (g_blocks[k].cap >= bytes && g_blocks[k].cap / 4 <= bytes + BLOCK_KEEP_MIN && (best < 0 || g_blocks[k].cap < g_blocks[best].cap)) best = k;
    if (best >= 0) {
        p = g_blocks[best].p; *cap_out = g_blocks[best].cap;
        g_block_bytes -= g_blocks[best].cap;
        g_blocks[best] = g_blocks[--g_nblocks];
    }
    pthread_mutex_unlock(&g_block_mu);
    if (!p) { p = hf_malloc(bytes); *cap_out = bytes; }
    return p;
}
static void block_put(void *p, size_t cap)
{
    int kept = 0;
    pthread_mutex_lock(&g_block_mu);
    if (cap >= BLOCK_KEEP_MIN && g_nblocks < BLOCK_POOL && g_block_bytes + cap <= block_keep_total()) {
        g_blocks[g_nblocks].p = p; g_blocks[g_nblocks].cap = cap; ++g_nblocks;
        g_block_bytes += cap;
        kept = 1;
    }
    pthread_mutex_unlock(&g_block_mu);
    if (!kept) free(p);
}
/* Give kept blocks back to the allocator, largest first, until at most keep_bytes are held; returns the bytes released.
   A long-lived process that loaded one large batch and will not load another calls this (round-5 advisor: the pool was
   only emptied when the library was unloaded). */
size_t freesasa_ingest_trim(size_t keep_bytes)
{
    size_t freed = 0;
    pthread_mutex_lock(&g_block_mu);
    while (g_nblocks > 0 && g_block_bytes > keep_bytes) {
        int big = 0;
        for (int k = 1; k < g_nblocks; ++k)
            if (g_blocks[k].cap > g_blocks[big].cap) big = k;
        free(g_blocks[big].p);
        g_block_bytes -= g_blocks[big].cap; freed += g_blocks[big].cap;
        g_blocks[big] = g_blocks[--g_nblocks];
    }
    pthread_mutex_unlock(&g_block_mu);
    return freed;
}
__attribute__((destructor)) static void block_pool_release(void) { (void)freesasa_ingest_trim(0); }

static size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }

/* internal (ingest_cache.c builds its batches with it too): the arrays of a batch of ns structures, na atoms, nr
   residues, uninitialised; 0 or -1 (out of memory, *b zeroed) */
int ingest_batch_alloc__(freesasa_ingest_batch *b, int32_t ns, int64_t na, int64_t nr)
{
    memset(b, 0, sizeof *b);
    const size_t n = (size_t)na, r = (size_t)nr, s = (size_t)ns;
    const size_t sz[14] = {24 * n, 8 * n, n, n, 4 * n, 2 * n, 8 * (s + 1), 8 * (r + 1), 8 * (s + 1), 2 * r, 4 * r, 6 * r, 4 * r, 4 * s};
    size_t total = 16;
    for (int k = 0; k < 14; ++k) total += up16(sz[k] ? sz[k] : 1);
    size_t cap = 0;
    char *blk = block_get(total, &cap);
    if (!blk) return -1;
    ((uint64_t *)blk)[0] = BLOCK_MAGIC; ((uint64_t *)blk)[1] = (uint64_t)cap;
    char *q = blk + 16;
    void *at[14];
    for (int k = 0; k < 14; ++k) { at[k] = q; q += up16(sz[k] ? sz[k] : 1); }
    b->n_structs = ns; b->n_atoms = na; b->n_residues = nr;
    b->xyz = at[0]; b->radii = at[1]; b->atom_class = at[2]; b->atom_backbone = at[3]; b->atom_name = at[4]; b->atom_symbol = at[5];
    b->offsets = at[6]; b->res_first = at[7]; b->res_offsets = at[8]; b->res_ref = at[9]; b->res_name = at[10]; b->res_number = at[11];
    b->res_chain = at[12]; b->status = at[13];
    return 0;
}

void freesasa_ingest_free(freesasa_ingest_batch *b)
{
    if (!b) return;
    if (b->xyz) { /* (the block starts 16 bytes before its first array) */
        char *blk = (char *)b->xyz - 16;
        if (((uint64_t *)blk)[0] == BLOCK_MAGIC) { ((uint64_t *)blk)[0] = 0; block_put(blk, (size_t)((uint64_t *)blk)[1]); }
    }
    memset(b, 0, sizeof *b);
}

/* CPUs this process may really use at once: its affinity mask, capped by the CPU quota of its cgroup (a GPU box can
 * show 256 logical CPUs and grant 16 of them: threads beyond the quota only take turns).  cgroup v2 "cpu.max" =
 * "<quota> <period>" or "max <period>"; v1: cpu.cfs_quota_us / cpu.cfs_period_us. */
int freesasa_ingest_usable_cpus(void)
{
    int n = (int)sysconf(_SC_NPROCESSORS_ONLN);
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0 && CPU_COUNT(&set) < n) n = CPU_COUNT(&set);
    double quota = -1, period = 0;
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        char q[32];
        if (fscanf(f, "%31s %lf", q, &period) == 2 && strcmp(q, "max") != 0) quota = atof(q);
        fclose(f);
    } else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))) {
        if (fscanf(f, "%lf", &quota) != 1) quota = -1;
        fclose(f);
        if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r"))) {
            if (fscanf(f, "%lf", &period) != 1) period = 0;
            fclose(f);
        }
    }
    if (quota > 0 && period > 0) {
        const int g = (int)(quota / period);
        if (g >= 1 && g < n) n = g;
        else if (g < 1) n = 1;
    }
    return n < 1 ? 1 : n;
}

static int run(job *j, int n_threads, freesasa_ingest_batch *out)
{
    memset(out, 0, sizeof *out);
    const int supported = FREESASA_INGEST_INCLUDE_HETATM | FREESASA_INGEST_INCLUDE_HYDROGEN | FREESASA_INGEST_JOIN_MODELS |
                          FREESASA_INGEST_HALT_AT_UNKNOWN | FREESASA_INGEST_SKIP_UNKNOWN | FREESASA_INGEST_RADIUS_FROM_OCCUPANCY;
    if (j->options & ~supported) return FREESASA_INGEST_EOPTION;
    if (j->n < 0) return FREESASA_INGEST_EOPTION;
    if (n_threads <= 0) { /* default: the cores, but no more threads than pay for their start-up */
        n_threads = freesasa_ingest_usable_cpus();
        /* one process per GPU (torchrun exports LOCAL_WORLD_SIZE): the ranks of a node share its cores */
        const char *lws = getenv("LOCAL_WORLD_SIZE");
        const int ranks = lws ? atoi(lws) : 1;
        if (ranks > 1) n_threads /= ranks;
        if (n_threads > 64) n_threads = 64;
        if (n_threads > j->n / 4) n_threads = j->n / 4;
    }
    if (n_threads > j->n) n_threads = j->n;
    if (n_threads < 1) n_threads = 1;
    j->out = out;
    j->slots = hf_calloc((size_t)(j->n > 0 ? j->n : 1), sizeof(slot));
    j->arena = hf_calloc((size_t)n_threads, sizeof(parsed));
    j->a_off = hf_malloc(sizeof(int64_t) * ((size_t)j->n + 1));
    j->r_off = hf_malloc(sizeof(int64_t) * ((size_t)j->n + 1));
    pthread_t *th = hf_calloc((size_t)n_threads, sizeof(pthread_t));
    if (!j->slots || !j->arena || !j->a_off || !j->r_off || !th) {
        free(j->slots); free(j->arena); free(j->a_off); free(j->r_off); free(th);
        return FREESASA_INGEST_ENOMEM;
    }
    pthread_mutex_init(&j->mu, NULL);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    /* the barrier needs the exact number of participants: start the threads first */
    int started = 0;
    pthread_barrier_t *bar = &j->bar;
    /* count how many threads can be created before anyone reaches the barrier: workers block on a
       start gate (the mutex) until the count is known */
    pthread_mutex_lock(&j->mu);
    for (; started < n_threads - 1; ++started)
        if (hf_thread_create(&th[started], NULL, worker, j)) break;
    pthread_barrier_init(bar, NULL, (unsigned)started + 1);
    pthread_mutex_unlock(&j->mu);
    worker(j); /* the calling thread works too */
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (getenv("FREESASA_INGEST_TIMING"))
        fprintf(stderr, "ingest: %d inputs, %d threads, %lld atoms, %.3f ms\n", j->n, started + 1, (long long)out->n_atoms,
                1e3 * (double)(t1.tv_sec - t0.tv_sec) + 1e-6 * (double)(t1.tv_nsec - t0.tv_nsec));
    pthread_barrier_destroy(bar);
    pthread_mutex_destroy(&j->mu);
    free(th); free(j->slots); free(j->arena); free(j->a_off); free(j->r_off);
    const int rc = j->rc;
    if (rc) freesasa_ingest_free(out);
    return rc;
}

int freesasa_ingest_pdb_files(const char *const *paths, int n_paths, int options, int n_threads,
                              freesasa_ingest_batch *out)
{
    if (!out) return FREESASA_INGEST_EOPTION;
    job j;
    memset(&j, 0, sizeof j);
    j.paths = paths; j.n = n_paths; j.options = options;
    return run(&j, n_threads, out);
}

int freesasa_ingest_pdb_texts(const char *const *texts, const size_t *lens, int n_texts, int options,
                              int n_threads, freesasa_ingest_batch *out)
{
    if (!out) return FREESASA_INGEST_EOPTION;
    job j;
    memset(&j, 0, sizeof j);
    j.texts = texts; j.lens = lens; j.n = n_texts; j.options = options;
    return run(&j, n_threads, out);
}
