/*
 * gpu_parse.hip — PDB / mmCIF text -> (xyz, radius, class) ON THE DEVICE, for the file sweep (BASELINE configs[3]).
 *
 * Why: the sweep of files was bound by its host parser (round 5: 8.7e7 atoms/s from 16 threads against 4.8e8 for the
 * kernels on the same structures).  Here the host only READS the files' bytes into page-locked staging (and, for mmCIF,
 * finds the _atom_site loop's header, a line scan of the part of the file before it); the text goes over PCIe
 * (~130 bytes per atom) and kernels do what src/structure.c:644-722, src/pdb.c:176-283, src/cif.cc:113-240 and
 * src/classifier.c:781-796,1002-1017 do for one file on one core - as the host loader (ingest.c) restates them, function
 * for function; every rule below names the host function it mirrors, which names the reference lines.
 *
 *   kp_count_nl / kp_scan_blocks / kp_line_starts   where the lines are (a line = up to a '\n'; every file ends with one)
 *   kp_parse_lines   one thread per line: record type, hydrogen test, alt-loc label, coordinates, ProtOr radius and
 *                    class (binary search in the table of protor_table.h), element fallback; mmCIF: the row's tokens
 *   kp_resolve       one wave per file, its lines in order: first ENDMDL / lowest model, the alt-loc rule (a scan: wave
 *                    ballots), the first error, which atoms are kept and where they land
 *   kp_scatter       kept atoms to the batch arrays the tile kernels read
 *
 * What the device REFUSES goes to the host parser, file by file, and is counted: coordinates not of the "%8.3f" form
 * (PDB) or not plain decimals of <= 15 digits (mmCIF) - the host's strtod path -, lines longer than the reference's
 * 119-byte fgets chunk, RADIUS_FROM_OCCUPANCY, mmCIF that is not ONE data block with its _atom_site category in one
 * loop with one row per line (pair form, several blocks, text fields or names with blanks inside the rows).
 * Output: coordinates, radii, classes, atoms per file, status per file - what a sweep needs.  Residue boundaries and
 * labels are not built here (the sweep's totals and class sums do not use them; freesasa_ingest_* builds them).
 */
#include <hip/hip_runtime.h>

#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "engine_internal.h"
#include "gpu_parse.h"
#include "protor_table.h"

namespace {

/* per-line record bits */
enum {
    PL_CAND = 1,        /* an atom line that passed the record-type and hydrogen filters: takes part in the alt-loc rule */
    PL_ENDMDL = 2,      /* PDB: an ENDMDL line */
    PL_ERR_FORMAT = 4,  /* kept atom lines only: too short for coordinates (host: FREESASA_INGEST_EFORMAT) */
    PL_UNKNOWN = 8,     /* the classifier does not know the atom */
    PL_HOST = 16,       /* something only the host parser reads like the reference: the file is refused */
    PL_TERM = 32,       /* mmCIF: a line that ends the loop's rows (a tag, loop_, data_, save_) */
    PL_ROW = 64,        /* mmCIF: a row of the loop (its model number counts, whatever its group) */
    PL_DATA = 128       /* mmCIF: a data_ line */
};
#define PL_ALT_SHIFT 8

struct ParseArgs {
    const unsigned char *text; /* the batch's files, one after the other; every file ends with '\n'; padded with blanks to a multiple of 16 */
    unsigned T;                /* bytes of text (without the padding) */
    const ParseFile *files;    /* [F + 1]: files[F].beg = T */
    int F, options;
    /* lines */
    unsigned *blk_cnt;         /* [blocks + 1] newlines per 4096-byte block, then their exclusive prefix; [blocks] = lines */
    int n_blocks;
    unsigned *lstart;          /* [L + 1] first byte of every line; lstart[L] = T */
    int L;                     /* (host: after kp_scan_blocks) */
    unsigned *lflag;           /* [L] */
    int *lmodel;               /* [L] mmCIF: the row's model number */
    int *lpos;                 /* [L] kept atoms: place inside their file; else -1 */
    double *lx, *ly, *lz, *lr; /* [L] */
    unsigned char *lcls;       /* [L] */
    /* per file */
    int *fatoms, *fstatus, *fhost; /* [F] */
    long long *foff;           /* [F + 1] first atom of every file in the output (host-computed) */
    /* tables */
    const unsigned long long *pkey; const double *prad; const unsigned char *pcls; /* [PROTOR_N] sorted by key */
    const unsigned short *esym; const double *erad;                                /* [ELEMENT_N] */
    /* output */
    double *xyz, *radii; unsigned char *cls;
};

#define PB 256
#define PBLK 4096 /* bytes per workgroup of the line kernels: 16 per thread */

__device__ __forceinline__ bool is_sp(unsigned c) { return c == ' ' || (c - 9u) < 5u; } /* ingest.c is_sp: isspace of the C locale */
__device__ __forceinline__ bool cif_ws(unsigned c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; } /* ingest.c cif_ws */

__device__ __forceinline__ unsigned nl_mask16(const unsigned char *p) /* bit k: byte k of the 16 is '\n' */
{
    const uint4 v = *(const uint4 *)p;
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    unsigned m = 0;
    for (int k = 0; k < 4; ++k)
        for (int b = 0; b < 4; ++b)
            if (((w[k] >> (8 * b)) & 0xffu) == '\n') m |= 1u << (4 * k + b);
    return m;
}

__global__ __launch_bounds__(PB) void kp_count_nl(ParseArgs a)
{
    __shared__ unsigned part[PB / 64];
    const size_t base = (size_t)blockIdx.x * PBLK + (size_t)threadIdx.x * 16;
    unsigned c = base < a.T ? __popc(nl_mask16(a.text + base)) : 0u;
    for (int d = 1; d < 64; d <<= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) a.blk_cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

/* one workgroup: exclusive prefix of the blocks' newline counts, the total behind them */
__global__ __launch_bounds__(1024) void kp_scan_blocks(ParseArgs a)
{
    __shared__ unsigned wsum[16], carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < a.n_blocks; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        const unsigned v = b < a.n_blocks ? a.blk_cnt[b] : 0u;
        unsigned incl = v;
        for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d, 64); if ((int)(threadIdx.x & 63) >= d) incl += o; }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned before = carry;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += wsum[w];
        if (b < a.n_blocks) a.blk_cnt[b] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) a.blk_cnt[a.n_blocks] = carry;
}

__global__ __launch_bounds__(PB) void kp_line_starts(ParseArgs a)
{
    __shared__ unsigned wsum[PB / 64];
    const size_t base = (size_t)blockIdx.x * PBLK + (size_t)threadIdx.x * 16;
    const unsigned m = base < a.T ? nl_mask16(a.text + base) : 0u;
    const unsigned c = __popc(m);
    unsigned incl = c;
    for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d, 64); if ((int)(threadIdx.x & 63) >= d) incl += o; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned idx = a.blk_cnt[blockIdx.x] + incl - c;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) idx += wsum[w];
    if (blockIdx.x == 0 && threadIdx.x == 0) a.lstart[0] = 0;
    for (unsigned mm = m; mm; mm &= mm - 1) a.lstart[++idx] = (unsigned)(base + __ffs(mm)); /* the line behind newline number idx */
}

/* ---- the classifier (ingest.c protor_lookup / freesasa_ingest_guess_radius; tables: protor_table.h) */
__device__ double protor_radius(const ParseArgs &a, const unsigned char *rt, int rl, const unsigned char *at, int al, int *cls)
{
    *cls = 2; /* FREESASA_INGEST_UNKNOWN */
    if (rl < 1 || rl > 3 || al < 1 || al > 4) return -1.0;
    unsigned long long k = 0;
    for (int i = 0; i < 3; ++i) k = (k << 8) | (i < rl ? rt[i] : (unsigned char)' ');
    for (int i = 0; i < 4; ++i) k = (k << 8) | (i < al ? at[i] : (unsigned char)' ');
    int lo = 0, hi = PROTOR_N - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const unsigned long long v = a.pkey[mid];
        if (v == k) { *cls = a.pcls[mid]; return a.prad[mid]; }
        if (v < k) lo = mid + 1; else hi = mid - 1;
    }
    return -1.0;
}
/* sym: the symbol's characters (n of them, at most 2 are looked at): right-justified to two, compared as they are */
__device__ double element_radius(const ParseArgs &a, const unsigned char *sym, int n)
{
    const unsigned c0 = n >= 2 ? sym[0] : (unsigned)' ', c1 = n >= 2 ? sym[1] : (n == 1 ? sym[0] : (unsigned)' ');
    const unsigned short key = (unsigned short)(c0 | (c1 << 8));
    for (int i = 0; i < ELEMENT_N; ++i)
        if (a.esym[i] == key) return a.erad[i];
    return -1.0;
}
/* the first whitespace-delimited token of a field of w bytes (ingest.c field_token) */
__device__ __forceinline__ int field_token(const unsigned char *f, int w, int *start)
{
    int i = 0;
    while (i < w && is_sp(f[i])) ++i;
    *start = i;
    int n = 0;
    while (i + n < w && !is_sp(f[i + n])) ++n;
    return n;
}
/* three fields "%8.3f" (ingest.c coords_8_3): exact integer / 1000, one correctly rounded division */
__device__ bool coords_8_3(const unsigned char *sec, double v[3])
{
    for (int f = 0; f < 3; ++f) {
        const unsigned char *c = sec + 8 * f;
        const unsigned d3 = c[3] - '0', d5 = c[5] - '0', d6 = c[6] - '0', d7 = c[7] - '0';
        if (c[4] != '.' || d3 > 9 || d5 > 9 || d6 > 9 || d7 > 9 || !(c[0] == ' ' || c[0] == '-')) return false;
        unsigned m = d3, mul = 10;
        int i = 2;
        for (; i >= 0 && (unsigned)(c[i] - '0') <= 9; --i) { m += (unsigned)(c[i] - '0') * mul; mul *= 10; }
        bool neg = false;
        if (i >= 0 && c[i] == '-') { neg = true; --i; }
        for (; i >= 0; --i)
            if (c[i] != ' ') return false;
        const double x = (double)(m * 1000u + d5 * 100u + d6 * 10u + d7) / 1000.0;
        v[f] = neg ? -x : x;
    }
    return true;
}
/* [+-]digits[.digits] filling the token, at most 15 digits (ingest.c tok_plain_double) */
__device__ bool plain_double(const unsigned char *q, int n, double *out)
{
    const double p10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
    int i = 0;
    bool neg = false;
    if (i < n && (q[i] == '-' || q[i] == '+')) neg = q[i++] == '-';
    unsigned long long m = 0;
    int digits = 0, frac = 0;
    while (i < n && (unsigned)(q[i] - '0') <= 9) { m = m * 10 + (q[i++] - '0'); ++digits; }
    if (i < n && q[i] == '.') {
        ++i;
        while (i < n && (unsigned)(q[i] - '0') <= 9) { m = m * 10 + (q[i++] - '0'); ++digits; ++frac; }
    }
    if (i != n || digits < 1 || digits > 15) return false;
    const double v = (double)m / p10[frac];
    *out = neg ? -v : v;
    return true;
}
__device__ int tok_int(const unsigned char *p, int n) /* ingest.c tok_int */
{
    int i = 0;
    bool neg = false;
    if (i < n && (p[i] == '-' || p[i] == '+')) neg = p[i++] == '-';
    long long v = 0;
    while (i < n && (unsigned)(p[i] - '0') <= 9 && v < 100000000LL) v = v * 10 + (p[i++] - '0');
    return (int)(neg ? -v : v);
}
__device__ __forceinline__ unsigned lower(unsigned c) { return c - 'A' < 26u ? c + 32u : c; }
__device__ bool ieq(const unsigned char *p, const char *kw, int n)
{
    for (int i = 0; i < n; ++i)
        if (lower(p[i]) != (unsigned)(unsigned char)kw[i]) return false;
    return true;
}

__device__ void parse_pdb_line(const ParseArgs &a, const unsigned char *line, int n, unsigned &flag, double v[3], double &r, int &cls)
{
    /* ingest.c parse_pdb, the body of its line loop; n counts the newline like the reference's fgets buffer */
    const int options = a.options;
    if (n > 119) { flag = PL_HOST; return; } /* (the reference reads such a line in chunks of 119: host) */
    if (n < 4 || (line[0] != 'A' && line[0] != 'H' && line[0] != 'E')) return;
    for (int i = 0; i < n; ++i)
        if (line[i] == 0) { n = i; break; } /* an embedded NUL ends the line */
    const bool want_het = (options & FREESASA_INGEST_INCLUDE_HETATM) != 0, want_h = (options & FREESASA_INGEST_INCLUDE_HYDROGEN) != 0;
    const bool is_atom = n >= 4 && line[0] == 'A' && line[1] == 'T' && line[2] == 'O' && line[3] == 'M';
    const bool is_het = n >= 6 && line[0] == 'H' && line[1] == 'E' && line[2] == 'T' && line[3] == 'A' && line[4] == 'T' && line[5] == 'M';
    if (is_atom || (want_het && is_het)) {
        const bool has_sym = n >= 78;
        const unsigned s0 = has_sym ? line[76] : 0u, s1 = has_sym ? line[77] : 0u;
        int hyd;
        if (n < 13) hyd = -1;
        else if (has_sym && s0 == ' ' && (s1 == 'H' || s1 == 'D')) hyd = 1;
        else if (!(has_sym && s0 == ' ' && s1 == ' ')) hyd = 0;
        else if (!(line[12] == ' ' || (line[12] >= '1' && line[12] <= '9'))) hyd = 0;
        else hyd = (line[12] == 'H' || line[13] == 'H' || line[12] == 'D' || line[13] == 'D') ? 1 : 0;
        if (hyd && !want_h) return;
        const bool has_name = n >= 16;
        const unsigned alt = has_name ? line[16] : 0u;
        flag = PL_CAND | (alt << PL_ALT_SHIFT);
        if (n < 54) { flag |= PL_ERR_FORMAT; return; }
        if (!coords_8_3(line + 30, v)) { flag |= PL_HOST; return; } /* (the host's scan_double / strtod path) */
        if (options & FREESASA_INGEST_RADIUS_FROM_OCCUPANCY) { flag |= PL_HOST; return; }
        unsigned char symbol[2];
        int nsym = 0;
        if (has_sym) { symbol[0] = (unsigned char)s0; symbol[1] = (unsigned char)s1; nsym = 2; }
        if (!has_sym || (s0 == ' ' && s1 == ' ')) {
            if (has_name) { /* ingest.c guess_symbol */
                const unsigned char *nm = line + 12;
                if (nm[0] == ' ' || (nm[0] >= '1' && nm[0] <= '9')) { symbol[0] = ' '; symbol[1] = nm[1]; }
                else if (nm[3] == ' ') { symbol[0] = nm[0]; symbol[1] = nm[1]; }
                else { symbol[0] = ' '; symbol[1] = nm[0]; }
                nsym = 2;
            }
        }
        int as_, rs_;
        const int al = field_token(line + 12, has_name ? 4 : 0, &as_), rl = field_token(line + 17, n >= 20 ? 3 : 0, &rs_);
        const double rc = protor_radius(a, line + 17 + rs_, rl, line + 12 + as_, al, &cls);
        if (rc >= 0) {
            r = rc;
        } else {
            flag |= PL_UNKNOWN;
            r = element_radius(a, symbol, nsym);
            if (r < 0) r = +0.0;
        }
        return;
    }
    if (n >= 6 && line[0] == 'E' && line[1] == 'N' && line[2] == 'D' && line[3] == 'M' && line[4] == 'D' && line[5] == 'L') flag = PL_ENDMDL;
}

__device__ void parse_cif_line(const ParseArgs &a, const ParseFile &pf, const unsigned char *line, int n, unsigned &flag, int &model,
                               double v[3], double &r, int &cls)
{
    /* one line of the region behind the _atom_site loop's header: a row (ingest.c cif_visit_atom), a terminator, or nothing.
       n does not count the newline here. */
    int tp[12], tn[12];
    for (int k = 0; k < 12; ++k) { tp[k] = 0; tn[k] = 0; }
    int i = 0, col = 0;
    unsigned have = 0;
    while (i < n) {
        while (i < n && cif_ws(line[i])) ++i;
        if (i >= n) break;
        if (line[i] == '#') break; /* a comment: the rest of the line */
        const int s = i;
        if (line[i] == ';' && i == 0) { flag = PL_HOST; return; } /* a text field among the rows */
        if (line[i] == '\'' || line[i] == '"') {
            const unsigned q = line[i++];
            while (i < n) {
                if (line[i] == q && (i + 1 >= n || cif_ws(line[i + 1]))) { ++i; break; }
                ++i;
            }
        } else {
            while (i < n && !cif_ws(line[i])) ++i;
            const int len = i - s;
            const bool kw = line[s] == '_' || (len == 5 && ieq(line + s, "loop_", 5)) ||
                            (len >= 5 && line[s + 4] == '_' && (ieq(line + s, "data_", 5) || ieq(line + s, "save_", 5)));
            if (kw) {
                if (col == 0) { flag = PL_TERM | ((len >= 5 && line[s + 4] == '_' && ieq(line + s, "data_", 5)) ? PL_DATA : 0); }
                else flag = PL_HOST; /* values and a keyword on one line */
                return;
            }
        }
        if (col < 64) {
            for (int k = 0; k < 12; ++k)
                if (pf.slot[k] == col) { tp[k] = s; tn[k] = i - s; have |= 1u << k; }
        }
        ++col;
    }
    if (col == 0) return;                                  /* blank, or a comment */
    if (col != pf.ncol || have != 0xfffu) { flag = PL_HOST; return; } /* a row over several lines, or one cut short */
    flag = PL_ROW;
    model = tok_int(line + tp[11], tn[11]);
    const int options = a.options;
    const bool atom = tn[0] == 4 && line[tp[0]] == 'A' && line[tp[0] + 1] == 'T' && line[tp[0] + 2] == 'O' && line[tp[0] + 3] == 'M';
    if (!atom && !(options & FREESASA_INGEST_INCLUDE_HETATM)) return;
    if (!(options & FREESASA_INGEST_INCLUDE_HYDROGEN) && tn[7] == 1 && line[tp[7]] == 'H') return;
    const unsigned alt = line[tp[6]];
    flag |= PL_CAND | ((alt == '.' ? (unsigned)' ' : alt) << PL_ALT_SHIFT); /* '.' is the blank label here */
    /* names cut to the reference's field widths; a token with a quote keeps it unless it is the atom name's "..." */
    const unsigned char *an = line + tp[5];
    int al = tn[5];
    if (al >= 2 && an[0] == '"') { ++an; al -= 2; }
    if (al > 4) al = 4;
    int rl = tn[4] < 3 ? tn[4] : 3, sl = tn[7] < 2 ? tn[7] : 2;
    const unsigned char *rn = line + tp[4], *sy = line + tp[7];
    for (int k = 0; k < al; ++k) if (is_sp(an[k]) || an[k] == 0) { flag |= PL_HOST; return; } /* (ingest.c: not "plain") */
    for (int k = 0; k < rl; ++k) if (is_sp(rn[k]) || rn[k] == 0) { flag |= PL_HOST; return; }
    for (int k = 0; k < sl; ++k) if (is_sp(sy[k]) || sy[k] == 0) { flag |= PL_HOST; return; }
    for (int k = 0; k < 3; ++k)
        if (!plain_double(line + tp[8 + k], tn[8 + k], &v[k])) { flag |= PL_HOST; return; } /* (the host's strtod path) */
    const double rc = protor_radius(a, rn, rl, an, al, &cls);
    if (rc >= 0) {
        r = rc;
    } else {
        flag |= PL_UNKNOWN;
        r = element_radius(a, sy, sl);
        if (r < 0) r = +0.0;
    }
}

__global__ __launch_bounds__(PB) void kp_parse_lines(ParseArgs a)
{
    const int l = blockIdx.x * PB + threadIdx.x;
    if (l >= a.L) return;
    const unsigned s = a.lstart[l], e = a.lstart[l + 1] - 1; /* [s, e): the line without its newline */
    int lo = 0, hi = a.F; /* the file of the line: the last one that begins at or before s */
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.files[mid].beg <= s) lo = mid; else hi = mid; }
    const ParseFile pf = a.files[lo];
    unsigned flag = 0;
    int model = 0, cls = 2;
    double v[3] = {0, 0, 0}, r = 0;
    const unsigned char *line = a.text + s;
    const int len = (int)(e - s);
    if (pf.kind == PARSE_PDB) {
        /* (n counts the newline, as the reference's buffer does - except behind a last line that had none) */
        const bool last = a.lstart[l + 1] == a.files[lo + 1].beg;
        parse_pdb_line(a, line, len + ((last && pf.no_final_nl) ? 0 : 1), flag, v, r, cls);
    } else if (pf.kind == PARSE_CIF) {
        if (s >= pf.row0) parse_cif_line(a, pf, line, len, flag, model, v, r, cls);
    }
    a.lflag[l] = flag;
    a.lmodel[l] = model;
    a.lx[l] = v[0]; a.ly[l] = v[1]; a.lz[l] = v[2]; a.lr[l] = r;
    a.lcls[l] = (unsigned char)cls;
}

/* One wave per file, the file's lines in order (ingest.c parse_pdb's loop state / cif_visit_atom's): which atoms are kept. */
__global__ __launch_bounds__(64) void kp_resolve(ParseArgs a)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    const ParseFile pf = a.files[f];
    if (pf.kind == PARSE_HOST) { if (lane == 0) { a.fatoms[f] = 0; a.fstatus[f] = 0; a.fhost[f] = 1; } return; }
    /* the file's lines: [l0, l1) */
    int l0, l1;
    {
        const unsigned b = pf.beg, e = a.files[f + 1].beg;
        int lo = 0, hi = a.L; /* first line that starts at or behind b */
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.lstart[mid] < b) lo = mid + 1; else hi = mid; }
        l0 = lo;
        lo = l0; hi = a.L;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.lstart[mid] < e) lo = mid + 1; else hi = mid; }
        l1 = lo;
    }
    const bool cif = pf.kind == PARSE_CIF, join = (a.options & FREESASA_INGEST_JOIN_MODELS) != 0;
    /* pass A: where the atoms end (first ENDMDL / the loop's terminator), the lowest model, refusals */
    int end = l1, min_model = 0x7fffffff, host = 0, later_data = 0;
    for (int l = l0 + lane; l - lane < l1; l += 64) {
        const unsigned fl = l < l1 ? a.lflag[l] : 0u;
        if (cif ? (fl & PL_TERM) != 0 : ((fl & PL_ENDMDL) != 0 && !join)) end = end < l ? end : l;
    }
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_xor(end, d, 64); end = end < o ? end : o; }
    for (int l = l0 + lane; l - lane < l1; l += 64) {
        const unsigned fl = l < l1 ? a.lflag[l] : 0u;
        if (l < end) {
            if (fl & PL_HOST) host = 1;
            if ((fl & PL_ROW) && a.lmodel[l] < min_model) min_model = a.lmodel[l];
        } else if (cif && (fl & PL_DATA)) {
            later_data = 1; /* a second data block: its atoms would count too (ingest.c cif_walk): the host reads such a file */
        }
    }
    host |= later_data;
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_xor(min_model, d, 64); min_model = min_model < o ? min_model : o;
        host |= __shfl_xor(host, d, 64);
    }
    if (host) { if (lane == 0) { a.fatoms[f] = 0; a.fstatus[f] = 0; a.fhost[f] = 1; } return; }
    /* pass B: the alt-loc rule in order, the first error, the places of the kept atoms */
    const bool skip_unknown_pdb = (a.options & FREESASA_INGEST_SKIP_UNKNOWN) && !(a.options & FREESASA_INGEST_HALT_AT_UNKNOWN);
    const bool halt_pdb = (a.options & FREESASA_INGEST_HALT_AT_UNKNOWN) != 0;
    const bool drop_unknown_cif = (a.options & (FREESASA_INGEST_SKIP_UNKNOWN | FREESASA_INGEST_HALT_AT_UNKNOWN)) != 0;
    unsigned the_alt = ' '; /* the label in force (blank: none) */
    int kept = 0, first_err = 0x7fffffff, err_code = 0;
    for (int lb = l0; lb < end; lb += 64) {
        const int l = lb + lane;
        const unsigned fl = l < end ? a.lflag[l] : 0u;
        bool cand = (fl & PL_CAND) != 0;
        if (cif && cand && !join && a.lmodel[l] != min_model) cand = false;
        const unsigned alt = (fl >> PL_ALT_SHIFT) & 0xffu;
        /* the_alt after a candidate line: blank label -> blank; a label while none is in force -> that label; else unchanged.
           For lane i the label in force BEFORE it is decided by the last blank candidate before it (in this chunk, or the
           state carried in): the first labelled candidate behind that one set the label. */
        const unsigned long long mc = __ballot(cand), mb = __ballot(cand && alt == ' ');
        const unsigned long long below = lane ? (~0ULL >> (64 - lane)) : 0ULL;
        const unsigned long long bb = mb & below;
        const int last_blank = bb ? 63 - __clzll(bb) : -1;
        /* labelled candidates behind the last blank one and before this lane (this lane included if labelled) */
        const unsigned long long labelled = mc & ~mb;
        const unsigned long long run = labelled & (last_blank >= 0 ? ~(~0ULL >> (63 - last_blank)) : ~0ULL) & (below | (1ULL << lane));
        /* the label in force when this lane's line is judged (after its own effect if it sets it) */
        const unsigned first_label = (unsigned)__shfl((int)alt, run ? __ffsll((long long)run) - 1 : lane, 64); /* (every lane shuffles: no divergence around it) */
        const unsigned in_force = (last_blank < 0 && the_alt != ' ') ? the_alt : (run ? first_label : (unsigned)' ');
        const bool alt_ok = cand && (alt == ' ' || alt == in_force);
        /* the state the next chunk starts from */
        unsigned next_alt = the_alt;
        if (mc) {
            const int last_c = 63 - __clzll(mc);
            const unsigned a_last = __shfl((int)alt, last_c, 64), f_last = __shfl((int)in_force, last_c, 64);
            next_alt = a_last == ' ' ? ' ' : f_last;
        }
        the_alt = next_alt;
        /* behind the alt rule: a line too short for coordinates fails the file; an unknown atom halts, is dropped or guessed */
        int err = 0;
        bool keep = alt_ok;
        if (alt_ok) {
            if (fl & PL_ERR_FORMAT) err = FREESASA_INGEST_EFORMAT;
            else if (fl & PL_UNKNOWN) {
                if (cif) { if (drop_unknown_cif) keep = false; }
                else if (halt_pdb) err = FREESASA_INGEST_EUNKNOWN;
                else if (skip_unknown_pdb) keep = false;
            }
        }
        if (err) keep = false;
        const unsigned long long me = __ballot(err != 0);
        if (me && first_err == 0x7fffffff) {
            const int le = __ffsll((long long)me) - 1;
            first_err = lb + le;
            err_code = __shfl(err, le, 64);
        }
        const unsigned long long mk = __ballot(keep);
        if (l < end) a.lpos[l] = keep ? kept + __popcll(mk & below) : -1;
        kept += __popcll(mk);
    }
    for (int l = (end > l0 ? end : l0) + lane; l < l1; l += 64) a.lpos[l] = -1;
    if (lane == 0) {
        a.fhost[f] = 0;
        if (first_err != 0x7fffffff) { a.fatoms[f] = 0; a.fstatus[f] = err_code; }
        else if (kept == 0) { a.fatoms[f] = 0; a.fstatus[f] = FREESASA_INGEST_EEMPTY; }
        else { a.fatoms[f] = kept; a.fstatus[f] = 0; }
    }
}

__global__ __launch_bounds__(PB) void kp_scatter(ParseArgs a)
{
    const int l = blockIdx.x * PB + threadIdx.x;
    if (l >= a.L) return;
    const int p = a.lpos[l];
    if (p < 0) return;
    const unsigned s = a.lstart[l];
    int lo = 0, hi = a.F;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.files[mid].beg <= s) lo = mid; else hi = mid; }
    if (a.fatoms[lo] == 0) return; /* (a file that failed keeps none of its atoms) */
    const long long o = a.foff[lo] + p;
    a.xyz[3 * o] = a.lx[l]; a.xyz[3 * o + 1] = a.ly[l]; a.xyz[3 * o + 2] = a.lz[l];
    a.radii[o] = a.lr[l];
    a.cls[o] = a.lcls[l];
}

/* the classifier's tables on the device, once per device */
struct Tables { void *p = nullptr; };
Tables g_tables[64];
std::mutex g_tab_mu;

} /* namespace */

static int tables_for(freesasa_gpu_ctx *c, ParseArgs &pa)
{
    const size_t b_key = 8 * PROTOR_N, b_rad = 8 * PROTOR_N, b_cls = (PROTOR_N + 7) & ~7, b_es = (2 * ELEMENT_N + 7) & ~7, b_er = 8 * ELEMENT_N;
    std::lock_guard<std::mutex> lk(g_tab_mu);
    const int d = c->device >= 0 && c->device < 64 ? c->device : 0;
    if (!g_tables[d].p) {
        std::vector<unsigned char> h(b_key + b_rad + b_cls + b_es + b_er, 0);
        unsigned long long *k = (unsigned long long *)h.data();
        double *r = (double *)(h.data() + b_key);
        unsigned char *cl = h.data() + b_key + b_rad;
        unsigned short *es = (unsigned short *)(h.data() + b_key + b_rad + b_cls);
        double *er = (double *)(h.data() + b_key + b_rad + b_cls + b_es);
        for (int i = 0; i < PROTOR_N; ++i) { k[i] = protor_table[i].key; r[i] = protor_table[i].radius; cl[i] = (unsigned char)protor_table[i].cls; }
        for (int i = 0; i < ELEMENT_N; ++i) { es[i] = (unsigned short)((unsigned char)element_table[i].sym[0] | ((unsigned)(unsigned char)element_table[i].sym[1] << 8)); er[i] = element_table[i].radius; }
        void *p = nullptr;
        if (dev_malloc(&p, h.size()) != hipSuccess) return ctx_fail(c, "out of device memory (classifier tables)");
        if (hipMemcpy(p, h.data(), h.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(p); return ctx_fail(c, "upload of the classifier tables failed"); }
        g_tables[d].p = p;
    }
    char *p = (char *)g_tables[d].p;
    pa.pkey = (const unsigned long long *)p; pa.prad = (const double *)(p + b_key); pa.pcls = (const unsigned char *)(p + b_key + b_rad);
    pa.esym = (const unsigned short *)(p + b_key + b_rad + b_cls); pa.erad = (const double *)(p + b_key + b_rad + b_cls + b_es);
    return 0;
}

/* (gpu_parse.h) */
int parse_batch_dev_begin(freesasa_gpu_ctx *c, unsigned char *h_text, size_t T, const ParseFile *files, int F, int options,
                          int *atoms_out, int *status_out, int *host_out, long long *total_atoms_out)
{
    if (T >= (1ULL << 31)) return ctx_fail(c, "batch text too large for the device parser");
    hipStream_t st = c->stream;
    ParseArgs a;
    memset(&a, 0, sizeof a);
    c->parse_lines = 0; c->parse_atoms = 0;
    if (tables_for(c, a)) return -1;
    const size_t Tp = (T + 15) & ~(size_t)15;
    for (size_t k = T; k < Tp; ++k) h_text[k] = ' ';
    a.T = (unsigned)T; a.F = F; a.options = options;
    a.n_blocks = (int)((Tp + PBLK - 1) / PBLK);
    DevBuf *B = c->parse;
    if (ensure(c, B[0], Tp + 16) || ensure(c, B[1], sizeof(ParseFile) * ((size_t)F + 1)) || ensure(c, B[2], 4 * ((size_t)a.n_blocks + 2)) ||
        ensure(c, B[3], 4 * 3 * (size_t)F + 16) || ensure(c, B[4], 8 * ((size_t)F + 1)))
        return -1;
    a.text = (const unsigned char *)B[0].p; a.files = (const ParseFile *)B[1].p; a.blk_cnt = (unsigned *)B[2].p;
    a.fatoms = (int *)B[3].p; a.fstatus = a.fatoms + F; a.fhost = a.fstatus + F; a.foff = (long long *)B[4].p;
    HIP_TRY(c, hipMemcpyAsync(B[0].p, h_text, Tp, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(B[1].p, files, sizeof(ParseFile) * ((size_t)F + 1), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(kp_count_nl, dim3(a.n_blocks), dim3(PB), 0, st, a);
    hipLaunchKernelGGL(kp_scan_blocks, dim3(1), dim3(1024), 0, st, a);
    int *words = c->pinned + 2 * (sasa::ST_WORDS + 4); /* (four ints behind the two status sets: gpu_engine.hip) */
    HIP_TRY(c, hipMemcpyAsync(words, a.blk_cnt + a.n_blocks, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    a.L = words[0];
    *total_atoms_out = 0;
    if (a.L <= 0) { /* no line at all */
        for (int f = 0; f < F; ++f) { atoms_out[f] = 0; status_out[f] = files[f].kind == PARSE_HOST ? 0 : FREESASA_INGEST_EEMPTY; host_out[f] = files[f].kind == PARSE_HOST; }
        return 0;
    }
    const size_t L = (size_t)a.L;
    if (ensure(c, B[5], 4 * (L + 2)) || ensure(c, B[6], 4 * L) || ensure(c, B[7], 4 * L) || ensure(c, B[8], 4 * L) ||
        ensure(c, B[9], 8 * 4 * L) || ensure(c, B[10], L))
        return -1;
    a.lstart = (unsigned *)B[5].p; a.lflag = (unsigned *)B[6].p; a.lmodel = (int *)B[7].p; a.lpos = (int *)B[8].p;
    a.lx = (double *)B[9].p; a.ly = a.lx + L; a.lz = a.ly + L; a.lr = a.lz + L; a.lcls = (unsigned char *)B[10].p;
    const int lblocks = (int)((L + PB - 1) / PB);
    hipLaunchKernelGGL(kp_line_starts, dim3(a.n_blocks), dim3(PB), 0, st, a);
    hipLaunchKernelGGL(kp_parse_lines, dim3(lblocks), dim3(PB), 0, st, a);
    hipLaunchKernelGGL(kp_resolve, dim3(F), dim3(64), 0, st, a);
    HIP_TRY(c, hipGetLastError());
    /* atoms / status / refused per file -> host */
    std::vector<int> h((size_t)3 * F);
    HIP_TRY(c, hipMemcpyAsync(h.data(), a.fatoms, 4 * 3 * (size_t)F, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    c->parse_off.resize((size_t)F + 1);
    long long run = 0;
    for (int f = 0; f < F; ++f) {
        atoms_out[f] = h[f]; status_out[f] = h[(size_t)F + f]; host_out[f] = h[2 * (size_t)F + f];
        c->parse_off[f] = run; run += h[f];
    }
    c->parse_off[F] = run;
    *total_atoms_out = run;
    c->parse_lines = a.L; c->parse_atoms = run; c->parse_files = F; c->parse_options = options; c->parse_T = (unsigned)T;
    return 0;
}

int parse_batch_dev_finish(freesasa_gpu_ctx *c, long long extra_atoms)
{
    const long long run = c->parse_atoms, cap = run + (extra_atoms > 0 ? extra_atoms : 0);
    if (cap <= 0) return 0;
    if (ensure(c, c->h_xyz, 24 * (size_t)cap) || ensure(c, c->h_radii, 8 * (size_t)cap) || ensure(c, c->h_counts, (size_t)cap)) return -1;
    if (run == 0) return 0;
    hipStream_t st = c->stream;
    ParseArgs a;
    memset(&a, 0, sizeof a);
    DevBuf *B = c->parse;
    const size_t L = (size_t)c->parse_lines;
    const int F = c->parse_files;
    a.T = c->parse_T; a.F = F; a.options = c->parse_options; a.L = (int)L;
    a.text = (const unsigned char *)B[0].p; a.files = (const ParseFile *)B[1].p;
    a.fatoms = (int *)B[3].p; a.fstatus = a.fatoms + F; a.fhost = a.fstatus + F; a.foff = (long long *)B[4].p;
    a.lstart = (unsigned *)B[5].p; a.lflag = (unsigned *)B[6].p; a.lmodel = (int *)B[7].p; a.lpos = (int *)B[8].p;
    a.lx = (double *)B[9].p; a.ly = a.lx + L; a.lz = a.ly + L; a.lr = a.lz + L; a.lcls = (unsigned char *)B[10].p;
    a.xyz = (double *)c->h_xyz.p; a.radii = (double *)c->h_radii.p; a.cls = (unsigned char *)c->h_counts.p;
    HIP_TRY(c, hipMemcpyAsync(B[4].p, c->parse_off.data(), 8 * ((size_t)F + 1), hipMemcpyHostToDevice, st)); /* (parse_off lives in the context: the copy may run later) */
    hipLaunchKernelGGL(kp_scatter, dim3((unsigned)((L + PB - 1) / PB)), dim3(PB), 0, st, a);
    HIP_TRY(c, hipGetLastError());
    return 0;
}
