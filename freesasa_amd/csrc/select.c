/* select.c — the reference's selection language on a loaded batch.
 *
 *     "name, expr"    expr := resn LIST | resi RANGES | symbol LIST | name LIST | chain RANGES
 *                             | expr and expr | expr or expr | not expr | ( expr )
 *
 * Restates the reference's lexer (src/lexer.l: longest match, first rule wins ties, unknown
 * characters are skipped), grammar (src/parser.y: or < and < not; '+' joins list items; '-' makes
 * ranges, open on either side for residue numbers; "\-" is a minus sign) and evaluation
 * (src/selection.c:283-660: identifiers are upper-cased, names/residue names/numbers/symbols are
 * compared with the structure's trimmed fields, ranges use atoi of the residue number or the
 * chain character, invalid identifiers are ignored with a warning).  The result is a byte mask over
 * the atoms of one structure of a freesasa_ingest_batch; the area of a selection is the masked sum
 * of per-atom SASA (freesasa_gpu_class_sums_dev with the mask as class gives it on the device).
 * Host code only.
 */
#include "freesasa_ingest.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hostfault.h"

/* ------------------------------------------------------------------ lexer */

enum { K_END, K_COMMA, K_DASH, K_PLUS, K_LPAR, K_RPAR, K_RESN, K_RESI, K_SYMBOL, K_NAME, K_CHAIN, K_AND, K_OR, K_NOT,
       K_MINUS, K_NUMBER, K_ID, K_SELID };
typedef struct { int type; char text[64]; } tok_t;
typedef struct { const char *s; size_t pos, len; } lex_t;

static size_t match_ci(const char *p, size_t left, const char *word)
{
    const size_t n = strlen(word);
    if (left < n) return 0;
    for (size_t i = 0; i < n; ++i)
        if (tolower((unsigned char)p[i]) != word[i]) return 0;
    return n;
}
static int is_idc(int c) { return isalnum(c) || c == '_'; }

/* the next token by flex's rules: among all patterns the longest match, the earlier rule on a tie */
static tok_t next_token(lex_t *lx)
{
    tok_t t;
    t.text[0] = '\0';
    for (;;) {
        if (lx->pos >= lx->len) { t.type = K_END; return t; }
        const char *p = lx->s + lx->pos;
        const size_t left = lx->len - lx->pos;
        size_t best = 0, take = 0;
        int type = -1;
#define CAND(n_, ty_) do { const size_t n__ = (n_); if (n__ > best) { best = n__; take = n__; type = (ty_); } } while (0)
        CAND(p[0] == ',', K_COMMA);
        CAND(p[0] == '-', K_DASH);
        CAND(p[0] == '+', K_PLUS);
        CAND(p[0] == '(', K_LPAR);
        CAND(p[0] == ')', K_RPAR);
        CAND(match_ci(p, left, "resn"), K_RESN);
        CAND(match_ci(p, left, "resi"), K_RESI);
        CAND(match_ci(p, left, "symbol"), K_SYMBOL);
        CAND(match_ci(p, left, "name"), K_NAME);
        CAND(match_ci(p, left, "chain"), K_CHAIN);
        CAND(p[0] == '&' ? 1 : match_ci(p, left, "and"), K_AND);
        CAND(p[0] == '|' ? 1 : match_ci(p, left, "or"), K_OR);
        CAND(p[0] == '!' ? 1 : match_ci(p, left, "not"), K_NOT);
        CAND(left >= 2 && p[0] == '\\' && p[1] == '-' ? 2 : 0, K_MINUS);
        size_t ws = 0;
        while (ws < left && (p[ws] == ' ' || p[ws] == '\t' || p[ws] == '\n' || p[ws] == '\r')) ++ws;
        CAND(ws, -2);
        size_t nd = 0;
        while (nd < left && isdigit((unsigned char)p[nd])) ++nd;
        CAND(nd, K_NUMBER);
        size_t ni = 0;
        while (ni < left && is_idc((unsigned char)p[ni])) ++ni;
        if (ni) while (ni < left && p[ni] == '\'') ++ni;
        CAND(ni, K_ID);
        size_t nsel = 0; /* [[:alnum:]_\-\+]+ followed by ',': the comma counts for the length, not for the text */
        while (nsel < left && (is_idc((unsigned char)p[nsel]) || p[nsel] == '-' || p[nsel] == '+')) ++nsel;
        if (nsel && nsel < left && p[nsel] == ',' && nsel + 1 > best) { best = nsel + 1; take = nsel; type = K_SELID; }
#undef CAND
        if (type == -1) { lx->pos += 1; continue; } /* flex's default rule: the character is echoed and dropped */
        lx->pos += take;
        if (type == -2) continue;
        t.type = type;
        const size_t n = take < sizeof t.text - 1 ? take : sizeof t.text - 1;
        memcpy(t.text, p, n);
        t.text[n] = '\0';
        return t;
    }
}

/* ------------------------------------------------------------------ parser */

enum { E_AND, E_OR, E_NOT, E_RESN, E_RESI, E_SYMBOL, E_NAME, E_CHAIN, E_PLUS, E_RANGE, E_RANGE_OPEN_L, E_RANGE_OPEN_R,
       E_NUMBER, E_ID };
typedef struct expr { int type; struct expr *l, *r; char value[64]; } expr;

typedef struct { lex_t lx; tok_t cur; int err; } parser;

static void advance(parser *ps) { ps->cur = next_token(&ps->lx); }
static expr *node(int type, expr *l, expr *r)
{
    expr *e = hf_calloc(1, sizeof *e);
    if (e) { e->type = type; e->l = l; e->r = r; }
    return e;
}
static void free_expr(expr *e)
{
    if (!e) return;
    free_expr(e->l); free_expr(e->r); free(e);
}

/* id: NUMBER | ID | "\-" NUMBER; values are upper-cased (ref: src/selection.c:100-131) */
static expr *parse_id(parser *ps)
{
    expr *e = NULL;
    if (ps->cur.type == K_NUMBER || ps->cur.type == K_ID) {
        e = node(ps->cur.type == K_NUMBER ? E_NUMBER : E_ID, NULL, NULL);
        if (e) snprintf(e->value, sizeof e->value, "%s", ps->cur.text);
        advance(ps);
    } else if (ps->cur.type == K_MINUS) {
        advance(ps);
        if (ps->cur.type != K_NUMBER) { ps->err = 1; return NULL; }
        e = node(E_NUMBER, NULL, NULL);
        if (e) snprintf(e->value, sizeof e->value, "-%.60s", ps->cur.text);
        advance(ps);
    } else {
        ps->err = 1;
        return NULL;
    }
    if (e) for (char *c = e->value; *c; ++c) *c = (char)toupper((unsigned char)*c);
    else ps->err = 1;
    return e;
}
static int starts_id(const parser *ps) { return ps->cur.type == K_NUMBER || ps->cur.type == K_ID || ps->cur.type == K_MINUS; }

static expr *parse_list(parser *ps) /* id ('+' id)*, right-nested like the grammar */
{
    expr *left = parse_id(ps);
    if (!left) return NULL;
    if (ps->cur.type == K_PLUS) {
        advance(ps);
        expr *right = parse_list(ps);
        if (!right) { free_expr(left); return NULL; }
        return node(E_PLUS, left, right);
    }
    return left;
}

/* one item of a range list: id | id '-' id | '-' id | id '-' (the open forms only for residues) */
static expr *parse_range_item(parser *ps, int allow_open)
{
    if (ps->cur.type == K_DASH) {
        if (!allow_open) { ps->err = 1; return NULL; }
        advance(ps);
        expr *right = parse_id(ps);
        return right ? node(E_RANGE_OPEN_L, NULL, right) : NULL;
    }
    expr *left = parse_id(ps);
    if (!left) return NULL;
    if (ps->cur.type == K_DASH) {
        advance(ps);
        if (starts_id(ps)) {
            expr *right = parse_id(ps);
            if (!right) { free_expr(left); return NULL; }
            return node(E_RANGE, left, right);
        }
        if (!allow_open) { ps->err = 1; free_expr(left); return NULL; }
        return node(E_RANGE_OPEN_R, left, NULL);
    }
    return left;
}
static expr *parse_ranges(parser *ps, int allow_open) /* item ('+' item)*, left-associative */
{
    expr *left = parse_range_item(ps, allow_open);
    while (left && ps->cur.type == K_PLUS) {
        advance(ps);
        expr *right = parse_range_item(ps, allow_open);
        if (!right) { free_expr(left); return NULL; }
        left = node(E_PLUS, left, right);
    }
    /* "1-2-3", "1-2-": a second dash after a complete item is a syntax error in the grammar */
    if (left && ps->cur.type == K_DASH) { ps->err = 1; free_expr(left); return NULL; }
    return left;
}

static expr *parse_or(parser *ps);
static expr *parse_primary(parser *ps)
{
    switch (ps->cur.type) {
    case K_LPAR: {
        advance(ps);
        expr *e = parse_or(ps);
        if (!e) return NULL;
        if (ps->cur.type != K_RPAR) { ps->err = 1; free_expr(e); return NULL; }
        advance(ps);
        return e;
    }
    case K_NOT: {
        advance(ps);
        expr *e = parse_primary(ps); /* not binds tighter than and / or */
        return e ? node(E_NOT, NULL, e) : NULL;
    }
    case K_RESN: { advance(ps); expr *l = parse_list(ps); return l ? node(E_RESN, l, NULL) : NULL; }
    case K_SYMBOL: { advance(ps); expr *l = parse_list(ps); return l ? node(E_SYMBOL, l, NULL) : NULL; }
    case K_NAME: { advance(ps); expr *l = parse_list(ps); return l ? node(E_NAME, l, NULL) : NULL; }
    case K_RESI: { advance(ps); expr *l = parse_ranges(ps, 1); return l ? node(E_RESI, l, NULL) : NULL; }
    case K_CHAIN: { advance(ps); expr *l = parse_ranges(ps, 0); return l ? node(E_CHAIN, l, NULL) : NULL; }
    default:
        ps->err = 1;
        return NULL;
    }
}
static expr *parse_and(parser *ps)
{
    expr *left = parse_primary(ps);
    while (left && ps->cur.type == K_AND) {
        advance(ps);
        expr *right = parse_primary(ps);
        if (!right) { free_expr(left); return NULL; }
        left = node(E_AND, left, right);
    }
    return left;
}
static expr *parse_or(parser *ps)
{
    expr *left = parse_and(ps);
    while (left && ps->cur.type == K_OR) {
        advance(ps);
        expr *right = parse_and(ps);
        if (!right) { free_expr(left); return NULL; }
        left = node(E_OR, left, right);
    }
    return left;
}

/* ------------------------------------------------------------------ evaluation */

typedef struct {
    const freesasa_ingest_batch *b;
    int64_t a0, n;     /* atoms of the structure */
    const int64_t *res; /* residue index of every atom of the structure */
    int warn;
} ctx_t;

static void trimmed(char *dst, size_t cap, const char *src, size_t w)
{
    size_t i = 0, n = 0;
    while (i < w && src[i] && isspace((unsigned char)src[i])) ++i;
    while (i < w && src[i] && !isspace((unsigned char)src[i]) && n + 1 < cap) dst[n++] = src[i++];
    dst[n] = '\0';
}
static void atom_field(const ctx_t *c, int64_t i, int parent, char *out, size_t cap)
{
    const freesasa_ingest_batch *b = c->b;
    const int64_t a = c->a0 + i, r = c->res[i];
    switch (parent) {
    case E_NAME: trimmed(out, cap, b->atom_name + 4 * a, 4); break;
    case E_SYMBOL: trimmed(out, cap, b->atom_symbol + 2 * a, 2); break;
    case E_RESN: trimmed(out, cap, b->res_name + 4 * r, 4); break;
    default: trimmed(out, cap, b->res_number + 6 * r, 6); break; /* E_RESI */
    }
}
static int resnum(const ctx_t *c, int64_t i) /* atoi of the residue number field */
{
    char buf[8];
    memcpy(buf, c->b->res_number + 6 * c->res[i], 6);
    buf[6] = '\0';
    return atoi(buf);
}

/* ref: src/selection.c:376-452 */
static int valid_id(int parent, const expr *e)
{
    const char *v = e->value;
    const size_t n = strlen(v);
    switch (parent) {
    case E_NAME: return n <= 4;
    case E_SYMBOL: return e->type == E_ID && n <= 2;
    case E_RESN: return n <= 3;
    case E_RESI:
        if (e->type == E_NUMBER) return 1;
        if (n > 5 || n == 1) return 0;
        if (toupper((unsigned char)v[n - 1]) < 'A' || toupper((unsigned char)v[n - 1]) > 'Z') return 0;
        for (size_t i = 0; i + 1 < n; ++i)
            if (v[i] < '0' || v[i] > '9') return 0;
        return 1;
    default: return n <= 1; /* E_CHAIN */
    }
}

static void select_id(ctx_t *c, int parent, const char *id, unsigned char *mask)
{
    char f[16];
    for (int64_t i = 0; i < c->n; ++i) {
        int m;
        if (parent == E_CHAIN) m = id[0] == c->b->res_chain[4 * c->res[i]];
        else { atom_field(c, i, parent, f, sizeof f); m = strcmp(f, id) == 0; }
        if (m) mask[i] = 1;
    }
}

/* ref: src/selection.c:454-505 */
static void select_range(ctx_t *c, int type, int parent, const expr *l, const expr *r, unsigned char *mask)
{
    if (parent == E_RESI) {
        if ((l && l->type != E_NUMBER) || (r && r->type != E_NUMBER)) { c->warn = 1; return; }
    } else {
        if (l->type != r->type || (l->type == E_ID && (strlen(l->value) > 1 || strlen(r->value) > 1))) { c->warn = 1; return; }
    }
    int lower, upper;
    if (type == E_RANGE_OPEN_L) { lower = resnum(c, 0); upper = atoi(r->value); }
    else if (type == E_RANGE_OPEN_R) { lower = atoi(l->value); upper = resnum(c, c->n - 1); }
    else if (l->type == E_NUMBER) { lower = atoi(l->value); upper = atoi(r->value); }
    else { lower = (int)l->value[0]; upper = (int)r->value[0]; }
    for (int64_t i = 0; i < c->n; ++i) {
        const int j = parent == E_RESI ? resnum(c, i) : (int)c->b->res_chain[4 * c->res[i]];
        if (j >= lower && j <= upper) mask[i] = 1;
    }
}

static void select_list(ctx_t *c, int parent, const expr *e, unsigned char *mask)
{
    switch (e->type) {
    case E_PLUS: select_list(c, parent, e->l, mask); select_list(c, parent, e->r, mask); break;
    case E_RANGE: case E_RANGE_OPEN_L: case E_RANGE_OPEN_R: select_range(c, e->type, parent, e->l, e->r, mask); break;
    default: /* E_ID, E_NUMBER */
        if (valid_id(parent, e)) select_id(c, parent, e->value, mask);
        else c->warn = 1;
    }
}

static int eval(ctx_t *c, const expr *e, unsigned char *mask)
{
    memset(mask, 0, (size_t)c->n);
    switch (e->type) {
    case E_RESN: case E_RESI: case E_SYMBOL: case E_NAME: case E_CHAIN:
        select_list(c, e->type, e->l, mask);
        return 0;
    case E_NOT:
        if (eval(c, e->r, mask)) return -1;
        for (int64_t i = 0; i < c->n; ++i) mask[i] = !mask[i];
        return 0;
    default: { /* E_AND, E_OR */
        unsigned char *tmp = hf_malloc((size_t)(c->n ? c->n : 1));
        if (!tmp) return -1;
        int rc = eval(c, e->l, mask) || eval(c, e->r, tmp);
        if (!rc)
            for (int64_t i = 0; i < c->n; ++i) mask[i] = e->type == E_AND ? (mask[i] && tmp[i]) : (mask[i] || tmp[i]);
        free(tmp);
        return rc ? -1 : 0;
    }
    }
}

int freesasa_ingest_select(const freesasa_ingest_batch *b, int structure, const char *command,
                           char name_out[FREESASA_INGEST_MAX_SELECTION_NAME + 1], unsigned char *mask_out)
{
    if (!b || !command || !name_out || !mask_out || structure < 0 || structure >= b->n_structs) return FREESASA_INGEST_SELECT_FAIL;
    name_out[0] = '\0';
    parser ps;
    memset(&ps, 0, sizeof ps);
    ps.lx.s = command; ps.lx.len = strlen(command);
    advance(&ps);
    if (ps.cur.type != K_SELID) return FREESASA_INGEST_SELECT_FAIL;
    char name[64];
    snprintf(name, sizeof name, "%s", ps.cur.text);
    advance(&ps);
    if (ps.cur.type != K_COMMA) return FREESASA_INGEST_SELECT_FAIL;
    advance(&ps);
    expr *e = parse_or(&ps);
    if (!e || ps.err || ps.cur.type != K_END) { free_expr(e); return FREESASA_INGEST_SELECT_FAIL; }

    ctx_t c;
    c.b = b; c.a0 = b->offsets[structure]; c.n = b->offsets[structure + 1] - c.a0; c.warn = 0;
    if (c.n == 0) { /* an input that failed to load: nothing to select from */
        free_expr(e);
        snprintf(name_out, FREESASA_INGEST_MAX_SELECTION_NAME + 1, "%.50s", name);
        return 0;
    }
    int64_t *res = hf_malloc(sizeof(int64_t) * (size_t)c.n);
    if (!res) { free_expr(e); return FREESASA_INGEST_SELECT_FAIL; }
    for (int64_t r = b->res_offsets[structure]; r < b->res_offsets[structure + 1]; ++r)
        for (int64_t a = b->res_first[r]; a < b->res_first[r + 1]; ++a) res[a - c.a0] = r;
    c.res = res;
    const int rc = eval(&c, e, mask_out);
    free(res);
    free_expr(e);
    if (rc) return FREESASA_INGEST_SELECT_FAIL;
    snprintf(name_out, FREESASA_INGEST_MAX_SELECTION_NAME + 1, "%.50s", name);
    return c.warn ? FREESASA_INGEST_SELECT_WARN : (int)c.n;
}
