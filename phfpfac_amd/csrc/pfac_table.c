/*
 * pfac_table.c -- host-side C: pattern file -> PHF-compressed PFAC transition table.
 *
 * Product code (libpfac_host.so).  Same observable semantics as the
 * reference's CreateTable/ + PHF/ path, built differently:
 *
 *   reader   create_table_reorder.c:53-122 (read_pattern): '\n'-separated raw
 *            bytes, id = 1-based line number, 1 <= length < 1023+1, file ends in
 *            '\n'.  Violations are reported as PFAC_E_PATTERN instead of exit(1)
 *            (or, for a missing final newline / empty line, instead of the
 *            reference's undefined behaviour).
 *   sort     create_table_reorder.c:21-45,116 (comp_pat + qsort): memcmp on the
 *            common prefix, shorter first; STABLE, so among identical lines the
 *            later one wins its final state (create_table_reorder.c:366).
 *   trie     create_table_reorder.c:277-378 (patternsToPFAC): identical state
 *            numbering (finals 0..n-1 = sorted index, n unused, root n+1,
 *            internals from n+2 in creation order), but built from the sorted
 *            list with an LCP stack into an edge list -- O(total bytes), no
 *            dense int[state][256] rows, no 4 GiB preallocation
 *            (create_table_reorder.c:10,306-311).
 *   PHF      phf.c:151-291 (FFDM): the same row-displacement perfect hash with
 *            the same lookup contract (r may be negative, -1 = empty row; HT
 *            holds the owning row; val the next state), rows placed in
 *            descending fullness, first fit.  The reference's O(rows^2)
 *            exchange sort (phf.c:126-139) is replaced by a counting sort and
 *            the first-fit scan keeps a first-free cursor; the resulting LAYOUT
 *            therefore differs from the reference's, the lookup function does
 *            not (tests/test_table.py checks every (state, byte) cell).
 */
#include "pfac.h"

#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define PFAC_MAX_PATTERN_LEN 1022   /* the reader stores the '\n' too and rejects str_len >= 1024 BEFORE looking at
                                       it (create_table_reorder.c:72-83): 1022 bytes is the longest pattern it accepts */
#define PFAC_COL_MAX 4096           /* phf.c:8 */

typedef struct {
    int32_t id;
    int32_t len;
    const unsigned char *pat;       /* points into the file image */
} pat_t;

typedef struct {
    int32_t from;
    int32_t ch;
    int32_t to;
} edge_t;

static void set_err(char *err, size_t n, const char *msg, long a) {
    if (err && n) snprintf(err, n, msg, a);
}

static int cmp_pat(const pat_t *a, const pat_t *b) {
    int32_t m = a->len < b->len ? a->len : b->len;
    int r = memcmp(a->pat, b->pat, (size_t)m);
    if (r) return r;
    return (a->len > b->len) - (a->len < b->len);
}

static void merge_sort(pat_t *a, pat_t *tmp, size_t n) {
    if (n < 2) return;
    if (n <= 8) {                       /* stable insertion sort for short runs */
        for (size_t i = 1; i < n; i++) {
            pat_t x = a[i];
            size_t j = i;
            while (j > 0 && cmp_pat(&x, &a[j - 1]) < 0) { a[j] = a[j - 1]; j--; }
            a[j] = x;
        }
        return;
    }
    size_t h = n / 2;
    merge_sort(a, tmp, h);
    merge_sort(a + h, tmp, n - h);
    if (cmp_pat(&a[h], &a[h - 1]) >= 0) return;
    size_t i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = (cmp_pat(&a[j], &a[i]) < 0) ? a[j++] : a[i++];
    while (i < h) tmp[k++] = a[i++];
    /* if the left run ran out first, the tail a[j..n) is already in place (k == j) */
    memcpy(a, tmp, k * sizeof(pat_t));
}

static int edge_key_cmp(const void *x, const void *y) {
    const edge_t *a = (const edge_t *)x, *b = (const edge_t *)y;
    if (a->from != b->from) return (a->from > b->from) - (a->from < b->from);
    return a->ch - b->ch;
}

static int is_pow2(int w) { return w > 0 && (w & (w - 1)) == 0; }

void pfac_table_free(pfac_table *t) {
    if (!t) return;
    free(t->s0); free(t->r); free(t->HT); free(t->val); free(t->idmap);
    free(t);
}

/* Row-displacement perfect hash over the edge list (sorted by key). */
static int build_phf(pfac_table *t, const edge_t *edges, int32_t n_edges, char *err, size_t err_len) {
    const int32_t width = t->width;
    const int wbit = t->width_bit;
    const int32_t max_row = (int32_t)(((int64_t)t->state_num * 256) / width) + 1;
    t->max_row = max_row;
    t->n_keys = n_edges;
    int32_t *row_cnt = (int32_t *)calloc((size_t)max_row + 1, sizeof(int32_t));
    int32_t *row_start = (int32_t *)malloc(((size_t)max_row + 1) * sizeof(int32_t));
    int32_t *order = (int32_t *)malloc((size_t)max_row * sizeof(int32_t));
    int32_t *bucket = (int32_t *)calloc((size_t)width + 2, sizeof(int32_t));
    t->r = (int32_t *)malloc((size_t)max_row * sizeof(int32_t));
    if (!row_cnt || !row_start || !order || !bucket || !t->r) { set_err(err, err_len, "out of memory (%ld rows)", max_row); return PFAC_E_NOMEM; }
    memset(t->r, 0xFF, (size_t)max_row * sizeof(int32_t));
    for (int32_t e = 0; e < n_edges; e++) {
        int64_t key = ((int64_t)edges[e].from << 8) + edges[e].ch;
        row_cnt[key >> wbit]++;
    }
    row_start[0] = 0;
    for (int32_t i = 0; i < max_row; i++) row_start[i + 1] = row_start[i] + row_cnt[i];
    /* counting sort of rows by fullness, descending; ties by row number ascending */
    for (int32_t i = 0; i < max_row; i++) bucket[row_cnt[i]]++;
    int32_t acc = 0;
    for (int32_t c = width; c >= 0; c--) { int32_t n = bucket[c]; bucket[c] = acc; acc += n; }
    for (int32_t i = 0; i < max_row; i++) order[bucket[row_cnt[i]]++] = i;

    int64_t cap = (int64_t)n_edges + 2 * (int64_t)width + 64;
    int32_t *HT = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
    int32_t *val = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
    if (!HT || !val) { set_err(err, err_len, "out of memory (%ld slots)", (long)cap); return PFAC_E_NOMEM; }
    memset(HT, 0xFF, (size_t)cap * sizeof(int32_t));
    memset(val, 0xFF, (size_t)cap * sizeof(int32_t));
    /* nxt[i] = a free slot >= i (union-find with path halving); lets the first-fit
     * scan visit only offsets whose first column lands on a free slot */
    int32_t *nxt = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
    if (!nxt) { set_err(err, err_len, "out of memory (%ld slots)", (long)cap); return PFAC_E_NOMEM; }
    for (int64_t i = 0; i < cap; i++) nxt[i] = (int32_t)i;
    int64_t max_used = -1, multi_cursor = 0;
    /* small automata: exact first fit (cursor never moves); large ones: bounded look-back */
    const int64_t lookback = n_edges <= 65536 ? INT64_MAX / 2 : 4 * (int64_t)width;
    const int32_t cmask = width - 1;
    for (int32_t o = 0; o < max_row; o++) {
        const int32_t row = order[o];
        const int32_t cnt = row_cnt[row];
        if (cnt == 0) break;
        const edge_t *re = edges + row_start[row];
        const int32_t col0 = (int32_t)((((int64_t)re[0].from << 8) + re[0].ch) & cmask);
        /* the reference tries every offset from -col0 upwards (phf.c:188); an offset
         * can only succeed if slot offset+col0 is free, so walk the free slots */
        /* multi-key rows resume a little before where the previous one fitted: the
         * slots further back are holes that rows of this fullness already failed on
         * (keeps the build near-linear; single-key rows still fill every hole) */
        int64_t p = cnt > 1 ? multi_cursor : 0, offset;
        for (;;) {
            while (nxt[p] != p) { nxt[p] = nxt[nxt[p]]; p = nxt[p]; }   /* find */
            offset = p - col0;
            if (offset + width + 1 >= cap) {
                int64_t ncap = cap * 2;
                HT = (int32_t *)realloc(HT, (size_t)ncap * sizeof(int32_t));
                val = (int32_t *)realloc(val, (size_t)ncap * sizeof(int32_t));
                nxt = (int32_t *)realloc(nxt, (size_t)ncap * sizeof(int32_t));
                if (!HT || !val || !nxt) { set_err(err, err_len, "out of memory (%ld slots)", (long)ncap); return PFAC_E_NOMEM; }
                memset(HT + cap, 0xFF, (size_t)(ncap - cap) * sizeof(int32_t));
                memset(val + cap, 0xFF, (size_t)(ncap - cap) * sizeof(int32_t));
                for (int64_t i = cap; i < ncap; i++) nxt[i] = (int32_t)i;
                cap = ncap;
            }
            int32_t i;
            for (i = 1; i < cnt; i++) {
                int32_t col = (int32_t)((((int64_t)re[i].from << 8) + re[i].ch) & cmask);
                if (HT[offset + col] != -1) break;
            }
            if (i == cnt) break;
            p++;
        }
        if (offset > INT32_MAX - width) { set_err(err, err_len, "hash table too large (%ld)", (long)offset); return PFAC_E_NOMEM; }
        t->r[row] = (int32_t)offset;
        if (cnt > 1 && p - lookback > multi_cursor) multi_cursor = p - lookback;
        for (int32_t i = 0; i < cnt; i++) {
            int32_t col = (int32_t)((((int64_t)re[i].from << 8) + re[i].ch) & cmask);
            HT[offset + col] = row;
            val[offset + col] = re[i].to;
            nxt[offset + col] = (int32_t)(offset + col + 1);
            if (offset + col > max_used) max_used = offset + col;
        }
    }
    free(nxt);
    t->ht_size = (int32_t)(max_used + 1);
    if (t->ht_size < 1) t->ht_size = 1;     /* keep the arrays non-empty */
    t->HT = (int32_t *)realloc(HT, (size_t)t->ht_size * sizeof(int32_t));
    t->val = (int32_t *)realloc(val, (size_t)t->ht_size * sizeof(int32_t));
    free(row_cnt); free(row_start); free(order); free(bucket);
    return PFAC_OK;
}

/* sort + trie + PHF over n patterns (ids and bytes filled in by a reader; pats is consumed) */
static int build_from_patterns(pat_t *pats, size_t n, size_t total_bytes, int32_t max_len, int width, int part,
                               int n_parts, pfac_table **out, char *err, size_t err_len);

static int build_mem_part(const void *patterns, size_t n_bytes, int width, int part, int n_parts, pfac_table **out,
                          char *err, size_t err_len) {
    if (!patterns || !out) { set_err(err, err_len, "null argument%ld", 0); return PFAC_E_ARG; }
    *out = NULL;
    if (n_parts < 1 || part < 0 || part >= n_parts) { set_err(err, err_len, "bad partition index %ld", part); return PFAC_E_ARG; }
    if (!is_pow2(width) || width > PFAC_COL_MAX) {
        set_err(err, err_len, "PHF width %ld must be a power of two <= 4096", width); return PFAC_E_ARG;
    }
    const unsigned char *buf = (const unsigned char *)patterns;
    if (n_bytes == 0 || buf[n_bytes - 1] != '\n') {
        set_err(err, err_len, "pattern file must end with a newline (%ld bytes)", (long)n_bytes); return PFAC_E_PATTERN;
    }
    /* ---- reader ---- */
    size_t n_lines = 0;
    for (size_t i = 0; i < n_bytes; i++) n_lines += (buf[i] == '\n');
    if (n_lines > (size_t)INT32_MAX / 2) { set_err(err, err_len, "too many patterns (%ld)", (long)n_lines); return PFAC_E_PATTERN; }
    pat_t *pats = (pat_t *)malloc(n_lines * sizeof(pat_t));
    if (!pats) { set_err(err, err_len, "out of memory (%ld patterns)", (long)n_lines); return PFAC_E_NOMEM; }
    size_t start = 0, n = 0;
    int32_t max_len = 0;
    for (size_t i = 0; i < n_bytes; i++) {
        if (buf[i] != '\n') continue;
        size_t len = i - start;
        if (len == 0) { free(pats); set_err(err, err_len, "pattern %ld is empty", (long)n + 1); return PFAC_E_PATTERN; }
        if (len > PFAC_MAX_PATTERN_LEN) { free(pats); set_err(err, err_len, "Pattern %ld length over 1024.", (long)n + 1); return PFAC_E_PATTERN; }
        pats[n].id = (int32_t)(n + 1);
        pats[n].len = (int32_t)len;
        pats[n].pat = buf + start;
        if ((int32_t)len > max_len) max_len = (int32_t)len;
        n++;
        start = i + 1;
    }
    return build_from_patterns(pats, n, n_bytes - n_lines, max_len, width, part, n_parts, out, err, err_len);
}

int pfac_table_build_mem(const void *patterns, size_t n_bytes, int width, pfac_table **out, char *err, size_t err_len) {
    return build_mem_part(patterns, n_bytes, width, 0, 1, out, err, err_len);
}

int pfac_table_build_mem_part(const void *patterns, size_t n_bytes, int width, int part, int n_parts, pfac_table **out,
                              char *err, size_t err_len) {
    return build_mem_part(patterns, n_bytes, width, part, n_parts, out, err, err_len);
}

/*
 * Escape-aware reader: the reference's read_pattern_ext() / fgetc_ext() (create_table_reorder.c:131-185,
 * ctdef.h:37-99; present but never called there).  Inside a pattern a backslash introduces
 *   \a \b \t \n \v \f \r   control characters      \' \" \\   the character itself
 *   \ooo   up to three octal digits      \xNN   up to two hex digits
 * any other "\c" is a literal backslash followed by c; only a REAL newline ends a pattern, so patterns may
 * contain '\n' bytes.  The reference scans the numbers with fscanf("%3o") / fscanf("%2x") on the stream; this is
 * an in-memory parser of the same grammar, odd inputs included (scan_escape_number() below follows glibc's
 * integer conversion step by step: "\8" is byte 0 followed by '8', "\x" without digits is byte 0, white space --
 * even a newline -- between "\x" and its digits is skipped, a sign or a "0x" prefix counts against the width).
 * tests/test_table.py pins it against the reference's own reader on hand-picked and on fuzzed pattern files.
 */
#define PFAC_EOL 0x10A
typedef struct { const unsigned char *p, *end; } mem_cursor;
static int cur_getc(mem_cursor *c) { return c->p < c->end ? *c->p++ : EOF; }
static void cur_ungetc(mem_cursor *c, int ch) { if (ch != EOF) c->p--; }

/* fscanf(fp, "%<width>o" or "%<width>x", value): *value is written only when a number was matched */
static void scan_escape_number(mem_cursor *c, unsigned base, int width, unsigned *value) {
    int ch;
    do ch = cur_getc(c); while (ch == ' ' || (ch >= '\t' && ch <= '\r'));   /* leading white space, not counted */
    if (ch == EOF) return;                                                    /* input failure */
    int negative = 0, have_sign = 0, n_digits = 0;
    unsigned acc = 0;
    if (ch == '-' || ch == '+') { negative = ch == '-'; have_sign = 1; width--; ch = cur_getc(c); }
    if (width != 0 && ch == '0') {
        width--; n_digits = 1;
        ch = cur_getc(c);
        if (width != 0 && (ch == 'x' || ch == 'X') && base == 16) { width--; ch = cur_getc(c); }
    }
    while (ch != EOF && width != 0) {
        unsigned d;
        if (ch >= '0' && ch <= '9') d = (unsigned)(ch - '0');
        else if (ch >= 'a' && ch <= 'f') d = (unsigned)(ch - 'a') + 10;
        else if (ch >= 'A' && ch <= 'F') d = (unsigned)(ch - 'A') + 10;
        else break;
        if (d >= base) break;
        acc = acc * base + d;
        n_digits++; width--;
        ch = cur_getc(c);
    }
    cur_ungetc(c, ch);                       /* the character after the number (a lone sign stays consumed) */
    (void)have_sign;
    if (n_digits == 0) return;               /* matching failure */
    *value = negative ? 0u - acc : acc;
}

static int getc_escaped(mem_cursor *c) {
    const int c0 = cur_getc(c);
    if (c0 == '\\') {
        const int c1 = cur_getc(c);
        unsigned value = 0;
        if (c1 == EOF) return c0;
        if (c1 >= '0' && c1 <= '9') {
            cur_ungetc(c, c1);
            scan_escape_number(c, 8, 3, &value);
            return (int)(char)value;
        }
        switch (c1) {
            case 'a': return '\a';
            case 'b': return '\b';
            case 't': return '\t';
            case 'n': return '\n';
            case 'v': return '\v';
            case 'f': return '\f';
            case 'r': return '\r';
            case '\'': case '"': case '\\': return c1;
            case 'x':
                scan_escape_number(c, 16, 2, &value);
                return (int)(char)value;
            default:
                cur_ungetc(c, c1);
                return c0;
        }
    }
    if (c0 == '\n') return PFAC_EOL;
    return c0;
}

static int build_escaped_mem(const unsigned char *img, size_t n_bytes, int width, pfac_table **out, char *err, size_t err_len) {
    mem_cursor cur = {img, img + n_bytes};
    size_t cap = 1024, n = 0, arena_cap = 1 << 16, arena_len = 0, total = 0;
    pat_t *pats = (pat_t *)malloc(cap * sizeof(pat_t));
    size_t *offs = (size_t *)malloc(cap * sizeof(size_t));
    unsigned char *arena = (unsigned char *)malloc(arena_cap);
    int32_t max_len = 0;
    int rc = PFAC_OK;
    unsigned char str[PFAC_MAX_PATTERN_LEN + 2];
    while (pats && offs && arena) {
        int len = 0, ch, eof_inside = 0;
        for (;;) {
            const int at_end = cur.p >= cur.end;
            ch = getc_escaped(&cur);
            if (ch == PFAC_EOL) break;
            if (ch == EOF && at_end) { eof_inside = 1; break; }
            str[len++] = (unsigned char)ch;
            if (len > PFAC_MAX_PATTERN_LEN) break;
        }
        if (eof_inside) { set_err(err, err_len, "pattern file must end with a newline (pattern %ld)", (long)n + 1); rc = PFAC_E_PATTERN; break; }
        if (len > PFAC_MAX_PATTERN_LEN) { set_err(err, err_len, "Pattern %ld length over 1024.", (long)n + 1); rc = PFAC_E_PATTERN; break; }
        if (len == 0) { set_err(err, err_len, "pattern %ld is empty", (long)n + 1); rc = PFAC_E_PATTERN; break; }
        if (n == cap) {
            cap *= 2;
            pats = (pat_t *)realloc(pats, cap * sizeof(pat_t));
            offs = (size_t *)realloc(offs, cap * sizeof(size_t));
            if (!pats || !offs) break;
        }
        if (arena_len + (size_t)len > arena_cap) {
            arena_cap *= 2;
            arena = (unsigned char *)realloc(arena, arena_cap);
            if (!arena) break;
        }
        memcpy(arena + arena_len, str, (size_t)len);
        offs[n] = arena_len;
        pats[n].id = (int32_t)(n + 1);
        pats[n].len = len;
        arena_len += (size_t)len;
        total += (size_t)len;
        if (len > max_len) max_len = len;
        n++;
        if (cur.p >= cur.end) break;                    /* end of file after a newline (ctr.c:174-180) */
    }
    if (!pats || !offs || !arena) { free(pats); free(offs); free(arena); set_err(err, err_len, "out of memory (%ld patterns)", (long)n); return PFAC_E_NOMEM; }
    if (rc) { free(pats); free(offs); free(arena); return rc; }
    for (size_t i = 0; i < n; i++) pats[i].pat = arena + offs[i];
    free(offs);
    rc = build_from_patterns(pats, n, total, max_len, width, 0, 1, out, err, err_len);
    free(arena);
    return rc;
}

int pfac_table_build_file_escaped(const char *pattern_file, int width, pfac_table **out, char *err, size_t err_len) {
    if (!pattern_file || !out) return PFAC_E_ARG;
    *out = NULL;
    if (!is_pow2(width) || width > PFAC_COL_MAX) {
        set_err(err, err_len, "PHF width %ld must be a power of two <= 4096", width); return PFAC_E_ARG;
    }
    FILE *f = fopen(pattern_file, "rb");
    if (!f) { if (err && err_len) snprintf(err, err_len, "cannot open pattern file %s", pattern_file); return PFAC_E_IO; }
    size_t cap = 1 << 16, n = 0;
    unsigned char *img = (unsigned char *)malloc(cap);
    while (img) {
        n += fread(img + n, 1, cap - n, f);
        if (n < cap) break;
        cap *= 2;
        img = (unsigned char *)realloc(img, cap);
    }
    const int io_error = ferror(f);
    fclose(f);
    if (!img) { set_err(err, err_len, "out of memory (%ld bytes of pattern file)", (long)n); return PFAC_E_NOMEM; }
    if (io_error) { free(img); if (err && err_len) snprintf(err, err_len, "cannot read pattern file %s", pattern_file); return PFAC_E_IO; }
    const int rc = build_escaped_mem(img, n, width, out, err, err_len);
    free(img);
    return rc;
}

static int same_bytes(const pat_t *a, const pat_t *b) {
    return a->len == b->len && memcmp(a->pat, b->pat, (size_t)a->len) == 0;
}

static int build_from_patterns(pat_t *pats, size_t n, size_t total_bytes, int32_t max_len, int width, int part,
                               int n_parts, pfac_table **out, char *err, size_t err_len) {
    pat_t *tmp = (pat_t *)malloc((n ? n : 1) * sizeof(pat_t));
    if (!tmp) { free(pats); set_err(err, err_len, "out of memory (%ld patterns)", (long)n); return PFAC_E_NOMEM; }
    merge_sort(pats, tmp, n);
    free(tmp);
    pat_t *const all = pats;                   /* what gets freed */
    if (n_parts > 1) {
        /* Pattern partitioning as the reference does it (create_table_reorder.c:217-247): the SORTED list is cut
         * into P runs of k = n / P patterns, the last one also takes the n % P left over; ids stay the 1-based line
         * numbers of the whole file and max_pat_len stays the global maximum (ctr.c:238,246).  One difference, on
         * purpose: a cut never separates identical strings (it moves past them), so "the last line wins"
         * (ctr.c:366) is decided inside one partition -- the reference lets such duplicates overflow the
         * position's result slots (main.cc:308-315), which is outside the parity domain. */
        const size_t k = n / (size_t)n_parts;
        size_t lo = (size_t)part * k;
        size_t hi = part == n_parts - 1 ? n : lo + k;
        while (lo > 0 && lo < n && same_bytes(&pats[lo - 1], &pats[lo])) lo++;
        while (hi > 0 && hi < n && same_bytes(&pats[hi - 1], &pats[hi])) hi++;
        if (hi < lo) hi = lo;
        pats += lo;
        n = hi - lo;
        total_bytes = 0;
        for (size_t i = 0; i < n; i++) total_bytes += (size_t)pats[i].len;
    }

    /* ---- trie as an edge list (LCP stack over the sorted list) ---- */
    if ((int64_t)n + 2 + (int64_t)total_bytes > INT32_MAX / 256) {
        /* keys are (state<<8)+ch in int32 on the device (master_kernel.cu:52) */
        free(all); set_err(err, err_len, "automaton too large (%ld pattern bytes)", (long)total_bytes); return PFAC_E_PATTERN;
    }
    edge_t *edges = (edge_t *)malloc((total_bytes + 1) * sizeof(edge_t));
    int32_t *path = (int32_t *)malloc((PFAC_MAX_PATTERN_LEN + 1) * sizeof(int32_t));
    int32_t *path_edge = (int32_t *)malloc((PFAC_MAX_PATTERN_LEN + 1) * sizeof(int32_t));
    pfac_table *t = (pfac_table *)calloc(1, sizeof *t);
    int32_t *idmap = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
    if (!edges || !path || !path_edge || !t || !idmap) {
        free(all); free(edges); free(path); free(path_edge); free(t); free(idmap);
        set_err(err, err_len, "out of memory (%ld bytes)", (long)total_bytes); return PFAC_E_NOMEM;
    }
    const int32_t root = (int32_t)n + 1;
    int32_t state_count = root + 1;
    int32_t n_edges = 0;
    const pat_t *prev = NULL;
    for (size_t i = 0; i < n; i++) {
        const pat_t *cur = &pats[i];
        idmap[i] = cur->id;
        int32_t l = 0;
        if (prev) {
            int32_t m = prev->len < cur->len ? prev->len : cur->len;
            while (l < m && prev->pat[l] == cur->pat[l]) l++;
        }
        if (prev && l == cur->len) {
            /* identical to the previous line: the edge into its final state is
             * re-pointed at the later pattern (create_table_reorder.c:366) */
            edges[path_edge[l - 1]].to = (int32_t)i;
            path[l - 1] = (int32_t)i;
            prev = cur;
            continue;
        }
        /* path[d] = state after d+1 bytes of prev; the first l are shared */
        int32_t state = l ? path[l - 1] : root;
        for (int32_t j = l; j < cur->len; j++) {
            int32_t to = (j == cur->len - 1) ? (int32_t)i : state_count++;
            edges[n_edges].from = state;
            edges[n_edges].ch = cur->pat[j];
            edges[n_edges].to = to;
            path[j] = to;
            path_edge[j] = n_edges;
            n_edges++;
            state = to;
        }
        prev = cur;
    }
    free(path); free(path_edge); free(all);

    t->width = width;
    for (t->width_bit = 0; (width >> t->width_bit) != 1; t->width_bit++) ;
    t->n_patterns = (int32_t)n;
    t->num_final = (int32_t)n;
    t->state_num = state_count;
    t->max_pat_len = max_len;
    t->idmap = idmap;
    t->s0 = (int32_t *)malloc(256 * sizeof(int32_t));
    if (!t->s0) { free(edges); pfac_table_free(t); return PFAC_E_NOMEM; }
    memset(t->s0, 0xFF, 256 * sizeof(int32_t));
    qsort(edges, (size_t)n_edges, sizeof(edge_t), edge_key_cmp);
    for (int32_t e = 0; e < n_edges; e++)
        if (edges[e].from == root) t->s0[edges[e].ch] = edges[e].to;
    int rc = build_phf(t, edges, n_edges, err, err_len);
    free(edges);
    if (rc) { pfac_table_free(t); return rc; }
    *out = t;
    return PFAC_OK;
}

int pfac_table_build_file_part(const char *pattern_file, int width, int part, int n_parts, pfac_table **out, char *err,
                               size_t err_len) {
    if (!pattern_file || !out) return PFAC_E_ARG;
    FILE *f = fopen(pattern_file, "rb");
    if (!f) { if (err && err_len) snprintf(err, err_len, "cannot open pattern file %s", pattern_file); return PFAC_E_IO; }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    rewind(f);
    unsigned char *buf = (unsigned char *)malloc(sz > 0 ? (size_t)sz : 1);
    if (!buf) { fclose(f); return PFAC_E_NOMEM; }
    if (sz > 0 && fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(buf); return PFAC_E_IO; }
    fclose(f);
    int rc = build_mem_part(buf, sz > 0 ? (size_t)sz : 0, width, part, n_parts, out, err, err_len);
    free(buf);
    return rc;
}

int pfac_table_build_file(const char *pattern_file, int width, pfac_table **out, char *err, size_t err_len) {
    return pfac_table_build_file_part(pattern_file, width, 0, 1, out, err, err_len);
}

int32_t pfac_table_lookup(const pfac_table *t, int32_t state, int32_t ch) {
    int32_t key = (state << 8) + ch;
    int32_t row = key >> t->width_bit;
    int32_t col = key & (t->width - 1);
    if (row < 0 || row >= t->max_row) return -1;
    int32_t idx = t->r[row] + col;
    if (idx < 0 || idx >= t->ht_size) return -1;
    return t->HT[idx] == row ? t->val[idx] : -1;
}

/* ---- flat image ---- */
size_t pfac_table_blob_words(const pfac_table *t) {
    return (size_t)PFAC_BLOB_HEADER_WORDS + 256 + (size_t)t->max_row + 2 * (size_t)t->ht_size + (size_t)t->num_final;
}

int pfac_table_to_blob(const pfac_table *t, int32_t *blob, size_t n_words) {
    if (!t || !blob || n_words < pfac_table_blob_words(t)) return PFAC_E_ARG;
    memset(blob, 0, PFAC_BLOB_HEADER_WORDS * sizeof(int32_t));
    blob[0] = PFAC_BLOB_MAGIC; blob[1] = PFAC_BLOB_VERSION;
    blob[2] = t->width; blob[3] = t->width_bit; blob[4] = t->n_patterns; blob[5] = t->num_final;
    blob[6] = t->state_num; blob[7] = t->max_pat_len; blob[8] = t->max_row; blob[9] = t->ht_size;
    blob[10] = t->n_keys;
    int32_t *p = blob + PFAC_BLOB_HEADER_WORDS;
    memcpy(p, t->s0, 256 * sizeof(int32_t)); p += 256;
    memcpy(p, t->r, (size_t)t->max_row * sizeof(int32_t)); p += t->max_row;
    memcpy(p, t->HT, (size_t)t->ht_size * sizeof(int32_t)); p += t->ht_size;
    memcpy(p, t->val, (size_t)t->ht_size * sizeof(int32_t)); p += t->ht_size;
    memcpy(p, t->idmap, (size_t)t->num_final * sizeof(int32_t));
    return PFAC_OK;
}

static int32_t *dup_words(const int32_t *src, size_t n) {
    int32_t *d = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
    if (d && n) memcpy(d, src, n * sizeof(int32_t));
    return d;
}

int pfac_table_from_blob(const int32_t *blob, size_t n_words, pfac_table **out) {
    if (!blob || !out || n_words < PFAC_BLOB_HEADER_WORDS) return PFAC_E_ARG;
    if (blob[0] != PFAC_BLOB_MAGIC || blob[1] != PFAC_BLOB_VERSION) return PFAC_E_ARG;
    pfac_table *t = (pfac_table *)calloc(1, sizeof *t);
    if (!t) return PFAC_E_NOMEM;
    t->width = blob[2]; t->width_bit = blob[3]; t->n_patterns = blob[4]; t->num_final = blob[5];
    t->state_num = blob[6]; t->max_pat_len = blob[7]; t->max_row = blob[8]; t->ht_size = blob[9];
    t->n_keys = blob[10];
    if (!is_pow2(t->width) || t->width > PFAC_COL_MAX || (1 << t->width_bit) != t->width || t->max_row < 1 ||
        t->ht_size < 1 || t->num_final < 0 || t->state_num < t->num_final + 2 || pfac_table_blob_words(t) > n_words) {
        free(t); return PFAC_E_ARG;
    }
    const int32_t *p = blob + PFAC_BLOB_HEADER_WORDS;
    t->s0 = dup_words(p, 256); p += 256;
    t->r = dup_words(p, (size_t)t->max_row); p += t->max_row;
    t->HT = dup_words(p, (size_t)t->ht_size); p += t->ht_size;
    t->val = dup_words(p, (size_t)t->ht_size); p += t->ht_size;
    t->idmap = dup_words(p, (size_t)t->num_final);
    if (!t->s0 || !t->r || !t->HT || !t->val || !t->idmap) { pfac_table_free(t); return PFAC_E_NOMEM; }
    *out = t;
    return PFAC_OK;
}

int pfac_table_from_reference_arrays(const int32_t *s0, const int32_t *r, const int32_t *HT, const int32_t *val,
                                     const int32_t *idmap, int32_t width, int32_t state_num, int32_t num_final,
                                     int32_t ht_size, int32_t max_pat_len, pfac_table **out) {
    if (!s0 || !r || !HT || !val || !out || !is_pow2(width) || width > PFAC_COL_MAX || ht_size < 0 ||
        num_final < 0 || state_num < num_final + 2)
        return PFAC_E_ARG;
    pfac_table *t = (pfac_table *)calloc(1, sizeof *t);
    if (!t) return PFAC_E_NOMEM;
    t->width = width;
    for (t->width_bit = 0; (width >> t->width_bit) != 1; t->width_bit++) ;
    t->n_patterns = t->num_final = num_final;
    t->state_num = state_num;
    t->max_pat_len = max_pat_len;
    t->max_row = (int32_t)(((int64_t)state_num * 256) / width) + 1;     /* master_kernel.cu:212 */
    t->ht_size = ht_size > 0 ? ht_size : 1;
    t->s0 = dup_words(s0, 256);
    t->r = dup_words(r, (size_t)t->max_row);
    if (ht_size > 0) { t->HT = dup_words(HT, (size_t)ht_size); t->val = dup_words(val, (size_t)ht_size); }
    else {
        t->HT = (int32_t *)malloc(sizeof(int32_t)); t->val = (int32_t *)malloc(sizeof(int32_t));
        if (t->HT) t->HT[0] = -1;
        if (t->val) t->val[0] = -1;
    }
    t->idmap = (int32_t *)malloc((num_final ? (size_t)num_final : 1) * sizeof(int32_t));
    if (!t->s0 || !t->r || !t->HT || !t->val || !t->idmap) { pfac_table_free(t); return PFAC_E_NOMEM; }
    for (int32_t i = 0; i < num_final; i++) t->idmap[i] = idmap ? idmap[i] : i;
    for (int32_t i = 0; i < t->ht_size; i++) t->n_keys += (t->HT[i] >= 0);
    *out = t;
    return PFAC_OK;
}

/* ---- text emitter (main.cc:335-350) ---- */
static inline char *put_uint(char *p, uint64_t v, int min_width) {
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    for (int i = n; i < min_width; i++) *p++ = ' ';
    while (n) *p++ = tmp[--n];
    return p;
}

/* Where the records come from: a sorted pfac_record array, or the compact device form (heap of 32-bit words + the
 * ordered tile index; tile_pre[t] = records of the tiles before t, tile_pre[n_tiles] = all). */
typedef struct {
    const pfac_record *rec;
    const void *words;              /* 16- or 32-bit words (word_bytes) */
    int word_bytes;
    const uint64_t *tile_index;
    const uint64_t *tile_pre;
    uint64_t n_tiles;
} rec_src;

/* tile of record k of the sorted sequence: the t with tile_pre[t] <= k < tile_pre[t+1] (empty tiles skipped) */
static uint64_t tile_of(const rec_src *s, uint64_t k) {
    uint64_t lo = 0, hi = s->n_tiles;               /* invariant: tile_pre[lo] <= k < tile_pre[hi] */
    while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (s->tile_pre[mid] <= k) lo = mid; else hi = mid;
    }
    return lo;
}

/* Lines of records [k0, k1) into p (pass 2) or only their byte count (p == NULL, pass 1). */
static char *format_records(const rec_src *s, uint64_t k0, uint64_t k1, uint64_t base, const int32_t *idmap, char *p,
                            uint64_t *bytes_out) {
    uint64_t bytes = 0, t = 0;
    if (s->words && k0 < k1) t = tile_of(s, k0);
    for (uint64_t k = k0; k < k1; k++) {
        uint64_t pos;
        uint32_t st;
        if (s->words) {
            while (k >= s->tile_pre[t + 1]) t++;
            const uint64_t at = PFAC_TIX_FIRST(s->tile_index[t]) + (k - s->tile_pre[t]);
            const uint32_t w = s->word_bytes == 2 ? ((const uint16_t *)s->words)[at] : ((const uint32_t *)s->words)[at];
            pos = base + t * PFAC_TILE_BYTES + PFAC_PACKED_POS(w);
            st = PFAC_PACKED_STATE(w);
        } else {
            pos = base + s->rec[k].pos;
            st = s->rec[k].state;
        }
        const int32_t id = idmap ? idmap[st] : (int32_t)st;
        if (!p) {
            int dp = 1, di = id < 0 ? 2 : 1;
            for (uint64_t v = pos; v >= 10; v /= 10) dp++;
            for (uint64_t v = id < 0 ? (uint64_t)(-(int64_t)id) : (uint64_t)id; v >= 10; v /= 10) di++;
            bytes += 12 + (uint64_t)(dp < 4 ? 4 : dp) + 16 + (uint64_t)di + 1;
            continue;
        }
        memcpy(p, "At position ", 12); p += 12;
        p = put_uint(p, pos, 4);                                  /* %4d */
        memcpy(p, ", match pattern ", 16); p += 16;
        if (id < 0) { *p++ = '-'; p = put_uint(p, (uint64_t)(-(int64_t)id), 1); }
        else p = put_uint(p, (uint64_t)id, 1);                    /* %d */
        *p++ = '\n';
    }
    if (bytes_out) *bytes_out = bytes;
    return p;
}

static int64_t emit_serial(FILE *f, const rec_src *s, uint64_t n, uint64_t base, const int32_t *idmap) {
    enum { CHUNK = 1 << 16, LINE_MAX_BYTES = 64 };
    char *buf = (char *)malloc((size_t)CHUNK * LINE_MAX_BYTES);
    if (!buf) return PFAC_E_NOMEM;
    int64_t total = 0;
    for (uint64_t k0 = 0; k0 < n; k0 += CHUNK) {
        const uint64_t k1 = k0 + CHUNK < n ? k0 + CHUNK : n;
        const size_t len = (size_t)(format_records(s, k0, k1, base, idmap, buf, NULL) - buf);
        if (fwrite(buf, 1, len, f) != len) { free(buf); return PFAC_E_IO; }
        total += (int64_t)len;
    }
    free(buf);
    return total;
}

int64_t pfac_emit_records(void *file, const pfac_record *rec, uint64_t n, uint64_t base, const int32_t *idmap) {
    if (!file || (!rec && n)) return PFAC_E_ARG;
    const rec_src s = {rec, NULL, 0, NULL, NULL, 0};
    return emit_serial((FILE *)file, &s, n, base, idmap);
}

/* ---- pattern-partition mode: merge of the per-partition match lists, the reference's
 * host merge (main.cc:304-324) on compact records.  There partition p's ids are appended, p ascending, after
 * whatever position i's slots already hold; the partitions are consecutive runs of the SORTED pattern list and
 * all patterns matching at one position are prefixes of one another, so that order is (position, pattern
 * length) again.  Here: K lists sorted by position -> one list ordered by (position, partition), stable inside
 * a partition; the output's `state` field holds the PATTERN ID (idmaps[k] applied; NULL = already ids). ---- */
int64_t pfac_merge_partitions(const pfac_record *const *lists, const uint64_t *counts, const int32_t *const *idmaps,
                              int n_parts, pfac_record *out, uint64_t out_cap) {
    if (n_parts < 0 || (n_parts && (!lists || !counts))) return PFAC_E_ARG;
    uint64_t total = 0;
    for (int k = 0; k < n_parts; k++) {
        if (counts[k] && !lists[k]) return PFAC_E_ARG;
        total += counts[k];
    }
    if (total > out_cap) return PFAC_E_OVERFLOW;
    if (total && !out) return PFAC_E_ARG;
    uint64_t *head = (uint64_t *)calloc(n_parts ? (size_t)n_parts : 1, sizeof(uint64_t));
    if (!head) return PFAC_E_NOMEM;
    uint64_t w = 0;
    while (w < total) {
        /* best = smallest (pos, partition) among the heads; limit = the runner-up: a whole run is copied at once */
        int best = -1, next = -1;
        for (int k = 0; k < n_parts; k++) {
            if (head[k] >= counts[k]) continue;
            const uint32_t p = lists[k][head[k]].pos;
            if (best < 0 || p < lists[best][head[best]].pos) { next = best; best = k; }
            else if (next < 0 || p < lists[next][head[next]].pos) next = k;
        }
        const pfac_record *src = lists[best];
        const int32_t *im = idmaps ? idmaps[best] : NULL;
        uint64_t h = head[best];
        if (next < 0) {
            for (; h < counts[best]; h++, w++) { out[w].pos = src[h].pos; out[w].state = im ? (uint32_t)im[src[h].state] : src[h].state; }
        } else {
            /* records of `best` go first while pos < limit, or pos == limit and best is the earlier partition */
            const uint32_t lim = lists[next][head[next]].pos;
            for (; h < counts[best] && (src[h].pos < lim || (src[h].pos == lim && best < next)); h++, w++) {
                out[w].pos = src[h].pos;
                out[w].state = im ? (uint32_t)im[src[h].state] : src[h].state;
            }
        }
        head[best] = h;
    }
    free(head);
    return (int64_t)total;
}

/* ---- parallel text emitter (once the scan runs at TB/s the serial fprintf loop of
 * main.cc:341-349 is the end-to-end wall).  Records are cut into blocks; pass 1 sizes every block (line
 * length depends on the digit counts), a prefix sum gives each block its file offset, pass 2 formats the
 * blocks and pwrite()s them in place from all threads.  Byte-identical to the serial emitter. ---- */
enum { EMIT_BLOCK = 1 << 17, EMIT_LINE_MAX = 64 };

typedef struct {
    const rec_src *src;
    uint64_t n, base;
    const int32_t *idmap;
    uint64_t n_blocks;
    uint64_t *block_bytes;          /* pass 1 out, then exclusive prefix = file offsets */
    int fd;
    int64_t file_base;
    int pass;
    int tid, n_threads;
    int rc;
} emit_job;

static void *emit_worker(void *arg) {
    emit_job *j = (emit_job *)arg;
    char *buf = NULL;
    if (j->pass == 2) {
        buf = (char *)malloc((size_t)EMIT_BLOCK * EMIT_LINE_MAX);
        if (!buf) { j->rc = PFAC_E_NOMEM; return NULL; }
    }
    for (uint64_t b = (uint64_t)j->tid; b < j->n_blocks; b += (uint64_t)j->n_threads) {
        const uint64_t k0 = b * EMIT_BLOCK, k1 = k0 + EMIT_BLOCK < j->n ? k0 + EMIT_BLOCK : j->n;
        if (j->pass == 1) {
            format_records(j->src, k0, k1, j->base, j->idmap, NULL, &j->block_bytes[b]);
        } else {
            const size_t len = (size_t)(format_records(j->src, k0, k1, j->base, j->idmap, buf, NULL) - buf);
            size_t done = 0;
            const int64_t off = j->file_base + (int64_t)j->block_bytes[b];
            while (done < len) {
                ssize_t w = pwrite(j->fd, buf + done, len - done, off + (int64_t)done);
                if (w <= 0) { j->rc = PFAC_E_IO; free(buf); return NULL; }
                done += (size_t)w;
            }
        }
    }
    free(buf);
    return NULL;
}

static int64_t emit_mt(FILE *f, const rec_src *src, uint64_t n, uint64_t base, const int32_t *idmap, int n_threads) {
    if (n_threads < 2 || n < 4 * (uint64_t)EMIT_BLOCK) return emit_serial(f, src, n, base, idmap);
    if (n_threads > 64) n_threads = 64;
    if (fflush(f)) return PFAC_E_IO;
    const int fd = fileno(f);
    const long at = ftell(f);
    /* pipes are not seekable, and Linux pwrite() ignores the offset on O_APPEND descriptors: serial path */
    if (fd < 0 || at < 0 || (fcntl(fd, F_GETFL) & O_APPEND)) return emit_serial(f, src, n, base, idmap);
    const uint64_t n_blocks = (n + EMIT_BLOCK - 1) / EMIT_BLOCK;
    uint64_t *bb = (uint64_t *)malloc((size_t)n_blocks * sizeof(uint64_t));
    emit_job *jobs = (emit_job *)calloc((size_t)n_threads, sizeof(emit_job));
    pthread_t *th = (pthread_t *)malloc((size_t)n_threads * sizeof(pthread_t));
    int *started = (int *)calloc((size_t)n_threads, sizeof(int));
    if (!bb || !jobs || !th || !started) { free(bb); free(jobs); free(th); free(started); return PFAC_E_NOMEM; }
    int64_t total = 0;
    int rc = 0;
    for (int pass = 1; pass <= 2 && !rc; pass++) {
        for (int t = 0; t < n_threads; t++) {
            emit_job j = {src, n, base, idmap, n_blocks, bb, fd, (int64_t)at, pass, t, n_threads, 0};
            jobs[t] = j;
            started[t] = pthread_create(&th[t], NULL, emit_worker, &jobs[t]) == 0;
            if (!started[t]) emit_worker(&jobs[t]);            /* no thread to be had: this share runs inline */
        }
        for (int t = 0; t < n_threads; t++) {
            if (started[t]) pthread_join(th[t], NULL);
            if (jobs[t].rc) rc = jobs[t].rc;
        }
        if (pass == 1) {                            /* exclusive prefix: block sizes -> offsets */
            uint64_t acc = 0;
            for (uint64_t b = 0; b < n_blocks; b++) { uint64_t v = bb[b]; bb[b] = acc; acc += v; }
            total = (int64_t)acc;
        }
    }
    free(bb); free(jobs); free(th); free(started);
    if (rc) return rc;
    if (fseek(f, at + (long)total, SEEK_SET)) return PFAC_E_IO;
    return total;
}

int64_t pfac_emit_records_mt(void *file, const pfac_record *rec, uint64_t n, uint64_t base, const int32_t *idmap,
                             int n_threads) {
    if (!file || (!rec && n)) return PFAC_E_ARG;
    const rec_src s = {rec, NULL, 0, NULL, NULL, 0};
    return emit_mt((FILE *)file, &s, n, base, idmap, n_threads);
}

int64_t pfac_emit_packed(void *file, const void *words, uint64_t n_words, int record_bytes, const uint64_t *tile_index,
                         uint64_t n_tiles, uint64_t base, const int32_t *idmap, int n_threads) {
    if (!file || (!tile_index && n_tiles) || (record_bytes != 2 && record_bytes != 4)) return PFAC_E_ARG;
    uint64_t *pre = (uint64_t *)malloc((size_t)(n_tiles + 1) * sizeof(uint64_t));
    if (!pre) return PFAC_E_NOMEM;
    uint64_t n = 0;
    int64_t rc = PFAC_E_ARG;
    /* the tile index comes from the device: an index copied after an overflowed or failed scan, or paired with a
     * shorter heap copy, must not send the formatter outside words[0, n_words) */
    for (uint64_t t = 0; t < n_tiles; t++) {
        const uint64_t cnt = PFAC_TIX_COUNT(tile_index[t]);
        if (cnt && PFAC_TIX_FIRST(tile_index[t]) + cnt > n_words) { free(pre); return PFAC_E_ARG; }
        pre[t] = n;
        n += cnt;
    }
    pre[n_tiles] = n;
    if (words || !n) {
        const rec_src s = {NULL, words, record_bytes, tile_index, pre, n_tiles};
        rc = emit_mt((FILE *)file, &s, n, base, idmap, n_threads);
    }
    free(pre);
    return rc;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Character-class patterns (second half): the front end the reference sketches in
 * CreateTable/charset_table_reorder.c (orphaned there: it is #included by nothing and does not compile).  Grammar,
 * restated from fgetc_set() / build_NFA() (:45-168): a pattern is a sequence of ELEMENTS up to a real newline; an
 * element is one (escape-aware, fgetc_ext) character or a class "[...]" / "[^...]" whose items are characters and
 * ranges "l-r" (items go through the escape reader too; a '-' with no character before it is a literal '-'; the
 * first ']' -- by VALUE, so "\x5d" as well -- closes the class, also right after "l-").  No repetition operators:
 * every pattern has a fixed length in elements, the NFA is one chain per pattern, and the subset construction
 * (NFA2DFA, :321-427) yields an acyclic DFA whose states are "the patterns still alive after d bytes".  That DFA IS
 * a PFAC table (no failure links), numbered as patternsToPFAC numbers a trie: final states first (0 .. F-1, in BFS
 * order as mark_DFA_id :429-471 does), F unused, root F+1, the other states behind it.  One difference from a plain
 * trie: a final state can stand for SEVERAL patterns ("[ab]c" and "ac" both end in the state reached by "ac");
 * pfac_outputs lists them, ascending pattern id, and pfac_emit_records_multi prints one line per pattern.
 * The scan kernel is untouched: it walks whatever lookup(state, byte) says.
 * Parity: UNPINNED against the reference (its code for this cannot be built and it holds no fixture); pinned against
 * the independent brute-force matcher oracle/charclass_oracle.py.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct { uint64_t w[4]; } cset_t;
static inline int cset_has(const cset_t *s, int c) { return (int)((s->w[c >> 6] >> (c & 63)) & 1u); }
static inline void cset_put(cset_t *s, int c, int on) {
    if (on) s->w[c >> 6] |= 1ull << (c & 63); else s->w[c >> 6] &= ~(1ull << (c & 63));
}

void pfac_outputs_free(pfac_outputs *o) {
    if (!o) return;
    free(o->first); free(o->ids); free(o);
}

typedef struct {
    int32_t first, count, depth;    /* members[first .. first+count): ascending pattern indices */
    int32_t id;                     /* final numbering */
} dstate_t;

static uint64_t hash_members(const int32_t *m, int32_t n, int32_t depth) {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)depth;
    for (int32_t i = 0; i < n; i++) { h ^= (uint64_t)(uint32_t)m[i]; h *= 1099511628211ull; }
    return h;
}

static int build_charclass_mem(const unsigned char *img, size_t n_bytes, int width, pfac_table **out, pfac_outputs **outs,
                               char *err, size_t err_len) {
    /* ---- parse: elems[] arena, pattern p = elems[poff[p] .. poff[p] + plen[p]) ---- */
    mem_cursor cur = {img, img + n_bytes};
    size_t ecap = 1024, n_elem = 0, pcap = 256, n_pat = 0;
    cset_t *elems = (cset_t *)malloc(ecap * sizeof(cset_t));
    int32_t *poff = (int32_t *)malloc(pcap * sizeof(int32_t)), *plen = (int32_t *)malloc(pcap * sizeof(int32_t));
    int rc = PFAC_OK;
    int32_t max_len = 0;
    while (elems && poff && plen && cur.p < cur.end && !rc) {
        int32_t len = 0;
        const size_t start = n_elem;
        for (;;) {
            const int at_end = cur.p >= cur.end;
            int ch = getc_escaped(&cur);
            if (ch == PFAC_EOL) break;
            if (ch == EOF && at_end) { set_err(err, err_len, "pattern file must end with a newline (pattern %ld)", (long)n_pat + 1); rc = PFAC_E_PATTERN; break; }
            cset_t s = {{0, 0, 0, 0}};
            if (ch == '[') {
                int setting = 1, have_l = 0, bad = 0;
                unsigned char ch_l = 0;
                ch = getc_escaped(&cur);
                if (ch == '^') { s.w[0] = s.w[1] = s.w[2] = s.w[3] = ~0ull; setting = 0; ch = getc_escaped(&cur); }
                while (ch != ']') {
                    if (ch == PFAC_EOL || (ch == EOF && cur.p >= cur.end)) { bad = 1; break; }
                    if (ch == '-' && have_l) {
                        const int r = getc_escaped(&cur);
                        if (r == PFAC_EOL || (r == EOF && cur.p >= cur.end)) { bad = 1; break; }
                        for (int i = ch_l; i <= (int)(unsigned char)r; i++) cset_put(&s, i, setting);
                    } else {
                        ch_l = (unsigned char)ch;
                        cset_put(&s, ch_l, setting);
                        have_l = 1;
                    }
                    ch = getc_escaped(&cur);
                }
                if (bad) { set_err(err, err_len, "pattern %ld: character class not closed before the end of the line", (long)n_pat + 1); rc = PFAC_E_PATTERN; break; }
            } else {
                cset_put(&s, (unsigned char)ch, 1);
            }
            if (n_elem == ecap) { ecap *= 2; elems = (cset_t *)realloc(elems, ecap * sizeof(cset_t)); if (!elems) break; }
            elems[n_elem++] = s;
            if (++len > PFAC_MAX_PATTERN_LEN) { set_err(err, err_len, "Pattern %ld length over 1024.", (long)n_pat + 1); rc = PFAC_E_PATTERN; break; }
        }
        if (rc || !elems) break;
        if (len == 0) { set_err(err, err_len, "pattern %ld is empty", (long)n_pat + 1); rc = PFAC_E_PATTERN; break; }
        if (n_pat == pcap) {
            pcap *= 2;
            poff = (int32_t *)realloc(poff, pcap * sizeof(int32_t));
            plen = (int32_t *)realloc(plen, pcap * sizeof(int32_t));
            if (!poff || !plen) break;
        }
        poff[n_pat] = (int32_t)start; plen[n_pat] = len; n_pat++;
        if (len > max_len) max_len = len;
    }
    if (!elems || !poff || !plen) { free(elems); free(poff); free(plen); set_err(err, err_len, "out of memory (%ld patterns)", (long)n_pat); return PFAC_E_NOMEM; }
    if (!rc && n_pat == 0) { set_err(err, err_len, "no patterns%ld", 0); rc = PFAC_E_PATTERN; }
    if (rc) { free(elems); free(poff); free(plen); return rc; }

    /* ---- subset construction, breadth first: a state = the patterns alive after `depth` bytes ---- */
    size_t scap = 1024, n_states = 0, mcap = 4096 + n_pat, n_mem = 0, ecap2 = 4096, n_edges = 0, hcap = 4096;
    dstate_t *st = (dstate_t *)malloc(scap * sizeof(dstate_t));
    int32_t *mem = (int32_t *)malloc(mcap * sizeof(int32_t));
    edge_t *edges = (edge_t *)malloc(ecap2 * sizeof(edge_t));
    int32_t *htab = (int32_t *)malloc(hcap * sizeof(int32_t));
    int32_t *next = (int32_t *)malloc((n_pat ? n_pat : 1) * sizeof(int32_t));
    int oom = !st || !mem || !edges || !htab || !next;
    if (!oom) {
        memset(htab, 0xFF, hcap * sizeof(int32_t));
        for (size_t p = 0; p < n_pat; p++) mem[n_mem++] = (int32_t)p;
        st[0].first = 0; st[0].count = (int32_t)n_pat; st[0].depth = 0; st[0].id = -1;
        n_states = 1;
        htab[hash_members(mem, (int32_t)n_pat, 0) & (hcap - 1)] = 0;
    }
    for (size_t si = 0; si < n_states && !oom && !rc; si++) {
        const dstate_t cs = st[si];
        for (int c = 0; c < 256 && !oom && !rc; c++) {
            int32_t nn = 0;
            for (int32_t k = 0; k < cs.count; k++) {
                const int32_t p = mem[cs.first + k];
                if (plen[p] > cs.depth && cset_has(&elems[poff[p] + cs.depth], c)) next[nn++] = p;
            }
            if (nn == 0) continue;
            /* known state? (same members at the same depth) */
            const uint64_t h = hash_members(next, nn, cs.depth + 1);
            size_t slot = (size_t)(h & (hcap - 1));
            int32_t found = -1;
            while (htab[slot] >= 0) {
                const dstate_t *o = &st[htab[slot]];
                if (o->depth == cs.depth + 1 && o->count == nn && memcmp(mem + o->first, next, (size_t)nn * sizeof(int32_t)) == 0) { found = htab[slot]; break; }
                slot = (slot + 1) & (hcap - 1);
            }
            if (found < 0) {
                if (n_states >= (size_t)(INT32_MAX / 256) - 8) { set_err(err, err_len, "automaton too large (%ld states)", (long)n_states); rc = PFAC_E_PATTERN; break; }
                if (n_states == scap) { scap *= 2; st = (dstate_t *)realloc(st, scap * sizeof(dstate_t)); if (!st) { oom = 1; break; } }
                if (n_mem + (size_t)nn > mcap) { while (n_mem + (size_t)nn > mcap) mcap *= 2; mem = (int32_t *)realloc(mem, mcap * sizeof(int32_t)); if (!mem) { oom = 1; break; } }
                memcpy(mem + n_mem, next, (size_t)nn * sizeof(int32_t));
                st[n_states].first = (int32_t)n_mem; st[n_states].count = nn; st[n_states].depth = cs.depth + 1; st[n_states].id = -1;
                n_mem += (size_t)nn;
                found = (int32_t)n_states++;
                htab[slot] = found;
                if (n_states * 2 > hcap) {             /* rehash */
                    hcap *= 4;
                    free(htab);
                    htab = (int32_t *)malloc(hcap * sizeof(int32_t));
                    if (!htab) { oom = 1; break; }
                    memset(htab, 0xFF, hcap * sizeof(int32_t));
                    for (size_t q = 0; q < n_states; q++) {
                        size_t s2 = (size_t)(hash_members(mem + st[q].first, st[q].count, st[q].depth) & (hcap - 1));
                        while (htab[s2] >= 0) s2 = (s2 + 1) & (hcap - 1);
                        htab[s2] = (int32_t)q;
                    }
                }
            }
            if (n_edges == ecap2) { ecap2 *= 2; edges = (edge_t *)realloc(edges, ecap2 * sizeof(edge_t)); if (!edges) { oom = 1; break; } }
            edges[n_edges].from = (int32_t)si; edges[n_edges].ch = c; edges[n_edges].to = found; n_edges++;
        }
    }
    free(next); free(htab);
    pfac_table *t = NULL;
    pfac_outputs *o = NULL;
    if (!oom && !rc) {
        /* ---- numbering: finals 0 .. F-1 in BFS (= creation) order, F unused, root F+1, the rest behind ---- */
        int32_t F = 0, n_out = 0;
        for (size_t q = 1; q < n_states; q++) {
            int fin = 0;
            for (int32_t k = 0; k < st[q].count; k++) if (plen[mem[st[q].first + k]] == st[q].depth) { fin = 1; n_out++; }
            if (fin) st[q].id = F++;
        }
        int32_t nf = F + 2;
        st[0].id = F + 1;
        for (size_t q = 1; q < n_states; q++) if (st[q].id < 0) st[q].id = nf++;
        t = (pfac_table *)calloc(1, sizeof *t);
        o = (pfac_outputs *)calloc(1, sizeof *o);
        if (t && o) {
            t->width = width;
            for (t->width_bit = 0; (width >> t->width_bit) != 1; t->width_bit++) ;
            t->n_patterns = (int32_t)n_pat;
            t->num_final = F;
            t->state_num = nf;
            t->max_pat_len = max_len;
            t->idmap = (int32_t *)malloc((F ? (size_t)F : 1) * sizeof(int32_t));
            t->s0 = (int32_t *)malloc(256 * sizeof(int32_t));
            o->n_states = F;
            o->first = (int32_t *)malloc(((size_t)F + 1) * sizeof(int32_t));
            o->ids = (int32_t *)malloc((n_out ? (size_t)n_out : 1) * sizeof(int32_t));
        }
        if (!t || !o || !t->idmap || !t->s0 || !o->first || !o->ids) oom = 1;
        else {
            /* outputs per final state: the patterns that END there, ascending id (members are ascending) */
            int32_t at = 0;
            for (size_t q = 1; q < n_states; q++) {            /* finals were numbered in this order */
                if (st[q].id >= F) continue;
                o->first[st[q].id] = at;
                for (int32_t k = 0; k < st[q].count; k++) {
                    const int32_t p = mem[st[q].first + k];
                    if (plen[p] == st[q].depth) o->ids[at++] = p + 1;          /* pattern id = 1-based line number */
                }
                t->idmap[st[q].id] = o->ids[o->first[st[q].id]];
            }
            o->first[F] = at;
            memset(t->s0, 0xFF, 256 * sizeof(int32_t));
            for (size_t e = 0; e < n_edges; e++) { edges[e].from = st[edges[e].from].id; edges[e].to = st[edges[e].to].id; }
            qsort(edges, n_edges, sizeof(edge_t), edge_key_cmp);
            for (size_t e = 0; e < n_edges; e++) if (edges[e].from == F + 1) t->s0[edges[e].ch] = edges[e].to;
            rc = build_phf(t, edges, (int32_t)n_edges, err, err_len);
        }
    }
    free(st); free(mem); free(edges); free(elems); free(poff); free(plen);
    if (oom) { set_err(err, err_len, "out of memory (%ld states)", (long)n_states); rc = PFAC_E_NOMEM; }
    if (rc) { pfac_table_free(t); pfac_outputs_free(o); return rc; }
    *out = t;
    *outs = o;
    return PFAC_OK;
}

int pfac_table_build_mem_charclass(const void *patterns, size_t n_bytes, int width, pfac_table **out, pfac_outputs **outputs,
                                   char *err, size_t err_len) {
    if (!patterns || !out || !outputs) return PFAC_E_ARG;
    *out = NULL; *outputs = NULL;
    if (!is_pow2(width) || width > PFAC_COL_MAX) { set_err(err, err_len, "PHF width %ld must be a power of two <= 4096", width); return PFAC_E_ARG; }
    return build_charclass_mem((const unsigned char *)patterns, n_bytes, width, out, outputs, err, err_len);
}

int pfac_table_build_file_charclass(const char *pattern_file, int width, pfac_table **out, pfac_outputs **outputs, char *err,
                                    size_t err_len) {
    if (!pattern_file || !out || !outputs) return PFAC_E_ARG;
    *out = NULL; *outputs = NULL;
    FILE *f = fopen(pattern_file, "rb");
    if (!f) { if (err && err_len) snprintf(err, err_len, "cannot open pattern file %s", pattern_file); return PFAC_E_IO; }
    size_t cap = 1 << 16, n = 0;
    unsigned char *img = (unsigned char *)malloc(cap);
    while (img) {
        n += fread(img + n, 1, cap - n, f);
        if (n < cap) break;
        cap *= 2;
        img = (unsigned char *)realloc(img, cap);
    }
    const int io_error = ferror(f);
    fclose(f);
    if (!img) { set_err(err, err_len, "out of memory (%ld bytes of pattern file)", (long)n); return PFAC_E_NOMEM; }
    if (io_error) { free(img); if (err && err_len) snprintf(err, err_len, "cannot read pattern file %s", pattern_file); return PFAC_E_IO; }
    const int rc = pfac_table_build_mem_charclass(img, n, width, out, outputs, err, err_len);
    free(img);
    return rc;
}

/* One line per (record, pattern that ends in the record's final state), patterns in ascending id. */
int64_t pfac_emit_records_multi(void *file, const pfac_record *rec, uint64_t n, uint64_t base, const pfac_outputs *outs) {
    if (!file || (!rec && n) || !outs) return PFAC_E_ARG;
    FILE *f = (FILE *)file;
    enum { CHUNK = 1 << 14, LINE_MAX_BYTES = 64 };
    size_t cap = (size_t)CHUNK * LINE_MAX_BYTES;
    char *buf = (char *)malloc(cap);
    if (!buf) return PFAC_E_NOMEM;
    int64_t total = 0;
    char *p = buf;
    for (uint64_t k = 0; k < n; k++) {
        if (rec[k].state >= (uint32_t)outs->n_states) { free(buf); return PFAC_E_ARG; }
        for (int32_t j = outs->first[rec[k].state]; j < outs->first[rec[k].state + 1]; j++) {
            if ((size_t)(p - buf) + LINE_MAX_BYTES > cap) {
                if (fwrite(buf, 1, (size_t)(p - buf), f) != (size_t)(p - buf)) { free(buf); return PFAC_E_IO; }
                total += p - buf;
                p = buf;
            }
            memcpy(p, "At position ", 12); p += 12;
            p = put_uint(p, base + rec[k].pos, 4);
            memcpy(p, ", match pattern ", 16); p += 16;
            p = put_uint(p, (uint64_t)outs->ids[j], 1);
            *p++ = '\n';
        }
    }
    if (fwrite(buf, 1, (size_t)(p - buf), f) != (size_t)(p - buf)) { free(buf); return PFAC_E_IO; }
    total += p - buf;
    free(buf);
    return total;
}
